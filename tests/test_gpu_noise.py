"""Stochastic stages on the GPU.  The reference draws from numpy's global RNG and
qarray's noise classes (source absent), so parity here is DISTRIBUTIONAL / structural:
reproducibility per seed, exact no-op where the reference adds nothing, N(0,1) replacement
beyond full_noise_distance, and the radial amplitude profile of qarray_base_class.py:470-493."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _env(noise, seed=5, B=4, N=4, R=32):
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=seed, noise=noise,
                              capacitance_model=SyntheticCapacitanceModel(2))
    env.reset()
    return env


def _observe_at(env, offsets):
    """Place env e's gate voltages `offsets[e]` volts from its ground truth (barriers at truth)."""
    from qadapt_hip import _lib
    st, steps = env.get_state()
    L, N = env.L, env.N
    for e, off in enumerate(offsets):
        st[e, L.s_gate_v:L.s_gate_v + N] = st[e, L.s_gate_gt:L.s_gate_gt + N] + off
        st[e, L.s_barrier_v:L.s_barrier_v + N - 1] = st[e, L.s_barrier_gt:L.s_barrier_gt + N - 1]
    env.set_state(st, steps)
    _lib.check(env._h, env._lib.qd_observe(env._h, None, 0, env._stream()), "qd_observe")
    return env.raw()[0], st


def test_radial_noise_profile_and_replacement():
    offs = [1.0, 27.0, 33.0, 60.0]
    det = _env(None); z0, _ = _observe_at(det, offs); det.close()
    env = _env(["radial"]); z1, st = _observe_at(env, offs)
    P = env._params_host; L = env.L; N = env.N; R = env.R
    for e, off in enumerate(offs):
        zr, rd, full, amp = P[e, L.noise + 4:L.noise + 8]
        w = P[e, L.scal + 2]
        assert 20 <= zr <= 30 and 30 <= full <= 40 and amp == 0.05
        d = z1[e] - z0[e]
        if off > full:                                    # replaced by randn
            assert abs(z1[e].mean()) < 0.1 and abs(z1[e].std() - 1.0) < 0.1
            continue
        # expected amplitude per pixel (channel 0): clip(alpha (dist - zero_radius), 0, max)
        v1 = st[e, L.s_gate_v + 0]; v2 = st[e, L.s_gate_v + 1]
        V1, V2 = np.meshgrid(np.linspace(v1 - w, v1 + w, R), np.linspace(v2 - w, v2 + w, R))
        dist = np.sqrt((V1 - st[e, L.s_gate_gt]) ** 2 + (V2 - st[e, L.s_gate_gt + 1]) ** 2).reshape(-1)
        a = np.clip(amp / rd * (dist - zr), 0, amp)
        if a.max() == 0:
            assert np.array_equal(z1[e], z0[e])           # inside zero_radius: exactly untouched
        else:
            nz = a > 1e-4
            r = d[0][nz] / a[nz]
            assert abs(r.mean()) < 0.15 and abs(r.std() - 1.0) < 0.15
            assert np.all(d[0][~nz] == 0)
    env.close()


def test_sensor_noise_is_small_reproducible_and_seeded():
    offs = [0.5, 1.0, 2.0, 3.0]
    det = _env(None); z0, _ = _observe_at(det, offs); det.close()
    a = _env(["sensor"], seed=5); za, _ = _observe_at(a, offs); a.close()
    b = _env(["sensor"], seed=5); zb, _ = _observe_at(b, offs); b.close()
    c = _env(["sensor"], seed=6); zc, _ = _observe_at(c, offs); c.close()
    assert np.array_equal(za, zb)                          # same seed, same call sequence: identical
    d = za - z0
    assert np.any(d != 0) and np.abs(d).max() < 1.0        # amplitudes <= 1e-4 (white), 0.012 (telegraph)
    assert not np.array_equal(za, z0)
    # a different seed samples a different device, so only check it runs and differs
    assert zc.shape == za.shape and not np.array_equal(zc, za)


def test_noise_changes_between_observations():
    env = _env(True)
    z1, _ = _observe_at(env, [1.0, 27.0, 33.0, 60.0])
    z2, _ = _observe_at(env, [1.0, 27.0, 33.0, 60.0])
    assert not np.array_equal(z1, z2)                      # new observation number -> new streams
    obs = env._obs()
    img = obs["image"].cpu().numpy()
    assert np.all((img >= 0) & (img <= 1))
    env.close()


def test_latching_identity_when_always_accepted_and_hysteresis_otherwise():
    """a14 (UNVERIFIED restatement of qarray's LatchingModel): with every acceptance probability
    equal to 1 latching is the identity; with small probabilities some pixels keep the previous
    pixel's occupations (hysteresis along the fast scan axis), rows restart clean."""
    import ctypes
    from qadapt_hip import _lib
    offs = [0.5, 1.0, 1.5, 2.0]
    det = _env(None); z0, _ = _observe_at(det, offs); det.close()
    env = _env(["latch"])
    L, N, R = env.L, env.N, env.R
    # acceptance probabilities all 1 -> identity
    eb = env.last_episode
    P1 = eb.params.copy(); P1[:, L.pleads:L.pleads + N] = 1.0; P1[:, L.pinter:L.pinter + N * N] = 1.0
    ids = np.arange(env.B, dtype=np.int32)
    st_now, steps_now = env.get_state()                       # keep the VGM / voltages reset() left behind
    _lib.check(env._h, env._lib.qd_load_episodes(env._h, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), env.B,
                                                 P1.ctypes.data, st_now.ctypes.data, 0, env._stream()), "load")
    z1, _ = _observe_at(env, offs)
    assert np.array_equal(z1, z0)
    # tiny acceptance probabilities -> held states
    P2 = eb.params.copy(); P2[:, L.pleads:L.pleads + N] = 0.05; P2[:, L.pinter:L.pinter + N * N] = 0.05
    _lib.check(env._h, env._lib.qd_load_episodes(env._h, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), env.B,
                                                 P2.ctypes.data, st_now.ctypes.data, 0, env._stream()), "load")
    z2, _ = _observe_at(env, offs)
    occ = env.occupations()
    changed = z2 != z0
    assert changed.any()
    # the first pixel of every row is never latched
    assert not changed.reshape(env.B, N - 1, R, R)[:, :, :, 0].any()
    # a latched pixel carries exactly the occupations of its left neighbour
    idx = np.argwhere(changed)
    for (e, ch, p) in idx[:50]:
        assert np.array_equal(occ[e, ch, p], occ[e, ch, p - 1])
    env.close()


# ---------------------------------------------------------------------------------------------------------
# Sample-by-sample: the kernels against the numpy restatement of the same rules on the same Philox stream
# (oracle/qd_noise_oracle.py).  Stays "unverified against qarray" (source absent), but no longer self-certified.
# ---------------------------------------------------------------------------------------------------------
def _noise_params(P, L, e):
    return dict(white_amp=P[e, L.noise + 0], tel_p01=P[e, L.noise + 1], tel_p10=P[e, L.noise + 2], tel_amp=P[e, L.noise + 3],
                zero_radius=P[e, L.noise + 4], ramp_distance=P[e, L.noise + 5],
                full_noise_distance=P[e, L.noise + 6] if P[e, L.noise + 6] > 0 else None, max_amplitude=P[e, L.noise + 7])


@pytest.mark.parametrize("flags", [["sensor"], ["radial"], ["latch"], ["sensor", "radial", "latch"]])
def test_noisy_observation_matches_numpy_restatement(flags):
    import qd_noise_oracle as NO
    import qd_oracle_c as OC
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    B, N, R, seed, off = 3, 4, 24, 99, 40
    env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=seed, env_id_offset=off, noise=flags, validate=True,
                              capacitance_model=SyntheticCapacitanceModel(2))
    env.reset()
    # env 0 near its ground truth, env 1 in the radial ramp, env 2 beyond full_noise_distance
    raw, st = _observe_at(env, [1.5, 27.0, 70.0])
    occ = env.occupations()
    ck = env.get_checkpoint()
    serial = ck["obs_serial"]                               # the observation just rendered
    P = env._params_host; L = env.L
    latched_any = False
    for e in range(B):
        dev = H.dev_view(N, P[e]); sv = H.state_view(N, st[e])
        s = NO.Stream(seed, off + e, serial)
        nz = _noise_params(P, L, e)
        p_leads = P[e, L.pleads:L.pleads + N]; p_inter = P[e, L.pinter:L.pinter + N * N].reshape(N, N)
        for ch in range(N - 1):
            det = OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R)
            z, used = NO.observe_channel(dev, nz, s, ch, R, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v,
                                         dev.window, sv.gate_gt, det["occ"], set(flags), p_leads, p_inter)
            ok = det["tc"].max(axis=1) < 1e6
            if "radial" in flags and NO.radial_replaced(sv.gate_v[ch], sv.gate_v[ch + 1], sv.gate_gt[ch], sv.gate_gt[ch + 1],
                                                        nz["full_noise_distance"]):
                assert np.allclose(raw[e, ch], z, rtol=1e-12, atol=1e-12), (flags, e, ch)     # pure N(0,1) image
                continue
            if "latch" in flags:
                # the latched occupations themselves: exact copies of the held pixel's values
                assert np.allclose(occ[e, ch][ok], used[ok], rtol=1e-6, atol=1e-6), (flags, e, ch)
                latched_any |= bool((used != det["occ"]).any())
            assert np.allclose(raw[e, ch][ok], z[ok], rtol=1e-6, atol=1e-9), (flags, e, ch, np.abs(raw[e, ch][ok] - z[ok]).max())
    assert latched_any or "latch" not in flags, "the scene was meant to contain latched pixels"
    env.close()


def test_mixed_n_with_latching_config5():
    """BASELINE config 5 in miniature: a ragged batch N in {2,4,6,8} with the latched model on, sharded two ways;
    every bucket's images equal a homogeneous env with the same global env ids, and latching changed something."""
    import torch
    from qadapt_hip.mixed import MixedVecQuantumDeviceEnv
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    counts = {2: 3, 4: 3, 6: 2, 8: 2}
    R = 16
    fac = lambda n: SyntheticCapacitanceModel(40 + n)
    shards = [MixedVecQuantumDeviceEnv(counts, resolution=R, seed=7, rank=r, world=2, capacitance_model_factory=fac, noise=["latch"])
              for r in range(2)]
    seen = []; changed = False
    for sh in shards:
        sh.reset()
        for n, e in sh.buckets.items():
            first, cnt = sh.assignment[n]
            seen += list(range(first, first + cnt))
            ref = VecQuantumDeviceEnv(cnt, num_dots=n, resolution=R, seed=7, env_id_offset=first, noise=["latch"],
                                      capacitance_model=fac(n))
            ref.reset()
            assert np.array_equal(e.global_image.cpu().numpy(), ref.global_image.cpu().numpy()), (n, first)
            plain = VecQuantumDeviceEnv(cnt, num_dots=n, resolution=R, seed=7, env_id_offset=first, capacitance_model=fac(n))
            plain.reset()
            changed |= not np.array_equal(e.raw()[0], plain.raw()[0])
            ref.close(); plain.close()
        acts = {n: torch.zeros((e.B, 2 * n - 1), device="cuda") for n, e in sh.buckets.items()}
        out = sh.step(acts, auto_reset=True)
        assert set(out) == set(sh.buckets)
    assert sorted(seen) == list(range(sum(counts.values())))
    assert changed, "latching was meant to change at least one bucket's raw signal"
    for sh in shards:
        sh.close()

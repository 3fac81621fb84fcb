"""The tile-shared candidate search (csrc/qd_tile.h) against the exact per-pixel search (same library, A/B
switch) and against the oracle's brute-force scan: kept charge states bit-exact, in every regime, including image
edges that cut tiles (R not a multiple of 8).  Also that the fast path really ran (search counters)."""
import numpy as np
import pytest

import qd_oracle_c as OC
import helpers as H

pytestmark = pytest.mark.gpu


def _pair(B, N, R, seed, mode, rng):
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    envs = []
    for px in (False, True):
        env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=seed, validate=True, pixel_search=px,
                                  capacitance_model=SyntheticCapacitanceModel(7))
        env.reset()
        envs.append(env)
    st, steps = envs[0].get_state()
    for e in range(B):
        if mode != "start":
            st[e] = H.place(N, st[e], mode, rng)
    for env in envs:
        env.set_state(st, steps)
        env.observe()
    return envs, st


RESID_ANY = 1e-12       # on-device residual ||H x - lam x|| / ||H||_inf the dense per-component solver reaches in EVERY pixel and regime


@pytest.mark.parametrize("N,R,mode", [(8, 64, "start"), (8, 64, "mid"), (8, 64, "near"), (6, 64, "mid"), (5, 44, "near"),
                                       (4, 64, "start"), (4, 64, "near"), (4, 36, "mid"), (7, 33, "start"), (8, 100, "near")])
def test_tile_search_equals_pixel_search(N, R, mode):
    """The default pipeline (one search per 8x8 tile + exact redo pass) against the per-pixel search (A/B switch)."""
    B = 2 if R <= 64 else 1
    rng = np.random.default_rng(31 * N + R)
    (tile, pix), st = _pair(B, N, R, 4000 + N, mode, rng)
    ct = tile.candidates(); cp = pix.candidates()
    assert np.array_equal(ct, cp), (N, R, mode, int((ct != cp).any(axis=(3, 4)).sum()))
    # identical records -> identical ground states
    assert np.array_equal(tile.raw()[0], pix.raw()[0])
    et = tile.eigen()
    assert et[..., 1].max() <= RESID_ANY, et[..., 1].max()
    for e in range(B):
        dev = H.dev_view(N, tile._params_host[e]); sv = H.state_view(N, st[e])
        for ch in range(N - 1):
            sp = H.pixel_spectrum(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R, states=cp[e, ch])
            assert np.all(np.abs(et[e, ch, :, 0] - sp["lam0"]) <= 1e-12 * sp["hnorm"]), (e, ch)   # vs the oracle's dense eigh
    s = tile.search_stats()
    print(f"[tile search] N={N} R={R} {mode}: {s}")
    print(f"[eigen solver] N={N} R={R} {mode}: {tile.solver_stats()}")
    # the fast path did the work (R=33 has 9 one-pixel-wide tiles of 25 per channel)
    assert s["tiles"] > 0 and s["tiles_redone"] < 0.5 * s["tiles"], s
    tile.close(); pix.close()


def test_tile_search_against_brute_force_oracle():
    """Directly against the plain-C 4^N scan (one channel per env to bound the CPU time)."""
    N, R, B = 8, 64, 2
    rng = np.random.default_rng(5)
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=77, validate=True, capacitance_model=SyntheticCapacitanceModel(7))
    env.reset()
    st, steps = env.get_state()
    st[1] = H.place(N, st[1], "mid", rng)
    env.set_state(st, steps); env.observe()
    cand = env.candidates()
    for e, ch in ((0, 2), (1, 5)):
        dev = H.dev_view(N, env._params_host[e]); sv = H.state_view(N, st[e])
        ref = OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R)
        assert np.array_equal(cand[e, ch], ref["states"]), (e, ch)
    env.close()


def test_product_mode_tile_search_equals_validate_mode():
    """The benched path (unsorted records, energies from the tile planes) gives the same images as the validate path."""
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    N, R, B = 6, 64, 3
    imgs = []
    for validate in (True, False):
        env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=91, validate=validate, capacitance_model=SyntheticCapacitanceModel(7))
        env.reset()
        st, steps = env.get_state()
        rng = np.random.default_rng(9)
        for e in range(B):
            st[e] = H.place(N, st[e], ("near", "mid", "near")[e], rng)    # where float64 resolves the ground vector
        env.set_state(st, steps); env.observe()
        raw, _ = env.raw()
        imgs.append((raw, env.global_image.cpu().numpy()))
        env.close()
    assert np.allclose(imgs[0][0], imgs[1][0], rtol=1e-9, atol=1e-12)
    assert np.abs(imgs[0][1] - imgs[1][1]).max() <= 1e-6

"""The bench's random-action regime (tc up to 1e44), pixel by pixel against the plain-C oracle -- the parity sweep
(scripts/parity_sweep.py) as part of the -m gpu tier: several seeds, the envs where round 2's plain-Lanczos kernel
failed (seed 777 env 1: occupations off by 0.26; seed 4242, 6 dots, env 11: 1e-2) and config 3 at its full batch.

Per pixel rule (the same as tests/test_gpu_parity.py::_check_channel): kept charge states bit-exact; on-device eigen
residual <= RESID_MAX and |lambda - lambda_oracle| <= 1e-12 ||H|| in EVERY pixel (whatever the gap); occupations and raw
sensor signal within 1e-6 wherever float64 resolves the ground vector (rel_gap > GAP_MIN), total charge integer in every
pixel.  The same envs are then rendered by a PRODUCT-mode handle (no validate flag, energies from the tile planes,
unsorted records) and its raw signal must agree with the oracle on the same pixels."""
import os

import numpy as np
import pytest

import qd_oracle_c as OC
import helpers as H
from test_gpu_parity import RESID_MAX

pytestmark = pytest.mark.gpu
os.environ.setdefault("OMP_NUM_THREADS", "16")


def _stepped(N, B, R, seed, steps, validate):
    import torch
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=seed, validate=validate, capacitance_model=SyntheticCapacitanceModel(99))
    env.reset()
    gen = torch.Generator(device="cpu").manual_seed(99 + seed - 1234)           # the action stream of scripts/parity_sweep.py
    for _ in range(steps):
        env.step((torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).cuda())
    st, _ = env.get_state()
    env.observe()                     # the last step's update changed the VGM after its image: render the stored state
    return env, st


@pytest.mark.parametrize("N,B,steps,seed,envs", [
    (8, 16, 6, 777, (1, 4, 9, 14)),          # env 1: near-degenerate lowest pair INSIDE the winning component at tc ~ 1e15
    (8, 16, 4, 1234, (0, 5, 8, 13)),         # envs 8, 13: the two defects the round-2 sweep found (tc up to 4e44)
    (8, 16, 5, 4242, (2, 6, 11, 15)),
    (6, 16, 8, 4242, (11, 3, 7, 12)),        # env 11: the 6-dot failure of round 2
])
def test_random_action_sweep_against_oracle(N, B, steps, seed, envs):
    R = 64
    env, st = _stepped(N, B, R, seed, steps, validate=True)
    cand = env.candidates(); occ = env.occupations(); raw, _ = env.raw(); eig = env.eigen()
    params = env._params_host.copy()
    stats = env.solver_stats(); search = env.search_stats()
    env.close()
    penv, pst = _stepped(N, B, R, seed, steps, validate=False)
    assert np.array_equal(pst, st)                                             # same trajectory in product mode
    praw, _ = penv.raw()
    penv.close()
    worst = dict(occ=0.0, sig=0.0, psig=0.0, lam=0.0, res=0.0, unres=0, tc=0.0)
    for e in envs:
        dev = H.dev_view(N, params[e]); sv = H.state_view(N, st[e])
        for ch in range(N - 1):
            tag = (seed, N, e, ch)
            ref = OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R)
            assert np.array_equal(cand[e, ch], ref["states"]), tag
            sp = H.pixel_spectrum(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R, states=ref["states"])
            ok = sp["rel_gap"] > H.GAP_MIN
            res = float(eig[e, ch, :, 1].max()); dl = float((np.abs(eig[e, ch, :, 0] - sp["lam0"]) / sp["hnorm"]).max())
            assert res <= RESID_MAX, (tag, res)
            assert dl <= 1e-12, (tag, dl)
            d_occ = np.abs(occ[e, ch] - ref["occ"]).max(axis=1)
            den = np.maximum(np.abs(ref["z"]), 1e-3)
            d_sig = np.abs(raw[e, ch] - ref["z"]) / den
            d_psig = np.abs(praw[e, ch] - ref["z"]) / den
            assert np.all(sp["rel_gap"][d_occ > 1e-6] <= H.GAP_MIN), (tag, float(d_occ[ok].max()))
            assert np.all(sp["rel_gap"][d_sig > 1e-6] <= H.GAP_MIN), (tag, float(d_sig[ok].max()))
            assert np.all(sp["rel_gap"][d_psig > 1e-6] <= H.GAP_MIN), (tag, float(d_psig[ok].max()))
            tot = occ[e, ch].sum(axis=1)
            assert np.abs(tot - np.round(tot)).max() < 1e-9, tag
            if ok.any():
                worst["occ"] = max(worst["occ"], float(d_occ[ok].max())); worst["sig"] = max(worst["sig"], float(d_sig[ok].max()))
                worst["psig"] = max(worst["psig"], float(d_psig[ok].max()))
            worst["lam"] = max(worst["lam"], dl); worst["res"] = max(worst["res"], res)
            worst["unres"] += int((~ok).sum()); worst["tc"] = max(worst["tc"], float(sp["tcmax"].max()))
    print(f"[sweep] seed {seed}, {N}-dot, envs {envs} after {steps} random-action steps: max tc {worst['tc']:.1e}, "
          f"{worst['unres']} of {len(envs) * (N - 1) * R * R} pixels unresolvable in float64; over the others max |occ - oracle| "
          f"{worst['occ']:.1e}, rel signal error {worst['sig']:.1e} (product mode {worst['psig']:.1e}); every pixel: eigen residual <= "
          f"{worst['res']:.1e}, |lam - lam_oracle| / ||H|| <= {worst['lam']:.1e}; solver {stats}; search {search}")


def test_config3_full_batch_product_mode():
    """BASELINE config 3 exactly: 8 dots, 4096 envs, 64x64, in product mode (what bench.py times), after two random-action
    steps with auto-reset.  Properties over the WHOLE batch (images finite, inside [0, 1], exactly the percentile
    normalisation of the raw signal; per-agent tensors are the documented views) and four envs spread over the launch
    chunks against the C oracle, pixel by pixel."""
    import torch
    import qd_oracle as O
    from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
    N, B, R, seed = 8, 4096, 64, 1234
    env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=seed, capacitance_model=SyntheticCapacitanceModel(99))
    env.reset()
    chunk = env.chunk_envs()
    assert chunk < B                                                           # several launch chunks
    gen = torch.Generator(device="cpu").manual_seed(5)
    for _ in range(2):
        env.step((torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).cuda(), auto_reset=True)
    st, _ = env.get_state()
    env.observe()
    img = env.global_image
    assert bool(torch.isfinite(img).all()) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    assert bool(torch.isfinite(env.plunger_images).all()) and bool(torch.isfinite(env.barrier_images).all())
    # per-agent tensors = views of the global image (whole batch, on the device)
    assert torch.equal(env.barrier_images[..., 0], img.permute(0, 3, 1, 2))
    assert torch.equal(env.plunger_images[:, 0, ..., 0], img[..., 0]) and torch.equal(env.plunger_images[:, N - 1, ..., 1], img[..., N - 2].transpose(1, 2))
    assert torch.equal(env.plunger_images[:, 3, ..., 0], img[..., 2]) and torch.equal(env.plunger_images[:, 3, ..., 1], img[..., 3].transpose(1, 2))
    raw, plohi = env.raw()
    assert np.isfinite(raw).all()
    picks = (0, chunk - 1, chunk, B - 1) if chunk + 1 < B else (0, 1, B - 2, B - 1)     # both sides of a chunk boundary, first and last env
    himg = img[list(picks)].cpu().numpy()
    worst = 0.0
    for k, e in enumerate(picks):
        assert plohi[e, 0] == np.percentile(raw[e], 0.5) and plohi[e, 1] == np.percentile(raw[e], 99.5)
        assert np.array_equal(himg[k], O.normalise_image(raw[e].reshape(N - 1, R, R).transpose(1, 2, 0)))
        dev = H.dev_view(N, env._params_host[e]); sv = H.state_view(N, st[e])
        for ch in range(N - 1):
            ref = OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R)
            sp = H.pixel_spectrum(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R, states=ref["states"])
            ok = sp["rel_gap"] > H.GAP_MIN
            d = np.abs(raw[e, ch] - ref["z"]) / np.maximum(np.abs(ref["z"]), 1e-3)
            assert np.all(sp["rel_gap"][d > 1e-6] <= H.GAP_MIN), (e, ch, float(d[ok].max()))
            if ok.any():
                worst = max(worst, float(d[ok].max()))
    print(f"[config 3, B = {B}, product mode, {chunk} envs per launch] envs {picks}: max relative signal error over resolvable pixels {worst:.1e}")
    env.close()

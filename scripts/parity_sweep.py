"""Parity sweep at the headline shape under the bench workload: B 8-dot 64x64 envs are stepped with random actions and
synthetic CNN outputs (validate mode), then EVERY channel of every env is compared with the plain-C oracle: kept charge
states bit-exact, eigen residual, eigenvalue vs the dense eigh, occupations / signal where float64 resolves the ground vector.
Prints one summary line per env and a total; profiles/r02_parity_sweep.txt keeps the output of the round's run."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np, torch
import helpers as H, qd_oracle_c as OC
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
only = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else None      # optional: compare these envs only
R = 64
seed = int(os.environ.get("QD_SWEEP_SEED", "1234"))      # other seeds: other devices and action sequences
env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=seed, validate=True, capacitance_model=SyntheticCapacitanceModel(99))
env.reset()
gen = torch.Generator(device="cpu").manual_seed(99 + seed - 1234)
for t in range(steps):
    env.step((torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).cuda())
st, _ = env.get_state()
# the last step's update changed the VGM after the image was rendered: re-render with the stored state
env.observe()
cand = env.candidates(); occ = env.occupations(); raw, _ = env.raw(); eig = env.eigen()
tot = dict(px=0, mism=0, unres=0, wocc=0.0, wsig=0.0, wres=0.0, wresok=0.0, wlam=0.0)
dec = {}     # unresolvable pixels by decade of rel_gap: [count, count with |occ - oracle| > 1e-3, > 1e-6]
t0 = time.time()
for e in (range(B) if only is None else only):
    dev = H.dev_view(N, env._params_host[e]); sv = H.state_view(N, st[e])
    m = u = 0; wo = ws = wr = wrk = wl = 0.0; tcm = 0.0
    for ch in range(N - 1):
        ref = OC.csd_channel(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R)
        m += int((cand[e, ch] != ref["states"]).any(axis=(1, 2)).sum())
        sp = H.pixel_spectrum(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R, states=ref["states"])
        ok = sp["rel_gap"] > H.GAP_MIN
        u += int((~ok).sum()); tcm = max(tcm, sp["tcmax"].max())
        wr = max(wr, eig[e, ch, :, 1].max()); wl = max(wl, (np.abs(eig[e, ch, :, 0] - sp["lam0"]) / sp["hnorm"]).max())
        dd = np.abs(occ[e, ch] - ref["occ"]).max(axis=1)
        for g, x in zip(np.floor(np.log10(np.maximum(sp["rel_gap"][~ok], 1e-20))).astype(int), dd[~ok]):
            c = dec.setdefault(int(g), [0, 0, 0]); c[0] += 1; c[1] += x > 1e-3; c[2] += x > 1e-6
        if ok.any():
            wo = max(wo, np.abs(occ[e, ch] - ref["occ"]).max(axis=1)[ok].max())
            wrk = max(wrk, eig[e, ch, :, 1][ok].max())
            ws = max(ws, (np.abs(raw[e, ch] - ref["z"]) / np.maximum(np.abs(ref["z"]), 1e-3))[ok].max())
    print(f"env {e:2d}: state-list mismatches {m}, unresolvable pixels {u:5d}/{(N-1)*R*R}, max |occ-oracle| {wo:.1e}, max rel signal err {ws:.1e}, "
          f"max eigen residual {wr:.1e} (resolvable pixels {wrk:.1e}), max |lam-lam_oracle|/|H| {wl:.1e}, max tc {tcm:.1e}", flush=True)
    tot["px"] += (N - 1) * R * R; tot["mism"] += m; tot["unres"] += u
    tot["wocc"] = max(tot["wocc"], wo); tot["wsig"] = max(tot["wsig"], ws); tot["wres"] = max(tot["wres"], wr); tot["wresok"] = max(tot["wresok"], wrk); tot["wlam"] = max(tot["wlam"], wl)
print(f"TOTAL {N}-dot {R}x{R}, {B} envs (seed {seed}) after {steps} random-action steps: {tot['px']} pixels, {tot['mism']} state-list mismatches, "
      f"{tot['unres']} unresolvable in float64 (rel_gap <= {H.GAP_MIN}), max |occ-oracle| {tot['wocc']:.2e}, max rel signal err {tot['wsig']:.2e}, "
      f"max eigen residual {tot['wres']:.2e} (resolvable pixels {tot['wresok']:.2e}), max |lam-lam_oracle|/|H| {tot['wlam']:.2e}; search stats {env.search_stats()}; oracle time {time.time()-t0:.0f}s")
print("unresolvable pixels by decade of rel_gap (10^d): count, |occ-oracle| > 1e-3, > 1e-6")
for g in sorted(dec): print(f"  1e{g:+d}: {dec[g][0]:7d} {dec[g][1]:7d} {dec[g][2]:7d}")
env.close()

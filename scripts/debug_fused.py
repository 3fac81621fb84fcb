import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as H
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel

def run(N, R, mode, seed, B=2):
    rng = np.random.default_rng(31 * N + R)
    envs = []
    for kw in (dict(fused=True), dict(pixel_search=True)):
        env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=seed, validate=True, capacitance_model=SyntheticCapacitanceModel(7), **kw)
        env.reset(); envs.append(env)
    st, steps = envs[0].get_state()
    for e in range(B):
        if mode != "start":
            st[e] = H.place(N, st[e], mode, rng)
    for env in envs:
        env.set_state(st, steps); env.observe()
    t, p = envs
    et, ep = t.eigen(), p.eigen(); ot, op = t.occupations(), p.occupations()
    cp = p.candidates()
    print(f"== N={N} R={R} {mode}: stats {t.search_stats()}")
    for e in range(B):
        dev = H.dev_view(N, t._params_host[e]); sv = H.state_view(N, st[e])
        for ch in range(N - 1):
            sp = H.pixel_spectrum(dev, sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, sv.barrier_v, dev.window, ch, R, states=cp[e, ch])
            dt = (et[e, ch, :, 0] - sp["lam0"]) / sp["hnorm"]; dp = (ep[e, ch, :, 0] - sp["lam0"]) / sp["hnorm"]
            do = np.abs(ot[e, ch] - op[e, ch]).max(1)
            ok = sp["rel_gap"] > 1e-7
            bad = np.nonzero(np.abs(dt) > 1e-11)[0]
            print(f" e{e} ch{ch}: fused (lam-lam0)/|H| min {dt.min():.1e} max {dt.max():.1e} | pixel min {dp.min():.1e} max {dp.max():.1e} | resid fused {et[e,ch,:,1].max():.1e} pixel {ep[e,ch,:,1].max():.1e} | occ diff max(resolvable) {do[ok].max() if ok.any() else 0:.1e} | tcmax {sp['tcmax'].max():.1e} nbad {len(bad)} first {bad[:5]}")
    for env in envs: env.close()

for a in sys.argv[1:]:
    N, R, mode = a.split(",")
    run(int(N), int(R), mode, 4000 + int(N))

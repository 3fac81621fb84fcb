"""Kernel micro-benchmark: times the candidate and ground-state kernels alone
(HIP events inside the library) for a batch placed in a chosen voltage regime."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
import helpers as H

ap = argparse.ArgumentParser()
ap.add_argument("--dots", type=int, default=8); ap.add_argument("--envs", type=int, default=128)
ap.add_argument("--resolution", type=int, default=64); ap.add_argument("--modes", default="start,near,mid")
ap.add_argument("--iters", type=int, default=3); ap.add_argument("--pixel-search", action="store_true")
ap.add_argument("--each", action="store_true", help="also time every hot kernel by itself")
a = ap.parse_args()
env = VecQuantumDeviceEnv(a.envs, num_dots=a.dots, resolution=a.resolution, seed=1234, capacitance_model=SyntheticCapacitanceModel(1),
                          pixel_search=a.pixel_search, env_chunk=a.envs)        # all envs in ONE launch (no split over the lanes)
env.reset()
st0, steps = env.get_state()
rng = np.random.default_rng(0)
for mode in a.modes.split(","):
    if mode.startswith("wild"):
        # the bench's regime: envs reset, then stepped with uniform random actions for e % K + 1 steps (staggered phases)
        K = int(mode[4:] or 12)
        env.set_state(st0, steps)
        gen = torch.Generator(device="cpu").manual_seed(7)
        act = torch.rand((K, a.envs, 2 * a.dots - 1), generator=gen) * 2 - 1
        phase = torch.arange(a.envs) % K
        for t in range(K):
            act_t = act[t].clone(); act_t[phase < t] = 0.0          # envs past their phase hold still (zero deltas are not a no-op for absolute actions, so:)
            stb, sb = env.get_state()
            env.step(act_t.cuda())
            sta, sa = env.get_state()
            hold = (phase < t).numpy()
            sta[hold] = stb[hold]
            env.set_state(sta, sb)
    else:
        st = st0.copy()
        for e in range(a.envs):
            st[e] = H.place(a.dots, st0[e], mode, rng, vgm_noise=0.0)
        env.set_state(st, steps)
    c = env.time_candidates_kernel(a.iters); g = env.time_ground_kernel(a.iters)
    px = a.envs * (a.dots - 1) * a.resolution ** 2
    print(f"{mode:6s} N={a.dots} B={a.envs}: candidates {c:8.3f} ms ({c*1e6/px:6.2f} ns/px)  ground {g:8.3f} ms ({g*1e6/px:6.2f} ns/px)"
          f"  => kernel-only {a.envs/((c+g)*1e-3):9.1f} env-steps/s", flush=True)
    if a.each:
        k = env.time_kernels(a.iters)
        print(f"{mode:6s} per kernel, us per env-step: " + ", ".join(f"{n} {v * 1e3 / min(a.envs, env.chunk_envs()):.2f}" for n, v in k.items()), flush=True)
env.close()

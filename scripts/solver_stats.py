"""Diagnostic (needs a -DQD_DEBUG_STATS build: scripts/ab_build.sh stats -DQD_DEBUG_STATS; QDSIM_LIB=ab/libqdsim_stats.so):
per-pixel statistics of the ground-state kernel's solve -- lanes in solved components, component sizes, Laguerre iterations
of the wave (two pixels) against those of the winning component -- after reset and in the bench's random-action regime."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"))
import numpy as np, torch
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
N = 8; B = int(sys.argv[1]) if len(sys.argv) > 1 else 48; K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
env = VecQuantumDeviceEnv(B, num_dots=N, resolution=64, seed=1234, validate=True, capacitance_model=SyntheticCapacitanceModel(1))
env.reset()
def report(tag):
    env.observe()
    v = env.eigen()[..., 1].reshape(-1).round().astype(np.int64)
    wits = v % 100; mits = (v // 100) % 100; kmax = (v // 10**4) % 100; kwin = (v // 10**6) % 100; nact = (v // 10**8) % 100; big = (v // 10**10) % 100
    print(f"{tag}: {v.size} pixels")
    print(f"  Laguerre iterations: wave mean {wits.mean():.2f} (hist by 5: {np.bincount(wits // 5, minlength=10).tolist()}), winning component's own {mits.mean():.2f}; "
          f"share of all wave iterations spent in waves with >= 10: {wits[wits >= 10].sum() / max(wits.sum(), 1):.2f}")
    print(f"  rows of T: wave max {kmax.mean():.2f}, winner {kwin.mean():.2f}")
    print(f"  lanes in solved components: mean {nact.mean():.1f}, <= 8: {(nact <= 8).mean():.3f}, <= 16: {(nact <= 16).mean():.3f}, 25..32: {(nact >= 25).mean():.3f}")
    print(f"  largest solved component, hist 0..: {np.bincount(big, minlength=15).tolist()}")
report("after reset")
gen = torch.Generator(device="cpu").manual_seed(7)
phase = torch.arange(B) % K
for t in range(K):
    a = torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1
    stb, sb = env.get_state(); env.step(a.cuda()); sta, sa = env.get_state()
    hold = (phase < t).numpy(); sta[hold] = stb[hold]; env.set_state(sta, sb)
report(f"staggered 1..{K} random steps")
env.close()

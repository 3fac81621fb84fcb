"""Tile-search counters under the bench workload (random actions, synthetic CNN outputs, a few steps)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"))
import numpy as np, torch
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 24
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
env = VecQuantumDeviceEnv(B, num_dots=N, resolution=64, seed=1234, validate=True, capacitance_model=SyntheticCapacitanceModel(99))
env.reset()
gen = torch.Generator(device="cpu").manual_seed(99)
prev = None
for t in range(steps):
    env.step((torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).cuda())
    s = env.search_stats()
    cur = np.array([s["tiles"], s["tiles_redone"], s["pixels_redone"]] + list(s["tiles_redone_by_reason"].values()))
    d = cur if prev is None else cur - prev
    prev = cur
    print(f"step {t}: tiles {d[0]} redone {d[1]} ({100*d[1]/max(d[0],1):.1f}%) px_redone {d[2]} reasons(ranges,seeds,frontier,leaves,superset) {d[3:].tolist()} mean|S| {s['mean_superset']:.1f}", flush=True)
env.close()

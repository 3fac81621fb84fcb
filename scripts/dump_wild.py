"""Dump the device parameters / state / kept states of a few envs of the parity-sweep workload (offline solver studies)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
N, R = 8, 64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seed = int(os.environ.get("QD_SWEEP_SEED", "1234"))
env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=seed, validate=True, capacitance_model=SyntheticCapacitanceModel(99))
env.reset()
gen = torch.Generator(device="cpu").manual_seed(99 + seed - 1234)
for t in range(steps):
    env.step((torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).cuda())
st, _ = env.get_state(); env.observe()
cand = env.candidates(); eig = env.eigen()
sel = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [3, 5, 9, 10]
out_name = sys.argv[4] if len(sys.argv) > 4 else "wild_dump.npz"
occ = env.occupations()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", out_name), occ=occ[sel], params=np.stack([env._params_host[e] for e in sel]),
                    state=np.stack([st[e] for e in sel]), cand=cand[sel], eig=eig[sel], sel=np.array(sel))
env.close()

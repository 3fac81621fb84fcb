"""Dump the device parameters / state / kept states of a few envs of the parity-sweep workload (offline solver studies)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
N, B, steps, R = 8, 12, 3, 64
env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=1234, validate=True, capacitance_model=SyntheticCapacitanceModel(99))
env.reset()
gen = torch.Generator(device="cpu").manual_seed(99)
for t in range(steps):
    env.step((torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).cuda())
st, _ = env.get_state(); env.observe()
cand = env.candidates(); eig = env.eigen()
sel = [3, 5, 9, 10]
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "wild_dump.npz"), params=np.stack([env._params_host[e] for e in sel]),
                    state=np.stack([st[e] for e in sel]), cand=cand[sel], eig=eig[sel], sel=np.array(sel))
env.close()

"""The same batched step with the observations handed to the HOST, as the RLlib-format boundary does
(qadapt_hip/multi_agent.py::_HostMirror: per-agent images, voltages, rewards and flags copied into a pinned ring, one stream
synchronisation per step) -- the PCIe-inclusive rate next to the device-resident one bench.py reports.
    python scripts/pcie_rate.py [envs] [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"))
import numpy as np, torch
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
from qadapt_hip.multi_agent import _HostMirror
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N = 8
env = VecQuantumDeviceEnv(B, num_dots=N, resolution=64, seed=1234, capacitance_model=SyntheticCapacitanceModel(1))
env.reset(); env.stagger_episodes()
gen = torch.Generator(device="cpu").manual_seed(3)
acts = [(torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).cuda() for _ in range(4)]
def run(label, mirror):
    for t in range(2):
        env.step(acts[t % 4], auto_reset=True)
        if mirror: mirror.pull()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nbytes = 0
    for t in range(steps):
        env.step(acts[t % 4], auto_reset=True)
        if mirror:
            h = mirror.pull(); nbytes = sum(a.nbytes for a in h.values())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {B * steps / dt:.1f} env-steps/s ({dt / steps * 1e3:.1f} ms per step" + (f", {nbytes / 1e6:.0f} MB to the host per step)" if mirror else ")"), flush=True)
run("observations stay in HBM (bench.py)", None)
run("observations to pinned host memory, views handed out (zero_copy)", _HostMirror(env, with_global=False, zero_copy=True))
run("observations to pinned host memory, fresh numpy arrays per step (default)", _HostMirror(env, with_global=False, zero_copy=False))
env.close()

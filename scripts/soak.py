"""Soak run: many batched steps with random actions, resets inside, every observation checked for NaN / range.
    python scripts/soak.py [envs] [steps] [noise: 0/1] [dots]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"))
import numpy as np, torch
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
noise = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False; N = int(sys.argv[4]) if len(sys.argv) > 4 else 8
env = VecQuantumDeviceEnv(B, num_dots=N, resolution=64, seed=77, capacitance_model=SyntheticCapacitanceModel(5),
                          noise=["sensor", "radial", "latch"] if noise else None)
obs = env.reset(); env.stagger_episodes()
gen = torch.Generator(device="cpu").manual_seed(11)
t0 = time.time(); bad = 0; resets = 0
for t in range(steps):
    a = (torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).cuda()
    out = env.step(a, auto_reset=True)
    obs, rew, term, trunc = out[0], out[1], out[2], out[3]
    img = obs["image"] if isinstance(obs, dict) else obs
    fin = torch.isfinite(img).all().item() and torch.isfinite(rew).all().item()
    rng_ok = (img.min().item() >= 0.0) and (img.max().item() <= 1.0)
    resets += int(trunc.sum().item())
    if not (fin and rng_ok):
        bad += 1; print(f"step {t}: finite {fin} range {rng_ok} min {img.min().item()} max {img.max().item()}", flush=True)
    if t % 20 == 19: print(f"step {t + 1}/{steps}: {B * (t + 1) / (time.time() - t0):.0f} env-steps/s, resets so far {resets}, bad steps {bad}", flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED", f"{B} envs x {steps} steps, noise {noise}, N {N}, resets {resets}")
env.close()
sys.exit(1 if bad else 0)

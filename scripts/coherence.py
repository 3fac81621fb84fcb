"""How often do x-adjacent pixels of a tile row carry the identical kept-state SET (validate mode: the lists come in the
reference's energy order, which differs from pixel to pixel; the product pipeline writes them in the order of each lane's
top-32 buffer, which differs as well)?  Decides whether the ground kernel could reuse hop pattern / components along a
tile row if the tile kernel emitted the states in a canonical order."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
from qadapt_hip.vec_env import VecQuantumDeviceEnv, SyntheticCapacitanceModel
import helpers as H
N, B, R = 8, 16, 64
env = VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=1234, validate=True, capacitance_model=SyntheticCapacitanceModel(1))
env.reset()
st0, steps = env.get_state(); rng = np.random.default_rng(0)
def report(tag):
    env.observe()
    c = env.candidates()                                   # [B, C, P, 32, N]
    c = c.reshape(B, N - 1, R, R // 8, 8, 32, N)
    same = (c[:, :, :, :, 1:] == c[:, :, :, :, :-1]).all(axis=(5, 6))       # x-adjacent inside a tile row
    setsame = np.zeros_like(same)
    code = (c.astype(np.int64) * (7 ** np.arange(N))).sum(-1)               # order-free comparison
    cs = np.sort(code, axis=-1)
    setsame = (cs[:, :, :, :, 1:] == cs[:, :, :, :, :-1]).all(axis=-1)
    run = same.all(axis=4)
    print(f"{tag}: adjacent pairs with identical list {same.mean():.3f} (identical set, any order {setsame.mean():.3f}); tile rows (8 px) with one list {run.mean():.3f}")
for mode in ("start", "mid", "near"):
    st = st0.copy()
    for e in range(B): st[e] = H.place(N, st0[e], mode, rng, vgm_noise=0.0)
    env.set_state(st, steps); report(mode)
env.set_state(st0, steps)
gen = torch.Generator(device="cpu").manual_seed(7)
for t in range(6): env.step((torch.rand((B, 2 * N - 1), generator=gen) * 2 - 1).cuda())
report("6 random steps")
env.close()

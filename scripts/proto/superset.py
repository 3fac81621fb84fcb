"""Prototype: size of tile-shared candidate supersets (round-2 candidates-kernel redesign)."""
import sys, os, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import qd_oracle as O
import helpers as H
from qadapt_hip.layout import layout

def run(N=8, R=64, seeds=(1234, 1235, 1236), mode="start", tile=8, ntiles=6, rng=np.random.default_rng(0)):
    eb = H.sample_blocks(N, seeds)
    L = layout(N)
    res = []
    for e in range(len(seeds)):
        par = eb.params[e]; st = eb.state[e]
        if mode != "start":
            st = H.place(N, st, mode, rng, vgm_noise=0.02)
        dev = H.dev_view(N, par); sv = H.state_view(N, st)
        A = dev.cdd_inv_full[:N, :N]
        for ch in rng.choice(N - 1, size=2, replace=False):
            vg = O.sweep_voltages(sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, ch, -dev.window, dev.window, R)
            v_ext = np.concatenate([vg, np.broadcast_to(sv.barrier_v, (R * R, N - 1))], axis=1)
            vd = v_ext @ dev.cgd_full[:N, :].T
            ncont = O.continuous_ground_state(v_ext, dev.cdd_inv_full, dev.cgd_full, N)
            fl = np.floor(ncont).astype(int)
            for _ in range(ntiles):
                ty, tx = rng.integers(0, R // tile, 2)
                ys, xs = np.meshgrid(np.arange(ty * tile, (ty + 1) * tile), np.arange(tx * tile, (tx + 1) * tile), indexing="ij")
                pix = (ys * R + xs).reshape(-1)
                flt = fl[pix]; vdt = vd[pix]
                lo = np.maximum(flt.min(0) - 1, 0); hi = flt.max(0) + 2
                ranges = [np.arange(lo[i], hi[i] + 1) for i in range(N)]
                grid = np.stack(np.meshgrid(*ranges, indexing="ij"), -1).reshape(-1, N).astype(float)
                # energies per pixel
                d = grid[None, :, :] - vdt[:, None, :]
                E = np.einsum("pci,ij,pcj->pc", d, A, d)
                valid = np.all((grid[None] >= np.maximum(flt[:, None, :] - 1, 0)) & (grid[None] <= flt[:, None, :] + 2), -1)
                E = np.where(valid, E, np.inf)
                order = np.argsort(E, axis=1, kind="stable")[:, :32]
                union = np.unique(order)
                # criterion: centre pixel = mean v'
                v0 = vdt.mean(0)
                d0 = grid - v0
                E0 = np.einsum("ci,ij,cj->c", d0, A, d0)
                # planes: E_p(c) - q_p = E0(c) - 2 (v_p - v0)^T A (c - v0)
                dv = vdt - v0                        # (64,N)
                lin = -2 * (dv @ A) @ d0.T           # (64, C)
                anyvalid_pre = valid.any(0)
                q = np.einsum("pi,ij,pj->p", dv, A, dv)
                Eapp = E0[None] + lin + q[:, None]
                err = np.nanmax(np.abs(np.where(valid, Eapp - E, 0)))
                rel = E0[None] + lin
                cref = np.argmin(np.where(anyvalid_pre, E0, np.inf))
                rel = rel - rel[:, cref:cref+1]
                M = rel.max(0); m = rel.min(0)
                anyvalid = valid.any(0)
                allvalid = valid.all(0)
                k0 = np.argsort(np.where(allvalid, E0, np.inf), kind="stable")[:32]
                Tp = M[k0].max()
                S = np.nonzero((m <= Tp) & anyvalid)[0]
                T = np.sort(np.where(allvalid, M, np.inf))[31]
                S2 = np.nonzero((m <= T) & anyvalid)[0]
                assert np.isin(union, S).all() and np.isin(union, S2).all()
                res.append((len(grid), len(union), len(S2), len(S), len(np.unique(flt, axis=0)), err, (ncont[pix] != vd[pix]).any(1).mean()))
    res = np.array(res)
    print(f"N={N} mode={mode} tile={tile}: box {res[:,0].mean():.0f}  union(ideal) mean {res[:,1].mean():.1f} max {res[:,1].max():.0f} | S(T) mean {res[:,2].mean():.1f} max {res[:,2].max():.0f} | S(T') mean {res[:,3].mean():.1f} max {res[:,3].max():.0f} | distinct floors/tile {res[:,4].mean():.1f} | approx err {res[:,5].max():.2e} | clipped frac {res[:,6].mean():.2f}")

if __name__ == "__main__":
    for N in (8, 4, 6):
      for mode in ("start", "mid", "near"):
        for tile in (8,):
            run(N=N, mode=mode, tile=tile)

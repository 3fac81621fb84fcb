"""Prototype 2: separable-bound superset + tile-level hop components (sizes)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import qd_oracle as O
import helpers as H
from qadapt_hip.layout import layout
from scipy.sparse.csgraph import connected_components
from scipy.sparse import csr_matrix

def run(N=8, R=64, seeds=(1234, 1235, 1236, 1237), mode="start", tile=8, ntiles=8, rng=None):
    rng = np.random.default_rng(0)
    eb = H.sample_blocks(N, seeds)
    rows = []
    for e in range(len(seeds)):
        par = eb.params[e]; st = eb.state[e]
        if mode != "start":
            st = H.place(N, st, mode, rng, vgm_noise=0.02)
        dev = H.dev_view(N, par); sv = H.state_view(N, st)
        A = dev.cdd_inv_full[:N, :N]
        for ch in rng.choice(N - 1, size=min(2, N - 1), replace=False):
            vg = O.sweep_voltages(sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, ch, -dev.window, dev.window, R)
            vb = np.broadcast_to(sv.barrier_v, (R * R, N - 1))
            v_ext = np.concatenate([vg, vb], axis=1)
            vd = v_ext @ dev.cgd_full[:N, :].T
            ncont = O.continuous_ground_state(v_ext, dev.cdd_inv_full, dev.cgd_full, N)
            fl = np.floor(ncont).astype(int)
            tc_all = O.tunnel_couplings(O.effective_barrier_potential(vg, vb, dev.Cbg, dev.Cbb), dev.tc_base, dev.alpha)
            for _ in range(ntiles):
                ty, tx = rng.integers(0, R // tile, 2)
                ys, xs = np.meshgrid(np.arange(ty * tile, (ty + 1) * tile), np.arange(tx * tile, (tx + 1) * tile), indexing="ij")
                pix = (ys * R + xs).reshape(-1)
                flt = fl[pix]; vdt = vd[pix]; tct = tc_all[pix]
                lo = np.maximum(flt.min(0) - 1, 0); hi = flt.max(0) + 2
                ranges = [np.arange(lo[i], hi[i] + 1) for i in range(N)]
                grid = np.stack(np.meshgrid(*ranges, indexing="ij"), -1).reshape(-1, N).astype(float)
                d = grid[None, :, :] - vdt[:, None, :]
                E = np.einsum("pci,ij,pcj->pc", d, A, d)
                valid = np.all((grid[None] >= np.maximum(flt[:, None, :] - 1, 0)) & (grid[None] <= flt[:, None, :] + 2), -1)
                E = np.where(valid, E, np.inf)
                order = np.argsort(E, axis=1, kind="stable")[:, :32]
                union = np.unique(order)
                ref = (tile // 2 - 1) * tile + (tile // 2 - 1)
                v0 = vdt[ref]
                d0 = grid - v0
                E0 = np.einsum("ci,ij,cj->c", d0, A, d0)
                allvalid = valid.all(0); anyvalid = valid.any(0)
                cref = np.argmin(np.where(allvalid, E0, np.inf))
                dE = E0 - E0[cref]
                lam = -2 * (vdt - v0) @ A                 # (64,N)
                Lam = np.abs(lam).max(0)                  # (N,)
                dc = np.abs(grid - grid[cref])
                slack = dc @ Lam
                Mh = dE + slack; mh = dE - slack
                T = np.sort(np.where(allvalid, Mh, np.inf))[31]
                S = np.nonzero((mh <= T) & anyvalid)[0]
                assert np.isin(union, S).all()
                # hop graph on S
                cs = grid[S]
                diff = cs[None, :, :] - cs[:, None, :]
                adj = np.zeros((len(S), len(S)), bool); pair = np.zeros((len(S), len(S)), int)
                for dd in range(N - 1):
                    ex = np.zeros(N); ex[dd] = -1; ex[dd + 1] = 1
                    f = np.all(diff == ex, -1) | np.all(diff == -ex, -1)
                    adj |= f; pair[f] = dd
                nc, lab = connected_components(csr_matrix(adj), directed=False)
                csize = np.bincount(lab, minlength=nc)
                # per pixel: kept set, components, gershgorin
                need = set(); maxpp = 0; nsolve_pp = []
                Sidx = {c: i for i, c in enumerate(S)}
                for p in range(len(pix)):
                    kept = np.array([Sidx[c] for c in order[p]])
                    Fp = E[p, order[p]]
                    sub = adj[np.ix_(kept, kept)]
                    Hoff = np.zeros(sub.shape)
                    ii, jj = np.nonzero(sub)
                    for a_, b_ in zip(ii, jj):
                        si = cs[kept[a_]]; dd = pair[kept[a_], kept[b_]]
                        if cs[kept[b_]][dd] < si[dd]:
                            Hoff[a_, b_] = -tct[p, dd] * np.sqrt(si[dd] * (si[dd + 1] + 1))
                        else:
                            Hoff[a_, b_] = -tct[p, dd] * np.sqrt(si[dd + 1] * (si[dd] + 1))
                    ncp, labp = connected_components(csr_matrix(sub), directed=False)
                    lower = Fp - np.abs(Hoff).sum(1)
                    ns = 0
                    for k in range(ncp):
                        mem = labp == k
                        if lower[mem].min() <= Fp.min():
                            if mem.sum() > 1:
                                ns += 1
                                maxpp = max(maxpp, mem.sum())
                            need.add(lab[kept[np.nonzero(mem)[0][0]]])
                    nsolve_pp.append(ns)
                needsz = [csize[k] for k in need]
                rows.append((len(S), len(union), nc, len(need), max(needsz), sum(needsz), maxpp, np.mean(nsolve_pp), max(nsolve_pp), np.log10(tct.max())))
    r = np.array(rows, float)
    print(f"N={N} {mode}: |S| mean {r[:,0].mean():.0f} max {r[:,0].max():.0f} (union {r[:,1].mean():.0f}) | tile comps {r[:,2].mean():.0f}, needed {r[:,3].mean():.1f} max {r[:,3].max():.0f} | largest needed comp mean {r[:,4].mean():.1f} max {r[:,4].max():.0f} p90 {np.percentile(r[:,4],90):.0f} | sum needed {r[:,5].mean():.1f} max {r[:,5].max():.0f} | per-pixel max comp {r[:,6].mean():.1f} max {r[:,6].max():.0f} | solves/pixel mean {r[:,7].mean():.2f} max {r[:,8].max():.0f} | log10 tcmax {r[:,9].mean():.1f}")

if __name__ == "__main__":
    for N in (8, 4, 6):
        for mode in ("start", "mid", "near"):
            run(N=N, mode=mode)

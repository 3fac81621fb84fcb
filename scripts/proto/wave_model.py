"""Offline model of the per-wave serial phases of qd_k_ground on dumped wild-regime pixels: which component sets the
Laguerre iteration count of a wave (two adjacent pixels), is it the winner, would bounds have removed it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from laguerre_study import *

def pixel_components(Hp):
    Hp = Hp - np.diag(Hp).min() * np.eye(32)
    F = np.diag(Hp); rad = np.abs(Hp).sum(axis=1) - np.abs(F)
    comps = components(Hp); out = []
    upper_all = F.min()
    for comp in comps:
        lower = (F[comp] - rad[comp]).min()
        active = lower <= upper_all
        if not active: continue
        if len(comp) == 1:
            out.append(dict(size=1, k=1, its=0, lam=F[comp[0]], ub=F[comp[0]], lo=F[comp[0]])); continue
        Hc = Hp[np.ix_(comp, comp)]
        al, be = lanczos(Hc)
        lam, its = laguerre(al, be)
        ub = al.min()
        if len(al) >= 2:
            ub = min(ub, 0.5 * (al[0] + al[1]) - np.hypot(0.5 * (al[0] - al[1]), be[0]))
        out.append(dict(size=len(comp), k=len(al), its=its, lam=lam, ub=ub))
    return out

if __name__ == "__main__":
    d = np.load(os.path.join(ROOT, "gpurun_out", "wild_dump.npz"))
    N, R = 8, 64
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.default_rng(1)
    tot = dict(waves=0, its=0, its_win=0, its_prune=0, kmax=0, kmax_win=0, setter_is_loser=0, ncomp=0)
    for which in range(4):
        dev = H.dev_view(N, d["params"][which]); sv = H.state_view(N, d["state"][which])
        for ch in rng.choice(N - 1, 2, replace=False):
            Hm, tc = hamiltonians(dev, sv, ch, R, d["cand"][which, ch])
            for p in rng.choice(R * R // 2, nw // 8, replace=False) * 2:
                comps = []
                for q in (p, p + 1):
                    cs = pixel_components(Hm[q]); best = min(c["lam"] for c in cs); ubmin = min(c["ub"] for c in cs)
                    for c in cs:
                        c["win"] = c["lam"] <= best + 1e-12 * max(1.0, abs(best)); c["prunable"] = c["lam"] > ubmin + 1e-14 * abs(ubmin)
                    comps += cs
                its = max(c["its"] for c in comps); itsw = max(c["its"] for c in comps if c["win"])
                itsp = max([c["its"] for c in comps if not c["prunable"]])
                tot["waves"] += 1; tot["its"] += its; tot["its_win"] += itsw; tot["its_prune"] += itsp
                tot["kmax"] += max(c["k"] for c in comps); tot["kmax_win"] += max(c["k"] for c in comps if c["win"])
                setter = max(comps, key=lambda c: c["its"]); tot["setter_is_loser"] += (not setter["win"]); tot["ncomp"] += len(comps)
    w = tot["waves"]
    print({k: (v / w if k != "waves" else v) for k, v in tot.items()})

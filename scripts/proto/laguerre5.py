"""Prototype 5: Laguerre started at the lowest eigenvalue of T's leading 2x2 block (an UPPER bound of the lowest root, usually
close to it) instead of the Gershgorin lower bound; the Sturm count c(x) makes that rigorous: c = 0 left of the spectrum
(step right), c = 1 between the two lowest roots (step left: converges to the lowest root from the right without crossing
it), c >= 2 (only possible at the start) -> fall back to the Gershgorin start."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from laguerre3 import evalc

def laguerre5(al, be, maxit=48, trace=False, stats=None):
    k = len(al)
    if k == 1: return al[0], 0
    bfull = np.concatenate([[0.0], np.abs(be), [0.0]])
    lo = (al - bfull[:-1] - bfull[1:]).min()
    anorm = max(np.abs(al).max(), np.abs(be).max()); tscale = max(abs(lo), 3 * anorm)
    xstart = lo - (1e-3 * tscale + 1e-300)
    hd = 0.5 * (al[0] - al[1]); x0 = 0.5 * (al[0] + al[1]) - np.hypot(hd, be[0])
    xl = x0; sprev = 0.0; left_prev = None; first = True
    for it in range(maxit):
        p1, d1, e1, c = evalc(al, be, xl)
        if trace: print(f"  it {it} x={xl:.17g} c={c} p1={p1:.2e}")
        if first and c >= 2:
            xl = xstart; first = False; sprev = 0.0; left_prev = None
            if stats is not None: stats["fallback"] = stats.get("fallback", 0) + 1
            continue
        first = False
        left = c > 0
        if p1 == 0.0: return xl, it + 1
        G = d1 / p1; E = e1 / p1
        disc = (k - 1.0) * ((k - 1.0) * G * G - k * E)
        sq = np.sqrt(disc) if disc > 0 else 0.0
        den = G - sq if G < 0 else G + sq
        if left != (den > 0):
            alt = G + sq if G < 0 else G - sq
            if left == (alt > 0) and alt != 0: den = alt
        xn = xl - k / den if den != 0 else xl
        st = abs(xn - xl); tol = 4e-16 * max(abs(xn), abs(xl), 0.5 * tscale)
        if xn == xl: return xl, it + 1
        done = st <= tol or (left_prev == left and 100.0 * st ** 4 <= tol * sprev ** 3 and sprev > 0)
        left_prev = left; sprev = st; xl = xn
        if done: return xl, it + 1
    return xl, maxit

if __name__ == "__main__":
    from laguerre_study import *
    rng = np.random.default_rng(5)
    d = np.load(os.path.join(ROOT, "gpurun_out", "wild_dump.npz"))
    N, R = 8, 64; comps = []
    for which in range(4):
        dev = H.dev_view(N, d["params"][which]); sv = H.state_view(N, d["state"][which])
        for ch in range(N - 1):
            Hm, tc = hamiltonians(dev, sv, ch, R, d["cand"][which, ch])
            for p in rng.choice(R * R, 40, replace=False):
                Hp = Hm[p] - np.diag(Hm[p]).min() * np.eye(32)
                for comp in components(Hp):
                    if len(comp) < 2: continue
                    al, be = lanczos(Hp[np.ix_(comp, comp)])
                    if len(al) >= 2: comps.append((al, be))
    # a "normal regime" set: same structures with the couplings scaled down to O(1e-2..1) of the diagonal spread
    norm = []
    for al, be in comps[:3000]:
        k = len(al); a2 = rng.normal(size=k) * 1.0; b2 = np.abs(rng.normal(size=k - 1)) * 10.0 ** rng.uniform(-2, 0)
        norm.append((a2, b2))
    for name, cs in (("dumped wild components", comps), ("synthetic normal-regime tridiagonals", norm)):
        base = np.array([laguerre(al, be)[1] for al, be in cs]); stats = {}
        its = []; errs = []
        for al, be in cs:
            T = np.diag(al) + np.diag(be, 1) + np.diag(be, -1); tv = np.linalg.eigvalsh(T); sc = max(abs(tv).max(), 1e-300)
            lam, n = laguerre5(al, be, stats=stats); its.append(n); errs.append(abs(lam - tv[0]) / sc)
        its = np.array(its)
        ks = np.array([len(a) for a, b in cs])
        print(f"{name}: {len(cs)}; plain mean {base.mean():.2f} (k>=3: {base[ks >= 3].mean():.2f}); start at the 2x2 Ritz value: mean {its.mean():.2f} (k>=3: {its[ks >= 3].mean():.2f}) max {its.max()}, "
              f"max err {max(errs):.1e}, {stats}; hist {np.bincount(np.minimum(its, 40) // 5).tolist()}")

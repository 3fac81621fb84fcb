"""Prototype 4 (hybrid): plain Laguerre from the left for the first NP iterations; lanes still running switch to the
Sturm-count iteration of laguerre3 (multiplicity steps from either side); when that has converged the iterate is probed
PROBE * ||T|| to its left (Sturm count must be 0 there, otherwise restart from the Gershgorin bound) and the plain
iteration FINISHES from the probe point -- the final convergence is the old one (lowest copy, monotone from the left)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from laguerre3 import evalc

def plain_step(k, p1, d1, e1, xl):
    G = d1 / p1; E = e1 / p1
    disc = (k - 1.0) * ((k - 1.0) * G * G - k * E)
    sq = np.sqrt(disc) if disc > 0 else 0.0
    den = G - sq if G < 0 else G + sq
    return xl - k / den if den != 0 else xl

def laguerre4(al, be, NP=6, PROBE=1e-13, maxit=48, stats=None, trace=False):
    k = len(al)
    if k == 1: return al[0], 0
    bfull = np.concatenate([[0.0], np.abs(be), [0.0]])
    lo = (al - bfull[:-1] - bfull[1:]).min(); hi = al.min()
    tscale = max(abs(lo), abs(hi), np.abs(be).max())
    xstart = lo - (1e-3 * tscale + 1e-300)
    xl = xstart; sprev = 0.0; n = 0
    def plain(xl, sprev, n, limit):
        while n < limit:
            p1, d1, e1, c = evalc(al, be, xl); n += 1
            if p1 == 0.0: return xl, sprev, n, True
            xn = plain_step(k, p1, d1, e1, xl)
            if not (xn > xl): return xl, sprev, n, True
            st = xn - xl; tol = 4e-16 * max(abs(xn), abs(xl))
            done = st <= tol or 100.0 * st ** 4 <= tol * sprev ** 3
            sprev = st; xl = xn
            if done: return xl, sprev, n, True
        return xl, sprev, n, False
    xl, sprev, n, ok = plain(xl, sprev, n, NP)
    if ok: return xl, n
    # phase 2: Sturm-count iteration
    slow = True; plain_prev = False; left_prev = False; mdead = k < 3; tol = 4e-16 * tscale
    while n < maxit:
        p1, d1, e1, c = evalc(al, be, xl); n += 1
        left = c > 0
        if p1 == 0.0: xn = xl; st = 0.0; m2 = False
        else:
            G = d1 / p1; E = e1 / p1; G2 = G * G
            m2 = slow and not mdead and E >= 0.375 * G2
            lf = k - 1.0
            if m2:
                lf = 0.5 * k - 1.0
                if k >= 4 and E >= 0.615 * G2: lf = k / 3.0 - 1.0
                if k >= 5 and E >= 0.722 * G2: lf = 0.25 * k - 1.0
            disc = lf * ((k - 1.0) * G2 - k * E)
            sq = np.sqrt(disc) if disc > 0 else 0.0
            den = G - sq if G < 0 else G + sq
            if left != (den > 0):
                alt = G + sq if G < 0 else G - sq
                if left == (alt > 0) and alt != 0: den = alt
            xn = xl - k / den if den != 0 else xl
            st = abs(xn - xl)
        if trace: print(f"   p2 it {n} x={xl:.17g} c={c} m2={m2} st={st:.3e}")
        done = xn == xl or st <= tol or (not m2 and plain_prev and left == left_prev and 100.0 * st ** 4 <= tol * sprev ** 3)
        slow = st > 0.1 * sprev and sprev > 0
        plain_prev = not m2; left_prev = left; sprev = st; xl = xn
        if done: break
    # probe
    xp = xl - PROBE * tscale
    p1, d1, e1, c = evalc(al, be, xp); n += 1
    if c != 0:
        if stats is not None: stats["restart"] = stats.get("restart", 0) + 1
        xp = xstart
    # phase 3: plain finish (the probe evaluation doubles as its first iteration in the kernel; counted separately here)
    xl, sprev, n, ok = plain(xp, 0.0, n, maxit)
    return xl, n

if __name__ == "__main__":
    from laguerre_study import *
    from laguerre3 import laguerre3
    d = np.load(os.path.join(ROOT, "gpurun_out", "wild_dump.npz"))
    N, R = 8, 64; rng = np.random.default_rng(5)
    comps = []
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
    base = np.array([laguerre(al, be)[1] for al, be in comps])
    print(f"dumped components {len(comps)}: plain mean {base.mean():.2f} max {base.max()}, hist {np.bincount(np.minimum(base, 60) // 5).tolist()}")
    for NP in (4, 5, 6, 8):
        for PROBE in (1e-13, 1e-11):
            its = []; errs = []; stats = {}
            for al, be in comps:
                T = np.diag(al) + np.diag(be, 1) + np.diag(be, -1); tv = np.linalg.eigvalsh(T); sc = max(abs(tv).max(), 1e-300)
                lam, n = laguerre4(al, be, NP=NP, PROBE=PROBE, stats=stats); its.append(n); errs.append((lam - tv[0]) / sc)
            its = np.array(its); errs = np.array(errs)
            print(f"  NP {NP} probe {PROBE:.0e}: mean {its.mean():.2f} max {its.max()}, hist {np.bincount(np.minimum(its, 60) // 5).tolist()}, err min {errs.min():.1e} max {errs.max():.1e}, {stats}")

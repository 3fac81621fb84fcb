"""Prototype of the two-sided Laguerre iteration with multiplicity-2 steps (see qd_groundstate.h, step 6a)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

def evalp(al, be, x):
    k = len(al)
    p0, p1, d0, d1, e0, e1 = 1.0, al[0] - x, 0.0, -1.0, 0.0, 0.0
    for i in range(1, k):
        a_ = al[i] - x; b2 = be[i - 1] ** 2
        p0, p1, d0, d1, e0, e1 = p1, a_ * p1 - b2 * p0, d1, a_ * d1 - b2 * d0 - p1, e1, a_ * e1 - b2 * e0 - 2 * d1
        s = abs(p1)
        if s > 1e100 or (0 < s < 1e-100):
            f = 1e-100 if s > 1e100 else 1e100
            p0 *= f; p1 *= f; d0 *= f; d1 *= f; e0 *= f; e1 *= f
    return p0, p1, d1, e1

def laguerre2(al, be, maxit=48, slow_ratio=0.1, trace=False):
    k = len(al)
    if k == 1: return al[0], 0
    bfull = np.concatenate([[0.0], np.abs(be), [0.0]])
    lo = (al - bfull[:-1] - bfull[1:]).min(); hi = al.min()
    tscale = max(abs(lo), abs(hi), np.abs(be).max())
    xl = lo - (1e-3 * tscale + 1e-300); sprev = 0.0; slow = False; plain_prev = False; left_prev = False
    for it in range(maxit):
        p0, p1, d1, e1 = evalp(al, be, xl)
        if p1 == 0.0: return xl, it
        left = not (p1 > 0 and p0 > 0)                     # not (known to be) left of the spectrum: move left
        G = d1 / p1; E = e1 / p1; G2 = G * G
        m2 = slow and k >= 3 and E >= 0.375 * G2          # G^2 / (G^2 - E) >= 1.6: two or more roots dominate
        lf = k - 1.0
        if m2:
            lf = 0.5 * k - 1.0
            if k >= 4 and E >= 0.615 * G2: lf = k / 3.0 - 1.0      # >= 2.6: three
            if k >= 5 and E >= 0.722 * G2: lf = 0.25 * k - 1.0     # >= 3.6: four
        disc = lf * ((k - 1.0) * G2 - k * E)
        sq = np.sqrt(disc) if disc > 0 else 0.0
        den = G - sq if G < 0 else G + sq
        if left and not den > 0:
            alt = G + sq if G < 0 else G - sq
            if alt > 0: den = alt
        xn = xl - k / den if den != 0 else xl
        st = abs(xn - xl); tol = 4e-16 * max(abs(xn), abs(xl))
        if trace: print(f"  it {it} x={xl:.17g} left={left} m2={m2} st={st:.3e} p1={p1:.2e} p0={p0:.2e}")
        if xn == xl: return xl, it + 1
        # the cubic-convergence prediction needs two successive plain steps in the same direction
        done = st <= tol or (not m2 and plain_prev and left == left_prev and 100.0 * st ** 4 <= tol * sprev ** 3)
        plain_prev = not m2; left_prev = left
        slow = st > slow_ratio * sprev and sprev > 0
        sprev = st; xl = xn
        if done: return xl, it + 1
    return xl, maxit

if __name__ == "__main__":
    from laguerre_study import *
    rng = np.random.default_rng(5)
    # synthetic doublets: T with two nearly equal lowest eigenvalues, separation delta
    worst = 0; its = []
    for trial in range(4000):
        k = int(rng.integers(3, 9)); delta = 10.0 ** rng.uniform(-17, -1)
        ev = np.sort(rng.uniform(0.5, 3.0, k)); ev[0] = 0.5 - delta * 0.0; ev[1] = ev[0] + delta * (1 + abs(ev[0]))
        scale = 10.0 ** rng.uniform(-3, 12); ev = ev * scale * (1 if rng.random() < 0.5 else -1); ev.sort()
        if ev[0] > 0 or True:
            pass
        Q, _ = np.linalg.qr(rng.normal(size=(k, k))); A = (Q * ev) @ Q.T; A = 0.5 * (A + A.T)
        # tridiagonalise by Lanczos with full reorthogonalisation from a random vector
        q = rng.normal(size=k); q /= np.linalg.norm(q); Qs = [q]; al = []; be = []
        for j in range(k):
            w = A @ Qs[j]; a = Qs[j] @ w; al.append(a)
            for qq in Qs: w -= (qq @ w) * qq
            b = np.linalg.norm(w)
            if j + 1 < k:
                if b < 1e-14 * abs(ev).max(): break
                be.append(b); Qs.append(w / b)
        al = np.array(al); be = np.array(be)
        if len(al) < 2: continue
        T = np.diag(al) + np.diag(be, 1) + np.diag(be, -1); tv = np.linalg.eigvalsh(T); sc = abs(tv).max()
        lam, n = laguerre2(al, be); its.append(n)
        err = abs(lam - tv[0]) / sc
        if err > worst:
            worst = err; wcase = (al.copy(), be.copy(), tv.copy(), lam, n)
    its = np.array(its)
    print(f"synthetic doublets: {len(its)} cases, mean its {its.mean():.2f} max {its.max()}, worst |lam-ev0|/|T| {worst:.2e}")
    if worst > 1e-13:
        al, be, tv, lam, n = wcase; print("worst case eigs", tv[:3], "lam", lam, "its", n); laguerre2(al, be, trace=True)
    # the dumped wild-regime components
    d = np.load(os.path.join(ROOT, "gpurun_out", "wild_dump.npz"))
    N, R = 8, 64; a_its = []; b_its = []; ea = []; eb = []
    for which in range(4):
        dev = H.dev_view(N, d["params"][which]); sv = H.state_view(N, d["state"][which])
        for ch in range(N - 1):
            Hm, tc = hamiltonians(dev, sv, ch, R, d["cand"][which, ch])
            for p in rng.choice(R * R, 40, replace=False):
                Hp = Hm[p] - np.diag(Hm[p]).min() * np.eye(32)
                for comp in components(Hp):
                    if len(comp) < 2: continue
                    al, be = lanczos(Hp[np.ix_(comp, comp)])
                    if len(al) < 2: continue
                    T = np.diag(al) + np.diag(be, 1) + np.diag(be, -1); tv = np.linalg.eigvalsh(T); sc = max(abs(tv).max(), 1e-300)
                    la, ia = laguerre(al, be); lb, ib = laguerre2(al, be)
                    a_its.append(ia); b_its.append(ib); ea.append(abs(la - tv[0]) / sc); eb.append(abs(lb - tv[0]) / sc)
    a_its = np.array(a_its); b_its = np.array(b_its)
    print(f"dumped components {len(a_its)}: plain mean its {a_its.mean():.2f} max {a_its.max()}, two-sided {b_its.mean():.2f} max {b_its.max()}; "
          f"max err plain {max(ea):.1e}, two-sided {max(eb):.1e}")
    print("hist plain    ", np.bincount(np.minimum(a_its, 60) // 5).tolist()); print("hist two-sided", np.bincount(np.minimum(b_its, 60) // 5).tolist())

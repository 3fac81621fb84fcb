"""
Scalar NumPy prototype of the ground-state algorithm the HIP kernel uses
(NOT the oracle, NOT shipped): connected components of the hopping graph ->
Gershgorin pruning -> per-component Lanczos with full re-orthogonalisation from
the all-ones vector -> lowest eigenpair of the small tridiagonal by bisection +
inverse iteration -> pick the lowest component.  Used by tests to validate the
numerics of that algorithm against numpy.linalg.eigh before it is transcribed
to wave-level HIP code (csrc/qd_groundstate.hip follows this file step by step).
"""
import numpy as np


def build_neighbours(states, tc):
    """H_t as neighbour lists: for row i, list of (j, h_ij)."""
    M, N = states.shape
    nbr = [[] for _ in range(M)]
    for i in range(M):
        for j in range(M):
            d = states[j] - states[i]
            nz = np.nonzero(d)[0]
            if len(nz) != 2 or nz[1] != nz[0] + 1:
                continue
            a, b = nz
            if d[a] == -1 and d[b] == 1:        # forward hop a -> b
                h = -tc[a] * np.sqrt(states[i, a] * (states[i, b] + 1.0))
            elif d[a] == 1 and d[b] == -1:      # backward hop b -> a
                h = -tc[a] * np.sqrt(states[i, b] * (states[i, a] + 1.0))
            else:
                continue
            nbr[i].append((j, h))
    return nbr


def components(nbr):
    M = len(nbr)
    label = list(range(M))
    changed = True
    while changed:                               # min-label propagation (what the wave does)
        changed = False
        for i in range(M):
            m = min([label[i]] + [label[j] for j, _ in nbr[i]])
            if m < label[i]:
                label[i] = m; changed = True
    return np.array(label)


def sturm_count(alpha, beta, lam):
    """number of eigenvalues of tridiag(alpha,beta) below lam"""
    cnt = 0
    d = alpha[0] - lam
    if d == 0.0:
        d = -1e-300
    if d < 0:
        cnt += 1
    for i in range(1, len(alpha)):
        d = alpha[i] - lam - beta[i - 1] * beta[i - 1] / d
        if d == 0.0:
            d = -1e-300
        if d < 0:
            cnt += 1
    return cnt


def tridiag_lowest(alpha, beta, max_iter=200):
    k = len(alpha)
    if k == 1:
        return alpha[0], np.array([1.0])
    r = np.zeros(k)
    r[:-1] += np.abs(beta); r[1:] += np.abs(beta)
    lo = np.min(alpha - r); hi = np.min(alpha)           # lambda0 in [lo, hi]
    scale = max(abs(lo), abs(hi), np.max(np.abs(beta)))
    lo -= 1e-15 * scale + 1e-300
    for _ in range(max_iter):
        mid = 0.5 * (lo + hi)
        if mid <= lo or mid >= hi:
            break
        if sturm_count(alpha, beta, mid) >= 1:
            hi = mid
        else:
            lo = mid
    lam = hi
    # inverse iteration with the SPD factorisation at sigma = lo (count(lo) == 0)
    sig = lo
    d = np.zeros(k); l = np.zeros(k - 1)
    tiny = 1e-300 + 1e-18 * scale
    d[0] = alpha[0] - sig
    if d[0] < tiny:
        d[0] = tiny
    for i in range(1, k):
        l[i - 1] = beta[i - 1] / d[i - 1]
        d[i] = alpha[i] - sig - l[i - 1] * beta[i - 1]
        if d[i] < tiny:
            d[i] = tiny
    y = np.ones(k) / np.sqrt(k)
    y[0] = 1.0
    for _ in range(3):
        z = y.copy()
        for i in range(1, k):
            z[i] -= l[i - 1] * z[i - 1]
        z /= d
        for i in range(k - 2, -1, -1):
            z[i] -= l[i] * z[i + 1]
        y = z / np.sqrt(np.sum(z * z))
    return lam, y


def lanczos_component(F, nbr, idx, max_steps=None, tol=1e-13):
    """Lanczos with full (twice) re-orthogonalisation on the component `idx`
    starting from the all-ones vector.  Returns (lambda0, x on idx, steps)."""
    s = len(idx)
    pos = {g: i for i, g in enumerate(idx)}
    rows = [[(pos[j], h) for j, h in nbr[g]] for g in idx]
    diag = F[idx]

    def matvec(v):
        w = diag * v
        for i in range(s):
            for j, h in rows[i]:
                w[i] += h * v[j]
        return w

    if s == 1:
        return diag[0], np.array([1.0]), 0
    Q = []
    alpha = []; beta = []
    q = np.ones(s) / np.sqrt(s)
    scale = 0.0
    for j in range(s if max_steps is None else min(s, max_steps)):
        Q.append(q)
        w = matvec(q)
        a = float(q @ w)
        alpha.append(a)
        for _ in range(2):
            for qi in Q:
                w = w - (qi @ w) * qi
        b = float(np.sqrt(w @ w))
        scale = max(scale, abs(a), b)
        if j == s - 1 or b <= tol * scale:
            break
        beta.append(b)
        q = w / b
    lam, y = tridiag_lowest(np.array(alpha), np.array(beta))
    x = np.zeros(s)
    for yi, qi in zip(y, Q):
        x += yi * qi
    x /= np.sqrt(x @ x)
    return lam, x, len(alpha)


def ground_state(F, states, tc, prune=True, stats=None):
    """<n> of the ground state of diag(F) + H_t(tc, states)."""
    M, N = states.shape
    nbr = build_neighbours(states, tc)
    label = components(nbr)
    labs = np.unique(label)
    radius = np.array([sum(abs(h) for _, h in nbr[i]) for i in range(M)])
    upper = np.min(F)                       # lambda0 <= min diagonal
    best = (np.inf, None, None)
    nsolved = 0; maxsteps = 0
    for L in labs:
        idx = np.nonzero(label == L)[0]
        if prune and np.min(F[idx] - radius[idx]) > upper:
            continue                        # Gershgorin: cannot host the ground state
        lam, x, steps = lanczos_component(F, nbr, list(idx))
        nsolved += len(idx) > 1; maxsteps = max(maxsteps, steps)
        if lam < best[0]:
            best = (lam, idx, x)
    lam, idx, x = best
    if stats is not None:
        stats.append((len(labs), nsolved, maxsteps, len(idx)))
    p = x * x
    return p @ states[idx].astype(float), lam

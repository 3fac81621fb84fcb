"""Offline study of the tridiagonal stage on wild-regime pixels (gpurun_out/wild_dump.npz from scripts/dump_wild.py):
mimics the kernel's plain Lanczos from the all-ones vector per hop component, then counts Laguerre iterations and
looks at the spectrum of T."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "rl-agent-for-qubit-array-tuning_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import helpers as H, qd_oracle as O

def hamiltonians(dev, sv, ch, R, states):
    N = dev.n_dot
    vg = O.sweep_voltages(sv.vgm, dev.origin, sv.gate_v, sv.sensor_gt, ch, -dev.window, dev.window, R)
    vb = np.broadcast_to(np.asarray(sv.barrier_v, float), (R * R, N - 1))
    v_ext = np.concatenate([vg, vb], axis=1)
    F = O.free_energy_states(v_ext, dev.cdd_inv_full, dev.cgd_full, states, N)
    tc = O.tunnel_couplings(O.effective_barrier_potential(vg, vb, dev.Cbg, dev.Cbb), dev.tc_base, dev.alpha)
    return F[:, :, None] * np.eye(states.shape[1]) + O.tunnel_hamiltonian(tc, states), tc

def components(Hm):
    n = Hm.shape[0]; A = (Hm != 0) | np.eye(n, dtype=bool)
    seen = np.zeros(n, bool); out = []
    for i in range(n):
        if seen[i]: continue
        comp = {i}; fr = [i]
        while fr:
            j = fr.pop()
            for t in np.nonzero(A[j])[0]:
                if t not in comp: comp.add(int(t)); fr.append(int(t))
        comp = sorted(comp); seen[comp] = True; out.append(comp)
    return out

def lanczos(Hc):
    n = Hc.shape[0]
    q = np.ones(n) / np.sqrt(n); qp = np.zeros(n); bp = 0.0; anorm = 0.0
    al = []; be = []
    for j in range(n):
        w = Hc @ q
        a = q @ w
        w = w - a * q - bp * qp
        b = np.sqrt(w @ w)
        anorm = max(anorm, abs(a), b)
        al.append(a); be.append(b)
        if j + 1 >= n or not (b > 1e-13 * anorm):
            break
        qp, bp, q = q, b, w / b
    be[-1] = 0.0
    return np.array(al), np.array(be[:-1])

def laguerre(al, be, maxit=60):
    k = len(al)
    if k == 1: return al[0], 0
    bfull = np.concatenate([[0.0], np.abs(be), [0.0]])
    lo = (al - bfull[:-1] - bfull[1:]).min(); hi = al.min()
    tscale = max(abs(lo), abs(hi), np.abs(be).max())
    xl = lo - (1e-3 * tscale + 1e-300); sprev = 0.0
    for it in range(maxit):
        p0, p1, d0, d1, e0, e1 = 1.0, al[0] - xl, 0.0, -1.0, 0.0, 0.0
        for i in range(1, k):
            a_ = al[i] - xl; b2 = be[i - 1] ** 2
            p2 = a_ * p1 - b2 * p0; d2 = a_ * d1 - b2 * d0 - p1; e2 = a_ * e1 - b2 * e0 - 2 * d1
            p0, p1, d0, d1, e0, e1 = p1, p2, d1, d2, e1, e2
            s = abs(p1)
            if s > 1e100 or (0 < s < 1e-100):
                f = 1e-100 if s > 1e100 else 1e100
                p0 *= f; p1 *= f; d0 *= f; d1 *= f; e0 *= f; e1 *= f
        if p1 == 0.0: return xl, it
        G = d1 / p1; E = e1 / p1
        disc = (k - 1.0) * ((k - 1.0) * G * G - k * E)
        sq = np.sqrt(disc) if disc > 0 else 0.0
        den = G - sq if G < 0 else G + sq
        xn = xl - k / den if den != 0 else xl
        if not (xn > xl): return xl, it + 1
        st = xn - xl; tol = 4e-16 * max(abs(xn), abs(xl))
        done = st <= tol or 100.0 * st ** 4 <= tol * sprev ** 3
        sprev = st; xl = xn
        if done: return xl, it + 1
    return xl, maxit

if __name__ == "__main__":
    d = np.load(os.path.join(ROOT, "gpurun_out", "wild_dump.npz"))
    N, R = 8, 64
    which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    npx = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    dev = H.dev_view(N, d["params"][which]); sv = H.state_view(N, d["state"][which])
    rng = np.random.default_rng(0)
    its_all = []; rows = []
    for ch in range(N - 1):
        states = d["cand"][which, ch]
        Hm, tc = hamiltonians(dev, sv, ch, R, states)
        for p in rng.choice(R * R, npx // (N - 1), replace=False):
            Hp = Hm[p]; Fm = np.diag(Hp).min()
            Hp = Hp - Fm * np.eye(32)
            for comp in components(Hp):
                if len(comp) < 2: continue
                Hc = Hp[np.ix_(comp, comp)]
                al, be = lanczos(Hc)
                lam, its = laguerre(al, be)
                ev = np.linalg.eigvalsh(Hc); 
                T = np.diag(al) + np.diag(be, 1) + np.diag(be, -1); tv = np.linalg.eigvalsh(T)
                its_all.append(its)
                rows.append((its, len(comp), len(al), tc[p].max(), (tv[1] - tv[0]) / max(abs(tv).max(), 1e-300) if len(tv) > 1 else 1.0,
                             (ev[1] - ev[0]) / max(abs(ev).max(), 1e-300), abs(lam - ev[0]) / max(abs(ev).max(), 1e-300), (np.abs(be).min() if len(be) else 0.0) / max(abs(tv).max(), 1e-300)))
    its_all = np.array(its_all)
    print("components solved", len(its_all), "mean its %.2f" % its_all.mean(), "hist", np.bincount(np.minimum(its_all, 60) // 5).tolist())
    rows.sort(key=lambda r: -r[0])
    print("its size k tcmax relgap(T) relgap(Hc) |lam-ev0|/|Hc| min beta/|T|")
    for r in rows[:25]: print("%3d %3d %3d %.1e %.1e %.1e %.1e %.1e" % r)
    print("...")
    for r in rows[len(rows) // 2: len(rows) // 2 + 5]: print("%3d %3d %3d %.1e %.1e %.1e %.1e %.1e" % r)

def truncation_table(which=1, npx=200, seed=0):
    """For every solved component: per truncation j the Ritz residual estimate beta_j |s_j| / ||H||, the true angle between
    the Ritz vector and the component's ground vector, and the relative gap of the component."""
    d = np.load(os.path.join(ROOT, "gpurun_out", "wild_dump.npz"))
    N, R = 8, 64
    dev = H.dev_view(N, d["params"][which]); sv = H.state_view(N, d["state"][which])
    rng = np.random.default_rng(seed)
    out = []
    for ch in range(N - 1):
        states = d["cand"][which, ch]
        Hm, tc = hamiltonians(dev, sv, ch, R, states)
        for p in rng.choice(R * R, npx // (N - 1), replace=False):
            Hp = Hm[p]; hn = np.abs(Hp).sum(axis=1).max()
            Hp = Hp - np.diag(Hp).min() * np.eye(32)
            for comp in components(Hp):
                if len(comp) < 3: continue
                Hc = Hp[np.ix_(comp, comp)]
                ev, V = np.linalg.eigh(Hc); v1 = V[:, 0]
                # Lanczos with basis kept
                n = len(comp); q = np.ones(n) / np.sqrt(n); qp = np.zeros(n); bp = 0.0; Q = []; al = []; be = []; anorm = 0
                for j in range(n):
                    Q.append(q); w = Hc @ q; a = q @ w; w = w - a * q - bp * qp; b = np.sqrt(w @ w); al.append(a); be.append(b)
                    anorm = max(anorm, abs(a), b)
                    if j + 1 >= n or not (b > 1e-13 * anorm): break
                    qp, bp, q = q, b, w / b
                k = len(al); Qm = np.array(Q).T
                rowsj = []
                for j in range(1, k + 1):
                    T = np.diag(al[:j]) + np.diag(be[:j - 1], 1) + np.diag(be[:j - 1], -1)
                    tv, Y = np.linalg.eigh(T); y = Y[:, 0]
                    x = Qm[:, :j] @ y; x /= np.linalg.norm(x)
                    ang = np.sqrt(max(0.0, 1 - min(1.0, abs(x @ v1)) ** 2))
                    rho = (be[j - 1] if j < k else 0.0) * abs(y[-1])
                    rowsj.append((j, rho / hn, ang, abs(tv[0] - ev[0]) / hn))
                out.append(dict(size=n, k=k, relgap=(ev[1] - ev[0]) / hn, tc=tc[p].max(), rows=rowsj))
    return out

if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "trunc":
    tab = truncation_table(which, npx)
    # for thresholds: first j with rho/hn <= thr; report resulting angle error distribution and saved steps
    for thr in (1e-13, 1e-14, 1e-15, 1e-16):
        angs = []; saved = []; kk = []
        for c in tab:
            jsel = next(j for (j, rho, ang, le) in c["rows"] if rho <= thr)
            angs.append(c["rows"][jsel - 1][2]); saved.append(c["k"] - jsel); kk.append(c["k"])
        angs = np.array(angs); saved = np.array(saved)
        full = np.array([c["rows"][-1][2] for c in tab])
        print(f"thr {thr:.0e}: comps {len(tab)}, truncated in {np.mean(saved > 0):.2f}, mean k {np.mean(kk):.2f} -> {np.mean(np.array(kk) - saved):.2f}, "
              f"max angle err {angs.max():.1e} (full T: {full.max():.1e}), 99.9pct {np.quantile(angs, 0.999):.1e} (full {np.quantile(full, 0.999):.1e})")
    worst = sorted(tab, key=lambda c: -c["rows"][-1][2])[:5]
    for c in worst:
        print("size", c["size"], "k", c["k"], "relgap %.1e tc %.1e" % (c["relgap"], c["tc"]))
        for r in c["rows"]: print("    j %2d rho/|H| %.1e angle %.1e lamerr %.1e" % r)

def laguerre_m(al, be, maxit=60):
    """laguerre() with the multiplicity-2 step of the kernel (slow flag + local multiplicity estimate + overshoot check)."""
    k = len(al)
    if k == 1: return al[0], 0
    bfull = np.concatenate([[0.0], np.abs(be), [0.0]])
    lo = (al - bfull[:-1] - bfull[1:]).min(); hi = al.min()
    tscale = max(abs(lo), abs(hi), np.abs(be).max())
    xl = lo - (1e-3 * tscale + 1e-300); sprev = 0.0
    slow = False; mstep = False; mdead = k < 3; xback = xl
    for it in range(maxit):
        p0, p1, d0, d1, e0, e1 = 1.0, al[0] - xl, 0.0, -1.0, 0.0, 0.0
        for i in range(1, k):
            a_ = al[i] - xl; b2 = be[i - 1] ** 2
            p2 = a_ * p1 - b2 * p0; d2 = a_ * d1 - b2 * d0 - p1; e2 = a_ * e1 - b2 * e0 - 2 * d1
            p0, p1, d0, d1, e0, e1 = p1, p2, d1, d2, e1, e2
            s = abs(p1)
            if s > 1e100 or (0 < s < 1e-100):
                f = 1e-100 if s > 1e100 else 1e100
                p0 *= f; p1 *= f; d0 *= f; d1 *= f; e0 *= f; e1 *= f
        if mstep and not (p1 > 0 and p0 > 0):
            xl = xback; mstep = False; mdead = True; slow = False; sprev = 0.0
            continue
        if p1 == 0.0: return xl, it
        G = d1 / p1; E = e1 / p1; G2 = G * G
        mstep = slow and not mdead and E >= 0.375 * G2
        lf = 0.5 * k - 1.0 if mstep else k - 1.0
        disc = lf * ((k - 1.0) * G2 - k * E)
        xback = xl
        sq = np.sqrt(disc) if disc > 0 else 0.0
        den = G - sq if G < 0 else G + sq
        xn = xl - k / den if den != 0 else xl
        if not (xn > xl): return xl, it + 1
        st = xn - xl; tol = 4e-16 * max(abs(xn), abs(xl))
        done = st <= tol or (not mstep and 100.0 * st ** 4 <= tol * sprev ** 3)
        slow = st > 0.1 * sprev and sprev > 0
        sprev = st; xl = xn
        if done: return xl, it + 1
    return xl, maxit

if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "mstep":
    d = np.load(os.path.join(ROOT, "gpurun_out", "wild_dump.npz"))
    N, R = 8, 64
    rng = np.random.default_rng(3)
    a_its = []; b_its = []; errs_a = []; errs_b = []; above = 0
    for which in range(4):
        dev = H.dev_view(N, d["params"][which]); sv = H.state_view(N, d["state"][which])
        for ch in range(N - 1):
            Hm, tc = hamiltonians(dev, sv, ch, R, d["cand"][which, ch])
            for p in rng.choice(R * R, npx // 28, replace=False):
                Hp = Hm[p] - np.diag(Hm[p]).min() * np.eye(32)
                for comp in components(Hp):
                    if len(comp) < 2: continue
                    al, be = lanczos(Hp[np.ix_(comp, comp)])
                    if len(al) < 2: continue
                    T = np.diag(al) + np.diag(be, 1) + np.diag(be, -1); tv = np.linalg.eigvalsh(T); sc = max(abs(tv).max(), 1e-300)
                    la, ia = laguerre(al, be); lb, ib = laguerre_m(al, be)
                    a_its.append(ia); b_its.append(ib); errs_a.append(abs(la - tv[0]) / sc); errs_b.append(abs(lb - tv[0]) / sc)
                    above += (lb - tv[0]) / sc > 1e-13
    a_its = np.array(a_its); b_its = np.array(b_its)
    print(f"components {len(a_its)}: plain mean its {a_its.mean():.2f} max {a_its.max()}, with m-step {b_its.mean():.2f} max {b_its.max()}; "
          f"max |lam - eig|/|T| plain {max(errs_a):.1e}, m-step {max(errs_b):.1e}; above the root by > 1e-13: {above}")
    print("hist plain ", np.bincount(np.minimum(a_its, 60) // 5).tolist()); print("hist m-step", np.bincount(np.minimum(b_its, 60) // 5).tolist())

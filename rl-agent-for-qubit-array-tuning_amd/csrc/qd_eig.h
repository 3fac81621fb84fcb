// qd_eig.h -- lowest eigenpair of ONE small dense real symmetric matrix PER LANE (reference:
// ground_state.py:149-162 calls a dense eigh on the 32x32 Hamiltonian and keeps column 0 only).
//
// H = diag(F) + H_t is block diagonal over the connected components of the hop graph (qd_groundstate.h); every
// component of 2..32 states that can hold the ground state becomes one TASK: its dense block, lower triangle packed
// row-major, A[i(i+1)/2 + j], j <= i.  A lane solves a whole task by itself, so the 64 lanes of a wavefront do 64
// different solves (round 2 ran the same serial recurrences redundantly in every member lane of a component):
//   1. scale by a power of two so that ||A||_inf is in [1, 2)         (exact; the couplings span 1e-22 .. 1e44)
//   2. Householder tridiagonalisation  Q^T A Q = T                     (backward stable, no Krylov ghosts: this is what
//      LAPACK's eigh does to the whole matrix, applied block by block)
//   3. lowest eigenvalue of T: Laguerre's iteration from the left of the spectrum (monotone, cubic at a simple root)
//   4. eigenvector of T from the twisted factorisation of T - lambda (one solve, accurate in every entry)
//   5. x = Q y, normalised.
// The same source compiles for the host (tests/hosttest) where it is checked against numpy.linalg.eigh.
// Sizes 2..QD_EIG_REG run on register arrays with every loop unrolled (template <S>; 9 and 11 padded to 10 and 12); larger
// blocks (0.1 % of the pixels have one) run the same algorithm with run-time loops on the task's record in memory.
#pragma once
#include <math.h>
#include "qd_common.h"

#define QD_EIG_REG 12          // largest block solved in registers (sizes 9, 11 padded to 10, 12)

#if defined(__HIP_DEVICE_COMPILE__)
#define QD_E_ANY(p) (__any((p)) != 0)
// v_rcp_f64 / v_rsq_f64 deliver 4.6e-8 / 5.2e-8 relative accuracy on gfx950, one Newton step 2e-15 / 4e-15, two steps
// 1.1e-16 / 2.4e-16 (measured in round 2).  One step is enough where the result only steers an iteration.
QD_HD double qd_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
QD_HD double qd_rcp1(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
QD_HD double qd_sqrt1(double x) {           // sqrt(x), x > 0, ~1 ulp
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    const double s = x * y;
    return fma(fma(-s, s, x), 0.5 * y, s);
}
// sqrt(x) and 1/sqrt(x) from one v_rsq_f64 + two Newton steps (x > 0); ~1 ulp
QD_HD void qd_sqrt_rsqrt(double x, double& s, double& r) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    r = y;
    s = x * y;
    s = fma(fma(-s, s, x), 0.5 * y, s);
}
#else
#define QD_E_ANY(p) (p)
QD_HD double qd_rcp(double x) { return 1.0 / x; }
QD_HD double qd_rcp1(double x) { return 1.0 / x; }
QD_HD double qd_sqrt1(double x) { return sqrt(x); }
QD_HD void qd_sqrt_rsqrt(double x, double& s, double& r) { s = sqrt(x); r = 1.0 / s; }
#endif

// 2^-E and 2^E with E the binary exponent of v (1.0, 1.0 for zero / denormal / inf)
QD_HD void qd_pow2_scale(double v, double& down, double& up) {
    down = 1.0; up = 1.0;
    union { double d; unsigned long long u; } c; c.d = v;
    const unsigned long long ef = (c.u >> 52) & 0x7FFull;
    if (ef != 0ull && ef < 2046ull) {
        c.u = (2046ull - ef) << 52; down = c.d;
        c.u = ef << 52; up = c.d;
    }
}

// One Laguerre step of p(x) = det(T - x) given p, p', p'' at xl (degree dk); returns false when converged or stalled.
// The iterate needs no correctly rounded sqrt / quotient (the fixed point does not depend on them).  Converged = the
// step no longer moves the iterate by more than 4e-16 relative (with an absolute floor of 2e-17 ||T||: the shift of the
// inverse iteration sits 2e-16 ||T|| below).  Round 2 also stopped one iteration early on a cubic-convergence
// prediction from the last two steps; next to a close second root the prediction is wrong (eigen residuals of 1e-9
// instead of 1e-16 in a few pixels), and with one task per lane the extra iteration costs next to nothing.
QD_HD bool qd_laguerre_step(double dk, double p1, double d1, double e1, double tscale, double& xl) {
    if (p1 == 0.0) return false;
    const double ip = qd_rcp1(p1);
    const double G = d1 * ip, E = e1 * ip;
    const double disc = (dk - 1.0) * ((dk - 1.0) * G * G - dk * E);
    double sq = 0.0;
    if (disc > 0.0) sq = qd_sqrt1(disc);
    const double den = (G < 0.0) ? G - sq : G + sq;
    const double xn = (den != 0.0) ? fma(-dk, qd_rcp1(den), xl) : xl;
    if (!(xn > xl)) return false;                          // the monotone sequence has stalled
    const double st = xn - xl, tol = 4e-16 * fmax(fmax(fabs(xn), fabs(xl)), 0.05 * tscale);
    xl = xn;
    return st > tol;
}

#define QD_EIG_MAXIT 64

// ---------------------------------------------------------------------------------------------------------------
// Register version.  Ain: packed lower triangle (sz (sz+1) / 2 doubles, any memory), sz <= S.  Outputs: lam (same units as
// Ain), x[S] (unit 2-norm; zero beyond sz), and with RESID the absolute residual ||A x - lam x||_2.  `iters` (optional)
// returns the number of Laguerre iterations this lane needed (statistics).
// ---------------------------------------------------------------------------------------------------------------
template <int S, bool RESID>
QD_HD void qd_eig_lowest(const double* Ain, double& lam_out, double* x, double& resid_out,
                         int* iters = nullptr, int sz = S) {
    constexpr int NE = S * (S + 1) / 2;
#define QD_IX(i, j) ((i) * ((i) + 1) / 2 + (j))
    // sz <= S: a smaller block is PADDED to S rows with decoupled states (after the scaling: diagonal 4 > ||A||, zero
    // couplings), which the reflectors leave alone and whose eigenvalues lie to the right of every real one
    double a[NE];
#if defined(__HIP_DEVICE_COMPILE__)
    if (sz == S) {
        // task records are 16-byte aligned: pairs of doubles per load (half the address work of the per-lane strided loads)
        const double2* A2 = reinterpret_cast<const double2*>(Ain);
#pragma unroll
        for (int e = 0; e + 1 < NE; e += 2) { const double2 v = A2[e >> 1]; a[e] = v.x; a[e + 1] = v.y; }
        if (NE & 1) a[NE - 1] = Ain[NE - 1];
    } else
#endif
    {
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) a[QD_IX(i, j)] = (i < sz) ? Ain[QD_IX(i, j)] : 0.0;
    }
    // ---- 1. scale ----
    double anorm = 0.0;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        double rs = 0.0;
#pragma unroll
        for (int j = 0; j < S; ++j) rs += fabs(a[j <= i ? QD_IX(i, j) : QD_IX(j, i)]);
        anorm = fmax(anorm, rs);
    }
    double tsc, tusc;
    qd_pow2_scale(anorm, tsc, tusc);
#pragma unroll
    for (int e = 0; e < NE; ++e) a[e] *= tsc;
#pragma unroll
    for (int i = 0; i < S; ++i) if (i >= sz) a[QD_IX(i, i)] = 4.0;
    // ---- 2. Householder: reflector k zeroes column k below the sub-diagonal; v_k (v_k[k+1] = 1) is kept in the
    // zeroed entries, tau_k beside it.  A block that is tridiagonal already (sigma == 0) is left alone. ----
    double al[S], be[S], tau[S];
#pragma unroll
    for (int k = 0; k + 2 < S; ++k) {
        double sigma = 0.0;
#pragma unroll
        for (int i = k + 2; i < S; ++i) sigma = fma(a[QD_IX(i, k)], a[QD_IX(i, k)], sigma);
        const double x0 = a[QD_IX(k + 1, k)];
        // ||A|| is in [1, 2): a column tail below 1e-100, or below 1e-17 |x0|, changes nothing that float64 can see, and
        // is dropped (zeroed) instead of reflected -- its reflector would need v0^2 ~ sigma^2 / x0^2, which underflows
        // (tc of different barriers differ by up to 60 decades inside one block: NaN in a handful of pixels at tc ~ 1e45)
        const bool refl = (sigma > 1e-200) & (sigma > 1e-34 * x0 * x0);
        const double mu = qd_sqrt1(fma(x0, x0, sigma) + 1e-300);
        // v0 = x0 - mu without cancellation
        const double v0 = (x0 <= 0.0) ? x0 - mu : -sigma * qd_rcp(x0 + mu);
        const double v0sq = v0 * v0;
        const double t = refl ? 2.0 * v0sq * qd_rcp(sigma + v0sq) : 0.0;
        const double iv0 = refl ? qd_rcp(v0) : 0.0;
#pragma unroll
        for (int i = k + 2; i < S; ++i) a[QD_IX(i, k)] *= iv0;            // v_i (v_{k+1} = 1 implied)
        tau[k] = t;
        be[k] = refl ? mu : x0;
        al[k] = a[QD_IX(k, k)];
        // trailing block B = a[k+1.., k+1..]:  p = t B v,  K = t/2 p.v,  w = p - K v,  B -= v w^T + w v^T
        double p[S], w[S];
        double pv = 0.0;
#pragma unroll
        for (int i = k + 1; i < S; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = k + 1; j < S; ++j) {
                const double vj = (j == k + 1) ? 1.0 : a[QD_IX(j, k)];
                acc = fma(a[j <= i ? QD_IX(i, j) : QD_IX(j, i)], vj, acc);
            }
            p[i] = t * acc;
            const double vi = (i == k + 1) ? 1.0 : a[QD_IX(i, k)];
            pv = fma(p[i], vi, pv);
        }
        const double K = 0.5 * t * pv;
#pragma unroll
        for (int i = k + 1; i < S; ++i) {
            const double vi = (i == k + 1) ? 1.0 : a[QD_IX(i, k)];
            w[i] = fma(-K, vi, p[i]);
        }
#pragma unroll
        for (int i = k + 1; i < S; ++i) {
            const double vi = (i == k + 1) ? 1.0 : a[QD_IX(i, k)];
#pragma unroll
            for (int j = k + 1; j <= i; ++j) {
                const double vj = (j == k + 1) ? 1.0 : a[QD_IX(j, k)];
                a[QD_IX(i, j)] = fma(-vi, w[j], fma(-w[i], vj, a[QD_IX(i, j)]));
            }
        }
    }
    if (S >= 2) { al[S - 2] = a[QD_IX(S - 2, S - 2)]; be[S - 2] = a[QD_IX(S - 1, S - 2)]; }
    al[S - 1] = a[QD_IX(S - 1, S - 1)]; be[S - 1] = 0.0;
    // ---- 3. lowest eigenvalue of T ----
    double lo = INFINITY, tscale = 0.0;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const double rad = (i > 0 ? fabs(be[i - 1]) : 0.0) + fabs(be[i]);
        lo = fmin(lo, al[i] - rad);
        tscale = fmax(tscale, fabs(al[i]) + rad);
    }
    double xl = lo - (1e-3 * tscale + 1e-300);
    {
        bool more = true;
        int myit = 0;
        for (int it = 0; it < QD_EIG_MAXIT; ++it) {
            if (!QD_E_ANY(more)) break;
            // p, p', p'' at xl by the three-term recurrences of the leading minors (|entries| <= ~4: no rescaling needed)
            double p0 = 1.0, p1 = al[0] - xl, d0 = 0.0, d1 = -1.0, e0 = 0.0, e1 = 0.0;
#pragma unroll
            for (int i = 1; i < S; ++i) {
                const double a_ = al[i] - xl, b2 = be[i - 1] * be[i - 1];
                const double p2 = fma(a_, p1, -(b2 * p0));
                const double d2 = fma(a_, d1, -(b2 * d0)) - p1;
                const double e2 = fma(a_, e1, -(b2 * e0)) - 2.0 * d1;
                p0 = p1; p1 = p2; d0 = d1; d1 = d2; e0 = e1; e1 = e2;
            }
            if (more) { more = qd_laguerre_step((double)S, p1, d1, e1, tscale, xl); ++myit; }
        }
        if (iters) *iters = myit;
    }
    const double lam = xl;
    // ---- 4. eigenvector of T by the twisted factorisation of T - lam (Parlett / Dhillon, LAPACK dlar1v):
    // forward pivots d+ (top down), backward pivots d- (bottom up), gamma_k = d+_k + d-_k - (alpha_k - lam); with
    // z_k = 1 and the two recurrences run outwards from k, (T - lam) z = gamma_k e_k, so the k with the smallest
    // |gamma_k| gives the eigenvector to working accuracy in ONE solve, whatever the size of its entries.  (Round 2
    // and the first version of this file ran inverse iteration on the top-down LDL^T alone: when the eigenvector's
    // last entry is small -- 3e-7 in the pixel that showed it -- the near-singular pivot does not appear where the
    // elimination ends and entries come out wrong by eps / |y_last|: eigen residuals of 1e-9 instead of 1e-16.) ----
    double y[S];
    {
        const double pivmin = 2.3e-16 * tscale + 1e-300;
        double um[S], lp[S];
        double dm = al[S - 1] - lam;
        double dmk[S];
        dmk[S - 1] = dm;
#pragma unroll
        for (int i = S - 2; i >= 0; --i) {
            if (fabs(dm) < pivmin) dm = -pivmin;
            um[i] = be[i] * qd_rcp(dm);
            dm = (al[i] - lam) - um[i] * be[i];
            dmk[i] = dm;
        }
        double dp = al[0] - lam;
        double gbest = fabs(dmk[0]);                       // gamma_0 = d-_0 (d+_0 = alpha_0 - lam)
        int kbest = 0;
#pragma unroll
        for (int i = 0; i + 1 < S; ++i) {
            if (fabs(dp) < pivmin) dp = -pivmin;
            lp[i] = be[i] * qd_rcp(dp);
            dp = (al[i + 1] - lam) - lp[i] * be[i];
            const double g = fabs(dp + dmk[i + 1] - (al[i + 1] - lam));
            if (g < gbest) { gbest = g; kbest = i + 1; }
        }
#pragma unroll
        for (int i = 0; i < S; ++i) y[i] = (i == kbest) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 1; i < S; ++i) if (i > kbest) y[i] = -um[i - 1] * y[i - 1];      // downwards from the twist
#pragma unroll
        for (int i = S - 2; i >= 0; --i) if (i < kbest) y[i] = -lp[i] * y[i + 1];      // upwards from the twist
    }
    // ---- 5. x = Q y = H_0 (H_1 (.. y)) ----
#pragma unroll
    for (int k = S - 3; k >= 0; --k) {
        double dot = y[k + 1];
#pragma unroll
        for (int i = k + 2; i < S; ++i) dot = fma(a[QD_IX(i, k)], y[i], dot);
        const double f = tau[k] * dot;
        y[k + 1] -= f;
#pragma unroll
        for (int i = k + 2; i < S; ++i) y[i] = fma(-f, a[QD_IX(i, k)], y[i]);
    }
    {
        double nrm = 0.0;
#pragma unroll
        for (int i = 0; i < S; ++i) nrm = fma(y[i], y[i], nrm);
        double sn = 0.0, inv = 1.0;
        if (nrm > 0.0 && nrm < INFINITY) qd_sqrt_rsqrt(nrm, sn, inv);
#pragma unroll
        for (int i = 0; i < S; ++i) y[i] *= inv;
    }
    lam_out = lam * tusc;
    resid_out = 0.0;
    if constexpr (RESID) {
        double r2 = 0.0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            double acc = -lam_out * y[i];
#pragma unroll
            for (int j = 0; j < S; ++j) if (i < sz && j < sz) acc = fma(Ain[j <= i ? QD_IX(i, j) : QD_IX(j, i)], y[j], acc);
            if (i < sz) r2 = fma(acc, acc, r2);
        }
        resid_out = sqrt(r2);
    }
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = y[i];
#undef QD_IX
}

// ---------------------------------------------------------------------------------------------------------------
// Memory version for blocks of QD_EIG_REG < s <= 32 states: the same algorithm with run-time loops, IN PLACE on the
// task's packed matrix M (destroyed: it ends up holding the reflectors) and a workspace of 4 s doubles.  When the
// residual is wanted the caller keeps a copy of the matrix (Morig, may be null).
// ---------------------------------------------------------------------------------------------------------------
QD_HD void qd_eig_lowest_mem(int s, double* M, double* work, const double* Morig,
                             double& lam_out, double& resid_out, int* iters = nullptr) {
#define QD_IX(i, j) ((i) * ((i) + 1) / 2 + (j))
#define QD_SYM(i, j) M[(j) <= (i) ? QD_IX(i, j) : QD_IX(j, i)]
    double* al = work; double* be = work + s; double* tau = work + 2 * s; double* y = work + 3 * s;
    const int ne = s * (s + 1) / 2;
    double anorm = 0.0;
    for (int i = 0; i < s; ++i) {
        double rs = 0.0;
        for (int j = 0; j < s; ++j) rs += fabs(QD_SYM(i, j));
        anorm = fmax(anorm, rs);
    }
    double tsc, tusc;
    qd_pow2_scale(anorm, tsc, tusc);
    for (int e = 0; e < ne; ++e) M[e] *= tsc;
    for (int k = 0; k + 2 < s; ++k) {
        double sigma = 0.0;
        for (int i = k + 2; i < s; ++i) sigma = fma(M[QD_IX(i, k)], M[QD_IX(i, k)], sigma);
        const double x0 = M[QD_IX(k + 1, k)];
        const bool refl = (sigma > 1e-200) & (sigma > 1e-34 * x0 * x0);   // (see the register version)
        const double mu = qd_sqrt1(fma(x0, x0, sigma) + 1e-300);
        const double v0 = (x0 <= 0.0) ? x0 - mu : -sigma * qd_rcp(x0 + mu);
        const double v0sq = v0 * v0;
        const double t = refl ? 2.0 * v0sq * qd_rcp(sigma + v0sq) : 0.0;
        const double iv0 = refl ? qd_rcp(v0) : 0.0;
        for (int i = k + 2; i < s; ++i) M[QD_IX(i, k)] *= iv0;
        tau[k] = t;
        be[k] = refl ? mu : x0;
        al[k] = M[QD_IX(k, k)];
        if (refl) {
            // p (kept in y[k+1..]) = t B v
            double pv = 0.0;
            for (int i = k + 1; i < s; ++i) {
                double acc = 0.0;
                for (int j = k + 1; j < s; ++j) {
                    const double vj = (j == k + 1) ? 1.0 : M[QD_IX(j, k)];
                    acc = fma(QD_SYM(i, j), vj, acc);
                }
                const double pi = t * acc;
                y[i] = pi;
                const double vi = (i == k + 1) ? 1.0 : M[QD_IX(i, k)];
                pv = fma(pi, vi, pv);
            }
            const double K = 0.5 * t * pv;
            for (int i = k + 1; i < s; ++i) {
                const double vi = (i == k + 1) ? 1.0 : M[QD_IX(i, k)];
                y[i] = fma(-K, vi, y[i]);                                   // w
            }
            for (int i = k + 1; i < s; ++i) {
                const double vi = (i == k + 1) ? 1.0 : M[QD_IX(i, k)];
                const double wi = y[i];
                for (int j = k + 1; j <= i; ++j) {
                    const double vj = (j == k + 1) ? 1.0 : M[QD_IX(j, k)];
                    M[QD_IX(i, j)] = fma(-vi, y[j], fma(-wi, vj, M[QD_IX(i, j)]));
                }
            }
        }
    }
    al[s - 2] = M[QD_IX(s - 2, s - 2)]; be[s - 2] = M[QD_IX(s - 1, s - 2)];
    al[s - 1] = M[QD_IX(s - 1, s - 1)]; be[s - 1] = 0.0;
    double lo = INFINITY, tscale = 0.0;
    for (int i = 0; i < s; ++i) {
        const double rad = (i > 0 ? fabs(be[i - 1]) : 0.0) + fabs(be[i]);
        lo = fmin(lo, al[i] - rad);
        tscale = fmax(tscale, fabs(al[i]) + rad);
    }
    double xl = lo - (1e-3 * tscale + 1e-300);
    {
        bool more = true;
        int myit = 0;
        const double dk = (double)s;
        for (int it = 0; it < QD_EIG_MAXIT; ++it) {
            if (!QD_E_ANY(more)) break;
            if (more) {
                // (up to 32 rows: the minors can grow like 4^32, far from overflow)
                double p0 = 1.0, p1 = al[0] - xl, d0 = 0.0, d1 = -1.0, e0 = 0.0, e1 = 0.0;
                for (int i = 1; i < s; ++i) {
                    const double a_ = al[i] - xl, b2 = be[i - 1] * be[i - 1];
                    const double p2 = fma(a_, p1, -(b2 * p0));
                    const double d2 = fma(a_, d1, -(b2 * d0)) - p1;
                    const double e2 = fma(a_, e1, -(b2 * e0)) - 2.0 * d1;
                    p0 = p1; p1 = p2; d0 = d1; d1 = d2; e0 = e1; e1 = e2;
                }
                more = qd_laguerre_step(dk, p1, d1, e1, tscale, xl);
                ++myit;
            }
        }
        if (iters) *iters = myit;
    }
    const double lam = xl;
    {
        // twisted factorisation (see the register version); l+ goes where the diagonal of M was, u- on its sub-diagonal,
        // the backward pivots through y
        const double pivmin = 2.3e-16 * tscale + 1e-300;
        double dm = al[s - 1] - lam;
        y[s - 1] = dm;
        for (int i = s - 2; i >= 0; --i) {
            if (fabs(dm) < pivmin) dm = -pivmin;
            const double u = be[i] * qd_rcp(dm);
            M[QD_IX(i + 1, i)] = u;
            dm = (al[i] - lam) - u * be[i];
            y[i] = dm;
        }
        double dp = al[0] - lam;
        double gbest = fabs(y[0]);
        int kbest = 0;
        for (int i = 0; i + 1 < s; ++i) {
            if (fabs(dp) < pivmin) dp = -pivmin;
            const double l = be[i] * qd_rcp(dp);
            M[QD_IX(i, i)] = l;
            dp = (al[i + 1] - lam) - l * be[i];
            const double g = fabs(dp + y[i + 1] - (al[i + 1] - lam));
            if (g < gbest) { gbest = g; kbest = i + 1; }
        }
        for (int i = 0; i < s; ++i) y[i] = (i == kbest) ? 1.0 : 0.0;
        for (int i = kbest + 1; i < s; ++i) y[i] = -M[QD_IX(i, i - 1)] * y[i - 1];
        for (int i = kbest - 1; i >= 0; --i) y[i] = -M[QD_IX(i, i)] * y[i + 1];
    }
    for (int k = s - 3; k >= 0; --k) {
        double dot = y[k + 1];
        for (int i = k + 2; i < s; ++i) dot = fma(M[QD_IX(i, k)], y[i], dot);
        const double f = tau[k] * dot;
        y[k + 1] -= f;
        for (int i = k + 2; i < s; ++i) y[i] = fma(-f, M[QD_IX(i, k)], y[i]);
    }
    {
        double nrm = 0.0;
        for (int i = 0; i < s; ++i) nrm = fma(y[i], y[i], nrm);
        double sn = 0.0, inv = 1.0;
        if (nrm > 0.0 && nrm < INFINITY) qd_sqrt_rsqrt(nrm, sn, inv);
        for (int i = 0; i < s; ++i) y[i] *= inv;
    }
    lam_out = lam * tusc;
    resid_out = 0.0;
    if (Morig) {
        double r2 = 0.0;
        for (int i = 0; i < s; ++i) {
            double acc = -lam_out * y[i];
            for (int j = 0; j < s; ++j) acc = fma(Morig[j <= i ? QD_IX(i, j) : QD_IX(j, i)], y[j], acc);
            r2 = fma(acc, acc, r2);
        }
        resid_out = sqrt(r2);
    }
#undef QD_SYM
#undef QD_IX
}

/*
 * qd_oracle.c -- plain-C (OpenMP) restatement of the reference's per-step CSD
 * simulation.  TEST INFRASTRUCTURE: the checker for the HIP path and the
 * "port" CPU baseline that bench.py times.  Nothing in the product links or
 * loads this file.  It is validated against oracle/qd_oracle.py (NumPy, the
 * literal restatement) in tests/test_oracle_c.py.
 *
 * Parity status: "parity unpinned" against the reference binary for the
 * physics rows (the reference cannot run here and holds no golden vectors,
 * SURVEY.md 8c); pinned to the reference-generated fixtures for the Kalman and
 * sweep-grid rows through the NumPy oracle.
 *
 * Algorithm = the reference's, literally: per pixel scan ALL 4^N candidate
 * charge states (charge_states.py:135-222), keep the 32 lowest by (energy,
 * index) with duplicate zero-state padding, build the 32x32 Hamiltonian
 * (hamiltonian_build.py:12-45, 75-137, 460-483), dense symmetric eigensolve
 * (ground_state.py:150; here cyclic Jacobi), expectation occupations
 * (ground_state.py:152-162), sensor Lorentzians
 * (TunnelCoupledChargeSensed.py:332-380).
 *
 * CANONICAL ARITHMETIC.  So that integer results (floor values, candidate
 * lists) can be compared bit-for-bit with the GPU, every float64 expression
 * that feeds an integer decision is evaluated in a fixed order with explicit
 * fma(); the HIP code evaluates the same expressions in the same order:
 *   dot(a,b,n)   : acc = 0; for j<n: acc = fma(a[j], b[j], acc)
 *   energy(A,d)  : for i: t_i = dot(A[i,:], d);  E = 0; for i: E = fma(d[i], t_i, E)
 *   linspace     : start + (double)i * step, step = (stop-start)/(R-1), last = stop
 * Build with -ffp-contract=off so nothing else is fused.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define QD_MAXN 8
#define QD_MAXG (QD_MAXN + 1)
#define QD_MAXV (2 * QD_MAXN)
#define QD_K 32
#define QD_NPEAK 5

static inline double qd_dot(const double *a, const double *b, int n) {
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc = fma(a[j], b[j], acc);
    return acc;
}

/* E = d^T A d, A row-major with leading dimension lda */
static inline double qd_energy(const double *A, int lda, const double *d, int n) {
    double E = 0.0;
    for (int i = 0; i < n; ++i) {
        double t = qd_dot(A + (size_t)i * lda, d, n);
        E = fma(d[i], t, E);
    }
    return E;
}

static inline double qd_linspace(double start, double stop, int R, int i) {
    if (R == 1) return start;
    if (i == R - 1) return stop;
    double step = (stop - start) / (double)(R - 1);
    return start + (double)i * step;
}

/* ---- a5: physical gate voltages of pixel (x,y) of channel ch ------------- */
static void qd_pixel_voltages(int N, const double *vgm, const double *origin,
                              const double *gate_v, double sensor_v, double window,
                              int ch, int R, int x, int y, double *vg /*G*/) {
    int G = N + 1;
    double Vd[QD_MAXG];
    for (int i = 0; i < N; ++i) Vd[i] = gate_v[i];
    Vd[N] = sensor_v;
    double v1 = gate_v[ch], v2 = gate_v[ch + 1];
    Vd[ch] = qd_linspace(v1 + (-window), v1 + window, R, x);
    Vd[ch + 1] = qd_linspace(v2 + (-window), v2 + window, R, y);
    for (int i = 0; i < G; ++i) vg[i] = qd_dot(vgm + (size_t)i * G, Vd, G) + origin[i];
}

/* ---- f4: linear voltage-dependent capacitances (voltage_dependent_capacitance.py:72-88,
 *      ground_state.py:53-57): cdd(V) = cdd_full*sa, cgd(V) = cgd_full*sb with
 *      sa = 1 + alpha*mean|v_ext|, sb = 1 + beta*mean|v_ext|, used by the ground-state stage only.
 *      Canonical form (shared with csrc/qd_pixel.h): v' scaled by sb, gradient step 0.1/sa,
 *      energies of the constant-A metric multiplied by 1/sa.  vc == NULL: sa = sb = 1 (exact). */
static void qd_cap_scales(int V, const double *v_ext, const double *vc, double *sa, double *sb) {
    *sa = 1.0; *sb = 1.0;
    if (vc) {
        double sum = 0.0;
        for (int j = 0; j < V; ++j) sum += fabs(v_ext[j]);
        double mabs = sum / (double)V;
        *sa = fma(vc[0], mabs, 1.0);
        *sb = fma(vc[1], mabs, 1.0);
    }
}

/* ---- a8: continuous ground state (charge_states.py:36-88) ---------------- */
static void qd_continuous(int N, int G, int V, const double *cdd_inv, const double *cgd,
                          const double *v_ext, double sa, double sb, double *vdash, double *n_cont) {
    int all_pos = 1;
    const double lr = 0.1 / sa;
    for (int i = 0; i < N; ++i) {
        vdash[i] = qd_dot(cgd + (size_t)i * V, v_ext, V) * sb;
        n_cont[i] = vdash[i];
        if (!(vdash[i] >= 0.0)) all_pos = 0;
    }
    if (!all_pos) {
        double n[QD_MAXN], g2[QD_MAXN], nn[QD_MAXN];
        for (int i = 0; i < N; ++i) n[i] = vdash[i] > 0.0 ? vdash[i] : 0.0;
        for (int i = 0; i < N; ++i) g2[i] = qd_dot(cdd_inv + (size_t)i * G, vdash, N);
        for (int it = 0; it < 50; ++it) {
            for (int i = 0; i < N; ++i) {
                double g1 = qd_dot(cdd_inv + (size_t)i * G, n, N);
                double grad = g1 - g2[i];
                double v = n[i] - lr * grad;
                nn[i] = v > 0.0 ? v : 0.0;
            }
            memcpy(n, nn, sizeof(double) * N);
        }
        for (int i = 0; i < N; ++i) n_cont[i] = n[i];
    }
    for (int i = 0; i < N; ++i) if (!(n_cont[i] > 0.0)) n_cont[i] = 0.0;
}

/* ---- a9: literal scan of all 4^N candidates, top-32 by (energy, index) --- */
static int qd_candidates(int N, int G, const double *cdd_inv, const double *vdash,
                         const double *n_cont, int32_t *states /*32*N*/, int32_t *fl_out) {
    static const int DELTA[4] = {-1, 0, 1, 2};
    double fl[QD_MAXN];
    for (int i = 0; i < N; ++i) { fl[i] = floor(n_cont[i]); if (fl_out) fl_out[i] = (int32_t)fl[i]; }
    double bestE[QD_K]; int32_t bestI[QD_K]; int nbest = 0;
    long total = 1L << (2 * N);
    double d[QD_MAXN], c[QD_MAXN];
    for (long idx = 0; idx < total; ++idx) {
        int valid = 1;
        for (int i = 0; i < N; ++i) {
            int dig = (int)((idx >> (2 * (N - 1 - i))) & 3);   /* MSD = dot 0 */
            c[i] = fl[i] + (double)DELTA[dig];
            if (c[i] < 0.0) valid = 0;
            d[i] = c[i] - vdash[i];
        }
        if (!valid) continue;
        double E = qd_energy(cdd_inv, G, d, N);
        if (E == INFINITY || E != E) continue;              /* +inf sorts with the invalid ones */
        if (nbest == QD_K && !(E < bestE[QD_K - 1])) continue;   /* ties keep the earlier index */
        int pos = nbest < QD_K ? nbest : QD_K - 1;
        while (pos > 0 && E < bestE[pos - 1]) { bestE[pos] = bestE[pos - 1]; bestI[pos] = bestI[pos - 1]; --pos; }
        bestE[pos] = E; bestI[pos] = (int32_t)idx;
        if (nbest < QD_K) ++nbest;
    }
    for (int m = 0; m < QD_K; ++m)
        for (int i = 0; i < N; ++i) {
            if (m < nbest) {
                int dig = (int)((bestI[m] >> (2 * (N - 1 - i))) & 3);
                states[m * N + i] = (int32_t)fl[i] + DELTA[dig];
            } else states[m * N + i] = 0;                     /* duplicate |0..0> padding */
        }
    return nbest;
}

/* ---- dense symmetric eigensolver: cyclic Jacobi, returns eigvec of min eig */
static void qd_jacobi_ground(double *A /*n*n, destroyed*/, int n, double *vec, double *lam) {
    double Vm[QD_K * QD_K];
    for (int i = 0; i < n * n; ++i) Vm[i] = 0.0;
    for (int i = 0; i < n; ++i) Vm[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) { diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j]; }
        if (off == 0.0 || off <= 1e-36 * diag) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                if (!isfinite(theta)) t = 0.0;
                double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = cs * akp - sn * akq; A[k * n + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = cs * apk - sn * aqk; A[q * n + k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = Vm[k * n + p], vkq = Vm[k * n + q];
                    Vm[k * n + p] = cs * vkp - sn * vkq; Vm[k * n + q] = sn * vkp + cs * vkq;
                }
            }
    }
    int best = 0;
    for (int i = 1; i < n; ++i) if (A[i * n + i] < A[best * n + best]) best = i;
    for (int k = 0; k < n; ++k) vec[k] = Vm[k * n + best];
    *lam = A[best * n + best];
}

/* ---- a11-a13: H = diag(F) + H_t, ground state, <n> ------------------------ */
static void qd_ground_occupation(int N, int G, const double *cdd_inv, const double *vdash, double isa,
                                 const int32_t *states, const double *tc, double *occ, double *lam_out) {
    double H[QD_K * QD_K];
    double d[QD_MAXN];
    memset(H, 0, sizeof(H));
    for (int m = 0; m < QD_K; ++m) {
        for (int i = 0; i < N; ++i) d[i] = (double)states[m * N + i] - vdash[i];
        H[m * QD_K + m] = qd_energy(cdd_inv, G, d, N) * isa;
    }
    for (int i = 0; i < QD_K; ++i)
        for (int j = 0; j < QD_K; ++j) {
            if (i == j) continue;
            int a = -1, b = -1, ok = 1;
            for (int k = 0; k < N; ++k) {
                int df = states[j * N + k] - states[i * N + k];
                if (df == 0) continue;
                if (a < 0) a = k; else if (b < 0) b = k; else { ok = 0; break; }
            }
            if (!ok || a < 0 || b != a + 1) continue;
            int da = states[j * N + a] - states[i * N + a], db = states[j * N + b] - states[i * N + b];
            double na = (double)states[i * N + a], nb = (double)states[i * N + b];
            if (da == -1 && db == 1) H[i * QD_K + j] += -tc[a] * sqrt(na * (nb + 1.0));
            else if (da == 1 && db == -1) H[i * QD_K + j] += -tc[a] * sqrt(nb * (na + 1.0));
        }
    double vec[QD_K], lam;
    qd_jacobi_ground(H, QD_K, vec, &lam);
    for (int i = 0; i < N; ++i) occ[i] = 0.0;
    for (int m = 0; m < QD_K; ++m) {
        double p = vec[m] * vec[m];
        for (int i = 0; i < N; ++i) occ[i] += p * (double)states[m * N + i];
    }
    if (lam_out) *lam_out = lam;
}

/* ---- a15: sensor stage ---------------------------------------------------- */
static double qd_sensor(int N, int G, int V, const double *cdd_inv, const double *cgd,
                        const double *v_ext, const double *occ, double gamma) {
    double vd[QD_MAXG], d[QD_MAXG], F[2 * QD_NPEAK + 1];
    for (int i = 0; i < G; ++i) vd[i] = qd_dot(cgd + (size_t)i * V, v_ext, V);
    double Ns = nearbyint(vd[N]);                       /* np.round: half to even */
    for (int i = 0; i < N; ++i) d[i] = occ[i] - vd[i];
    for (int k = -QD_NPEAK; k <= QD_NPEAK; ++k) {
        d[N] = (Ns + (double)k) - vd[N];
        F[k + QD_NPEAK] = qd_energy(cdd_inv, G, d, G);
    }
    double s = 0.0;
    for (int k = 0; k < 2 * QD_NPEAK; ++k) {
        double x = (F[k + 1] - F[k]) / gamma;
        s += 1.0 / (x * x + 1.0);
    }
    return s;
}

/* One CSD channel of one env.  All matrices row-major float64.
 *   cdd_inv G*G, cgd G*V, Cbg nb*G, alpha nb, vgm G*G, origin G, gate_v N, barrier_v nb
 * Outputs (any may be NULL): states P*32*N, floors P*N, occ P*N, z P, tc_out P*nb. */
int qdo_csd_channel_vc(int N, int R, const double *cdd_inv, const double *cgd, const double *Cbg,
                       const double *alpha, double tc_base, double gamma,
                       const double *vgm, const double *origin, const double *gate_v, double sensor_v,
                       const double *barrier_v, double window, int ch,
                       int32_t *states_out, int32_t *floors_out, double *occ_out, double *z_out,
                       double *tc_out, int pix_begin, int pix_end, const double *vc /* NULL or {alpha, beta} */) {
    if (N < 2 || N > QD_MAXN || ch < 0 || ch >= N - 1 || R < 1) return 1;
    int G = N + 1, nb = N - 1, V = G + nb;
    if (pix_end < 0) pix_end = R * R;
#pragma omp parallel for schedule(dynamic, 8)
    for (int p = pix_begin; p < pix_end; ++p) {
        int y = p / R, x = p % R;
        double v_ext[QD_MAXV], vdash[QD_MAXN], n_cont[QD_MAXN], tc[QD_MAXN], occ[QD_MAXN];
        int32_t st[QD_K * QD_MAXN];
        qd_pixel_voltages(N, vgm, origin, gate_v, sensor_v, window, ch, R, x, y, v_ext);
        for (int b = 0; b < nb; ++b) v_ext[G + b] = barrier_v[b];
        double sa, sb;
        qd_cap_scales(V, v_ext, vc, &sa, &sb);
        qd_continuous(N, G, V, cdd_inv, cgd, v_ext, sa, sb, vdash, n_cont);
        qd_candidates(N, G, cdd_inv, vdash, n_cont, st, floors_out ? floors_out + (size_t)p * N : NULL);
        for (int b = 0; b < nb; ++b) {
            double vb_eff = barrier_v[b] + qd_dot(Cbg + (size_t)b * G, v_ext, G);
            tc[b] = tc_base * exp(-alpha[b] * vb_eff);
        }
        qd_ground_occupation(N, G, cdd_inv, vdash, 1.0 / sa, st, tc, occ, NULL);
        double z = qd_sensor(N, G, V, cdd_inv, cgd, v_ext, occ, gamma);
        if (states_out) memcpy(states_out + (size_t)p * QD_K * N, st, sizeof(int32_t) * QD_K * N);
        if (occ_out) memcpy(occ_out + (size_t)p * N, occ, sizeof(double) * N);
        if (tc_out) memcpy(tc_out + (size_t)p * nb, tc, sizeof(double) * nb);
        if (z_out) z_out[p] = z;
    }
    return 0;
}

int qdo_csd_channel(int N, int R, const double *cdd_inv, const double *cgd, const double *Cbg,
                    const double *alpha, double tc_base, double gamma,
                    const double *vgm, const double *origin, const double *gate_v, double sensor_v,
                    const double *barrier_v, double window, int ch,
                    int32_t *states_out, int32_t *floors_out, double *occ_out, double *z_out,
                    double *tc_out, int pix_begin, int pix_end) {
    return qdo_csd_channel_vc(N, R, cdd_inv, cgd, Cbg, alpha, tc_base, gamma, vgm, origin, gate_v, sensor_v,
                              barrier_v, window, ch, states_out, floors_out, occ_out, z_out, tc_out,
                              pix_begin, pix_end, NULL);
}

/* ---- a17: percentile normalisation, numpy 'linear' method ---------------- */
static int qd_cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    if (x != x) return (y != y) ? 0 : 1;                 /* NaN last */
    if (y != y) return -1;
    return (x > y) - (x < y);
}
static double qd_lerp(double a, double b, double t) {
    double diff = b - a;
    double r = a + diff * t;
    if (t >= 0.5) r = b - diff * (1.0 - t);
    return r;
}
static double qd_percentile_sorted(const double *s, long n, double q_percent) {
    double q = q_percent / 100.0;
    double virt = (double)(n - 1) * q;      /* numpy 'linear' method: (n-1)*quantile */
    double prev = floor(virt);
    long ip = (long)prev; if (ip < 0) ip = 0; if (ip > n - 1) ip = n - 1;
    long in = ip + 1; if (in > n - 1) in = n - 1;
    double g = virt - prev;
    if (s[n - 1] != s[n - 1]) return NAN;
    return qd_lerp(s[ip], s[in], g);
}
/* z: n values (any layout) -> out float32 same layout; returns p_low/p_high */
int qdo_normalise(const double *z, long n, float *out, double *plo_hi) {
    double *s = (double *)malloc(sizeof(double) * (size_t)n);
    if (!s) return 2;
    memcpy(s, z, sizeof(double) * (size_t)n);
    qsort(s, (size_t)n, sizeof(double), qd_cmp_double);
    double lo = qd_percentile_sorted(s, n, 0.5), hi = qd_percentile_sorted(s, n, 99.5);
    free(s);
    if (plo_hi) { plo_hi[0] = lo; plo_hi[1] = hi; }
    for (long i = 0; i < n; ++i) {
        double v = 0.0;
        if (hi > lo) { v = (z[i] - lo) / (hi - lo); v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
        out[i] = (float)v;
    }
    return 0;
}

/* whole raw observation of one env: z_out laid out [C][R*R].
 * gammas: NULL (constant peak width `gamma`) or one width per channel (f4 variable peak width) */
int qdo_env_images_vc(int N, int R, const double *cdd_inv, const double *cgd, const double *Cbg,
                      const double *alpha, double tc_base, double gamma, const double *vgm,
                      const double *origin, const double *gate_v, double sensor_v,
                      const double *barrier_v, double window, double *z_out, double *occ_out,
                      const double *vc, const double *gammas) {
    for (int ch = 0; ch < N - 1; ++ch) {
        int rc = qdo_csd_channel_vc(N, R, cdd_inv, cgd, Cbg, alpha, tc_base, gammas ? gammas[ch] : gamma, vgm, origin,
                                    gate_v, sensor_v, barrier_v, window, ch, NULL, NULL,
                                    occ_out ? occ_out + (size_t)ch * R * R * N : NULL,
                                    z_out + (size_t)ch * R * R, NULL, 0, -1, vc);
        if (rc) return rc;
    }
    return 0;
}

int qdo_env_images(int N, int R, const double *cdd_inv, const double *cgd, const double *Cbg,
                   const double *alpha, double tc_base, double gamma, const double *vgm,
                   const double *origin, const double *gate_v, double sensor_v,
                   const double *barrier_v, double window, double *z_out, double *occ_out) {
    return qdo_env_images_vc(N, R, cdd_inv, cgd, Cbg, alpha, tc_base, gamma, vgm, origin, gate_v, sensor_v,
                             barrier_v, window, z_out, occ_out, NULL, NULL);
}

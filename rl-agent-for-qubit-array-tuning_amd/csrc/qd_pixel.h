// qd_pixel.h -- the pixel-per-lane part of the hot path as __host__ __device__
// code: sweep-voltage synthesis (a5), continuous ground state (a8), EXACT
// k-best candidate search (a9), barrier couplings (a10), sensor stage (a15).
// The same source is compiled into the HIP kernels and into the CPU-only test
// harness (tests/hosttest/qd_hosttest.cpp) so its integer results can be checked against the
// oracle without a GPU.
//
// Reference rows (file:line under /root/reference):
//   a5  qarray_base_class.py:95-168, GateVoltageComposer.py:170-211
//   a8  charge_states.py:36-88        a9  charge_states.py:135-222
//   a10 barrier_voltage_model.py:55-151   a15 TunnelCoupledChargeSensed.py:332-380
//
// CANONICAL ARITHMETIC (must match oracle/qd_oracle.c bit for bit, because the
// results feed integer decisions: floor(), candidate order):
//   dot(a,b,n)  : acc = 0; for j<n: acc = fma(a[j], b[j], acc)
//   energy(A,d) : t_i = dot(A[i,:], d); E = 0; for i: E = fma(d[i], t_i, E)
//   linspace    : start + (double)i*step, step=(stop-start)/(R-1), last point = stop
// Compile with -ffp-contract=off: only the explicit fma() calls fuse.
#pragma once
#include <math.h>
#include "qd_common.h"

QD_HD double qd_dot(const double* a, const double* b, int n) {
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc = fma(a[j], b[j], acc);
    return acc;
}

template <int N>
QD_HD double qd_dotN(const double* a, const double* b) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) acc = fma(a[j], b[j], acc);
    return acc;
}

QD_HD double qd_linspace(double start, double stop, int R, int i) {
    if (R == 1) return start;
    if (i == R - 1) return stop;
    double step = (stop - start) / (double)(R - 1);
    return start + (double)i * step;
}

// ---------------------------------------------------------------------------
// a5 + a10 front end of one pixel: voltages and tunnel couplings.
//   par: env parameter block, st: env state block (layout qd_layout(N)).
// Outputs: v_ext[V] = [physical gate voltages (G), barrier voltages (nb)],
//          vpp[G] = cgd_full @ v_ext (constant matrices: what the sensor stage uses), tc[nb].
// ---------------------------------------------------------------------------
template <int N>
QD_HD void qd_pixel_voltages(const double* par, const double* st, int ch, int R, int x, int y,
                             double* v_ext, double* vpp, double* tc) {
    constexpr int G = N + 1, NB = N - 1, V = 2 * N;
    const QdLayout L = qd_layout(N);
    const double* vgm = st + L.s_vgm;
    const double* gate_v = st + L.s_gate_v;
    const double* barrier_v = st + L.s_barrier_v;
    const double sensor_v = st[L.s_sensor_gt];
    const double window = par[L.scal + 2];
    double Vd[G];
#pragma unroll
    for (int i = 0; i < N; ++i) Vd[i] = gate_v[i];
    Vd[N] = sensor_v;
    const double v1 = gate_v[ch], v2 = gate_v[ch + 1];
    const double sx = qd_linspace(v1 + (-window), v1 + window, R, x);
    const double sy = qd_linspace(v2 + (-window), v2 + window, R, y);
#pragma unroll
    for (int i = 0; i < N; ++i) {            // static indices only (keeps Vd in registers)
        if (i == ch) Vd[i] = sx;
        if (i == ch + 1) Vd[i] = sy;
    }
#pragma unroll
    for (int i = 0; i < G; ++i) { v_ext[i] = qd_dotN<G>(vgm + i * G, Vd) + par[L.origin + i]; QD_ROW_FENCE(); }
#pragma unroll
    for (int b = 0; b < NB; ++b) v_ext[G + b] = barrier_v[b];
#pragma unroll
    for (int i = 0; i < G; ++i) { vpp[i] = qd_dotN<V>(par + L.cgd + i * V, v_ext); QD_ROW_FENCE(); }
    // a10 tunnel couplings: vb_eff = vb + Cbg @ vg (the Cbb cross term is identically 0)
    const double tc_base = par[L.scal + 0];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        double vb_eff = barrier_v[b] + qd_dotN<G>(par + L.cbg + b * G, v_ext);
        tc[b] = tc_base * exp(-par[L.alpha + b] * vb_eff);
        QD_ROW_FENCE();
    }
}

// ---------------------------------------------------------------------------
// a8 continuous ground state, with the optional linear voltage-dependent
// capacitance model (f4; voltage_dependent_capacitance.py:72-88, ground_state.py:53-57):
//   cdd(V) = cdd_full (1 + alpha mean|v_ext|),  cgd(V) = cgd_full (1 + beta mean|v_ext|)
// so, for the ground-state stage only (the sensor stage keeps the constant matrices),
//   v' -> sb v',  A -> A / sa   with sa, sb the two scale factors.
// Canonical form used here and in the oracle: v' is scaled, the projected-gradient step
// uses lr = 0.1 / sa with the constant A, the search runs in the constant-A metric (a
// positive factor does not change the order) and the kept energies are multiplied by
// isa = 1 / sa afterwards.  With the model off sa = sb = 1 and every operation is exact,
// so results are bit-identical to the constant-capacitance path.
//   vd[N]: in  v' = (cgd_full @ v_ext)[:N];  out  the scaled v'.
// ---------------------------------------------------------------------------
template <int N>
QD_HD void qd_pixel_continuous(const double* par, const double* v_ext, double* vd, double* ncont, double* isa) {
    constexpr int G = N + 1, V = 2 * N;
    const QdLayout L = qd_layout(N);
    double sa = 1.0, sb = 1.0;
    if (par[L.scal + 4] != 0.0) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < V; ++j) sum += fabs(v_ext[j]);
        const double mabs = sum / (double)V;
        sa = fma(par[L.scal + 5], mabs, 1.0);
        sb = fma(par[L.scal + 6], mabs, 1.0);
    }
    *isa = 1.0 / sa;
    const double lr = 0.1 / sa;
    bool all_pos = true;
#pragma unroll
    for (int i = 0; i < N; ++i) { vd[i] = vd[i] * sb; ncont[i] = vd[i]; if (!(vd[i] >= 0.0)) all_pos = false; }
#if defined(QD_CAND_ABLATE) && QD_CAND_ABLATE == 1
    all_pos = true;                                        // diagnostic: skip the projected-gradient loop
#endif
    if (!all_pos) {
        const double* A = par + L.cdd_inv;
        double n[N], g2[N], nn[N];
#pragma unroll
        for (int i = 0; i < N; ++i) n[i] = vd[i] > 0.0 ? vd[i] : 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) { g2[i] = qd_dotN<N>(A + i * G, vd); QD_ROW_FENCE(); }
        for (int it = 0; it < 50; ++it) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double g1 = qd_dotN<N>(A + i * G, n);
                double grad = g1 - g2[i];
                double v = n[i] - lr * grad;
                nn[i] = v > 0.0 ? v : 0.0;
                QD_ROW_FENCE();
            }
#pragma unroll
            for (int i = 0; i < N; ++i) n[i] = nn[i];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) ncont[i] = n[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) if (!(ncont[i] > 0.0)) ncont[i] = 0.0;
}

// Whole front end (host harness / tests): vpp stays the constant-matrix product, vd is the
// (possibly scaled) v' the candidate search uses.
template <int N>
QD_HD void qd_pixel_front(const double* par, const double* st, int ch, int R, int x, int y,
                          double* v_ext, double* vpp, double* vd, double* ncont, double* tc, double* isa) {
    qd_pixel_voltages<N>(par, st, ch, R, x, y, v_ext, vpp, tc);
#pragma unroll
    for (int i = 0; i < N; ++i) vd[i] = vpp[i];
    qd_pixel_continuous<N>(par, v_ext, vd, ncont, isa);
}

// a15 peak width of one channel: constant, or (f4, qarray_base_class.py:192-196 +
// utils/vary_peak_width.py) clip(gamma0 - |alpha (|v_ch| + |v_ch+1|) / 2|, 0, 1) with v the
// current VIRTUAL plunger voltages of the swept pair.
QD_HD double qd_peak_width(const double* par, const double* st, const QdLayout& L, int ch) {
    const double g0 = par[L.scal + 1], a = par[L.scal + 3];
    if (a < 0.0) return g0;
    const double vavg = (fabs(st[L.s_gate_v + ch]) + fabs(st[L.s_gate_v + ch + 1])) / 2.0;
    const double w = g0 - fabs(a * vavg);
    return fmin(fmax(w, 0.0), 1.0);
}

// ---------------------------------------------------------------------------
// a9  exact k-best candidate search.
// Minimise E(c) = (c-v')^T A (c-v') over c = floor(n_cont) + delta,
// delta in {-1,0,1,2}^N, c >= 0, keeping the 32 smallest by (E, index) where
// index = sum_i (delta_i+1) << 2(N-1-i).  Equivalent to the reference's scan of
// all 4^N candidates with stable sorts (proved against the oracle in tests), but
// done as a depth-first branch and bound:
//   E(c) = E(m) + g.(c-m) + (c-m)^T A (c-m),   m = n_cont,  g = 2 A (m - v')
// (exact for any m).  With A = U U^T (U upper) the quadratic part is
// sum_i ((U^T (c-m))_i)^2 where term i depends on dots 0..i only, and the linear
// part is separable, so  sum_{i<=L} [t_i^2 + g_i (c_i-m_i)] + tail_L  with
// tail_L = sum_{j>L} min_c g_j (c_j-m_j)  is a lower bound of E - E(m) for the
// whole sub-tree.  m is the (approximate) constrained continuous minimiser, so
// the bound stays tight when dots are clipped empty (g_i > 0 there) -- without
// the linear term the search degenerates to ~3^N leaves in that regime.
// Bounds are only used to PRUNE (with a safety margin that covers round-off);
// every candidate that reaches a leaf gets its canonical energy, so the kept
// list is bit-identical to the brute-force scan.
// The top-k buffer is caller-provided strided storage (LDS on the GPU).
// ---------------------------------------------------------------------------
template <int N>
struct QdSearch {
    const double* A; int lda;        // cdd_inv (row-major, lda = G)
    const double* U;                 // N*N row-major upper factor
    const double* uinv;              // 1/U[i][i]
    double fl[N], vdash[N], m[N], g[N], tail[N];
    double dm[N], dv[N];             // current path: c - m, c - v'
    double pre1[N], pre2[N];         // canonical row sums over dots 0..A and 0..B of the current path
    double Em;
    double* e; int es;               // energies, stride
    uint16_t* id; int is;            // indices, stride
    int count;
    double gE[4]; unsigned gI[4]; int gS[4];    // per-group maxima of the kept buffer
    double maxE; unsigned maxI; int maxS, maxG;  // overall maximum (the 32nd best)
    double lim;                      // prune when (partial + tail) > lim   (relative to Em)
    unsigned idx;
    unsigned long long nodes, leaves, inserts, shifts;   // statistics (host harness only)
};

// The kept set lives in caller-provided strided storage (LDS on the GPU) as an UNSORTED buffer
// split into 4 groups of 8 slots; the lexicographic (E, idx) maximum of each group is cached in
// registers.  Replacing the overall maximum costs one write + an 8-slot rescan (independent
// reads) instead of a chain of dependent compare-and-shift steps.  qd_search_sort() orders the
// final list when the caller wants the reference order.
template <int N>
QD_HD bool qd_lex_less(double ea, unsigned ia, double eb, unsigned ib) {
    // bitwise on purpose: short-circuit operators become divergent branches in the leaf loop
    return (bool)((int)(ea < eb) | ((int)(ea == eb) & (int)(ia < ib)));
}

template <int N>
QD_HD void qd_search_rescan_group(QdSearch<N>& S, int g) {
    double me = -INFINITY; unsigned mi = 0; int ms = g * 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int sl = g * 8 + t;
        const double e = S.e[sl * S.es]; const unsigned i = S.id[sl * S.is];
        if (t == 0 || qd_lex_less<N>(me, mi, e, i)) { me = e; mi = i; ms = sl; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j == g) { S.gE[j] = me; S.gI[j] = mi; S.gS[j] = ms; }
}

template <int N>
QD_HD void qd_search_set_bound(QdSearch<N>& S) {
    // overall maximum = lexicographic max of the 4 group maxima
    int g = 0; double bound = S.gE[0]; unsigned bi = S.gI[0]; int bs = S.gS[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (qd_lex_less<N>(bound, bi, S.gE[j], S.gI[j])) { bound = S.gE[j]; bi = S.gI[j]; bs = S.gS[j]; g = j; }
    S.maxE = bound; S.maxI = bi; S.maxS = bs; S.maxG = g;
    S.lim = (bound - S.Em) + ((fabs(bound) + fabs(S.Em)) * 1e-11 + 1e-300);
}

template <int N>
QD_HD void qd_search_insert(QdSearch<N>& S, double E, unsigned idx) {
    if (!(E < INFINITY)) return;                          // inf / NaN: treated as invalid
#ifndef __HIP_DEVICE_COMPILE__
    S.inserts++;
#endif
    if (S.count < QD_K) {
        S.e[S.count * S.es] = E; S.id[S.count * S.is] = (uint16_t)idx;
        S.count++;
        if (S.count == QD_K) {
#pragma unroll
            for (int g = 0; g < 4; ++g) qd_search_rescan_group(S, g);
            qd_search_set_bound(S);
        }
        return;
    }
    if (!qd_lex_less<N>(E, idx, S.maxE, S.maxI)) return;
    S.e[S.maxS * S.es] = E; S.id[S.maxS * S.is] = (uint16_t)idx;
    qd_search_rescan_group(S, S.maxG);
    qd_search_set_bound(S);
}

// in-place insertion sort of the kept entries by (E, idx)
template <int N>
QD_HD void qd_search_sort(QdSearch<N>& S) {
    for (int i = 1; i < S.count; ++i) {
        const double E = S.e[i * S.es]; const unsigned idx = S.id[i * S.is];
        int pos = i;
        while (pos > 0) {
            const double ep = S.e[(pos - 1) * S.es]; const unsigned ip = S.id[(pos - 1) * S.is];
            if (!qd_lex_less<N>(E, idx, ep, ip)) break;
            S.e[pos * S.es] = ep; S.id[pos * S.is] = (uint16_t)ip;
            --pos;
        }
        S.e[pos * S.es] = E; S.id[pos * S.is] = (uint16_t)idx;
    }
}

// split points of the shared canonical row sums: dots 0..A, A+1..B, and B+1..N-1 at the leaf
template <int N>
struct QdSplit {
    static constexpr int B = N - 3;                       // leaf finishes the last two dots
    static constexpr int A = (B - 3 >= 0) ? B - 3 : -1;
};

template <int N, int L>
struct QdLevel {
    static QD_HD void run(QdSearch<N>& S, double partial) {
#ifndef __HIP_DEVICE_COMPILE__
        S.nodes++;
#endif
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < L; ++j) s = fma(S.U[j * N + L], S.dm[j], s);
        const double uLL = S.U[L * N + L];
        const double flL = S.fl[L];
        const double base = flL - S.m[L];                  // c - m for delta = 0
        const double gL = S.g[L];
        // The increment f(k) = (u x + s)^2 + g x, x = base + (k-1), is a convex parabola in k, so
        // visiting k in order of distance from its real minimiser k* (Schnorr-Euchner zig-zag)
        // visits the choices in increasing f; the first one that fails the bound ends the node.
        const double ui = S.uinv[L];
        const double kstar = (fma(-s, ui, -(0.5 * gL) * (ui * ui)) - base) + 1.0;
        const int kmin = (flL > 0.0) ? 0 : 1;              // delta = -1 would give c < 0
        double kc = rint(kstar);
        kc = fmin(fmax(kc, (double)kmin), 3.0);
        if (!(kc == kc)) kc = (double)kmin;                // NaN guard
        int k = (int)kc, lo = k - 1, hi = k + 1;
        const unsigned sh = 2u * (unsigned)(N - 1 - L);
        const double tl = S.tail[L];
        // ONE call site per level (a 4x unrolled visit would inline 4^N leaves)
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const double x = base + (double)(k - 1);
            const double t = fma(uLL, x, s);
            const double pn = partial + fma(t, t, gL * x);
            if (pn + tl > S.lim) return;
            S.dm[L] = x;
            S.dv[L] = (flL + (double)(k - 1)) - S.vdash[L];
            S.idx = (S.idx & ~(3u << sh)) | ((unsigned)k << sh);
            // canonical energy rows t_i = fma-chain over j = 0..N-1 of A[i][j] dv[j]: the part of the
            // chain that only involves dots already fixed is shared by the whole sub-tree
            if constexpr (L == QdSplit<N>::A) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j <= QdSplit<N>::A; ++j) acc = fma(S.A[i * S.lda + j], S.dv[j], acc);
                    S.pre1[i] = acc;
                }
            }
            if constexpr (L == QdSplit<N>::B) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    double acc = (QdSplit<N>::A >= 0) ? S.pre1[i] : 0.0;
#pragma unroll
                    for (int j = QdSplit<N>::A + 1; j <= QdSplit<N>::B; ++j) acc = fma(S.A[i * S.lda + j], S.dv[j], acc);
                    S.pre2[i] = acc;
                }
            }
            QdLevel<N, L + 1>::run(S, pn);
            const bool can_lo = lo >= kmin, can_hi = hi <= 3;
            if (!((int)can_lo | (int)can_hi)) return;
            const bool take_lo = (bool)((int)can_lo & ((int)!can_hi | (int)((kstar - (double)lo) <= ((double)hi - kstar))));
            k = take_lo ? lo : hi;
            lo -= take_lo ? 1 : 0;
            hi += take_lo ? 0 : 1;
        }
    }
};

template <int N>
struct QdLevel<N, N> {
    static QD_HD void run(QdSearch<N>& S, double) {
#ifndef __HIP_DEVICE_COMPILE__
        S.leaves++;
#endif
        // fetch every matrix entry the leaf needs before the arithmetic starts (on the GPU they are scalar
        // loads: issued back to back they overlap, interleaved with the FMAs each one is waited for separately)
        constexpr int J0 = QdSplit<N>::B + 1, NJ = N - J0;
        double a[N][NJ > 0 ? NJ : 1];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) a[i][j] = S.A[i * S.lda + J0 + j];
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        double E = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double t = (QdSplit<N>::B >= 0) ? S.pre2[i] : 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) t = fma(a[i][j], S.dv[J0 + j], t);
            E = fma(S.dv[i], t, E);
        }
        qd_search_insert(S, E, S.idx);
    }
};

// Returns the number of valid candidates found (<= 32).  With sort_output the list is in the
// reference order (increasing (E, idx)); otherwise it is the same SET in search order.
template <int N>
QD_HD int qd_candidates(const double* par, const double* vpp, const double* ncont,
                        double* e, int es, uint16_t* id, int is, int32_t* fl_out, bool sort_output,
                        unsigned long long* stats = nullptr) {
    const QdLayout L = qd_layout(N);
    QdSearch<N> S;
    S.A = par + L.cdd_inv; S.lda = N + 1; S.U = par + L.ufac; S.uinv = par + L.uinv;
    bool shifted = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        S.fl[i] = floor(ncont[i]); S.vdash[i] = vpp[i]; S.m[i] = ncont[i];
        S.dm[i] = 0.0; S.dv[i] = 0.0; S.g[i] = 0.0; S.tail[i] = 0.0;
        fl_out[i] = (int32_t)S.fl[i];
        if (ncont[i] != vpp[i]) shifted = true;
    }
    S.Em = 0.0;
    if (shifted) {                                         // some dot is clipped: m != v'
        double mv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) mv[i] = S.m[i] - S.vdash[i];
        double Em = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double t = qd_dotN<N>(S.A + i * S.lda, mv);
            S.g[i] = 2.0 * t;
            Em = fma(mv[i], t, Em);
            QD_ROW_FENCE();
        }
        S.Em = Em;
        double acc = 0.0;
#pragma unroll
        for (int j = N - 1; j >= 0; --j) {
            S.tail[j] = acc;                               // sum over levels > j
            const double cmin = S.fl[j] > 0.0 ? S.fl[j] - 1.0 : 0.0, cmax = S.fl[j] + 2.0;
            const double lo = S.g[j] * (cmin - S.m[j]), hi = S.g[j] * (cmax - S.m[j]);
            acc += lo < hi ? lo : hi;
        }
    }
    S.e = e; S.es = es; S.id = id; S.is = is;
    S.count = 0; S.lim = INFINITY; S.idx = 0;
    S.nodes = S.leaves = S.inserts = S.shifts = 0;
#if !(defined(QD_CAND_ABLATE) && QD_CAND_ABLATE == 2)
    QdLevel<N, 0>::run(S, 0.0);                            // (ablate 2: diagnostic, front end only)
#endif
    if (sort_output) qd_search_sort(S);
#ifndef __HIP_DEVICE_COMPILE__
    if (stats) { stats[0] += S.nodes; stats[1] += S.leaves; stats[2] += S.inserts; stats[3] += S.shifts; }
#endif
    return S.count;
}

// ---------------------------------------------------------------------------
// a15  sensor stage of one pixel given the expectation occupations.
// F_k = q^T A q with q = [occ - v''[:N], (Ns+k) - v''[N]], k = -5..5;
// signal = sum over the 10 first differences of 1/((dF/gamma)^2 + 1).
// dF is evaluated in closed form: with a = A[N][N], b = A[N,:N] . (occ - v''),
// x_k = Ns + k - v''[N]:  F_{k+1} - F_k = 2 b + a (2 x_k + 1)   (A symmetric).
// This is the same quantity as the reference's difference of two quadratic
// forms without the cancellation; agreement is to round-off (float output).
// ---------------------------------------------------------------------------
template <int N>
QD_HD double qd_sensor(const double* par, const double* vpp, const double* occ, double gamma) {
    constexpr int G = N + 1;
    const QdLayout L = qd_layout(N);
    const double* A = par + L.cdd_inv;
    const double Ns = rint(vpp[N]);                       // np.round: half to even
    double b = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) b = fma(A[N * G + i], occ[i] - vpp[i], b);
    const double a = A[N * G + N];
    double s = 0.0;
#pragma unroll
    for (int k = -QD_NPEAK; k < QD_NPEAK; ++k) {
        double xk = (Ns + (double)k) - vpp[N];
        double dF = 2.0 * b + a * (2.0 * xk + 1.0);
        double r = dF / gamma;
        s += 1.0 / (r * r + 1.0);
    }
    return s;
}

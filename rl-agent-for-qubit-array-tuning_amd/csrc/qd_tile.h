// qd_tile.h -- tile-shared candidate search: ONE wavefront per 8x8 pixel tile.
//
// The reference scans all 4^N candidate charge states of every pixel (charge_states.py:135-222).  Neighbouring
// pixels of a CSD differ only by a small shift d_p of v' = cgd[:N] @ v_ext, and
//     E_p(c) - E_p(c0) = [E_ref(c) - E_ref(c0)] - 2 d_p^T A (c - c0),        d_p = v'_p - v'_ref,
// so with d_p = x dx + y dy (+ a rounding-level residual) the energies of a candidate over the whole tile are a
// PLANE over (x, y).  A candidate can be among the 32 lowest of SOME pixel of the tile only if its minimum over
// the tile m(c) does not exceed T = the 32nd smallest of the maxima M(c) taken over candidates valid in every
// pixel.  The wave therefore searches ONCE per tile:
//   1. per lane (= pixel): sweep voltages, couplings, continuous ground state, floor (as before)
//   2. seeds: the product set of the cheapest per-dot options around the greedy lattice point, one per lane,
//      gives an upper bound T'' >= T
//   3. level-synchronous branch and bound over dots 0..N-1, 64 children at a time, keeps every partial state
//      whose lower bound of m(c) is <= T''  ->  list S' (typically 50-130 states instead of 64 x ~100 leaves)
//   4. T by bisection on the count of M(c) <= t, superset S = {m(c) <= T} (typically 40-90 states)
//   5. per lane: energies of S by two FMAs each, own validity box, top-32 buffer in LDS
// Every bound carries explicit margins (affine residual, arithmetic); a lane whose 32nd / 33rd energies are closer
// than the margin, and any tile that overflows a capacity, is flagged (nvalid = -1) and redone by the exact
// per-pixel search (qd_k_candidates), so the kept SETS are bit-identical to the brute-force scan in every case.
#pragma once
#include "qd_pixel.h"
#include "qd_groundstate.h"        // qd_rcp, qd_sqrt_rsqrt

#if defined(__HIPCC__)

// Capacities, chosen so that the wave's LDS stays within 32 KB (5 waves per CU; measured: 3 waves per CU cost 14 %, 2 cost 50 %):
// the frontier ping-pong (2 x 832 x 12 B) hides under the per-lane kept sets (20 KB), the superset arrays take 383 x 28 B.
// Round 2 ran 512 / 255: across a batch of random devices 20-25 % of the tiles overflowed one of the two IN EVERY REGIME
// (a tile of some devices spans several electrons) and went to the per-pixel search at 2x the cost of a tile; 832 / 383
// halves that (8-dot 64x64, 242 envs per launch: tile + redo 51.6 -> 49.4 us per env-step in the random-action regime,
// 55.8 -> 51.5 near the ground truth); 832 / 511 needs 36 KB (4 waves per CU) and is slower again (55.0).
#ifndef QD_T_FCAP
#define QD_T_FCAP 832          // frontier capacity (partial states per level)
#endif
#ifndef QD_T_SCAP
#define QD_T_SCAP 383          // superset capacity
#endif
#define QD_T_REDO (-1)         // QdPixelRec.nvalid marker: pixel left to the exact per-pixel search

struct QdTileLds {
    // per-level constants of the search (dot i = level i)
    double m[QD_MAXN], g[QD_MAXN], lamx[QD_MAXN], lamy[QD_MAXN], tail[QD_MAXN], sla[QD_MAXN], slb[QD_MAXN], q[QD_MAXN];
    int lo[QD_MAXN], nd[QD_MAXN], cg[QD_MAXN];
    int optval[QD_MAXN][4];
    union {
        struct { double pn[2][QD_T_FCAP]; uint32_t code[2][QD_T_FCAP]; } f;      // frontier ping-pong
        struct { double e[QD_K][64]; uint16_t id[QD_K][64]; } k;                   // per-lane kept sets
    } u;
    double sD[QD_T_SCAP], sa[QD_T_SCAP], sb[QD_T_SCAP];
    uint32_t scode[QD_T_SCAP];
#if defined(QD_TILE_PAD_LDS)
    char pad[QD_TILE_PAD_LDS];                 // diagnostic builds: occupancy sensitivity
#endif
};

__device__ __forceinline__ double qd_rl(double v, int lane) {                       // uniform lane index
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int qd_wmin_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int qd_wmax_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ double qd_wmin_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double qd_wmax_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int qd_lane_prefix(unsigned long long mask) {           // set bits below my lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// max / min over the tile rectangle [x0,x1] x [y0,y1] of x a + y b
__device__ __forceinline__ double qd_plane_max(double a, double b, double x0, double x1, double y0, double y1) {
    return fmax(x0 * a, x1 * a) + fmax(y0 * b, y1 * b);
}
__device__ __forceinline__ double qd_plane_min(double a, double b, double x0, double x1, double y0, double y1) {
    return fmin(x0 * a, x1 * a) + fmin(y0 * b, y1 * b);
}

// Partial sums of one lattice point given as nibble code (dot j in nibble N-1-j, digit = c_j - lo_j):
//   pn = sum_i [t_i^2 + g_i x_i],  x = c - m,  t_i = sum_{j<=i} U[j][i] x_j       (= E(c) - E(m))
//   pa = lamx . (c - cg),  pb = lamy . (c - cg)
template <int N>
__device__ __forceinline__ void qd_tile_point(const QdTileLds& T, const double* __restrict__ U, uint32_t code,
                                              double& pn, double& pa, double& pb) {
    double xv[N];
    pn = 0.0; pa = 0.0; pb = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = T.lo[i] + (int)((code >> (4 * (N - 1 - i))) & 15u);
        xv[i] = (double)c - T.m[i];
        double t = 0.0;
#pragma unroll
        for (int j = 0; j <= i; ++j) t = fma(U[j * N + i], xv[j], t);
        pn += fma(t, t, T.g[i] * xv[i]);
        const double dc = (double)(c - T.cg[i]);
        pa = fma(T.lamx[i], dc, pa);
        pb = fma(T.lamy[i], dc, pb);
    }
}

// A pixel the tile search leaves to the exact per-pixel search: its front end is done already (v', the reference's 50-step
// projected gradient: ~5 k instructions of the 44 k a redone tile costs), so the results travel in the record's energy slots
// (E[0..N-1] = n_cont, E[8..8+N-1] = v', E[16] = 1 / s_a; rec->vpp / rec->tc are written by the front end) and the redo pass
// (qd_k_candidates, only_flagged) starts at the search.  Same code, same bits.
template <int N>
__device__ __forceinline__ void qd_tile_hand_over(QdPixelRec* rec, const double* vd, const double* ncont, double isa) {
#pragma unroll
    for (int i = 0; i < N; ++i) { rec->E[i] = ncont[i]; rec->E[8 + i] = vd[i]; }
    rec->E[16] = isa;
    rec->nvalid = QD_T_REDO;
}

// ---------------------------------------------------------------------------------------------
// grid = (tiles, C, n_env), block = 64 (one wavefront = one 8x8 pixel tile)
// stats (optional, 16 counters): tiles, tiles redone whole, lanes redone, sum of |S|, lanes redone for < 32 valid
// states in S; [8 + reason]: tiles redone by reason (1 ranges, 2 seeds, 3 frontier overflow, 4 too few leaves, 5 |S|)
// ---------------------------------------------------------------------------------------------
// Writes one QdPixelRec per pixel for the ground-state kernels (qd_k_gs_*).
template <int N>
__global__ void __launch_bounds__(64)
qd_k_tile(const int* __restrict__ env_ids, int env_base, int R, const double* __restrict__ params,
          const double* __restrict__ state, QdPixelRec* __restrict__ recs, int sort_output, int noise_flags,
          unsigned long long* __restrict__ stats) {
    constexpr int G = N + 1, NB = N - 1, V = 2 * N;
    const QdLayout L = qd_layout(N);
    const int slot = blockIdx.z;
    const int e = env_ids ? env_ids[env_base + slot] : env_base + slot;
    const int ch = blockIdx.y;
    const int P = R * R;
    const int tiles_x = (R + 7) >> 3;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int lane = threadIdx.x;
    const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    const bool inside = px < R && py < R;
    const int x = px < R ? px : R - 1, y = py < R ? py : R - 1;          // lanes off the image repeat an edge pixel
    const int p = y * R + x;
    __shared__ QdTileLds T;
    const double* spar = params + (size_t)e * L.size;
    const double* sst = state + (size_t)e * L.s_size;
    if (qd_radial_replaced(spar, sst, L, ch, noise_flags)) return;        // image will be pure noise: nothing to solve
    QdPixelRec* rec = recs + ((size_t)slot * (N - 1) + ch) * P + p;
    const double* A = spar + L.cdd_inv;                                   // row-major, lda = G
    const double* U = spar + L.ufac;

    // ---- 1. per-lane front end ------------------------------------------------------------
    double vd[N], ncont[N], isa;
    double vpp[G], tc[NB];
    {
        double v_ext[V];
        qd_pixel_voltages<N>(spar, sst, ch, R, x, y, v_ext, vpp, tc);
        if (inside) {
#pragma unroll
            for (int i = 0; i < G; ++i) rec->vpp[i] = vpp[i];
#pragma unroll
            for (int b = 0; b < NB; ++b) rec->tc[b] = tc[b];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) vd[i] = vpp[i];
        qd_pixel_continuous<N>(spar, v_ext, vd, ncont, &isa);
    }
    int fl[N];
#pragma unroll
    for (int i = 0; i < N; ++i) fl[i] = (int)floor(ncont[i]);

#if defined(QD_TILE_STOP) && QD_TILE_STOP == 1
    if (inside) rec->E[0] = ncont[0] + vd[N - 1] + tc[0] + (double)fl[1];                 // diagnostic build: time split of the kernel (scripts/ab_build.sh)
    return;
#endif
    // ---- 2. tile geometry and the affine model of v' ------------------------------------------
    const int xr = min(tx * 8 + 3, R - 1), yr = min(ty * 8 + 3, R - 1);  // reference pixel = lane 27 (clamped)
    const double xs = (double)(x - xr), ys = (double)(y - yr);
    const double x0 = (double)(tx * 8 - xr), x1 = (double)(min(tx * 8 + 7, R - 1) - xr);
    const double y0 = (double)(ty * 8 - yr), y1 = (double)(min(ty * 8 + 7, R - 1) - yr);
    const double Xh = fmax(-x0, x1), Yh = fmax(-y0, y1);
    bool fail = false;
    int why = 0;                                                           // first reason a tile was handed over (statistics)
    double rho;
    double v0[N];                                                          // v' of the reference pixel (uniform)
    {
        double r1 = 0.0;
        double dxv[N], dyv[N], mref[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            v0[i] = qd_rl(vd[i], 27);
            mref[i] = qd_rl(ncont[i], 27);
            // one pixel step in x / y from the tile's first pixel, which is always on the image (lanes off the image
            // repeat an edge pixel, so the step is 0 exactly when the tile is one column / row wide)
            dxv[i] = qd_rl(vd[i], 1) - qd_rl(vd[i], 0);
            dyv[i] = qd_rl(vd[i], 8) - qd_rl(vd[i], 0);
            const double r = ((vd[i] - v0[i]) - xs * dxv[i]) - ys * dyv[i];
            r1 += fabs(r);
            if (lane == 0) T.m[i] = mref[i];
        }
        // per-level constants: lane i < N does row i of A
        double colsum = 0.0;
        {
            const int row = lane < N ? lane : 0;
            double tg = 0.0, tx_ = 0.0, ty_ = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double a = A[row * G + j];
                tg = fma(a, mref[j] - v0[j], tg);
                tx_ = fma(a, dxv[j], tx_);
                ty_ = fma(a, dyv[j], ty_);
                colsum += fabs(a);
            }
            if (lane < N) { T.g[lane] = 2.0 * tg; T.lamx[lane] = -2.0 * tx_; T.lamy[lane] = -2.0 * ty_; }
        }
        const double anorm = qd_wmax_d(lane < N ? colsum : 0.0);
        rho = qd_wmax_d(r1) * anorm;
    }
    // floor ranges: candidate digits of the tile [lo, hi], valid in EVERY pixel [alo, ahi]
    int lo_[N], alo_[N], ahi_[N], nd_[N];
    unsigned long long allvalid_count = 1;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int fmn = qd_wmin_i(fl[i]), fmx = qd_wmax_i(fl[i]);
        lo_[i] = max(fmn - 1, 0); const int hi = fmx + 2;
        alo_[i] = max(fmx - 1, 0); ahi_[i] = fmn + 2;
        nd_[i] = hi - lo_[i] + 1;
        if (nd_[i] > 8 || ahi_[i] < alo_[i]) { fail = true; why = 1; }      // digits must fit 3 bits (packed range tests)
        allvalid_count *= (unsigned long long)(ahi_[i] >= alo_[i] ? ahi_[i] - alo_[i] + 1 : 0);
        if (lane == 0) { T.lo[i] = lo_[i]; T.nd[i] = nd_[i]; }
    }
    if (allvalid_count < (unsigned long long)QD_K) { fail = true; why = 1; }
    __builtin_amdgcn_wave_barrier();

#if defined(QD_TILE_STOP) && QD_TILE_STOP == 2
    if (inside) rec->E[0] = rho + v0[0];                 // diagnostic build: time split of the kernel (scripts/ab_build.sh)
    return;
#endif
    // ---- 3. greedy lattice point cg (uniform), option costs, product set of seeds -------------------
    double Tpp = INFINITY, pn_ref = 0.0, margin = 0.0, Ecg = 0.0;
    if (!fail) {
        int cg[N];
        {
            double dm[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double s = 0.0;
#pragma unroll
                for (int j = 0; j < i; ++j) s = fma(U[j * N + i], dm[j], s);
                const double ui = spar[L.uinv + i];
                const double cstar = T.m[i] + fma(-s, ui, -(0.5 * T.g[i]) * (ui * ui));
                double c = rint(cstar);
                c = fmin(fmax(c, (double)alo_[i]), (double)ahi_[i]);
                if (!(c == c)) c = (double)alo_[i];
                cg[i] = (int)c; dm[i] = c - T.m[i];
                if (lane == 0) { T.cg[i] = cg[i]; T.optval[i][0] = cg[i]; }
            }
            // q = A (cg - v0) by rows
            {
                const int row = lane < N ? lane : 0;
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) t = fma(A[row * G + j], (double)cg[j] - v0[j], t);
                if (lane < N) T.q[lane] = t;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // canonical energy of cg at my pixel: the absolute scale of the record's energies and of their round-off
        {
            double dd[N];
#pragma unroll
            for (int i = 0; i < N; ++i) dd[i] = (double)cg[i] - vd[i];
            double E = 0.0;
#pragma unroll 1
            for (int i = 0; i < N; ++i) {
                const double t = qd_dotN<N>(A + i * G, dd);
                double di = dd[0];
#pragma unroll
                for (int j = 1; j < N; ++j) di = (j == i) ? dd[j] : di;
                E = fma(di, t, E);
            }
            Ecg = E;
        }
        // suffix sums: tail (linear term), sla / slb (tile-width term) over the levels after i
        {
            double acc = 0.0, sa = 0.0, sb = 0.0;
#pragma unroll
            for (int j = N - 1; j >= 0; --j) {
                if (lane == 0) { T.tail[j] = acc; T.sla[j] = sa; T.slb[j] = sb; }
                const double gj = T.g[j], mj = T.m[j];
                const double cmin = (double)lo_[j], cmax = (double)(lo_[j] + nd_[j] - 1);
                acc += fmin(gj * (cmin - mj), gj * (cmax - mj));
                const double span = fmax((double)(cg[j] - lo_[j]), (double)(lo_[j] + nd_[j] - 1 - cg[j]));
                sa += fabs(T.lamx[j]) * span; sb += fabs(T.lamy[j]) * span;
            }
        }
        // option cost of (dot i = lane >> 3, value alo_i + (lane & 7)): exact energy change from cg
        double cost = INFINITY; int oi = lane >> 3, oc = 0;
        {
            int a_lo = 0, a_hi = -1, cgi = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) if (i == oi) { a_lo = alo_[i]; a_hi = ahi_[i]; cgi = cg[i]; }
            oc = a_lo + (lane & 7);
            if (oi < N && oc <= a_hi && oc != cgi) {
                const double dc = (double)(oc - cgi);
                cost = fma(2.0 * dc, T.q[oi], A[oi * G + oi] * dc * dc);
            }
        }
        // cheapest options first, as long as the product set stays within 64 lanes
        int nopt[N];
#pragma unroll
        for (int i = 0; i < N; ++i) nopt[i] = 1;
        int prod = 1;
        for (int it = 0; it < 24; ++it) {
            const double mn = qd_wmin_d(cost);
            if (!(mn < INFINITY)) break;
            const unsigned long long who = __ballot(cost == mn);
            const int wl = __builtin_ctzll(who);
            const int wi = wl >> 3;
            int ni = 1;
#pragma unroll
            for (int i = 0; i < N; ++i) if (i == wi) ni = nopt[i];
            const int wc = __builtin_amdgcn_readlane(oc, wl);
            if (ni < 4 && (prod / ni) * (ni + 1) <= 64) {
                if (lane == 0) T.optval[wi][ni] = wc;
#pragma unroll
                for (int i = 0; i < N; ++i) if (i == wi) nopt[i] = ni + 1;
                prod = (prod / ni) * (ni + 1);
            }
            if (lane == wl) cost = INFINITY;
        }
        __builtin_amdgcn_wave_barrier();
        // lane -> one member of the product set (mixed radix, dot 0 slowest)
        uint32_t scode_seed = 0;
        {
            int rem = lane;
#pragma unroll
            for (int i = N - 1; i >= 0; --i) {
                const int sel = rem % nopt[i]; rem /= nopt[i];
                const int c = T.optval[i][sel];
                scode_seed |= (uint32_t)(c - lo_[i]) << (4 * (N - 1 - i));
            }
        }
        double pn, pa, pb;
        qd_tile_point<N>(T, U, scode_seed, pn, pa, pb);
        pn_ref = qd_rl(pn, 0);                                              // lane 0 = cg itself
        const double escale = qd_wmax_d(fabs(pn)) + fabs(pn_ref) + 1.0;
        int maxdc = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) maxdc = max(maxdc, max(cg[i] - lo_[i], lo_[i] + nd_[i] - 1 - cg[i]));
        // margin of every tile bound: affine residual of v', arithmetic of the relative energies, and the round-off
        // of the CANONICAL absolute energies that define the reference order (~N^2 ulps of |E|)
        margin = 2.0 * rho * (double)maxdc + 1e-11 * escale + 1e-12 * qd_wmax_d(fabs(Ecg));
        const double Mh = (lane < prod) ? (pn - pn_ref) + qd_plane_max(pa, pb, x0, x1, y0, y1) + margin : INFINITY;
        if (prod < QD_K) { fail = true; why = 2; }
        else {
            // 32nd smallest of the seeds' maxima
            int below = 0;
            for (int j = 0; j < 64; ++j) {
                const double o = qd_rl(Mh, j);
                below += (o < Mh || (o == Mh && j < lane)) ? 1 : 0;
            }
            const unsigned long long sel = __ballot(below == QD_K - 1 && lane < prod);
            Tpp = sel ? qd_rl(Mh, __builtin_ctzll(sel)) : INFINITY;
            if (!(Tpp < INFINITY)) { fail = true; why = 2; }
        }
    }

#if defined(QD_TILE_STOP) && QD_TILE_STOP == 3
    if (inside) rec->E[0] = Tpp + pn_ref + margin + Ecg;                 // diagnostic build: time split of the kernel (scripts/ab_build.sh)
    return;
#endif
    // ---- 4. level-synchronous branch and bound -------------------------------------------------
    int nfront = 0, cur = 0;
    if (!fail) {
        if (lane == 0) { T.u.f.pn[0][0] = 0.0; T.u.f.code[0][0] = 0u; }
        nfront = 1;
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int i = 0; i < N && !fail; ++i) {
            const double mi = T.m[i], gi = T.g[i], lxi = T.lamx[i], lyi = T.lamy[i];
            const double tl = T.tail[i], sla = T.sla[i], slb = T.slb[i];
            const double uii = U[i * N + i];
            const int loi = T.lo[i], ndi = T.nd[i], cgi = T.cg[i];
            const unsigned sh = 4u * (unsigned)(N - 1 - i);
            int nout = 0;
#pragma unroll 1
            for (int base = 0; base < nfront; base += 64) {
                const int node = base + lane;
                const bool act = node < nfront;
                const uint32_t code = act ? T.u.f.code[cur][node] : 0u;
                const double pn0 = act ? T.u.f.pn[cur][node] : 0.0;
                // s = sum_{j<i} U[j][i] (c_j - m_j), pa / pb of the fixed part
                double s = 0.0, pa0 = 0.0, pb0 = 0.0;
#pragma unroll 1
                for (int j = 0; j < i; ++j) {
                    const int c = T.lo[j] + (int)((code >> (4 * (N - 1 - j))) & 15u);
                    s = fma(U[j * N + i], (double)c - T.m[j], s);
                    const double dc = (double)(c - T.cg[j]);
                    pa0 = fma(T.lamx[j], dc, pa0); pb0 = fma(T.lamy[j], dc, pb0);
                }
#pragma unroll 1
                for (int k = 0; k < ndi; ++k) {
                    const double xk = (double)(loi + k) - mi;
                    const double t = fma(uii, xk, s);
                    const double pn2 = pn0 + fma(t, t, gi * xk);
                    const double dc = (double)(loi + k - cgi);
                    const double pa2 = fma(lxi, dc, pa0), pb2 = fma(lyi, dc, pb0);
                    const double W = fma(Xh, fabs(pa2) + sla, Yh * (fabs(pb2) + slb));
                    const double LB = ((pn2 - pn_ref) + tl) - W - margin;
                    const bool keep = act && (LB <= Tpp);
                    const unsigned long long km = __ballot(keep);
                    const int pos = nout + qd_lane_prefix(km);
                    if (keep && pos < QD_T_FCAP) {
                        T.u.f.pn[cur ^ 1][pos] = pn2;
                        T.u.f.code[cur ^ 1][pos] = code | ((uint32_t)k << sh);
                    }
                    nout += __builtin_popcountll(km);
                }
            }
            if (nout > QD_T_FCAP) { fail = true; why = 3; }
            nfront = nout; cur ^= 1;
            __builtin_amdgcn_wave_barrier();
        }
    }

#if defined(QD_TILE_STOP) && QD_TILE_STOP == 4
    if (inside) rec->E[0] = (double)nfront + T.u.f.pn[cur][lane];                 // diagnostic build: time split of the kernel (scripts/ab_build.sh)
    return;
#endif
    // ---- 5. T (bisection on the count of maxima) and the superset S --------------------------------
    int nS = 0, nSfront = 0;
    if (!fail && nfront < QD_K) { fail = true; why = 4; }
    if (!fail) {
        constexpr int CH = QD_T_FCAP / 64;
        double Mh[CH];
        // packed all-valid range test: digit >= alo - lo and <= ahi - lo in every nibble (digits < 8)
        uint32_t vlo = 0, vhi = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            vlo |= (uint32_t)(alo_[i] - lo_[i]) << (4 * (N - 1 - i));
            vhi |= (uint32_t)(ahi_[i] - lo_[i]) << (4 * (N - 1 - i));
        }
        double mlo = INFINITY;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            Mh[c] = INFINITY;
            const int node = c * 64 + lane;
            if (c * 64 < nfront && node < nfront) {
                const uint32_t code = T.u.f.code[cur][node];
                const bool av = ((((code | 0x88888888u) - vlo) & 0x88888888u) == 0x88888888u) &&
                                ((((vhi | 0x88888888u) - code) & 0x88888888u) == 0x88888888u);
                if (av) {
                    double pn, pa, pb;
                    qd_tile_point<N>(T, U, code, pn, pa, pb);
                    Mh[c] = (pn - pn_ref) + qd_plane_max(pa, pb, x0, x1, y0, y1) + margin;
                }
                mlo = fmin(mlo, Mh[c]);
            }
        }
        double tlo = qd_wmin_d(mlo), thi = Tpp;                             // count(M <= thi) >= 32 holds by construction
        for (int it = 0; it < 14; ++it) {
            const double mid = 0.5 * (tlo + thi);
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) cnt += __builtin_popcountll(__ballot(Mh[c] <= mid));
            if (cnt >= QD_K) thi = mid; else tlo = mid;
        }
        const double Tt = thi;
        // S = {m <= T}: the states whose maximum is within T (kept in most pixels) fill S from the front, the others
        // from the back, so that the per-lane buffers below fill up with likely keepers first
        int nfrontS = 0, nbackS = 0;
#pragma unroll 1
        for (int c = 0; c < CH; ++c) {
            if (c * 64 >= nfront) break;
            const int node = c * 64 + lane;
            bool take = false, likely = false;
            uint32_t code = 0; double D = 0.0, pa = 0.0, pb = 0.0;
            if (node < nfront) {
                code = T.u.f.code[cur][node];
                double pn;
                qd_tile_point<N>(T, U, code, pn, pa, pb);
                D = pn - pn_ref;
                take = (D + qd_plane_min(pa, pb, x0, x1, y0, y1) - margin) <= Tt;
                double mhc = INFINITY;
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) if (cc == c) mhc = Mh[cc];
                likely = take && (mhc <= Tt);
            }
            const unsigned long long kf = __ballot(likely), kb = __ballot(take && !likely);
            const int tot = nfrontS + nbackS + __builtin_popcountll(kf) + __builtin_popcountll(kb);
            if (tot <= QD_T_SCAP) {
                int pos = -1;
                if (likely) pos = nfrontS + qd_lane_prefix(kf);
                else if (take) pos = QD_T_SCAP - 1 - (nbackS + qd_lane_prefix(kb));
                if (pos >= 0) { T.scode[pos] = code; T.sD[pos] = D; T.sa[pos] = pa; T.sb[pos] = pb; }
            }
            nfrontS += __builtin_popcountll(kf); nbackS += __builtin_popcountll(kb);
        }
        nS = nfrontS + nbackS;
        nSfront = nfrontS;
        if (nS > QD_T_SCAP || nS < QD_K) { fail = true; why = 5; }
        __builtin_amdgcn_wave_barrier();
    }
    if (stats && lane == 0) {
        atomicAdd(&stats[0], 1ull);
        if (fail) { atomicAdd(&stats[1], 1ull); atomicAdd(&stats[8 + why], 1ull); }
        atomicAdd(&stats[3], (unsigned long long)nS);
    }
    if (fail) {                                                            // the whole tile goes to the exact per-pixel search
        if (inside) qd_tile_hand_over<N>(rec, vd, ncont, isa);
        return;
    }

#if defined(QD_TILE_STOP) && QD_TILE_STOP == 5
    if (inside) rec->E[0] = (double)nS + T.sD[lane];                 // diagnostic build: time split of the kernel (scripts/ab_build.sh)
    return;
#endif
    // ---- 6. per lane: energies of S, own validity box, 32 lowest ------------------------------------
    uint32_t blo = 0, bhi = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        blo |= (uint32_t)(max(fl[i] - 1, 0) - lo_[i]) << (4 * (N - 1 - i));
        bhi |= (uint32_t)(fl[i] + 2 - lo_[i]) << (4 * (N - 1 - i));
    }
    double gE[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}; int gS[4] = {0, 8, 16, 24};
    double maxE = -INFINITY; int maxS = 0, maxG = 0, count = 0;
    double eout = INFINITY;                                                  // lowest energy NOT kept
    for (int si = 0; si < nS; ++si) {
        const int s = si < nSfront ? si : QD_T_SCAP - 1 - (si - nSfront);      // front part, then the back part
        const uint32_t code = T.scode[s];
        const bool valid = ((((code | 0x88888888u) - blo) & 0x88888888u) == 0x88888888u) &&
                           ((((bhi | 0x88888888u) - code) & 0x88888888u) == 0x88888888u);
        const double en = fma(xs, T.sa[s], fma(ys, T.sb[s], T.sD[s]));
        if (!valid) continue;
        if (count < QD_K) {
            T.u.k.e[count][lane] = en; T.u.k.id[count][lane] = (uint16_t)s;
            count++;
            if (count == QD_K) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    double me = -INFINITY; int ms = g * 8;
#pragma unroll
                    for (int t = 0; t < 8; ++t) { const double v = T.u.k.e[g * 8 + t][lane]; if (v > me) { me = v; ms = g * 8 + t; } }
                    gE[g] = me; gS[g] = ms;
                }
                maxE = gE[0]; maxS = gS[0]; maxG = 0;
#pragma unroll
                for (int g = 1; g < 4; ++g) if (gE[g] > maxE) { maxE = gE[g]; maxS = gS[g]; maxG = g; }
            }
        } else if (en < maxE) {
            eout = fmin(eout, maxE);                                        // the evicted state
            T.u.k.e[maxS][lane] = en; T.u.k.id[maxS][lane] = (uint16_t)s;
            {
                double me = -INFINITY; int ms = maxG * 8;
#pragma unroll
                for (int t = 0; t < 8; ++t) { const double v = T.u.k.e[maxG * 8 + t][lane]; if (v > me) { me = v; ms = maxG * 8 + t; } }
#pragma unroll
                for (int g = 0; g < 4; ++g) if (g == maxG) { gE[g] = me; gS[g] = ms; }
            }
            maxE = gE[0]; maxS = gS[0]; maxG = 0;
#pragma unroll
            for (int g = 1; g < 4; ++g) if (gE[g] > maxE) { maxE = gE[g]; maxS = gS[g]; maxG = g; }
        } else {
            eout = fmin(eout, en);
        }
    }
    // a lane whose boundary is closer than what the arithmetic can tell apart is redone exactly
    const double amb = 2.0 * margin;
    const bool redo = (count < QD_K) || !(eout - maxE > amb);
    if (stats) {
        const unsigned long long rm = __ballot(redo && inside), rc = __ballot(count < QD_K && inside);
        if (lane == 0 && rm) { atomicAdd(&stats[2], (unsigned long long)__builtin_popcountll(rm)); atomicAdd(&stats[4], (unsigned long long)__builtin_popcountll(rc)); }
    }
    if (!inside) return;
    if (redo) { qd_tile_hand_over<N>(rec, vd, ncont, isa); return; }

#if defined(QD_TILE_STOP) && QD_TILE_STOP == 6
    if (inside) rec->E[0] = maxE + eout + (double)count;                 // diagnostic build: time split of the kernel (scripts/ab_build.sh)
    return;
#endif
    // ---- 7. the record ---------------------------------------------------------------------------------------
    {
    // reference index: digit = c - floor + 1 in base 4, dot 0 most significant
    uint32_t off = 0;                                                         // lo - fl + 1 per dot, biased by +8 in nibbles
#pragma unroll
    for (int i = 0; i < N; ++i) off |= (uint32_t)(lo_[i] - fl[i] + 1 + 4) << (4 * (N - 1 - i));
    if (sort_output) {
        // validate mode: canonical energies, reference order (E, idx)
        for (int k = 0; k < QD_K; ++k) {
            const uint32_t code = T.scode[T.u.k.id[k][lane]];
            double dd[N];
#pragma unroll
            for (int i = 0; i < N; ++i) dd[i] = (double)(lo_[i] + (int)((code >> (4 * (N - 1 - i))) & 15u)) - vd[i];
            double E = 0.0;
#pragma unroll 1
            for (int i = 0; i < N; ++i) {
                const double t = qd_dotN<N>(A + i * G, dd);
                double di = dd[0];
#pragma unroll
                for (int j = 1; j < N; ++j) di = (j == i) ? dd[j] : di;
                E = fma(di, t, E);
            }
            T.u.k.e[k][lane] = E;
        }
        for (int i = 1; i < QD_K; ++i) {                                      // insertion sort by (E, code)
            const double E = T.u.k.e[i][lane]; const uint16_t id = T.u.k.id[i][lane];
            const uint32_t cd = T.scode[id];
            int pos = i;
            while (pos > 0) {
                const double ep = T.u.k.e[pos - 1][lane]; const uint16_t ip = T.u.k.id[pos - 1][lane];
                const bool less = (E < ep) || (E == ep && cd < T.scode[ip]);
                if (!less) break;
                T.u.k.e[pos][lane] = ep; T.u.k.id[pos][lane] = ip;
                --pos;
            }
            T.u.k.e[pos][lane] = E; T.u.k.id[pos][lane] = id;
        }
    }
    for (int k = 0; k < QD_K; ++k) {
        const uint32_t code = T.scode[T.u.k.id[k][lane]];
        const uint32_t dg = (code + off) - 0x44444444u;                       // nibbles: c - fl + 1 in 0..3 (no carries: each sum < 16)
        // 2-bit digits at 4-bit spacing -> packed base-4 index (dot 0 most significant): three mask-shift steps
        unsigned idx = dg & 0x33333333u;
        idx = (idx | (idx >> 2)) & 0x0F0F0F0Fu;
        idx = (idx | (idx >> 4)) & 0x00FF00FFu;
        idx = (idx | (idx >> 8)) & 0x0000FFFFu;
        rec->idx[k] = (uint16_t)idx;
        const double E = sort_output ? T.u.k.e[k][lane] : T.u.k.e[k][lane] + Ecg;
        rec->E[k] = E * isa;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) rec->fl[i] = fl[i];
    rec->nvalid = QD_K;
    }
}

#endif  // __HIPCC__

// qd_tile_ground.h -- ground state of the tunnel-coupled Hamiltonian, ONE PIXEL PER LANE,
// inside the tile kernel (qd_tile.h).  Reference: hamiltonian_build.py:12-137, ground_state.py:149-162.
//
// After the tile search every lane holds its 32 kept charge states as a bit mask over the tile's shared list S, and
// the diagonal of its Hamiltonian comes from the tile planes (two FMAs per state).  What differs between the pixels
// of a tile is only that diagonal, the tunnel couplings t_d and which states near the 32nd energy are kept -- the
// hop graph over S is the same for all of them.  So the wave builds the hop table of S once and then works through
// the distinct connected components ("structures") that any lane still needs:
//   * one lane (the leader) names a seed state; the component of the seed inside the leader's kept set is found by
//     a breadth-first walk over the shared hop table (uniform work);
//   * every lane whose kept set contains exactly this component joins in; the structure (members, edges, sqrt
//     factors) is uniform, only F_i and t_d are per lane;
//   * Gershgorin: a lane skips the solve if the component's lower bound exceeds its current upper bound of the
//     ground energy (lowest diagonal entry, or an eigenvalue already found) -- most components, in every regime;
//   * the solve is the one validated in round 1 (qd_groundstate.h): Lanczos from the all-ones vector, lowest
//     eigenvalue of the tridiagonal by Laguerre iteration from the left, eigenvector by inverse iteration, second
//     Lanczos pass for x = Q y -- but with one PIXEL per lane: the serial recurrences that every lane of a half-wave
//     used to repeat for one pixel now serve 64 pixels per instruction.
// The lowest eigenvalue over all components wins (ties: the component holding the lowest candidate index, as the
// reference order puts it first); <n> = sum_m x_m^2 s_m.
#pragma once

#if defined(__HIPCC__)

#define QD_T_NMAXC 14          // largest component solved here (a state has at most 2(N-1) = 14 hop neighbours)
#define QD_T_MAXSTRUCT 160     // structures examined per tile before the tile is handed to the per-pixel kernels

struct QdMask256 { unsigned long long w[4]; };
__device__ __forceinline__ bool qd_m_test(const QdMask256& m, int b) {
    unsigned long long w = m.w[0];
    w = (b >> 6) == 1 ? m.w[1] : w; w = (b >> 6) == 2 ? m.w[2] : w; w = (b >> 6) == 3 ? m.w[3] : w;
    return (w >> (b & 63)) & 1ull;
}
__device__ __forceinline__ void qd_m_set(QdMask256& m, int b) {
    const unsigned long long bit = 1ull << (b & 63);
    const int k = b >> 6;
    m.w[0] |= k == 0 ? bit : 0ull; m.w[1] |= k == 1 ? bit : 0ull; m.w[2] |= k == 2 ? bit : 0ull; m.w[3] |= k == 3 ? bit : 0ull;
}
__device__ __forceinline__ void qd_m_clear(QdMask256& m, int b) {
    const unsigned long long bit = ~(1ull << (b & 63));
    const int k = b >> 6;
    m.w[0] &= k == 0 ? bit : ~0ull; m.w[1] &= k == 1 ? bit : ~0ull; m.w[2] &= k == 2 ? bit : ~0ull; m.w[3] &= k == 3 ? bit : ~0ull;
}
__device__ __forceinline__ bool qd_m_any(const QdMask256& m) { return (m.w[0] | m.w[1] | m.w[2] | m.w[3]) != 0ull; }
__device__ __forceinline__ int qd_m_first(const QdMask256& m) {            // lowest set bit (mask must be non-empty)
    if (m.w[0]) return __builtin_ctzll(m.w[0]);
    if (m.w[1]) return 64 + __builtin_ctzll(m.w[1]);
    if (m.w[2]) return 128 + __builtin_ctzll(m.w[2]);
    return 192 + __builtin_ctzll(m.w[3]);
}
__device__ __forceinline__ int qd_m_count(const QdMask256& m) {
    return __builtin_popcountll(m.w[0]) + __builtin_popcountll(m.w[1]) + __builtin_popcountll(m.w[2]) + __builtin_popcountll(m.w[3]);
}
__device__ __forceinline__ int qd_m_rank(const QdMask256& m, int b) {      // set bits strictly below b
    int r = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int lo = k * 64;
        unsigned long long w = m.w[k];
        if (b <= lo) w = 0ull; else if (b < lo + 64) w &= (1ull << (b - lo)) - 1ull;
        r += __builtin_popcountll(w);
    }
    return r;
}
__device__ __forceinline__ QdMask256 qd_m_uniform(const QdMask256& m, int lane) {   // another lane's mask, as scalars
    QdMask256 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)m.w[k], lane);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(m.w[k] >> 32), lane);
        o.w[k] = ((unsigned long long)hi << 32) | lo;
    }
    return o;
}

// S is stored two-ended (likely keepers from the front, the rest from the back of the arrays); everything in the
// ground phase numbers its states logically 0..nS-1 and converts only to read the arrays.
__device__ __forceinline__ int qd_s_phys(int si, int nSfront) { return si < nSfront ? si : QD_T_SCAP - 1 - (si - nSfront); }
__device__ __forceinline__ int qd_s_logical(int phys, int nSfront) { return phys < nSfront ? phys : nSfront + (QD_T_SCAP - 1 - phys); }

struct QdTileGroundLds {
    uint32_t nbw[QD_T_SCAP][4];        // hop table of S: byte 2d = forward partner over pair (d, d+1), 2d+1 = backward; 255 = none
    int mS[16];                        // members of the current structure (indices into S, ascending)
    int ecnt[16];                      // directed edges per row
    unsigned char ej[QD_T_NMAXC][16], ed[QD_T_NMAXC][16];
    double esq[QD_T_NMAXC][16];
};

// lanes' view of the union area during the ground phase
struct QdTileGroundVec { double q[QD_T_NMAXC][64]; double tl[QD_MAXN][64]; };

// Build the hop table of S (once per tile).  codes: nibble N-1-i = c_i - lo_i.  pairnz: bit d set iff the pair (d, d+1)
// couples (t_d != 0; zero couplings do not link states, so tc = 0 stays exactly diagonal).
template <int N>
__device__ __forceinline__ void qd_tile_hop_table(const QdTileLds& T, QdTileGroundLds& Gd, int nS, int nSfront, unsigned pairnz) {
    const int lane = threadIdx.x;
    for (int base = 0; base < nS; base += 64) {
        const int j = base + lane;
        const uint32_t cj = j < nS ? T.scode[qd_s_phys(j, nSfront)] : 0u;
        uint32_t row[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (int k = 0; k < nS; ++k) {
            const uint32_t ck = T.scode[qd_s_phys(k, nSfront)];              // uniform
            const uint32_t Z = ((ck | 0x88888888u) - cj) ^ 0x88888888u;    // nibble 0: equal, 1: +1, 0xF: -1 (digits < 8)
            const int tz = __builtin_ctz(Z | 0x80000000u);
            const uint32_t Zs = Z >> tz;
            const bool fwd = Zs == 0xF1u, bwd = Zs == 0x1Fu;               // k = j - e_d + e_{d+1}  /  k = j + e_d - e_{d+1}
            if ((fwd || bwd) && (tz & 3) == 0) {
                const int qn = tz >> 2;                                      // nibble of dot d+1
                const int d = N - 2 - qn;
                if (d >= 0 && ((pairnz >> d) & 1u)) {
                    const int slot = 2 * d + (fwd ? 0 : 1);
                    const uint32_t sh = 8u * (unsigned)(slot & 3);
#pragma unroll
                    for (int w = 0; w < 4; ++w) if (w == (slot >> 2)) row[w] = (row[w] & ~(0xFFu << sh)) | ((uint32_t)k << sh);
                }
            }
        }
        if (j < nS) {
#pragma unroll
            for (int w = 0; w < 4; ++w) Gd.nbw[j][w] = row[w];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// occupation product of the hop in `slot` (2d forward: electron d -> d+1, 2d+1 backward) for the state with tile code `code`:
// H_ij = -t_d sqrt(n_from (n_to + 1)) with the occupations of the ROW state (hamiltonian_build.py:125-131)
template <int N>
__device__ __forceinline__ int qd_tile_hop_product(uint32_t code, int slot, const int* lo_) {
    const int d = slot >> 1;                                  // slot is a compile-time constant at every call site
    const int qn = N - 2 - d;                                 // nibble of dot d+1
    const int nd = (int)((code >> (4 * (qn + 1))) & 15u) + lo_[d], nd1 = (int)((code >> (4 * qn)) & 15u) + lo_[d + 1];
    return (slot & 1) ? nd1 * (nd + 1) : nd * (nd1 + 1);
}

// ---------------------------------------------------------------------------------------------------------
// Solve one uniform structure (n members, edges in Gd) for the lanes in `solve`: per lane lowest eigenvalue and
// eigenvector of  H = diag(F) + sum_edges -t_d sq (|i><j| + |j><i|).  Vectors with static indices live in
// registers; q is mirrored in LDS for the gathers of the matvec.
// Returns lam, x[] (normalised), and (VALIDATE) the residual ||Hx - lam x||_2.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int qd_uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double qd_uni_d(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// w = H v for the current structure: diagonal F, edges from the uniform lists in Gd, v mirrored in LDS for the gathers
template <int M>
__device__ __forceinline__ void qd_tile_matvec(const QdTileGroundLds& Gd, QdTileGroundVec& Vv, int n, const double (&F)[M],
                                               const double (&v)[M], double (&w)[M]) {
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < M; ++i) if (i < n) Vv.q[i][lane] = v[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < M; ++i) {
        w[i] = 0.0;
        if (i < n) {
            double acc = F[i] * v[i];
            const int cnt = qd_uni_i(Gd.ecnt[i]);
            for (int e = 0; e < cnt; ++e) {
                const int j = qd_uni_i((int)Gd.ej[i][e]), d = qd_uni_i((int)Gd.ed[i][e]);
                const double sq = qd_uni_d(Gd.esq[i][e]);
                acc = fma(-(Vv.tl[d][lane] * sq), Vv.q[j][lane], acc);
            }
            w[i] = acc;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// One Lanczos run from the all-ones vector.  PASS 1 records T (al, be, k); PASS 2 repeats exactly the same
// arithmetic (same operands, same bits) and accumulates x = sum_j y_j q_j for the first k steps.
template <int M, int PASS>
__device__ __forceinline__ void qd_tile_lanczos(const QdTileGroundLds& Gd, QdTileGroundVec& Vv, int n, bool solve,
                                                const double (&F)[M], double (&al)[M], double (&be)[M], int& k,
                                                const double (&y)[M], double (&x)[M]) {
    double q[M], qp[M], w[M];
    double q0;
    { double s_, r_; qd_sqrt_rsqrt((double)n, s_, r_); q0 = r_; }
#pragma unroll
    for (int i = 0; i < M; ++i) { q[i] = (i < n) ? q0 : 0.0; qp[i] = 0.0; }
    double bp = 0.0, anorm = 0.0;
    bool done = !solve;
    if (PASS == 1) k = solve ? 0 : 1;
    for (int j = 0; j < n; ++j) {
        if (!__any(!done)) break;
        qd_tile_matvec<M>(Gd, Vv, n, F, q, w);
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) if (i < n) a = fma(q[i], w[i], a);
        double b2 = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) if (i < n) { w[i] = w[i] - a * q[i] - bp * qp[i]; b2 = fma(w[i], w[i], b2); }
        double b = 0.0, ib = 0.0;
        if (b2 > 0.0) qd_sqrt_rsqrt(b2, b, ib);
        if (!done) {
            anorm = fmax(anorm, fmax(fabs(a), b));
            bool last;
            if (PASS == 1) {
                last = (j + 1 >= n) || !(b > 1e-13 * anorm);
#pragma unroll
                for (int r = 0; r < M; ++r) if (r == j) { al[r] = a; be[r] = last ? 0.0 : b; }     // uniform j: one static store
                k = j + 1;
            } else {
                double yj = 0.0;
#pragma unroll
                for (int r = 0; r < M; ++r) if (r == j) yj = y[r];
#pragma unroll
                for (int i = 0; i < M; ++i) if (i < n) x[i] = fma(yj, q[i], x[i]);
                last = j + 1 >= k;
            }
            if (last) done = true;
            else {
#pragma unroll
                for (int i = 0; i < M; ++i) if (i < n) { qp[i] = q[i]; q[i] = w[i] * ib; }
                bp = b;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Solve one uniform structure (n members, edges in Gd) for the lanes in `solve`: per lane lowest eigenvalue and
// eigenvector of  H = diag(F) + sum_edges -t_d sq (|i><j| + |j><i|).  Vectors with static indices live in
// registers; the vector being multiplied is mirrored in LDS for the gathers of the matvec.
// Returns lam, x[] (normalised), and (VALIDATE) the residual ||Hx - lam x||_2.
// ---------------------------------------------------------------------------------------------------------
template <int N, bool VALIDATE>
__device__ __forceinline__ void qd_tile_solve(const QdTileGroundLds& Gd, QdTileGroundVec& Vv, int n, bool solve,
                                              const double (&F)[QD_T_NMAXC], double& lam_out, double (&x)[QD_T_NMAXC],
                                              double& resid_out) {
    constexpr int M = QD_T_NMAXC;
    double al[M], be[M], y[M];
    int k = 1;
#pragma unroll
    for (int i = 0; i < M; ++i) { al[i] = 0.0; be[i] = 0.0; y[i] = 0.0; x[i] = 0.0; }
    qd_tile_lanczos<M, 1>(Gd, Vv, n, solve, F, al, be, k, y, x);
    if (!solve) al[0] = F[0];
#if defined(QD_TILE_ABLATE) && QD_TILE_ABLATE == 4
    lam_out = al[0]; resid_out = 0.0; x[0] = 1.0; return;         // diagnostic: Lanczos pass 1 only
#endif
    // ---- lowest eigenvalue of T: Laguerre from the left (as qd_groundstate.h 6a) ----
    double lo = INFINITY, hi = INFINITY, bmax = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        if (i < k) {
            const double bprev = i > 0 ? fabs(be[i > 0 ? i - 1 : 0]) : 0.0;
            const double bme = fabs(be[i]);
            lo = fmin(lo, al[i] - bprev - bme);
            hi = fmin(hi, al[i]);
            bmax = fmax(bmax, bme);
        }
    }
    const double tscale = fmax(fmax(fabs(lo), fabs(hi)), bmax);
    double xl = lo - (1e-3 * tscale + 1e-300);
    {
        bool conv = k <= 1;
        const double dk = (double)k;
        double sprev = 0.0;
        for (int it = 0; it < 48; ++it) {
            if (!__any(!conv)) break;
            double p0 = 1.0, p1 = 1.0, d0 = 0.0, d1 = 0.0, e0 = 0.0, e1 = 0.0, bprev = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                if (i >= n) break;
                if (i < k) {
                    const double a_ = al[i] - xl;
                    const double b2_ = bprev * bprev;
                    double p2_, d2_, e2_;
                    if (i == 0) { p2_ = a_; d2_ = -1.0; e2_ = 0.0; }
                    else {
                        p2_ = fma(a_, p1, -(b2_ * p0));
                        d2_ = fma(a_, d1, -(b2_ * d0)) - p1;
                        e2_ = fma(a_, e1, -(b2_ * e0)) - 2.0 * d1;
                    }
                    p0 = p1; p1 = p2_; d0 = d1; d1 = d2_; e0 = e1; e1 = e2_;
                    bprev = be[i];
                    if (i & 1) {                         // every second row: two rows of a T with entries ~1e44 stay inside the double range, four do not
                        const double ap_ = fabs(p1);
                        double sc_ = 1.0;
                        if (ap_ > 1e100) sc_ = 1e-100; else if (ap_ < 1e-100 && ap_ > 0.0) sc_ = 1e100;
                        if (sc_ != 1.0) { p0 *= sc_; p1 *= sc_; d0 *= sc_; d1 *= sc_; e0 *= sc_; e1 *= sc_; }
                    }
                }
            }
            if (!conv) {
                if (p1 == 0.0) conv = true;
                else {
                    const double ip = qd_rcp(p1);
                    const double G_ = d1 * ip, E_ = e1 * ip;
                    double disc = (dk - 1.0) * ((dk - 1.0) * G_ * G_ - dk * E_);
                    double sq = 0.0, rs_ = 0.0;
                    if (disc > 0.0) qd_sqrt_rsqrt(disc, sq, rs_);
                    const double den = (G_ < 0.0) ? G_ - sq : G_ + sq;
                    const double xn = (den != 0.0) ? fma(-dk, qd_rcp(den), xl) : xl;
                    if (!(xn > xl)) conv = true;
                    else {
                        const double st = xn - xl, tol = 4e-16 * fmax(fabs(xn), fabs(xl));
                        if (st <= tol) conv = true;
                        const double s2 = st * st, p3 = sprev * sprev * sprev;
                        if (100.0 * s2 * s2 <= tol * p3) conv = true;
                        sprev = st;
                        xl = xn;
                    }
                }
            }
        }
    }
    const double lam = (k <= 1) ? al[0] : xl;
#if defined(QD_TILE_ABLATE) && QD_TILE_ABLATE == 5
    lam_out = lam; resid_out = 0.0; x[0] = 1.0; return;           // diagnostic: + Laguerre
#endif
    // ---- eigenvector of T: inverse iteration on (T - sigma) = L D L^T, sigma just below lam ----
    {
        const double sig = (k <= 1) ? lam : xl - 2e-16 * tscale;
        const double tiny = 1e-300 + 1e-18 * fmax(fabs(sig), fabs(lam));
        double rd[M], lf[M];
        double bprev = 0.0, rdp = 1.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            rd[i] = 1.0; lf[i] = 0.0; y[i] = 0.0;
            if (i < k) {
                double di_ = al[i] - sig;
                if (i > 0) { const double l_ = bprev * rdp; di_ = di_ - l_ * bprev; lf[i] = l_; }
                if (!(di_ > tiny)) di_ = tiny;
                rdp = qd_rcp(di_);
                rd[i] = rdp;
                bprev = be[i];
            }
        }
#pragma unroll
        for (int iter = 0; iter < 2; ++iter) {
            double zprev = 0.0;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                if (i < k) {
                    const double rhs_ = (iter == 0) ? (i == 0 ? 1.0 : 0.0) : y[i];      // from e_1, not ones: ghost copies then add up coherently (qd_groundstate.h, 6b)
                    const double z_ = (i == 0) ? rhs_ : rhs_ - lf[i] * zprev;
                    y[i] = z_ * rd[i];
                    zprev = z_;
                }
            }
            double ynext = 0.0, lnext = 0.0, nrm = 0.0;            // L^T y = w, rows k-1 .. 0: y_i = w_i - l_{i+1} y_{i+1}
#pragma unroll
            for (int i = M - 1; i >= 0; --i) {
                if (i < k) {
                    const double y_ = y[i] - lnext * ynext;
                    y[i] = y_;
                    nrm = fma(y_, y_, nrm);
                    ynext = y_;
                    lnext = lf[i];
                }
            }
            double inv = 1.0, sn_ = 0.0;
            if (nrm > 0.0) qd_sqrt_rsqrt(nrm, sn_, inv);
#pragma unroll
            for (int i = 0; i < M; ++i) y[i] = y[i] * inv;
        }
    }
#if defined(QD_TILE_ABLATE) && QD_TILE_ABLATE == 6
    lam_out = lam; resid_out = 0.0; x[0] = y[0]; return;          // diagnostic: + inverse iteration
#endif
    // ---- Lanczos pass 2: x = sum_j y_j q_j ----
    qd_tile_lanczos<M, 2>(Gd, Vv, n, solve, F, al, be, k, y, x);
    if (solve) {
        double nx = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) nx = fma(x[i], x[i], nx);
        double snx = 0.0, inx = 1.0;
        if (nx > 0.0) qd_sqrt_rsqrt(nx, snx, inx);
#pragma unroll
        for (int i = 0; i < M; ++i) x[i] *= inx;
    } else {
#pragma unroll
        for (int i = 0; i < M; ++i) x[i] = (i == 0) ? 1.0 : 0.0;
    }
    lam_out = lam;
    resid_out = 0.0;
    if constexpr (VALIDATE) {
        double w[M];
        qd_tile_matvec<M>(Gd, Vv, n, F, x, w);
        double r2 = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) { const double r = w[i] - lam * x[i]; r2 = fma(r, r, r2); }
        resid_out = sqrt(r2);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Ground phase of one tile.  Per lane in: kept set KM (bits over S), `alive` (lane takes part), xs / ys (plane
// coordinates), isa (energy scale of the voltage-dependent capacitance model), emin = lowest kept plane energy.
// Per lane out: occ[N] expectation occupations, lam (relative to the lane's reference energy, times isa) and, with
// VALIDATE, the relative residual.  Returns false if the tile must be handed to the per-pixel kernels.
// ---------------------------------------------------------------------------------------------------------
template <int N, bool VALIDATE>
__device__ bool qd_tile_ground(const QdTileLds& T, QdTileGroundLds& Gd, QdTileGroundVec& Vv, int nS, int nSfront, const int* lo_,
                               const QdMask256& KM, bool alive, double xs, double ys, double isa, double emin, double eshift,
                               const double* tcv, double (&occ)[N], double& lam_best, double& resid_best,
                               unsigned long long* stats = nullptr) {
    constexpr int NB = N - 1, M = QD_T_NMAXC;
    const int lane = threadIdx.x;
    // couplings: per lane in LDS (gathered by uniform pair index in the matvec); pairs that do not couple at all
    unsigned pz = 0;
#pragma unroll
    for (int d = 0; d < NB; ++d) { Vv.tl[d][lane] = tcv[d]; pz |= (tcv[d] != 0.0 ? 1u : 0u) << d; }
    unsigned pairnz = 0;
#pragma unroll
    for (int d = 0; d < NB; ++d) pairnz |= (__any(alive && ((pz >> d) & 1u)) ? 1u : 0u) << d;
    qd_tile_hop_table<N>(T, Gd, nS, nSfront, pairnz);

    // States that can sit in a ground-candidate component: the state's own Gershgorin disc (kept partners only, my
    // couplings; the sqrt factors rounded up) must reach below my lowest diagonal entry.  By Gershgorin's theorem the
    // component that holds my ground state contains such a state, so every other component is never looked at.
    const double ub0 = emin * isa;
    QdMask256 cand; cand.w[0] = cand.w[1] = cand.w[2] = cand.w[3] = 0ull;
    double seedlb = INFINITY; int seed0 = 0;
    for (int s = 0; s < nS; ++s) {
        const bool kept = qd_m_test(KM, s);
        if (!__any(kept)) continue;
        const int ph = qd_s_phys(s, nSfront);
        const double e = fma(xs, T.sa[ph], fma(ys, T.sb[ph], T.sD[ph])) * isa;
        const uint32_t cs = (uint32_t)qd_uni_i((int)T.scode[ph]);
        double rad = 0.0;
#pragma unroll
        for (int wv = 0; wv < (2 * NB + 3) / 4; ++wv) {
            const uint32_t rw = (uint32_t)qd_uni_i((int)Gd.nbw[s][wv]);
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {
                if (wv * 4 + bq < 2 * NB) {
                    const int kk = (int)((rw >> (8 * bq)) & 255u);
                    if (kk != 255) {
                        const float sq = sqrtf((float)qd_tile_hop_product<N>(cs, wv * 4 + bq, lo_)) * 1.000001f;
                        const double td = fabs(tcv[(wv * 4 + bq) >> 1]);
                        rad = fma(qd_m_test(KM, kk) ? td : 0.0, (double)sq, rad);
                    }
                }
            }
        }
        if (kept) {
            const double lb = e - rad;
            if (lb <= ub0) qd_m_set(cand, s);
            if (lb < seedlb) { seedlb = lb; seed0 = s; }
        }
    }
#if defined(QD_TILE_ABLATE) && QD_TILE_ABLATE == 2
#pragma unroll
    for (int i = 0; i < N; ++i) occ[i] = seedlb;              // diagnostic: hop table + candidate discs only
    lam_best = 0.0; resid_best = 0.0;
    return true;
#endif
    QdMask256 resolved; resolved.w[0] = resolved.w[1] = resolved.w[2] = resolved.w[3] = 0ull;
    double ub = ub0;                                        // upper bound of my ground energy
    lam_best = INFINITY; resid_best = 0.0;
    uint32_t best_code = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < N; ++i) occ[i] = 0.0;
    bool first = true;
    bool ok = true;
    for (int iter = 0; iter < QD_T_MAXSTRUCT; ++iter) {
        // pending: candidate states of mine whose component has not been dealt with
        QdMask256 pend;
#pragma unroll
        for (int k = 0; k < 4; ++k) pend.w[k] = cand.w[k] & ~resolved.w[k];
        const bool need = alive && qd_m_any(pend);
        const unsigned long long nm = __ballot(need);
        if (!nm) break;
        if (iter == QD_T_MAXSTRUCT - 1) { ok = false; break; }
        const int leader = __builtin_ctzll(nm);
        // the leader's seed: its most promising state first (lowest loose bound), then simply the lowest index
        int myseed = (first && qd_m_test(pend, seed0)) ? seed0 : (qd_m_any(pend) ? qd_m_first(pend) : 0);
        const int seed = __builtin_amdgcn_readlane(myseed, leader);
        const QdMask256 KL = qd_m_uniform(KM, leader);
        // component of the seed inside the leader's kept set (uniform breadth-first walk over the hop table)
        QdMask256 Mm, front, ANB;
#pragma unroll
        for (int k = 0; k < 4; ++k) { Mm.w[k] = 0ull; front.w[k] = 0ull; ANB.w[k] = 0ull; }
        qd_m_set(Mm, seed); qd_m_set(front, seed);
        int guard = 0;
        while (qd_m_any(front) && guard++ < 64) {
            const int s = qd_m_first(front);
            qd_m_clear(front, s);
#pragma unroll
            for (int wv = 0; wv < (2 * NB + 3) / 4; ++wv) {
                const uint32_t rw = (uint32_t)__builtin_amdgcn_readfirstlane((int)Gd.nbw[s][wv]);
#pragma unroll
                for (int bq = 0; bq < 4; ++bq) {
                    if (wv * 4 + bq < 2 * NB) {
                        const int kk = (int)((rw >> (8 * bq)) & 255u);
                        if (kk != 255) {
                            qd_m_set(ANB, kk);
                            if (qd_m_test(KL, kk) && !qd_m_test(Mm, kk)) { qd_m_set(Mm, kk); qd_m_set(front, kk); }
                        }
                    }
                }
            }
        }
        const int n = qd_m_count(Mm);
        if (n > M) { ok = false; break; }
        // lanes whose kept graph has exactly this component: all members kept, no kept partner outside, not yet done
        bool has = alive;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            has = has && ((Mm.w[k] & ~KM.w[k]) == 0ull) && ((ANB.w[k] & ~Mm.w[k] & KM.w[k]) == 0ull) && ((Mm.w[k] & resolved.w[k]) == 0ull);
        }
        // (the leader always has it)
        // members in ascending S order, local edges with sqrt factors (lane i < n does row i)
        {
            QdMask256 rest = Mm;
            for (int i = 0; i < n; ++i) { const int s = qd_m_first(rest); qd_m_clear(rest, s); if (lane == 0) Gd.mS[i] = s; }
            __builtin_amdgcn_wave_barrier();
            if (lane < n) {
                const int s = Gd.mS[lane];
                const uint32_t cj = T.scode[qd_s_phys(s, nSfront)];
                int cnt = 0;
#pragma unroll
                for (int slot = 0; slot < 2 * NB; ++slot) {
                    const int kk = (int)((Gd.nbw[s][slot >> 2] >> (8 * (slot & 3))) & 255u);
                    if (kk != 255 && qd_m_test(Mm, kk)) {
                        const int prod = qd_tile_hop_product<N>(cj, slot, lo_);
                        double sq_ = 0.0, rs_ = 0.0;
                        if (prod > 0) qd_sqrt_rsqrt((double)prod, sq_, rs_);
                        Gd.ej[lane][cnt] = (unsigned char)qd_m_rank(Mm, kk);
                        Gd.ed[lane][cnt] = (unsigned char)(slot >> 1);
                        Gd.esq[lane][cnt] = sq_;
                        cnt++;
                    }
                }
                Gd.ecnt[lane] = cnt;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // per-lane diagonal, Gershgorin bound of the component
        double F[M];
        double lbc = INFINITY, hnorm = 0.0;
        uint32_t code0 = 0xFFFFFFFFu;                         // lowest candidate code of the component (tie rule)
#pragma unroll
        for (int i = 0; i < M; ++i) {
            F[i] = 0.0;
            if (i < n) {
                const int s = qd_s_phys(qd_uni_i(Gd.mS[i]), nSfront);
                const uint32_t cd = T.scode[s];
                code0 = cd < code0 ? cd : code0;
                F[i] = fma(xs, T.sa[s], fma(ys, T.sb[s], T.sD[s])) * isa;
                double rad = 0.0;
                const int cnt = qd_uni_i(Gd.ecnt[i]);
                for (int e = 0; e < cnt; ++e) rad = fma(fabs(Vv.tl[qd_uni_i((int)Gd.ed[i][e])][lane]), qd_uni_d(Gd.esq[i][e]), rad);
                lbc = fmin(lbc, F[i] - rad);
                hnorm = fmax(hnorm, fabs(F[i] + eshift) + rad);
            }
        }
        const bool solve = has && (n > 1) && (lbc <= ub);
        double lam = INFINITY, resid = 0.0;
        double x[M];
        if (stats && lane == 0) { atomicAdd(&stats[5], 1ull); if (n > 1) atomicAdd(&stats[6], 1ull); if (n > 1 && __any(solve)) atomicAdd(&stats[7], 1ull); }
#if defined(QD_TILE_ABLATE) && QD_TILE_ABLATE == 3
        if (false) {                                           // diagnostic: structure loop without the solves
#else
        if (n > 1 && __any(solve)) {
#endif
            qd_tile_solve<N, VALIDATE>(Gd, Vv, n, solve, F, lam, x, resid);
        } else {
#pragma unroll
            for (int i = 0; i < M; ++i) x[i] = (i == 0) ? 1.0 : 0.0;
        }
        if (n == 1) lam = F[0];
        const bool contender = has && (n == 1 || solve);
        if (contender) {
            const bool better = (lam < lam_best) || (lam == lam_best && code0 < best_code);
            if (better) {
                lam_best = lam; best_code = code0; resid_best = resid / (hnorm > 0.0 ? hnorm : 1.0);
#pragma unroll
                for (int d = 0; d < N; ++d) occ[d] = 0.0;
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    if (i < n) {
                        const uint32_t cd = T.scode[qd_s_phys(qd_uni_i(Gd.mS[i]), nSfront)];
                        const double p = x[i] * x[i];
#pragma unroll
                        for (int d = 0; d < N; ++d) occ[d] = fma(p, (double)(lo_[d] + (int)((cd >> (4 * (N - 1 - d))) & 15u)), occ[d]);
                    }
                }
            }
            ub = fmin(ub, lam);
        }
        if (has) {
#pragma unroll
            for (int k = 0; k < 4; ++k) resolved.w[k] |= Mm.w[k];
        }
        first = false;
    }
    return ok;
}

#endif  // __HIPCC__

// qd_groundstate.h -- ground state of H = diag(F) + H_t over the 32 kept charge states of a pixel (SURVEY rows
// a11-a13; reference: hamiltonian_build.py:12-45, 75-137, 460-483 and ground_state.py:149-162, where a dense 32x32
// eigh is called per pixel and only column 0 is used).
//
// Three kernels per launch chunk (qd_kernels.h), the image cut into batches of QD_GS_PPB pixels with one slab of
// scratch per batch:
//
//   A  STRUCTURE (qd_k_gs_structure), one pixel per half-wave, one basis state per lane (qd_ground_structure):
//      hop neighbours (states i, j couple over the adjacent pair d iff s_j - s_i = -+e_d +-e_{d+1}; with 4-bit-spaced
//      delta codes that is a borrow-free nibble difference of 0x1F << 4q or 0xF1 << 4q), coefficients
//      H_ij = -t_d sqrt(n_from (n_to + 1)) with the occupations of the ROW state (hamiltonian_build.py:125-131),
//      connected components (hopping conserves the total charge, so H is block diagonal; the padding copies of
//      |0..0> are always isolated), Gershgorin pruning (a component whose lower bound min(F - sum|H_ij|) exceeds
//      min F cannot hold the ground state).  Every surviving component of >= 2 states becomes a TASK: its dense
//      block is written, lower triangle packed, into the batch's slab and its offset appended to the list of its
//      size class; the batch's 64-task tiles are appended to the launch-wide tile list of the class.
//   B  SOLVE (qd_k_gs_solve<class>, one launch per size class with its own register budget), one TASK per lane
//      (qd_eig.h): Householder tridiagonalisation, Laguerre, twisted factorisation; 64 lanes = 64 different blocks
//      of the same size.  (Round 2 ran Lanczos + the serial tridiagonal recurrences with one state per lane: every
//      member lane of a component repeated the identical computation and a wave paid the maximum over its ~11
//      components; plain Lanczos also lost near-degenerate lowest pairs at tc >~ 1e14.)
//   C  SELECT (qd_k_gs_select), one pixel per lane (qd_ground_select): the component with the lowest eigenvalue wins
//      (ties: lowest candidate index), <n> = sum_m x_m^2 s_m, the sensor constant c0.
// Solving block by block is at least as accurate as one dense 32x32 eigh (no rounding-level mixing of different charge
// sectors); exactly-zero couplings do not link states, so tc == 0 gives exact integer occupations.
#pragma once
#include "qd_pixel.h"
#include "qd_eig.h"

#if defined(__HIPCC__)

#ifndef QD_NBREG
#define QD_NBREG 6                  // neighbour slots kept in registers; the other 2*(N-1) - 6 live in LDS (8 dots: 21 KB per block)
#endif
#define QD_GS_BLOCK 256
#define QD_GS_PPB 256               // pixels per batch (4 waves x 32 iterations x 2 pixels)
// size classes: 2 .. 8 states exactly, 9-10 (solved as 10), 11-12 (as 12), 13-32 (memory solver)
#define QD_GS_NBIN 10
__host__ __device__ inline int qd_gs_bin(int s) { return s <= 8 ? s - 2 : (s <= 10 ? 7 : (s <= 12 ? 8 : 9)); }
__host__ __device__ inline int qd_gs_bin_min(int bin) { return bin <= 6 ? bin + 2 : (bin == 7 ? 9 : (bin == 8 ? 11 : 13)); }
#define QD_LINK_NONE 0xFFFFFFFFu    // state whose component cannot hold the ground state (Gershgorin)
#define QD_LINK_SINGLE 0xFFFFFFFEu  // isolated state that can: T = [F]

// LDS-qualified volatile pointers: a plain `volatile double*` into __shared__ memory stays a generic
// pointer (address-space inference skips volatile accesses) and every access becomes a FLAT load;
// with the address space spelled out they are ds_read / ds_write.
typedef __attribute__((address_space(3))) double qd_lds_double;
typedef volatile qd_lds_double* qd_lds_vptr;
typedef const volatile qd_lds_double* qd_lds_cvptr;

template <int N>
struct QdWaveLds {
    static constexpr int NS = (2 * (N - 1) - QD_NBREG) > 1 ? (2 * (N - 1) - QD_NBREG) : 1;   // a state has at most 2*(N-1) hop neighbours
    double coef[NS][64];            // H_ij of neighbour slot QD_NBREG + s of lane
    unsigned char nidx[NS][64];
    double buf[64];                 // publish buffer for per-component reductions
    double pv[2][16];               // per half: tc[0..N-2] at offset 9
    short pfl[2][8];                // per half: floor(n_cont)
};
struct QdBlockLds {
    unsigned pool_top;              // bump allocator of the batch's slab (doubles)
    unsigned cnt[QD_GS_NBIN];       // tasks per size class
};

// One batch's scratch in HBM/L2:
//   pool   task records: [0] unused, [1] residual (out, validate mode; in: the size, for the padded / memory solvers),
//          [2 ..] packed lower triangle (in), overwritten by x[0..s-1] (out); blocks of > 12 states carry a 4 s workspace
//          (+ a copy of the matrix in validate mode)
//   link   per pixel and state: its component's position in `lists` (size classes back to back), QD_LINK_SINGLE or QD_LINK_NONE;
//          rank: index inside the component
//   lists  per size class: record offsets;  cnt: their lengths
//   lam    the tasks' eigenvalues, dense, same index as `lists`: written coalesced by the solve launches and read by the select
//          kernel with the locality of neighbouring pixels' tasks (reading them out of the records cost a 64-byte sector each)
//   aux    validate mode, per pixel: ||H||_inf (scale of the residual), then the lowest free energy (offset of the eigenvalues)
struct QdSlab {
    double* pool; unsigned* link; unsigned char* rank; unsigned* lists; double* aux; unsigned* cnt; double* lam;
};
__host__ __device__ inline int qd_gs_list_cap(int bin) { return QD_GS_PPB * (32 / qd_gs_bin_min(bin)); }
__host__ __device__ inline int qd_gs_list_off(int bin) { int o = 0; for (int b = 0; b < bin; ++b) o += qd_gs_list_cap(b); return o; }
__host__ __device__ inline int qd_gs_task_doubles(int s, bool validate) {
    const int ne = s * (s + 1) / 2;
    const int n = 2 + ne + (s > QD_EIG_REG ? 4 * s + (validate ? ne : 0) : 0);
    return (n + 1) & ~1;                                  // records stay 16-byte aligned: the solvers load / store pairs of doubles
}
// worst case per pixel: one component of all 32 states
__host__ __device__ inline size_t qd_gs_pool_doubles(bool validate) { return (size_t)QD_GS_PPB * (size_t)qd_gs_task_doubles(32, validate); }
__host__ __device__ inline size_t qd_gs_slab_bytes(bool validate) {
    size_t b = qd_gs_pool_doubles(validate) * 8;
    b += (size_t)QD_GS_PPB * 32 * 4;                      // link
    b += (size_t)qd_gs_list_off(QD_GS_NBIN) * 4;          // lists
    b += (size_t)QD_GS_PPB * 16;                          // aux
    b += (size_t)QD_GS_PPB * 32;                          // rank
    b += 64;                                              // cnt
    b += (size_t)qd_gs_list_off(QD_GS_NBIN) * 8;          // lam
    return (b + 255) & ~(size_t)255;
}
__host__ __device__ inline QdSlab qd_gs_slab(unsigned char* base, bool validate) {
    QdSlab s;
    s.pool = (double*)base; base += qd_gs_pool_doubles(validate) * 8;
    s.link = (unsigned*)base; base += (size_t)QD_GS_PPB * 32 * 4;
    s.lists = (unsigned*)base; base += (size_t)qd_gs_list_off(QD_GS_NBIN) * 4;
    s.aux = (double*)base; base += (size_t)QD_GS_PPB * 16;
    s.rank = base; base += (size_t)QD_GS_PPB * 32;
    s.cnt = (unsigned*)base; base += 64;
    s.lam = (double*)base;
    return s;
}
// launch-wide tile lists: tile descriptor = batch << 12 | tile index inside the batch's list << 6 | tasks in the tile - 1
__host__ __device__ inline size_t qd_gs_tile_cap(int bin, size_t batches) { return batches * (size_t)((qd_gs_list_cap(bin) + 63) / 64); }
__host__ __device__ inline size_t qd_gs_tile_off(int bin, size_t batches) { size_t o = 0; for (int b = 0; b < bin; ++b) o += qd_gs_tile_cap(b, batches); return o; }

__device__ __forceinline__ unsigned qd_half_ballot(bool p) {
    unsigned long long b = __ballot(p);
    return (unsigned)(b >> (threadIdx.x & 32));
}
__device__ __forceinline__ double qd_half_min(double v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 32));
    return v;
}
// wave-wide maximum, returned through readfirstlane so that the compiler knows it is uniform: loop bounds
// and slot guards built from it become scalar branches instead of exec-mask juggling
__device__ __forceinline__ int qd_wave_max_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

// ---------------------------------------------------------------------------------------------------------------
// Phase A.  rec: this half's pixel record (states, their free energies from the candidate search, tunnel couplings);
// ps: the pixel's slot in the batch; live: false for the clamped duplicate beyond the image (nothing is emitted).
// ---------------------------------------------------------------------------------------------------------------
template <int N, bool VALIDATE>
__device__ __forceinline__ void qd_ground_structure(const QdPixelRec* __restrict__ rec, bool live, int ps, QdWaveLds<N>& W,
                                                    QdBlockLds& S, const QdSlab& sl) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 31;
    const int hb = lane & 32;
    const unsigned lt = (1u << m) - 1u;
    qd_lds_vptr buf = (qd_lds_vptr)W.buf;

    // ---- 1. my state -------------------------------------------------------
    // pixel-uniform record fields go through LDS once (keeps them out of registers)
    const int hh = hb >> 5;
    if (m >= 9 && m < 9 + N - 1) W.pv[hh][m] = rec->tc[m - 9];
    if (m >= 16 && m < 16 + N) W.pfl[hh][m - 16] = (short)rec->fl[m - 16];
    __builtin_amdgcn_wave_barrier();
    const double* pvv = W.pv[hh];
    const int nvalid = rec->nvalid;
    const bool valid = m < nvalid;
    const unsigned code = valid ? (unsigned)rec->idx[m] : 0u;
    // digit of dot i (base 4, dot 0 most significant) -> nibble N-1-i of ecode: spread the 2-bit digits to 4-bit spacing
    unsigned ecode = code;
    ecode = (ecode | (ecode << 8)) & 0x00FF00FFu;
    ecode = (ecode | (ecode << 4)) & 0x0F0F0F0Fu;
    ecode = (ecode | (ecode << 2)) & 0x33333333u;
    // F_m: the candidate search already evaluated the canonical energy of every kept state, and of the |0..0> padding
    // when fewer than 32 candidates are valid (N <= 3).  The diagonal enters RELATIVE to the pixel's lowest free energy:
    // H - c I has the same eigenvectors, and the common offset (|F| ~ 1e3..1e5 far from the ground truth, against
    // spreads of O(1)) would cost ~eps |F| in every subtraction.
    const double Fabs = rec->E[m];
    const double fshift = qd_half_min(Fabs);
    const double F = Fabs - fshift;

    // ---- 2. hop neighbours -------------------------------------------------
    // pairs whose coupling is exactly zero do not link states (keeps the classical
    // limit tc == 0 exactly diagonal, as a dense eigh of a diagonal matrix would)
    unsigned tcnz = 0;
#pragma unroll
    for (int d = 0; d < N - 1; ++d) tcnz |= (pvv[9 + d] != 0.0 ? 1u : 0u) << d;
    // nibble q of ecode is the digit of dot N-1-q.  (cj | 0x8..8) - ecode holds 8 + (digit_j - digit_i)
    // in every nibble (no borrows), so Z = that ^ 0x8..8 has nibble 0 where the digits agree, 1 for +1
    // and 0xF for -1: j is a hop neighbour iff Z == 0x1F << 4q or 0xF1 << 4q (one electron moved
    // between the adjacent dots of pair N-2-q) and that pair's coupling is non-zero.
    unsigned tcq = 0;                                     // bit 4q set iff pair N-2-q couples
#pragma unroll
    for (int q = 0; q < N - 1; ++q) tcq |= ((tcnz >> (N - 2 - q)) & 1u) << (4 * q);
    // The relation is symmetric, so every pair is tested once: in round j lane m tests its partner (m + j) mod 32 and hands the
    // verdict to that partner as well (which receives it from lane (m - j) mod 32) -- 16 rounds instead of 32 tests per lane.
    unsigned nbrmask = 0;
#pragma unroll 4
    for (int j = 1; j <= 16; ++j) {
        const int pj = (m + j) & 31;
        const unsigned cj = __shfl(ecode, pj, 32);
        const unsigned Z = ((cj | 0x88888888u) - ecode) ^ 0x88888888u;
        const int tz = __builtin_ctz(Z | 0x80000000u);     // Z == 0 (same state): tz = 31, Zs = 0, no hop (a nibble of Z is never 8)
        const unsigned Zs = Z >> tz;
        // branch-free on purpose (bitwise, not short-circuit): the compiler otherwise builds a divergent branch per j
        const unsigned hop = ((unsigned)(Zs == 0x1Fu) | (unsigned)(Zs == 0xF1u)) & (tcq >> tz) & 1u;
        nbrmask |= hop << pj;
        if (j < 16) {                                      // (round 16 pairs m with m + 16 from both sides already)
            const int qj = (m - j) & 31;
            const unsigned back = (unsigned)__shfl((int)hop, qj, 32);
            nbrmask |= back << qj;
        }
    }
    // states beyond the valid count (|0..0> padding) neither hop nor are hopped to
    nbrmask = (valid && live) ? (nbrmask & (nvalid >= 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u))) : 0u;
#if defined(QD_ABLATE) && QD_ABLATE == 4
    nbrmask = 0;                                          // diagnostic: no hopping at all
#endif
    const int cnt = __popc(nbrmask);
    const int maxcnt = qd_wave_max_int(cnt);
    double radius = 0.0;
    double creg[QD_NBREG]; int jreg[QD_NBREG];
#pragma unroll
    for (int i = 0; i < QD_NBREG; ++i) { creg[i] = 0.0; jreg[i] = m; }
    {
        unsigned rem = nbrmask;
        for (int s = 0; s < maxcnt; ++s) {
            const bool has = rem != 0;
            const int j = has ? __builtin_ctz(rem) : m;
            rem &= rem - 1;
            const unsigned cj = __shfl(ecode, j, 32);
            double c = 0.0;
            if (has) {
                const int Y = (int)cj - (int)ecode;
                const unsigned ay = (unsigned)(Y < 0 ? -Y : Y);
                const int q = __builtin_ctz(ay) >> 2;
                const int d = N - 2 - q;                   // adjacent pair (d, d+1)
                // occupations of the pair's two dots straight from the code's digits and the floors in LDS
                const int nd = (int)W.pfl[hh][d] + (int)((ecode >> (4 * (q + 1))) & 3u) - 1;
                const int nd1 = (int)W.pfl[hh][d + 1] + (int)((ecode >> (4 * q)) & 3u) - 1;
                const double t = pvv[9 + d];
                // Y < 0: s_j = s_i - e_d + e_{d+1} (forward); else backward
                const double prod = (Y < 0) ? (double)nd * ((double)nd1 + 1.0)
                                            : (double)nd1 * ((double)nd + 1.0);
                double sq_ = 0.0;
                if (prod > 0.0) sq_ = qd_sqrt1(prod);
                c = -t * sq_;
            }
            if (s < QD_NBREG) {                            // (s is uniform: scalar branches)
#pragma unroll
                for (int i = 0; i < QD_NBREG; ++i) if (i == s) { creg[i] = c; jreg[i] = j; }
            } else { W.coef[s - QD_NBREG][lane] = c; W.nidx[s - QD_NBREG][lane] = (unsigned char)j; }
            radius += fabs(c);
        }
    }

    // ---- 3. connected components (reach masks) -----------------------------
    unsigned seg = (1u << m) | nbrmask;
    for (int guard = 0; guard < 32; ++guard) {
        unsigned nw = seg, rem = nbrmask;
        for (int s = 0; s < maxcnt; ++s) {
            const bool has = rem != 0;
            const int j = has ? __builtin_ctz(rem) : m;
            rem &= rem - 1;
            nw |= __shfl(seg, j, 32);                     // (own mask for exhausted lanes: no change)
        }
        const bool changed = nw != seg;
        seg = nw;
        if (!__any(changed)) break;
    }
    const int ssz = __popc(seg);
    const int r = __popc(seg & lt);                        // my index inside the component
    const int smax = qd_wave_max_int(ssz);

    // ---- 4. Gershgorin pruning ---------------------------------------------
    // upper bound of the pixel's ground energy: min F = 0 (the diagonal is relative to the lowest free energy)
    double comp_lower;
    {
        buf[lane] = F - radius;
        __builtin_amdgcn_wave_barrier();
        double acc = INFINITY;
        unsigned mm = seg;
        for (int it = 0; it < smax; ++it) {
            if (mm) { const int b = __builtin_ctz(mm); mm &= mm - 1; acc = fmin(acc, buf[hb + b]); }
        }
        __builtin_amdgcn_wave_barrier();
        comp_lower = acc;
    }
    const bool active = live && comp_lower <= 0.0;
#if defined(QD_ABLATE) && QD_ABLATE == 3
    const bool solve = false;                              // diagnostic: no tasks
#else
    const bool solve = active && ssz > 1;
#endif

    // ---- 5. tasks: one record per surviving component ------------------------
    unsigned base = 0, gi = 0;
    if (solve && r == 0) {
        base = atomicAdd(&S.pool_top, (unsigned)qd_gs_task_doubles(ssz, VALIDATE));
        const int bin = qd_gs_bin(ssz);
        gi = (unsigned)qd_gs_list_off(bin) + atomicAdd(&S.cnt[bin], 1u);
        sl.lists[gi] = base;
        if (ssz > 8) sl.pool[base + 1] = (double)ssz;
    }
    base = (unsigned)__shfl((int)base, __builtin_ctz(seg), 32);
    gi = (unsigned)__shfl((int)gi, __builtin_ctz(seg), 32);
    if (live) {
        sl.link[ps * 32 + m] = solve ? gi : (active ? QD_LINK_SINGLE : QD_LINK_NONE);
        sl.rank[ps * 32 + m] = (unsigned char)r;
        if (VALIDATE) {
            const double hn = -qd_half_min(-(fabs(Fabs) + radius));       // ||H||_inf over the 32 states (unshifted)
            if (m == 0) { sl.aux[ps] = hn; sl.aux[QD_GS_PPB + ps] = fshift; }
        }
    }
    {
        // my row of the lower triangle: couplings to members of lower rank, zeros elsewhere, F on the diagonal
        double* row = sl.pool + base + 2 + (r * (r + 1)) / 2;
        unsigned nrm = 0;                                  // ranks that carry a coupling
        for (int s = 0; s < maxcnt; ++s) {
            if (s < cnt && solve) {
                int j = jreg[0]; double c = creg[0];
                if (s >= QD_NBREG) { j = (int)W.nidx[s - QD_NBREG][lane]; c = W.coef[s - QD_NBREG][lane]; }
#pragma unroll
                for (int i = 1; i < QD_NBREG; ++i) if (i == s) { j = jreg[i]; c = creg[i]; }
                const int rj = __popc(seg & ((1u << j) - 1u));
                if (rj < r) { row[rj] = c; nrm |= 1u << rj; }
            }
        }
        for (int c = 0; c < smax - 1; ++c)
            if (solve && c < r && !((nrm >> c) & 1u)) row[c] = 0.0;
        if (solve) row[r] = F;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Phase B: one task per lane.  rec: the task's record in the slab.  Returns the Laguerre iterations (statistics).
// ---------------------------------------------------------------------------------------------------------------
template <int S, bool PADDED, bool VALIDATE>
__device__ __forceinline__ int qd_eig_task(double* rec, double& lam) {
    double resid, x[S];
    int its = 0;
    const int sz = PADDED ? (int)rec[1] : S;
    qd_eig_lowest<S, VALIDATE>(rec + 2, lam, x, resid, VALIDATE ? &its : nullptr, sz);
    if (VALIDATE) rec[1] = resid;                          // (the eigenvalue goes to the slab's dense array: qd_k_gs_solve)
    if (sz == S) {
        double2* X2 = reinterpret_cast<double2*>(rec + 2);
#pragma unroll
        for (int i = 0; i + 1 < S; i += 2) X2[i >> 1] = make_double2(x[i], x[i + 1]);
        if (S & 1) rec[2 + S - 1] = x[S - 1];
    } else {
#pragma unroll
        for (int i = 0; i < S; ++i) if (i < sz) rec[2 + i] = x[i];
    }
    return its;
}
// Blocks of 13..32 states (0.1 % of the pixels have one): the same algorithm with run-time loops in the task's record -- a serial
// O(s^3) chain of dependent memory operations, ~1 ms per wave whatever the batch, run on a side stream.  Measured and not kept
// (bench, env-steps/s; this version 11 500-11 600): solving blocks of <= 16 states in an LDS tile ([element][lane], 100 KB per
// wave, one wave per CU): 11 000-11 100 (a tenth of the latency per operation, a twentieth of the waves in flight); listing the
// class's single tasks launch-wide so that its waves are full: 11 400 (64 different records per memory instruction).
template <bool VALIDATE>
__device__ __forceinline__ int qd_eig_task_mem(double* rec, double& lam) {
    const int s = (int)rec[1];
    const int ne = s * (s + 1) / 2;
    double* M = rec + 2;
    double* work = M + ne;
    double* copy = VALIDATE ? work + 4 * s : nullptr;
    if (VALIDATE) for (int e = 0; e < ne; ++e) copy[e] = M[e];
    double resid;
    int its = 0;
    qd_eig_lowest_mem(s, M, work, copy, lam, resid, VALIDATE ? &its : nullptr);
    if (VALIDATE) rec[1] = resid;                          // (the eigenvalue goes to the slab's dense array: qd_k_gs_solve)
    for (int i = 0; i < s; ++i) M[i] = work[3 * s + i];
    return its;
}

// ---------------------------------------------------------------------------------------------------------------
// Phase C: one pixel per lane.  Returns the occupations, the ground energy (absolute) and, with VALIDATE, the
// relative residual ||H x - lam x||_2 / ||H||_inf of the winning component's eigenpair.  Written as three rounds of
// INDEPENDENT loads (links and energies; eigenvalues; the winner's vector) -- a lane's chain of 64 dependent loads was
// what the first version of this phase spent its time on.
// ---------------------------------------------------------------------------------------------------------------
template <int N, bool VALIDATE>
__device__ __forceinline__ void qd_ground_select(const QdPixelRec* __restrict__ rec, int ps, const QdSlab& sl,
                                                 double* occ, double& lam_out, double& resid_out) {
    const unsigned* __restrict__ link = sl.link + ps * 32;
    const unsigned char* __restrict__ rank = sl.rank + ps * 32;
    const double* __restrict__ pool = sl.pool;
    const double* __restrict__ lamd = sl.lam;
    const int nvalid = rec->nvalid;
    unsigned lk[QD_K];
    double lam[QD_K];
    {
        const uint4* l4 = reinterpret_cast<const uint4*>(link);
#pragma unroll
        for (int q = 0; q < QD_K / 4; ++q) { const uint4 v = l4[q]; lk[4 * q] = v.x; lk[4 * q + 1] = v.y; lk[4 * q + 2] = v.z; lk[4 * q + 3] = v.w; }
    }
    // eigenvalues are relative to the pixel's lowest free energy; an isolated state that survived the Gershgorin test
    // is a state of exactly that energy (its bound F - 0 must not exceed min F): lambda = 0
#pragma unroll
    for (int m = 0; m < QD_K; ++m) {
        const bool task = lk[m] < QD_LINK_SINGLE;
        const double lt = lamd[task ? lk[m] : 0u];         // (always a valid address; the value is used for tasks only)
        lam[m] = task ? lt : (lk[m] == QD_LINK_SINGLE ? 0.0 : INFINITY);
    }
    // the lowest component; tie between components (exactly equal energies): the state with the lowest candidate
    // index wins -- the reference order puts it first -- independent of the buffer order
    unsigned idx2[QD_K / 2];
    {
        const uint4* i4 = reinterpret_cast<const uint4*>(rec->idx);
#pragma unroll
        for (int q = 0; q < QD_K / 8; ++q) { const uint4 v = i4[q]; idx2[4 * q] = v.x; idx2[4 * q + 1] = v.y; idx2[4 * q + 2] = v.z; idx2[4 * q + 3] = v.w; }
    }
    double best = INFINITY; unsigned bestkey = 0xFFFFFFFFu, wl = QD_LINK_NONE; int bm = 0;
#pragma unroll
    for (int m = 0; m < QD_K; ++m) {
        const unsigned code = (idx2[m >> 1] >> (16 * (m & 1))) & 0xFFFFu;
        const unsigned key = (m < nvalid) ? code : 0xFFFFFFFEu;
        const bool better = (lam[m] < best) | ((lam[m] == best) & (key < bestkey));
        if (better) { best = lam[m]; bestkey = key; wl = lk[m]; bm = m; }
    }
    const bool wtask = wl < QD_LINK_SINGLE;
    const unsigned woff = wtask ? sl.lists[wl] : 0u;          // the winner's record
    // the winner's vector: one (predicated) load per member state
    unsigned rk8[QD_K / 4];
    {
        const uint4* r4 = reinterpret_cast<const uint4*>(rank);
#pragma unroll
        for (int q = 0; q < QD_K / 16; ++q) { const uint4 v = r4[q]; rk8[4 * q] = v.x; rk8[4 * q + 1] = v.y; rk8[4 * q + 2] = v.z; rk8[4 * q + 3] = v.w; }
    }
    double xs[QD_K];
#pragma unroll
    for (int m = 0; m < QD_K; ++m) {
        const bool member = wtask ? (lk[m] == wl) : (m == bm);
        const unsigned rk = (rk8[m >> 2] >> (8 * (m & 3))) & 0xFFu;
        double x = 0.0;
        if (member) x = wtask ? pool[woff + 2 + rk] : 1.0;
        xs[m] = x;
    }
    int fl[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { occ[i] = 0.0; fl[i] = rec->fl[i]; }
#pragma unroll
    for (int m = 0; m < QD_K; ++m) {
        const unsigned code = (idx2[m >> 1] >> (16 * (m & 1))) & 0xFFFFu;
        const double p = (m < nvalid) ? xs[m] * xs[m] : 0.0;          // (the padding lanes are |0..0>)
#pragma unroll
        for (int i = 0; i < N; ++i)
            occ[i] = fma(p, (double)(fl[i] + (int)((code >> (2 * (N - 1 - i))) & 3u) - 1), occ[i]);
    }
    lam_out = best;
    resid_out = 0.0;
    if (VALIDATE) {
        const double hn = sl.aux[ps];
        lam_out = best + sl.aux[QD_GS_PPB + ps];
        if (wtask) resid_out = pool[woff + 1] / (hn > 0.0 ? hn : 1.0);
    }
}

#endif  // __HIPCC__

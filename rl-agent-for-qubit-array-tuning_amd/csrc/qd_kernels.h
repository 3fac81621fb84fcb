// qd_kernels.h -- the HIP kernels of the batched env step (gfx950, wave64).
//
//   qd_k_actions     a2, a3   one thread per env
//   qd_k_candidates  a5, a8, a9, a10   one pixel per lane, exact k-best search
//   qd_k_gs_*        a11-a13, a15      structure per half-wave -> dense tasks per lane -> selection per lane (qd_groundstate.h)
//   qd_k_percentile  a17 (exact 0.5 / 99.5 percentiles, radix select)
//   qd_k_write_obs   a17, a22 normalise + global / per-agent images + voltages
//   qd_k_update      a19, a20, a21     Kalman, VGM (SVD pseudo-inverse), ground truth
//
// Data layout in HBM (all per handle):
//   params [B][L.size] f64, state [B][L.s_size] f64, steps [B] i32
//   recs   [chunk][C][P] QdPixelRec (scratch between candidates and ground kernels)
//   slabs  [batches of a launch] scratch of the ground-state kernels (qd_gs_slab_bytes each), tile lists per size class
//   zraw   [B][C][P] f64 raw sensor signal,  plohi [B][2] f64
#pragma once
#include "qd_groundstate.h"
#include "qd_rng.h"

#if defined(__HIPCC__)

// ---------------------------------------------------------------------------
// a2 + a3: actions -> voltages, reward vs previous ground truth, step counter
// (env.py:260-285, 350-462, 861-876)
// ---------------------------------------------------------------------------
struct QdRewardCfg {
    double gate_ramp_start, gate_quadratic_start, barrier_ramp_start;
    int max_steps;
    // config variants (env.py:393-441, 861-876)
    int use_deltas, sparse, curve;           // curve: 0 constant, 1 polynomial, 2 exponential, 3 linear
    double delta_max, curve_exponent;
    double plunger_radius, outer_plunger_radius, outer_plunger_reward_max, barrier_radius;
};

// env.py:416-441 (dense) and :393-414 (sparse) for one gate distance
__device__ __forceinline__ double qd_gate_reward(double dist, const QdRewardCfg& rc) {
    if (rc.sparse) {
        if (dist <= rc.plunger_radius) return 1.0;
        if (dist <= rc.outer_plunger_radius) {
            const double nd = (dist - rc.plunger_radius) / (rc.outer_plunger_radius - rc.plunger_radius);
            return rc.outer_plunger_reward_max * (1.0 - nd);
        }
        return 0.0;
    }
    double r;
    if (dist >= rc.gate_ramp_start) r = 0.0;
    else if (dist > rc.gate_quadratic_start)
        r = 0.5 * ((rc.gate_ramp_start - dist) / (rc.gate_ramp_start - rc.gate_quadratic_start));
    else {
        const double nrm = (rc.gate_quadratic_start - dist) / rc.gate_quadratic_start;
        double cv;
        if (rc.curve == 1) cv = pow(nrm, rc.curve_exponent);
        else if (rc.curve == 2) cv = (exp(rc.curve_exponent * nrm) - 1.0) / (exp(rc.curve_exponent) - 1.0);
        else if (rc.curve == 3) cv = nrm;
        else cv = 1.0;                                            // "constant"
        r = 0.5 + 0.5 * cv;
    }
    return fmin(fmax(r, 0.0), 1.0);
}

template <int N>
__global__ void qd_k_actions(int B, const double* __restrict__ params, double* __restrict__ state,
                             int* __restrict__ steps, const float* __restrict__ actions,
                             double* __restrict__ rewards, uint8_t* __restrict__ truncated, QdRewardCfg rc) {
    constexpr int NB = N - 1, V = 2 * N, NA = 2 * N - 1;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    const QdLayout L = qd_layout(N);
    const double* par = params + (size_t)e * L.size;
    double* st = state + (size_t)e * L.s_size;
    const float* act = actions + (size_t)e * NA;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float a = act[i];
        a = fminf(fmaxf(a, -1.0f), 1.0f);
        const float h = (a + 1.0f) / 2.0f;                       // float32, as the reference
        const double lo = par[L.pmin + i], hi = par[L.pmax + i];
        double v;
        if (rc.use_deltas) {
            // env.py:864-867: the increment is formed in float32, added to the float64 current voltage
            // with the sum stored back as float32 (in-place +=), then clipped against the float64 range
            const float span = (float)(rc.delta_max - (-rc.delta_max)), dmin = (float)(-rc.delta_max);
            const float d = h * span + dmin;
            const float s32 = (float)((double)d + st[L.s_gate_v + i]);
            v = fmin(fmax((double)s32, lo), hi);
        } else {
            v = (double)h * (hi - lo) + lo;
        }
        st[L.s_gate_v + i] = v;
        const double dist = fabs(st[L.s_gate_gt + i] - v) * fabs(par[L.cgd + i * V + i]);
        if (rewards) rewards[(size_t)e * NA + i] = qd_gate_reward(dist, rc);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float a = act[N + b];
        a = fminf(fmaxf(a, -1.0f), 1.0f);
        const float h = (a + 1.0f) / 2.0f;
        const double lo = par[L.bmin + b], hi = par[L.bmax + b];
        const double v = (double)h * (hi - lo) + lo;
        st[L.s_barrier_v + b] = v;
        const double dist = fabs(st[L.s_barrier_gt + b] - v) * par[L.alpha + b];
        double r;
        if (rc.sparse) r = dist <= rc.barrier_radius ? 1.0 : 0.0;
        else {
            r = (dist >= rc.barrier_ramp_start) ? 0.0 : (rc.barrier_ramp_start - dist) / rc.barrier_ramp_start;
            r = fmin(fmax(r, 0.0), 1.0);
        }
        if (rewards) rewards[(size_t)e * NA + N + b] = r;
    }
    const int s = steps[e] + 1;
    steps[e] = s;
    if (truncated) truncated[e] = s >= rc.max_steps ? 1 : 0;
}

// ---------------------------------------------------------------------------
// a16: stochastic stages.  flags = QD_NOISE_* (0: deterministic).
// ---------------------------------------------------------------------------
struct QdNoiseCfg {
    int flags;
    uint32_t seed, env_off, ser_lo, ser_hi;       // Philox key / counter words
    const unsigned long long* tel;                // telegraph state bits [B][C][tel_words]
    int tel_words;
};

// qarray_base_class.py:462-468: the whole channel image is replaced by white noise when either
// swept gate is further than full_noise_distance from its ground truth.
__device__ __forceinline__ bool qd_radial_replaced(const double* par, const double* st, const QdLayout& L, int ch, int flags) {
    if (!(flags & 2)) return false;
    const double full = par[L.noise + 6];
    if (!(full > 0.0)) return false;
    const double d1 = fabs(st[L.s_gate_v + ch] - st[L.s_gate_gt + ch]);
    const double d2 = fabs(st[L.s_gate_v + ch + 1] - st[L.s_gate_gt + ch + 1]);
    return d1 > full || d2 > full;
}

#include "qd_tile.h"        // tile-shared candidate search (uses qd_radial_replaced)

// Random telegraph process along the row-major raster of one (env, channel): two-state Markov
// chain, P(0->1) = p01, P(1->0) = p10, started from its stationary distribution; bit p of the
// output = state at pixel p.  One thread per (env, channel) -- the chain is serial by nature.
__global__ void qd_k_telegraph(const int* __restrict__ env_ids, int n_env, int C, int P, int psize, int noise_off,
                               const double* __restrict__ params, unsigned long long* __restrict__ tel, QdNoiseCfg nz) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_env * C) return;
    const int slot = t / C, ch = t - slot * C;
    const int e = env_ids ? env_ids[slot] : slot;
    const double* par = params + (size_t)e * psize + noise_off;
    const double p01 = par[1], p10 = par[2];
    unsigned long long* out = tel + ((size_t)e * C + ch) * nz.tel_words;
    const uint32_t k1 = nz.env_off + (uint32_t)e;
    QdPhilox r = qd_philox4x32_10(0xFFFFFFFFu, (uint32_t)ch | (QD_RNG_TELEGRAPH << 16), nz.ser_lo, nz.ser_hi, nz.seed, k1);
    const double pst = (p01 + p10 > 0.0) ? p01 / (p01 + p10) : 0.0;
    int state = qd_u01(r.v[0], r.v[1]) < pst ? 1 : 0;
    unsigned long long word = 0;
    for (int p = 0; p < P; ++p) {
        if ((p & 1) == 0) r = qd_philox4x32_10((uint32_t)(p >> 1), (uint32_t)ch | (QD_RNG_TELEGRAPH << 16), nz.ser_lo, nz.ser_hi, nz.seed, k1);
        const double u = (p & 1) ? qd_u01(r.v[2], r.v[3]) : qd_u01(r.v[0], r.v[1]);
        if (state == 0) { if (u < p01) state = 1; } else { if (u < p10) state = 0; }
        word |= (unsigned long long)state << (p & 63);
        if ((p & 63) == 63 || p == P - 1) { out[p >> 6] = word; word = 0; }
    }
}

// ---------------------------------------------------------------------------
// a5/a8/a9/a10: one pixel per lane.  grid = (ceil(P/BLOCK), C, n_env).
// ---------------------------------------------------------------------------
#ifndef QD_CAND_BLOCK
#define QD_CAND_BLOCK 64        // one tile (wave) per block: a finished wave frees its 20 KB of LDS at once; behind the tile search most
#endif                          // blocks have nothing to do and the flagged tiles run for ~0.5 ms (2 tiles per block: redo pass 13.6 -> 12.1 us)
#ifndef QD_CAND_WAVES
#define QD_CAND_WAVES 2         // <= 256 VGPRs: 2 waves per SIMD, which is also what the 20 KB of LDS per wave allows
#endif

// REDO: second pass behind the tile search (only the pixels it flagged; their front end comes with the record)
template <int N, bool REDO>
__global__ void __launch_bounds__(QD_CAND_BLOCK, QD_CAND_WAVES)
qd_k_candidates(const int* __restrict__ env_ids, int env_base, int R, const double* __restrict__ params,
                const double* __restrict__ state, QdPixelRec* __restrict__ recs, int sort_output, int noise_flags) {
    constexpr int only_flagged = REDO ? 1 : 0;
    constexpr int G = N + 1, NB = N - 1, V = 2 * N;
    const QdLayout L = qd_layout(N);
    const int slot = blockIdx.z;
    const int e = env_ids ? env_ids[env_base + slot] : env_base + slot;
    const int ch = blockIdx.y;
    const int P = R * R;
    // each wave works on an 8x8 pixel tile (not a 64x1 row): neighbouring pixels in BOTH sweep
    // directions have nearly the same search tree, so the lanes of a wave diverge less
    const int tiles_x = (R + 7) >> 3;
    const int tile = blockIdx.x * (QD_CAND_BLOCK / 64) + (threadIdx.x >> 6);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x = tx * 8 + (threadIdx.x & 7), y = ty * 8 + ((threadIdx.x >> 3) & 7);
    const bool inside = x < R && y < R;
    const int p = y * R + x;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // the parameter / state blocks are block-uniform and read-only: the compiler turns these into scalar
    // loads (SGPR operands through the constant cache) -- no LDS copy, no VGPRs, and the LDS left over
    // lets 4 blocks share a CU (measured: 9.4 -> 4.3 ms per 76-env launch together with QD_CAND_WAVES=2)
    const double* spar = params + (size_t)e * L.size;
    const double* sst = state + (size_t)e * L.s_size;
    double* se = (double*)smem_raw;                           // [32][BLOCK]
    uint16_t* sid = (uint16_t*)(se + QD_K * QD_CAND_BLOCK);   // [32][BLOCK]
    if (!inside) return;
    if (qd_radial_replaced(spar, sst, L, ch, noise_flags)) return;   // image will be pure noise: nothing to solve
    QdPixelRec* rec = recs + ((size_t)slot * (N - 1) + ch) * P + p;
    // second pass behind the tile search (qd_tile.h): only the pixels it left to the exact per-pixel search
    if (only_flagged == 1 && rec->nvalid != QD_T_REDO) return;
    double vd[N], ncont[N], isa;
    if (only_flagged == 1) {
        // behind the tile search: the front end of this pixel is in the record already (qd_tile_hand_over)
#pragma unroll
        for (int i = 0; i < N; ++i) { ncont[i] = rec->E[i]; vd[i] = rec->E[8 + i]; }
        isa = rec->E[16];
    } else {
        double v_ext[V], vpp[G], tc[NB];
        qd_pixel_voltages<N>(spar, sst, ch, R, x, y, v_ext, vpp, tc);
        // the sensor stage wants the constant-matrix product: hand it over before v' is rescaled
#pragma unroll
        for (int i = 0; i < G; ++i) rec->vpp[i] = vpp[i];
#pragma unroll
        for (int b = 0; b < NB; ++b) rec->tc[b] = tc[b];
#pragma unroll
        for (int i = 0; i < N; ++i) vd[i] = vpp[i];
        qd_pixel_continuous<N>(spar, v_ext, vd, ncont, &isa);
    }
    int32_t fl[N];
    const int nv = qd_candidates<N>(spar, vd, ncont, se + threadIdx.x, QD_CAND_BLOCK,
                                    sid + threadIdx.x, QD_CAND_BLOCK, fl, sort_output != 0);
#pragma unroll
    for (int m = 0; m < QD_K; ++m) rec->idx[m] = m < nv ? sid[m * QD_CAND_BLOCK + threadIdx.x] : 0;
    // fewer than 32 valid candidates (N <= 3): the list is padded with |0..0> states (a9 quirk), whose
    // free energy the ground-state kernel needs too
    double F0 = 0.0;
    if (nv < QD_K) {
        double dd[N];
#pragma unroll
        for (int i = 0; i < N; ++i) dd[i] = 0.0 - vd[i];
#pragma unroll 1
        for (int i = 0; i < N; ++i) {
            const double t = qd_dotN<N>(spar + L.cdd_inv + i * G, dd);
            double di = dd[0];
#pragma unroll
            for (int j = 1; j < N; ++j) di = (j == i) ? dd[j] : di;
            F0 = fma(di, t, F0);
        }
    }
#pragma unroll
    for (int m = 0; m < QD_K; ++m) rec->E[m] = (m < nv ? se[m * QD_CAND_BLOCK + threadIdx.x] : F0) * isa;
#pragma unroll
    for (int i = 0; i < N; ++i) rec->fl[i] = fl[i];
    rec->nvalid = nv;
}

// ---------------------------------------------------------------------------
// a11-a13 + a15 in three kernels (qd_groundstate.h).  The images of a launch chunk are cut into batches of QD_GS_PPB
// pixels of one (env, channel); batch b owns slab b.
//   qd_k_gs_structure  grid = batches, 256 threads: hop structure, one pixel per half-wave -> dense tasks in the slab,
//                      the batch's 64-task tiles appended to the launch-wide tile list of each size class
//   qd_k_gs_solve<K>   one launch per size class K (own register budget / occupancy), persistent waves striding over
//                      the class's tile list: one task per lane
//   qd_k_gs_select     grid = batches, one pixel per lane: lowest component, occupations, sensor constant
// stats (validate mode, counters at stats[16..]): tasks, sum of Laguerre iterations, wave tiles, sum of the tiles'
// maxima, then tasks per size class.
// ---------------------------------------------------------------------------
struct QdGsGeom { int n_env, C, P, nb; };               // nb = batches per (env, channel) image
__device__ __forceinline__ void qd_gs_locate(const QdGsGeom& g, int batch, int& slot, int& ch, int& p0) {
    slot = batch / (g.C * g.nb);
    const int rem = batch - slot * g.C * g.nb;
    ch = rem / g.nb; p0 = (rem - ch * g.nb) * QD_GS_PPB;
}

#ifndef QD_GS_WAVES
#define QD_GS_WAVES 7            // 72 VGPRs (28 B of scratch per lane) and 7 x 21 KB of LDS per CU (8 dots).  The kernel is latency
                                 // bound (ds_bpermute / LDS chains): measured per env-step 4 waves per SIMD 26.8 us, 5: 22.6, 6: 20.5, 7: 19.6,
                                 // 8 (64 VGPRs, 116 B of scratch): 31.2
#endif
// WPB waves per block: 4 in batch work; 16 for small launches (fewer batches than the chip has CUs: the launch takes as long as
// ONE block, so the batch's 128 wave iterations are spread over 16 waves instead of 4 -- 2-dot 32x32, 1 env: 158 -> 45 us)
template <int N, bool VALIDATE, int WPB>
__global__ void __launch_bounds__(64 * WPB, WPB == 4 ? QD_GS_WAVES : 4)
qd_k_gs_structure(const int* __restrict__ env_ids, int env_base, int rec_slot0, QdGsGeom g, int R, const double* __restrict__ params,
                  const QdPixelRec* __restrict__ recs, const double* __restrict__ state, int noise_flags,
                  unsigned char* __restrict__ slabs, unsigned* __restrict__ gtiles, unsigned* __restrict__ tilelist, size_t batches_cap) {
    const QdLayout L = qd_layout(N);
    __shared__ QdWaveLds<N> sW[WPB];
    __shared__ QdBlockLds sB;
    constexpr int PPW = QD_GS_PPB / WPB;                                  // pixels per wave
    const int batch = blockIdx.x;
    const QdSlab sl = qd_gs_slab(slabs + (size_t)batch * qd_gs_slab_bytes(VALIDATE), VALIDATE);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int slot, ch, p0;
    qd_gs_locate(g, batch, slot, ch, p0);
    const int e = env_ids ? env_ids[env_base + slot] : env_base + slot;
    const double* par = params + (size_t)e * L.size;
    const double* st = state + (size_t)e * L.s_size;
    if (qd_radial_replaced(par, st, L, ch, noise_flags)) {               // qd_k_sensor writes pure noise: no tasks
        if (threadIdx.x < QD_GS_NBIN) sl.cnt[threadIdx.x] = 0;
        return;
    }
    const QdPixelRec* rbase = recs + ((size_t)(rec_slot0 + slot) * g.C + ch) * g.P;
    if (threadIdx.x <= QD_GS_NBIN) { if (threadIdx.x == 0) sB.pool_top = 0; else sB.cnt[threadIdx.x - 1] = 0; }
    __syncthreads();
    QdWaveLds<N>& W = sW[wave];
    for (int it = 0; it < PPW / 2; ++it) {
        const int ps = wave * PPW + it * 2 + (lane >> 5);
        if (p0 + wave * PPW + it * 2 >= g.P) break;                              // uniform for the wave
        const int p = p0 + ps;
        // both halves of a wave run in lock step: clamp instead of exiting
        const int pc = p < g.P ? p : g.P - 1;
        qd_ground_structure<N, VALIDATE>(rbase + pc, p < g.P, ps, W, sB, sl);
    }
    __syncthreads();
    if (threadIdx.x < QD_GS_NBIN) {
        const int bin = threadIdx.x;
        const unsigned nt = sB.cnt[bin];
        sl.cnt[bin] = nt;
        const unsigned ntile = (nt + 63u) >> 6;
        if (ntile) {
            const unsigned start = atomicAdd(&gtiles[bin], ntile);
            // tile descriptor: batch (20 bits) | tile index in the batch's list (6) | tasks in the tile - 1 (6)
            unsigned* tl = tilelist + qd_gs_tile_off(bin, batches_cap) + start;
            for (unsigned t = 0; t < ntile; ++t) {
                const unsigned here = nt - t * 64u < 64u ? nt - t * 64u : 64u;
                tl[t] = ((unsigned)batch << 12) | (t << 6) | (here - 1u);
            }
        }
    }
}

// size class -> solver
template <int BIN, bool VALIDATE>
__device__ __forceinline__ int qd_gs_solve_task(double* rec, double& lam) {
    if constexpr (BIN <= 6) return qd_eig_task<BIN + 2, false, VALIDATE>(rec, lam);
    else if constexpr (BIN == 7) return qd_eig_task<10, true, VALIDATE>(rec, lam);
    else if constexpr (BIN == 8) return qd_eig_task<12, true, VALIDATE>(rec, lam);
    else return qd_eig_task_mem<VALIDATE>(rec, lam);
}
// (register budgets: the unrolled solvers keep the whole packed block live -- 2: 36, 4: 98, 6: 168, 8: 248 VGPRs)
template <int BIN> struct QdGsSolveWaves { static constexpr int v = BIN <= 1 ? 6 : (BIN == 2 ? 4 : (BIN == 3 ? 3 : (BIN <= 6 ? 2 : (BIN == 9 ? 4 : 1)))); };

template <int BIN, bool VALIDATE>
__global__ void __launch_bounds__(256, QdGsSolveWaves<BIN>::v)
qd_k_gs_solve(unsigned char* __restrict__ slabs, const unsigned* __restrict__ gtiles, const unsigned* __restrict__ tilelist,
              size_t batches_cap, unsigned long long* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const unsigned ntile = gtiles[BIN];
    const unsigned* tl = tilelist + qd_gs_tile_off(BIN, batches_cap);
    const unsigned nwaves = gridDim.x * 4u;
    unsigned gt = blockIdx.x * 4u + (threadIdx.x >> 6);
    // Software pipeline over the wave's tiles: a tile's descriptor is fetched two tiles ahead and its task offsets one tile
    // ahead, so only the loads of the block itself are waited for (the kernels are latency bound: waves wait > 80 % of the time).
    const size_t slab_bytes = qd_gs_slab_bytes(VALIDATE);
    auto offset_of = [&](unsigned d) -> unsigned {
        const QdSlab sl = qd_gs_slab(slabs + (size_t)(d >> 12) * slab_bytes, VALIDATE);
        return lane <= (int)(d & 63u) ? sl.lists[qd_gs_list_off(BIN) + (int)((d >> 6) & 63u) * 64 + lane] : 0u;
    };
    unsigned desc = gt < ntile ? tl[gt] : 0u;
    unsigned desc1 = gt + nwaves < ntile ? tl[gt + nwaves] : 0u;
    unsigned off = gt < ntile ? offset_of(desc) : 0u;
    for (; gt < ntile; gt += nwaves) {
        const unsigned desc2 = gt + 2u * nwaves < ntile ? tl[gt + 2u * nwaves] : 0u;
        const unsigned off1 = gt + nwaves < ntile ? offset_of(desc1) : 0u;
        const unsigned batch = desc >> 12;
        const int here = (int)(desc & 63u) + 1;
        const QdSlab sl = qd_gs_slab(slabs + (size_t)batch * slab_bytes, VALIDATE);
        int its = 0;
        if (lane < here) {
            double lam;
            its = qd_gs_solve_task<BIN, VALIDATE>(sl.pool + off, lam);
            sl.lam[qd_gs_list_off(BIN) + (int)((desc >> 6) & 63u) * 64 + lane] = lam;         // dense, coalesced: what the select kernel reads
        }
        desc = desc1; desc1 = desc2; off = off1;
        if (VALIDATE && stats) {
            int sum = its, mx = its;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); mx = max(mx, __shfl_xor(mx, o, 64)); }
            if (lane == 0) {
                atomicAdd(&stats[16], (unsigned long long)here); atomicAdd(&stats[17], (unsigned long long)sum);
                atomicAdd(&stats[18], 1ull); atomicAdd(&stats[19], (unsigned long long)mx);
                atomicAdd(&stats[20 + BIN], (unsigned long long)here);
            }
        }
    }
}

#ifndef QD_SEL_WAVES
#define QD_SEL_WAVES 3
#endif
template <int N, bool VALIDATE>
__global__ void __launch_bounds__(QD_GS_BLOCK, QD_SEL_WAVES)
qd_k_gs_select(const int* __restrict__ env_ids, int env_base, int rec_slot0, QdGsGeom g, int R, const double* __restrict__ params,
               const QdPixelRec* __restrict__ recs, double* __restrict__ zraw, double* __restrict__ occ_out,
               const double* __restrict__ state, int noise_flags, double* __restrict__ eig_out, unsigned char* __restrict__ slabs) {
    constexpr int G = N + 1;
    const QdLayout L = qd_layout(N);
    const int batch = blockIdx.x;
    const QdSlab sl = qd_gs_slab(slabs + (size_t)batch * qd_gs_slab_bytes(VALIDATE), VALIDATE);
    int slot, ch, p0;
    qd_gs_locate(g, batch, slot, ch, p0);
    const int e = env_ids ? env_ids[env_base + slot] : env_base + slot;
    const double* par = params + (size_t)e * L.size;
    const double* st = state + (size_t)e * L.s_size;
    if (qd_radial_replaced(par, st, L, ch, noise_flags)) return;
    const int ps = threadIdx.x, p = p0 + ps;
    if (p >= g.P) return;
    const QdPixelRec* rec = recs + ((size_t)(rec_slot0 + slot) * g.C + ch) * g.P + p;
    double occ[N], lam, resid;
    qd_ground_select<N, VALIDATE>(rec, ps, sl, occ, lam, resid);
    const size_t gp = ((size_t)e * g.C + ch) * g.P + p;
    // hand the sensor stage (qd_k_sensor) the pixel's constant c0 = 2 b + a (2 (Ns - v''_s) + 1):
    // F_{k+1} - F_k = c0 + 2 a (k + eta)   (closed form of the reference's energy differences),
    // b = sum_i A[N][i] (<n_i> - v''_i)
    double b = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) b = fma(par[L.cdd_inv + N * G + i], occ[i] - rec->vpp[i], b);
    const double vs = rec->vpp[N];
    const double Ns = rint(vs);                                 // np.round: half to even
    zraw[gp] = 2.0 * b + par[L.cdd_inv + N * G + N] * (2.0 * (Ns - vs) + 1.0);
    if (occ_out) {
#pragma unroll
        for (int i = 0; i < N; ++i) occ_out[gp * N + i] = occ[i];
    }
    if (VALIDATE && eig_out) { eig_out[gp * 2] = lam; eig_out[gp * 2 + 1] = resid; }
}

// ---------------------------------------------------------------------------
// a14: charge latching (ground_state.py:164 -> qarray LatchingModel.add_latching, source absent).
// UNVERIFIED restatement of its documented behaviour: walk the raster of one channel image in
// row-major order; the state is reset at the start of each row; a new pixel whose occupations
// differ (numpy.isclose sense) from the held state in exactly one dot is accepted with
// probability p_leads[dot], in exactly two dots with p_inter[a][b], otherwise always; a rejected
// pixel keeps the previous (latched) occupations.  One thread per (env, channel): the walk is
// serial.  Rewrites occ in place and corrects the sensor constant c0 of latched pixels:
//   c0 += 2 * sum_i A[N][i] (n_latched_i - n_i).
// ---------------------------------------------------------------------------
template <int N>
__global__ void qd_k_latch(const int* __restrict__ env_ids, int n_env, int R, const double* __restrict__ params,
                           const double* __restrict__ state, double* __restrict__ occ, double* __restrict__ zraw,
                           QdNoiseCfg nz) {
    constexpr int G = N + 1, C = N - 1;
    const QdLayout L = qd_layout(N);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_env * C) return;
    const int slot = t / C, ch = t - slot * C;
    const int e = env_ids ? env_ids[slot] : slot;
    const double* par = params + (size_t)e * L.size;
    const double* st = state + (size_t)e * L.s_size;
    if (qd_radial_replaced(par, st, L, ch, nz.flags)) return;
    const int P = R * R;
    const uint32_t k1 = nz.env_off + (uint32_t)e;
    double* oc = occ + ((size_t)e * C + ch) * P * N;
    double* zc = zraw + ((size_t)e * C + ch) * P;
    double hold[N];
    for (int p = 0; p < P; ++p) {
        double nn[N];
#pragma unroll
        for (int i = 0; i < N; ++i) nn[i] = oc[(size_t)p * N + i];
        bool accept = true;
        if (p % R != 0) {
            int cnt = 0, a = 0, b = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool differ = !(fabs(hold[i] - nn[i]) <= 1e-8 + 1e-5 * fabs(nn[i]));
                if (differ) { if (cnt == 0) a = i; else if (cnt == 1) b = i; cnt++; }
            }
            if (cnt == 1 || cnt == 2) {
                const QdPhilox r = qd_philox4x32_10((uint32_t)p, (uint32_t)ch | (QD_RNG_LATCH << 16), nz.ser_lo, nz.ser_hi, nz.seed, k1);
                const double u = qd_u01(r.v[0], r.v[1]);
                const double pa = (cnt == 1) ? par[L.pleads + a] : par[L.pinter + a * N + b];
                accept = u < pa;
            }
        }
        if (accept) {
#pragma unroll
            for (int i = 0; i < N; ++i) hold[i] = nn[i];
        } else {
            double corr = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) { corr = fma(par[L.cdd_inv + N * G + i], hold[i] - nn[i], corr); oc[(size_t)p * N + i] = hold[i]; }
            zc[p] += 2.0 * corr;
        }
    }
}

// ---------------------------------------------------------------------------
// a15 + a16: sensor stage, one pixel per lane, in place on zraw.
//   in : c0 from qd_k_gs_select     out: signal = sum_{k=-5..4} 1 / (((c0 + 2 a (k + eta)) / gamma)^2 + 1)
// eta = sensor-potential noise (white + telegraph), then radial image noise / replacement.
// grid = (ceil(P/256), C, n_env)
// ---------------------------------------------------------------------------
template <int N>
__global__ void qd_k_sensor(const int* __restrict__ env_ids, int R, const double* __restrict__ params,
                            const double* __restrict__ state, double* __restrict__ zraw, QdNoiseCfg nz) {
    constexpr int G = N + 1;
    const QdLayout L = qd_layout(N);
    const int e = env_ids ? env_ids[blockIdx.z] : blockIdx.z;
    const int ch = blockIdx.y;
    const int P = R * R;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const double* par = params + (size_t)e * L.size;
    const double* st = state + (size_t)e * L.s_size;
    const uint32_t k1 = nz.env_off + (uint32_t)e;
    double* zp = zraw + ((size_t)e * (N - 1) + ch) * P + p;
    if (qd_radial_replaced(par, st, L, ch, nz.flags)) {
        // qarray_base_class.py:466-468: np.random.randn(*z.shape)
        const QdPhilox r = qd_philox4x32_10((uint32_t)p, (uint32_t)ch | (QD_RNG_RADIAL << 16), nz.ser_lo, nz.ser_hi, nz.seed, k1);
        double z0, z1; qd_normal2(r, z0, z1);
        *zp = z0;
        return;
    }
    const double c0 = *zp;
    const double a = par[L.cdd_inv + N * G + N];
    const double gamma = qd_peak_width(par, st, L, ch);
    double eta = 0.0;
    if (nz.flags & 1) {
        // TunnelCoupledChargeSensed.py:354: input noise on the sensor potential (white + telegraph)
        const QdPhilox r = qd_philox4x32_10((uint32_t)p, (uint32_t)ch | (QD_RNG_WHITE << 16), nz.ser_lo, nz.ser_hi, nz.seed, k1);
        double z0, z1; qd_normal2(r, z0, z1);
        eta = par[L.noise + 0] * z0;
        const unsigned long long w = nz.tel[((size_t)e * (N - 1) + ch) * nz.tel_words + (p >> 6)];
        if ((w >> (p & 63)) & 1ull) eta += par[L.noise + 3];
    }
    double s = 0.0;
#pragma unroll
    for (int k = -QD_NPEAK; k < QD_NPEAK; ++k) {
        const double dF = c0 + 2.0 * a * ((double)k + eta);
        const double rr = dF / gamma;
        s += 1.0 / (rr * rr + 1.0);
    }
    if (nz.flags & 2) {
        // qarray_base_class.py:470-493: z + randn * clip(alpha (dist - zero_radius), 0, max_amplitude)
        const double w_ = par[L.scal + 2];
        const int y_ = p / R, x_ = p - y_ * R;
        const double v1 = st[L.s_gate_v + ch], v2 = st[L.s_gate_v + ch + 1];
        const double V1 = qd_linspace(v1 + (-w_), v1 + w_, R, x_), V2 = qd_linspace(v2 + (-w_), v2 + w_, R, y_);
        const double g1 = V1 - st[L.s_gate_gt + ch], g2 = V2 - st[L.s_gate_gt + ch + 1];
        const double dist = sqrt(g1 * g1 + g2 * g2);
        const double alpha = par[L.noise + 7] / par[L.noise + 5];
        const double amp = fmin(fmax(alpha * (dist - par[L.noise + 4]), 0.0), par[L.noise + 7]);
        const QdPhilox r = qd_philox4x32_10((uint32_t)p, (uint32_t)ch | (QD_RNG_RADIAL << 16), nz.ser_lo, nz.ser_hi, nz.seed, k1);
        double z0, z1; qd_normal2(r, z0, z1);
        s += z0 * amp;
    }
    *zp = s;
}

// ---------------------------------------------------------------------------
// a17: exact percentiles (numpy 'linear' method) of the C*P raw values of one
// env by MSB-first radix select on order-preserving 64-bit keys.
// One block per env.  plohi[e] = (p_low, p_high); NaN anywhere -> (NaN, NaN).
// ---------------------------------------------------------------------------
#define QD_PCT_BLOCK 1024

__device__ __forceinline__ unsigned long long qd_key(double x) {
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double qd_unkey(unsigned long long k) {
    unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// The keys of one env, visited by the whole block in step (every thread calls f the same number of times, with `in` false
// past the end, so f may vote across the wave).  CACHED: thread t keeps the keys of values t, t + 1024, ... in registers
// (images of up to 32 768 values: the 16 histogram passes and the rank pass run without touching memory again).
#define QD_PCT_KPT 32
template <bool CACHED>
struct QdPctKeys {
    const double* z; long n;
    unsigned long long keys[CACHED ? QD_PCT_KPT : 1];
    template <class F>
    __device__ __forceinline__ void each(F&& f) const {
        if constexpr (CACHED) {
#pragma unroll
            for (int j = 0; j < QD_PCT_KPT; ++j) {
                if ((long)j * QD_PCT_BLOCK >= n) break;
                f((long)j * QD_PCT_BLOCK + threadIdx.x < n, keys[j]);
            }
        } else {
            for (long i0 = 0; i0 < n; i0 += QD_PCT_BLOCK) {
                const long i = i0 + threadIdx.x;
                f(i < n, i < n ? qd_key(z[i]) : 0ull);
            }
        }
    }
};

// values of the ranks t[0], t[1] (0-based) among n keys, both selections in the same eight passes
template <bool CACHED>
__device__ __forceinline__ void qd_radix_select2(const QdPctKeys<CACHED>& K, long t0, long t1, unsigned (*hist)[256],
                                                 unsigned long long* sh_prefix /*2*/, long* sh_t /*2*/, unsigned long long* out /*2*/) {
    unsigned long long prefix[2] = {0, 0}, mask = 0;
    long t[2] = {t0, t1};
    for (int pass = 7; pass >= 0; --pass) {
        for (int i = threadIdx.x; i < 512; i += blockDim.x) hist[i >> 8][i & 255] = 0;
        __syncthreads();
        const int shift = pass * 8;
        K.each([&](bool valid, unsigned long long k) {
            const unsigned d = (unsigned)(k >> shift) & 255u;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const bool in = valid && (k & mask) == prefix[w];
                // the leading bytes of an image's values are mostly equal: one add per wave instead of 64 colliding LDS atomics
                const unsigned long long act = __ballot(in);
                if (act) {
                    const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)d, __builtin_ctzll(act));
                    if (__ballot(in && d == d0) == act) {
                        if ((int)(threadIdx.x & 63) == __builtin_ctzll(act)) atomicAdd(&hist[w][d0], (unsigned)__builtin_popcountll(act));
                    } else if (in) atomicAdd(&hist[w][d], 1u);
                }
            }
        });
        __syncthreads();
        if (threadIdx.x < 128) {
            // first bin b with count(bins <= b) > t: wave w scans histogram w, 4 bins per lane, wave prefix sum
            const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
            const long tw = w ? t[1] : t[0];
            const unsigned long long pw = w ? prefix[1] : prefix[0];
            const unsigned h4[4] = {hist[w][4 * l], hist[w][4 * l + 1], hist[w][4 * l + 2], hist[w][4 * l + 3]};
            const long own = (long)h4[0] + (long)h4[1] + (long)h4[2] + (long)h4[3];
            long inc = own;
            for (int o = 1; o < 64; o <<= 1) { const long v = (long)__shfl_up((long long)inc, o, 64); if (l >= o) inc += v; }
            const long exc = inc - own;
            const unsigned long long hit = __ballot(inc > tw);
            const int wl = hit ? __builtin_ctzll(hit) : 63;
            if (l == wl) {
                long acc = exc; int b = 4 * l;
                for (int j = 0; j < 4; ++j, ++b) { if (acc + (long)h4[j] > tw) break; acc += h4[j]; }
                if (b > 255) b = 255;
                sh_prefix[w] = pw | ((unsigned long long)b << shift);
                sh_t[w] = tw - acc;
            }
        }
        __syncthreads();
        prefix[0] = sh_prefix[0]; prefix[1] = sh_prefix[1]; t[0] = sh_t[0]; t[1] = sh_t[1];
        mask |= 0xffull << shift;
        __syncthreads();
    }
    out[0] = prefix[0]; out[1] = prefix[1];
}

__device__ __forceinline__ double qd_lerp(double a, double b, double t) {
    const double diff = b - a;
    double r = a + diff * t;
    if (t >= 0.5) r = b - diff * (1.0 - t);
    return r;
}

template <bool CACHED>
__global__ void __launch_bounds__(QD_PCT_BLOCK)
qd_k_percentile(const int* __restrict__ env_ids, long n, const double* __restrict__ zraw, double* __restrict__ plohi) {
    const int e = env_ids ? env_ids[blockIdx.x] : blockIdx.x;
    __shared__ unsigned hist[2][256];
    __shared__ unsigned long long sh_prefix[2];
    __shared__ long sh_t[2];
    __shared__ unsigned long long sh_red[4][QD_PCT_BLOCK / 64];
    __shared__ int sh_nan;
    if (threadIdx.x == 0) sh_nan = 0;
    __syncthreads();
    QdPctKeys<CACHED> K;
    K.z = zraw + (size_t)e * n; K.n = n;
    int has_nan = 0;
    if constexpr (CACHED) {
#pragma unroll
        for (int j = 0; j < QD_PCT_KPT; ++j) {
            K.keys[j] = 0ull;
            const long i = (long)j * QD_PCT_BLOCK + threadIdx.x;
            if (i < n) { const double v = K.z[i]; has_nan |= (v != v); K.keys[j] = qd_key(v); }
        }
    } else {
        for (long i = threadIdx.x; i < n; i += blockDim.x) has_nan |= (K.z[i] != K.z[i]);
    }
    if (has_nan) sh_nan = 1;
    __syncthreads();
    if (sh_nan) { if (threadIdx.x == 0) { plohi[2 * e] = NAN; plohi[2 * e + 1] = NAN; } return; }
    long ip[2], in[2]; double g[2];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const double q = (which == 0 ? 0.5 : 99.5) / 100.0;
        const double virt = (double)(n - 1) * q;             // numpy 'linear' method: (n-1)*quantile
        const double prev = floor(virt);
        ip[which] = (long)prev; if (ip[which] < 0) ip[which] = 0; if (ip[which] > n - 1) ip[which] = n - 1;
        in[which] = ip[which] + 1; if (in[which] > n - 1) in[which] = n - 1;
        g[which] = virt - prev;
    }
    unsigned long long ka[2];
    qd_radix_select2<CACHED>(K, ip[0], ip[1], hist, sh_prefix, sh_t, ka);
    // rank ip+1: equal to ka if enough values <= ka, else the smallest key > ka
    long cnt_le[2] = {0, 0}; unsigned long long mn[2] = {~0ull, ~0ull};
    K.each([&](bool valid, unsigned long long k) {
        if (valid) {
#pragma unroll
            for (int w = 0; w < 2; ++w) { cnt_le[w] += k <= ka[w]; if (k > ka[w] && k < mn[w]) mn[w] = k; }
        }
    });
    // block reductions (sums of cnt_le, minima of mn)
#pragma unroll
    for (int w = 0; w < 2; ++w)
        for (int o = 32; o > 0; o >>= 1) {
            cnt_le[w] += __shfl_xor((long long)cnt_le[w], o, 64);
            const unsigned long long other = (unsigned long long)__shfl_xor((long long)mn[w], o, 64);
            mn[w] = other < mn[w] ? other : mn[w];
        }
    if ((threadIdx.x & 63) == 0) {
        sh_red[0][threadIdx.x >> 6] = (unsigned long long)cnt_le[0]; sh_red[1][threadIdx.x >> 6] = (unsigned long long)cnt_le[1];
        sh_red[2][threadIdx.x >> 6] = mn[0]; sh_red[3][threadIdx.x >> 6] = mn[1];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < 2; ++w) {
            long tot = 0; unsigned long long gm = ~0ull;
            for (int v = 0; v < QD_PCT_BLOCK / 64; ++v) { tot += (long)sh_red[w][v]; gm = sh_red[2 + w][v] < gm ? sh_red[2 + w][v] : gm; }
            unsigned long long kb = ka[w];
            if (in[w] != ip[w]) kb = (tot > in[w]) ? ka[w] : gm;
            plohi[2 * e + w] = qd_lerp(qd_unkey(ka[w]), qd_unkey(kb), g[w]);
        }
    }
}

// ---------------------------------------------------------------------------
// a17 + a22: normalise and write every observation tensor.
// grid = (ceil(P/256), n_env); thread = one (y,x).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float qd_norm(double z, double lo, double hi) {
    if (!(hi > lo)) return 0.0f;
    double v = (z - lo) / (hi - lo);
    v = fmin(fmax(v, 0.0), 1.0);
    return (float)v;
}

template <int N>
__global__ void qd_k_write_obs(const int* __restrict__ env_ids, int R, const double* __restrict__ params,
                               const double* __restrict__ state, const double* __restrict__ zraw,
                               const double* __restrict__ plohi, float* __restrict__ gimg,
                               float* __restrict__ pimg, float* __restrict__ bimg, float* __restrict__ volt) {
    constexpr int C = N - 1, NA = 2 * N - 1;
    const QdLayout L = qd_layout(N);
    const int e = env_ids ? env_ids[blockIdx.y] : blockIdx.y;
    const int P = R * R;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const double lo = plohi[2 * e], hi = plohi[2 * e + 1];
    if (blockIdx.x == 0 && threadIdx.x < NA && volt) {
        const double* par = params + (size_t)e * L.size;
        const double* st = state + (size_t)e * L.s_size;
        const int i = threadIdx.x;
        double v, vlo, vhi;
        if (i < N) { v = st[L.s_gate_v + i]; vlo = par[L.pmin + i]; vhi = par[L.pmax + i]; }
        else { v = st[L.s_barrier_v + i - N]; vlo = par[L.bmin + i - N]; vhi = par[L.bmax + i - N]; }
        const float v32 = (float)v;                               // env.py:512 astype(float32) first
        double t = ((double)v32 - vlo) / (vhi - vlo);
        t = t * 2 - 1;
        volt[(size_t)e * NA + i] = (float)t;
    }
    if (p >= P) return;
    const int y = p / R, x = p - y * R;
    const int pt = x * R + y;                                     // transposed pixel
    const double* ze = zraw + (size_t)e * C * P;
    float v[C], vt[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { v[c] = qd_norm(ze[(size_t)c * P + p], lo, hi); vt[c] = qd_norm(ze[(size_t)c * P + pt], lo, hi); }
    if (gimg) {
#pragma unroll
        for (int c = 0; c < C; ++c) gimg[((size_t)e * P + p) * C + c] = v[c];
    }
    if (bimg) {
#pragma unroll
        for (int c = 0; c < C; ++c) bimg[((size_t)e * C + c) * P + p] = v[c];
    }
    if (pimg) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float a0, a1;
            if (i == 0) { a0 = v[0]; a1 = v[0]; }
            else if (i == N - 1) { a0 = vt[C - 1]; a1 = vt[C - 1]; }
            else { a0 = v[i - 1]; a1 = vt[i]; }
            float2 o; o.x = a0; o.y = a1;
            reinterpret_cast<float2*>(pimg)[((size_t)e * N + i) * P + p] = o;
        }
    }
}

// ---------------------------------------------------------------------------
// a19 + a20 + a21: one thread per env.
// ---------------------------------------------------------------------------
struct QdKalmanCfg { double variance_threshold, process_noise; int direct, n_out; };

// pseudo-inverse of an n x n matrix (row-major, n <= 9) by one-sided Jacobi SVD, numpy.linalg.pinv semantics: singular
// values <= 1e-15 * s_max are dropped.  ONE WAVE per matrix, everything in LDS (M, U, V, P: n*n doubles each, sv: n):
// a sweep is n - 1 (n even) or n rounds of a round-robin schedule, each round rotating up to n / 2 DISJOINT column pairs
// side by side (slot = lane / 12 owns a pair, its lane k < n owns row k of U and V and forms the pair's three column sums
// itself, in row order) -- the serial chain of a sweep is 9 rotations instead of 36 for n = 9.  (One thread per env took
// 490 us per call whatever the batch: ~290 rotations of 3 square roots and 4 divisions each.)
template <int n>
__device__ void qd_pinv_wave(const double* M, double* U, double* V, double* P, double* sv, int lane) {
    constexpr int m = (n + 1) & ~1;                     // players of the schedule (one dummy when n is odd)
    constexpr int SLOTW = 12;
    static_assert(n <= 9 && (m / 2) * SLOTW <= 64, "pair slots must fit one wave");
    for (int i = lane; i < n * n; i += 64) { U[i] = M[i]; V[i] = ((i / n) == (i % n)) ? 1.0 : 0.0; }
    __builtin_amdgcn_wave_barrier();
    const int slot = lane / SLOTW, k = lane - slot * SLOTW;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double offmax = 0.0;
        for (int r = 0; r < m - 1; ++r) {
            int a, b;
            if (slot == 0) { a = m - 1; b = r; }
            else { a = (r + slot) % (m - 1); b = (r - slot + (m - 1)) % (m - 1); }
            const bool mine = slot < m / 2 && a < n && b < n && k < n;
            const int p = a < b ? a : b, q = a < b ? b : a;
            bool rot = false;
            double nup = 0, nuq = 0, nvp = 0, nvq = 0;
            if (mine) {
                double app = 0, aqq = 0, apq = 0;
                for (int kk = 0; kk < n; ++kk) { const double up = U[kk * n + p], uq = U[kk * n + q]; app += up * up; aqq += uq * uq; apq += up * uq; }
                if (apq != 0.0) {
                    const double rel = fabs(apq) / sqrt(app * aqq);
                    if (rel > offmax) offmax = rel;
                    if (rel > 1e-16) {
                        const double zeta = (aqq - app) / (2.0 * apq);
                        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                        const double up = U[k * n + p], uq = U[k * n + q];
                        nup = cs * up - sn * uq; nuq = sn * up + cs * uq;
                        const double vp = V[k * n + p], vq = V[k * n + q];
                        nvp = cs * vp - sn * vq; nvq = sn * vp + cs * vq;
                        rot = true;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();                 // every column sum of the round is formed before a column changes
            if (rot) { U[k * n + p] = nup; U[k * n + q] = nuq; V[k * n + p] = nvp; V[k * n + q] = nvq; }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) offmax = fmax(offmax, __shfl_xor(offmax, o, 64));
        if (!(offmax > 1e-15)) break;
    }
    if (lane < n) {
        double s = 0.0;
        for (int kk = 0; kk < n; ++kk) s += U[kk * n + lane] * U[kk * n + lane];
        sv[lane] = sqrt(s);
    }
    __builtin_amdgcn_wave_barrier();
    double smax = 0.0;
    for (int j = 0; j < n; ++j) if (sv[j] > smax) smax = sv[j];
    const double cutoff = 1e-15 * smax;
    // M = U S V^T with U = Um / sv  =>  pinv = V S^-1 U^T = sum_j V[:,j] Um[:,j]^T / sv_j^2
    for (int idx = lane; idx < n * n; idx += 64) {
        const int i = idx / n, kk = idx - i * n;
        double acc = 0.0;
        for (int j = 0; j < n; ++j)
            if (sv[j] > cutoff) acc += V[i * n + j] * U[kk * n + j] / (sv[j] * sv[j]);
        P[idx] = acc;
    }
    __builtin_amdgcn_wave_barrier();
}

// solve A x = b (n <= 9), Gaussian elimination with partial pivoting; M (n*n) and r (n): the caller's work space
__device__ void qd_solve(const double* A, const double* b, int n, double* x, double* M, double* r) {
    for (int i = 0; i < n * n; ++i) M[i] = A[i];
    for (int i = 0; i < n; ++i) r[i] = b[i];
    for (int c = 0; c < n; ++c) {
        int piv = c; double best = fabs(M[c * n + c]);
        for (int i = c + 1; i < n; ++i) if (fabs(M[i * n + c]) > best) { best = fabs(M[i * n + c]); piv = i; }
        if (piv != c) {
            for (int k = 0; k < n; ++k) { const double t = M[c * n + k]; M[c * n + k] = M[piv * n + k]; M[piv * n + k] = t; }
            const double t = r[c]; r[c] = r[piv]; r[piv] = t;
        }
        const double d = M[c * n + c];
        for (int i = c + 1; i < n; ++i) {
            const double f = M[i * n + c] / d;
            for (int k = c; k < n; ++k) M[i * n + k] -= f * M[c * n + k];
            r[i] -= f * r[c];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = r[i];
        for (int k = i + 1; k < n; ++k) s -= M[i * n + k] * x[k];
        x[i] = s / M[i * n + i];
    }
}

template <int N>
__device__ bool qd_kalman_update(double* mean, double* var, int i, int j, double delta, double R, QdKalmanCfg kc) {
    const int r = i < j ? i : j, c = i < j ? j : i;
    if (R > kc.variance_threshold) return false;
    double nm, nv;
    if (kc.direct) {                                     // DirectUpdater.py:110-112: the prediction replaces the state
        nm = delta; nv = R;
    } else {
        const double Pn = var[r * N + c] + kc.process_noise;
        const double x = mean[r * N + c];
        const double K = Pn / (Pn + R);
        nm = x + K * delta;
        nv = (1 - K) * Pn;
    }
    nm = fmin(fmax(nm, -1.0), 1.0);
    mean[r * N + c] = nm; mean[c * N + r] = nm;
    var[r * N + c] = nv; var[c * N + r] = nv;
    return true;
}

// one WAVE per env (block = 64): Kalman updates by lane 0 on an LDS copy of the estimate, the VGM product and its
// pseudo-inverse by the wave, the ground truth (a 9 x 9 solve) by lane 0
#define QD_UPD_BLOCK 64
template <int N>
__global__ void __launch_bounds__(QD_UPD_BLOCK)
qd_k_update(const int* __restrict__ env_ids, int n_env, const double* __restrict__ params,
                            double* __restrict__ state, const float* __restrict__ values,
                            const float* __restrict__ log_vars, int recompute_gt, QdKalmanCfg kc) {
    constexpr int G = N + 1, C = N - 1;
    __shared__ double sMean[N * N], sVar[N * N], sM[G * G], sU[G * G], sV[G * G], sP[G * G], sSv[G], sX[G];
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= n_env) return;
    const int e = env_ids ? env_ids[t] : t;
    const QdLayout L = qd_layout(N);
    const double* par = params + (size_t)e * L.size;
    double* st = state + (size_t)e * L.s_size;
    double* mean = st + L.s_kmean;
    double* var = st + L.s_kvar;
    if (values && log_vars) {
        for (int i = lane; i < N * N; i += 64) { sMean[i] = mean[i]; sVar[i] = var[i]; }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const int K = kc.n_out;                                   // 3, or 2 in the legacy nearest-neighbour mode
            for (int i = 0; i < C; ++i) {
                const float* vv = values + ((size_t)e * C + i) * K;
                const float* lv = log_vars + ((size_t)e * C + i) * K;
                // env.py:592-618: predictions negated; KalmanUpdater.py:87-90 clamp then exp
                double Rv[3], dl[3];
                for (int k = 0; k < K; ++k) {
                    dl[k] = -(double)vv[k];
                    const double c = fmin(fmax((double)lv[k], -6.0), 2.0);
                    Rv[k] = exp(c);
                }
                if (K == 2) {                                         // KalmanUpdater.py:183-205: [RL, LR], both land on (i, i+1)
                    qd_kalman_update<N>(sMean, sVar, i + 1, i, dl[0], Rv[0], kc);
                    qd_kalman_update<N>(sMean, sVar, i, i + 1, dl[1], Rv[1], kc);
                } else {
                    qd_kalman_update<N>(sMean, sVar, i, i + 1, dl[0], Rv[0], kc);
                    if (i + 2 < N) qd_kalman_update<N>(sMean, sVar, i, i + 2, dl[1], Rv[1], kc);
                    if (i - 1 >= 0) qd_kalman_update<N>(sMean, sVar, i + 1, i - 1, dl[2], Rv[2], kc);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < N * N; i += 64) { mean[i] = sMean[i]; var[i] = sVar[i]; }
        // a20: VGM = pinv(cdd_inv_full @ (-E)), E = [[cgd_est, 0], [0, 1]]  (electrons sign folded in)
        for (int idx = lane; idx < G * G; idx += 64) {
            const int i = idx / G, j = idx - i * G;
            double acc = 0.0;
            for (int k = 0; k < G; ++k) {
                double ekj;
                if (k < N && j < N) ekj = (k == j) ? 1.0 : sMean[k * N + j];
                else ekj = (k == N && j == N) ? 1.0 : 0.0;
                acc += par[L.cdd_inv + i * G + k] * (-ekj);
            }
            sM[idx] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        qd_pinv_wave<G>(sM, sU, sV, sP, sSv, lane);
        for (int idx = lane; idx < G * G; idx += 64) st[L.s_vgm + idx] = sP[idx];
    } else if (recompute_gt) {
        for (int idx = lane; idx < G * G; idx += 64) sP[idx] = st[L.s_vgm + idx];
        __builtin_amdgcn_wave_barrier();
    }
    if (recompute_gt && lane == 0) {
        // a21: virtual = inv(VGM) (vopt - origin)
        double rhs[G];
        for (int i = 0; i < G; ++i) rhs[i] = par[L.vopt + i] - par[L.origin + i];
        qd_solve(sP, rhs, G, sX, sU, sSv);                            // (work space: the pseudo-inverse is done with sU / sSv)
        for (int i = 0; i < N; ++i) st[L.s_gate_gt + i] = (double)(float)sX[i];
        for (int b = 0; b < C; ++b) st[L.s_barrier_gt + b] = (double)(float)par[L.vbopt + b];
        st[L.s_sensor_gt] = sX[N];
    }
}

#endif  // __HIPCC__

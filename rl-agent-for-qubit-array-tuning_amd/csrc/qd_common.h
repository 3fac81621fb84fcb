// qd_common.h -- constants and per-env block layouts shared by the HIP kernels,
// the C-ABI and the CPU-only test harness (qd_hosttest.cpp).  No torch, no STL.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define QD_HD __host__ __device__ __forceinline__
#else
#define QD_HD inline
#endif
// keeps the scheduler from hoisting a whole matrix of LDS loads in front of an
// unrolled row loop (register pressure); no-op on the host
#if defined(__HIP_DEVICE_COMPILE__)
#define QD_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define QD_ROW_FENCE() ((void)0)
#endif

#define QD_MAXN 8            // dots
#define QD_K 32              // kept charge states   (qarray_config.yaml:129)
#define QD_NPEAK 5           // sensor peaks         (TunnelCoupledChargeSensed.py:77)

// Per-env PARAMETER block (float64, constant during an episode), offsets in doubles.
//   cdd_inv  G*G   inverse Maxwell matrix of dots+sensor      (a6)
//   cgd      G*V   negative gate/barrier -> dot/sensor matrix (a6)
//   cbg      nb*G  barrier-gate cross capacitance (positive)  (a10)
//   ufac     N*N   upper U with cdd_inv[:N,:N] = U U^T        (k-best search)
//   uinv     N     1 / U[i][i]                                 (k-best search)
//   alpha    nb    barrier lever arms                         (a10)
//   origin   G     virtual gate origin                        (a5)
//   vopt     G     optimal physical gate voltages             (a21)
//   vbopt    nb    barrier ground truth                       (a21)
//   pmin,pmax N ; bmin,bmax nb   action ranges                (a2)
//   scal     8     tc_base, gamma (coulomb_peak_width), window, vpw_alpha (<0: constant peak width),
//                  vc_on (0/1), vc_alpha, vc_beta (linear voltage-dependent capacitances), spare   (f4)
//   noise    8     white_amp, tel_p01, tel_p10, tel_amp, radial zero_radius, ramp_distance,
//                  full_noise_distance (<=0: none), radial max_amplitude       (a16)
//   pleads   N     latching: lead acceptance probability per dot                (a14)
//   pinter   N*N   latching: inter-dot acceptance probabilities                 (a14)
struct QdLayout {
    int N, G, nb, V;
    int cdd_inv, cgd, cbg, ufac, uinv, alpha, origin, vopt, vbopt, pmin, pmax, bmin, bmax, scal, noise, pleads, pinter, size;
    // STATE block (float64, mutable): vgm G*G, gate_v N, barrier_v nb, gate_gt N,
    // barrier_gt nb, sensor_gt 1, kal_mean N*N, kal_var N*N
    int s_vgm, s_gate_v, s_barrier_v, s_gate_gt, s_barrier_gt, s_sensor_gt, s_kmean, s_kvar, s_size;
};

QD_HD QdLayout qd_layout(int N) {
    QdLayout L;
    L.N = N; L.G = N + 1; L.nb = N - 1; L.V = 2 * N;
    int o = 0;
    L.cdd_inv = o; o += L.G * L.G;
    L.cgd = o;     o += L.G * L.V;
    L.cbg = o;     o += L.nb * L.G;
    L.ufac = o;    o += N * N;
    L.uinv = o;    o += N;
    L.alpha = o;   o += L.nb;
    L.origin = o;  o += L.G;
    L.vopt = o;    o += L.G;
    L.vbopt = o;   o += L.nb;
    L.pmin = o;    o += N;
    L.pmax = o;    o += N;
    L.bmin = o;    o += L.nb;
    L.bmax = o;    o += L.nb;
    L.scal = o;    o += 8;           // tc_base, gamma, window, vpw_alpha, vc_on, vc_alpha, vc_beta, spare
    L.noise = o;   o += 8;
    L.pleads = o;  o += N;
    L.pinter = o;  o += N * N;
    L.size = (o + 1) & ~1;           // keep blocks 16-byte aligned
    o = 0;
    L.s_vgm = o;        o += L.G * L.G;
    L.s_gate_v = o;     o += N;
    L.s_barrier_v = o;  o += L.nb;
    L.s_gate_gt = o;    o += N;
    L.s_barrier_gt = o; o += L.nb;
    L.s_sensor_gt = o;  o += 1;
    L.s_kmean = o;      o += N * N;
    L.s_kvar = o;       o += N * N;
    L.s_size = (o + 1) & ~1;
    return L;
}

// Per-pixel record handed from the candidate kernel to the ground-state kernel
// (global scratch, pixel-major):
//   uint16 idx[32]   candidate indices (base-4 digits, dot 0 most significant), sorted by (E, idx)
//   int32  floor[8]  floor(n_continuous)
//   int32  nvalid    number of valid candidates (<32 => rest are |0..0> padding)
//   double vpp[9]    cgd_full @ v_ext  (first N entries = v')
//   double tc[7]     tunnel couplings of the N-1 adjacent pairs
//   double E[32]     canonical energies of the kept candidates (= the diagonal F of H)
struct __attribute__((aligned(8))) QdPixelRec {
    uint16_t idx[QD_K];
    int32_t fl[QD_MAXN];
    int32_t nvalid;
    int32_t pad;
    double vpp[QD_MAXN + 1];
    double tc[QD_MAXN - 1];
    double E[QD_K];
};

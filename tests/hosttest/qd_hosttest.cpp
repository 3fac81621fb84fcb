// qd_hosttest.cpp -- CPU-only TEST HARNESS around the __host__ __device__ code
// in qd_pixel.h and qd_eig.h.  It exists so the per-pixel device code (sweep,
// continuous state, exact k-best search, sensor stage) and the per-task dense
// eigen-solver of the ground-state kernel can be checked against the oracle /
// numpy in the no-GPU test tier.  It is NOT part of the product: nothing in
// qadapt_hip loads it (the hop structure / task emission / selection phases of the
// ground-state kernel only exist as HIP code).  Built by tests/hosttest/Makefile.
#include <string.h>
#include "qd_pixel.h"
#include "qd_rng.h"
#include "qd_eig.h"
#include <stdlib.h>

template <int N>
static int run_front(const double* par, const double* st, int ch, int R, int p0, int p1,
                     int32_t* states, int32_t* floors, double* vpp_out, double* tc_out,
                     int32_t* nvalid, unsigned long long* stats, double* vd_out, double* e_out) {
    constexpr int G = N + 1, NB = N - 1, V = 2 * N;
    for (int p = p0; p < p1; ++p) {
        int y = p / R, x = p % R;
        double v_ext[V], vpp[G], vd[N], ncont[N], tc[NB > 0 ? NB : 1], isa;
        qd_pixel_front<N>(par, st, ch, R, x, y, v_ext, vpp, vd, ncont, tc, &isa);
        double e[QD_K]; uint16_t id[QD_K]; int32_t fl[N];
        int nv = qd_candidates<N>(par, vd, ncont, e, 1, id, 1, fl, true, stats);
        if (vd_out) memcpy(vd_out + (size_t)p * N, vd, sizeof(double) * N);
        if (e_out) for (int m = 0; m < QD_K; ++m) e_out[(size_t)p * QD_K + m] = m < nv ? e[m] * isa : 0.0;
        static const int DELTA[4] = {-1, 0, 1, 2};
        for (int m = 0; m < QD_K; ++m)
            for (int i = 0; i < N; ++i) {
                int dig = (id[m] >> (2 * (N - 1 - i))) & 3;
                states[((size_t)p * QD_K + m) * N + i] = m < nv ? fl[i] + DELTA[dig] : 0;
            }
        memcpy(floors + (size_t)p * N, fl, sizeof(int32_t) * N);
        memcpy(vpp_out + (size_t)p * G, vpp, sizeof(double) * G);
        memcpy(tc_out + (size_t)p * NB, tc, sizeof(double) * NB);
        nvalid[p] = nv;
    }
    return 0;
}

extern "C" int qdh_front(int N, const double* par, const double* st, int ch, int R, int p0, int p1,
                         int32_t* states, int32_t* floors, double* vpp_out, double* tc_out,
                         int32_t* nvalid, unsigned long long* stats, double* vd_out, double* e_out) {
    switch (N) {
#define C(n) case n: return run_front<n>(par, st, ch, R, p0, p1, states, floors, vpp_out, tc_out, nvalid, stats, vd_out, e_out);
        C(2) C(3) C(4) C(5) C(6) C(7) C(8)
#undef C
    }
    return 1;
}

extern "C" double qdh_peak_width(int N, const double* par, const double* st, int ch) {
    return qd_peak_width(par, st, qd_layout(N), ch);
}

extern "C" double qdh_sensor(int N, const double* par, const double* vpp, const double* occ, double gamma) {
    switch (N) {
#define C(n) case n: return qd_sensor<n>(par, vpp, occ, gamma);
        C(2) C(3) C(4) C(5) C(6) C(7) C(8)
#undef C
    }
    return -1.0;
}

extern "C" void qdh_layout(int N, int* out) {
    QdLayout L = qd_layout(N);
    memcpy(out, &L, sizeof(L));
}
extern "C" int qdh_layout_ints() { return (int)(sizeof(QdLayout) / sizeof(int)); }

extern "C" void qdh_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4) {
    QdPhilox r = qd_philox4x32_10(c0, c1, c2, c3, k0, k1);
    for (int i = 0; i < 4; ++i) out4[i] = r.v[i];
}
extern "C" void qdh_normals(uint32_t k0, uint32_t k1, int n, double* out) {
    for (int i = 0; i < n; i += 2) {
        QdPhilox r = qd_philox4x32_10((uint32_t)(i / 2), 7u, 0u, 0u, k0, k1);
        double a, b; qd_normal2(r, a, b);
        out[i] = a; if (i + 1 < n) out[i + 1] = b;
    }
}

// lowest eigenpair of one packed symmetric block with the kernel's own solver (csrc/qd_eig.h): register version for
// s <= QD_EIG_REG, in-memory version above
extern "C" int qdh_eig_lowest(int s, const double* packed, double* lam, double* x, double* resid, int* iters) {
    switch (s) {
#define C(n) case n: qd_eig_lowest<n, true>(packed, *lam, x, *resid, iters); return 0;
        C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(10) C(12)
#undef C
        // odd sizes above 8 run padded, exactly as the kernel's size classes do
        case 9: { double xx[10]; qd_eig_lowest<10, true>(packed, *lam, xx, *resid, iters, 9); memcpy(x, xx, sizeof(double) * 9); return 0; }
        case 11: { double xx[12]; qd_eig_lowest<12, true>(packed, *lam, xx, *resid, iters, 11); memcpy(x, xx, sizeof(double) * 11); return 0; }
    }
    if (s < 2 || s > 32) return 1;
    const int ne = s * (s + 1) / 2;
    double* M = (double*)malloc(sizeof(double) * (ne + 4 * s));
    if (!M) return 2;
    memcpy(M, packed, sizeof(double) * ne);
    qd_eig_lowest_mem(s, M, M + ne, packed, *lam, *resid, iters);
    memcpy(x, M + ne + 3 * s, sizeof(double) * s);
    free(M);
    return 0;
}

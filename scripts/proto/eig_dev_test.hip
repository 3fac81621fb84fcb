// Debug aid: runs the kernel's eigen-solver (csrc/qd_eig.h) on ONE packed block read from a file, on the device, and prints
// what it returns next to the host build of the same source.  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../../rl-agent-for-qubit-array-tuning_amd/csrc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qd_eig.h"

__global__ void k_mem(int s, double* M, double* work, const double* Morig, double* out) {
    double lam, res; int its = 0;
    qd_eig_lowest_mem(s, M, work, Morig, lam, res, &its);
    if (threadIdx.x == 0) { out[0] = lam; out[1] = res; out[2] = its; }
}
template <int S> __global__ void k_reg(const double* A, double* out) {
    double lam, res, x[S]; int its = 0;
    qd_eig_lowest<S, true>(A, lam, x, res, &its);
    if (threadIdx.x == 0) { out[0] = lam; out[1] = res; out[2] = its; for (int i = 0; i < S; ++i) out[3 + i] = x[i]; }
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    double hdr; if (fread(&hdr, 8, 1, f) != 1) return 1;
    const int s = (int)hdr, ne = s * (s + 1) / 2;
    double* A = (double*)malloc(8 * ne); if (fread(A, 8, ne, f) != (size_t)ne) return 1;
    double *dM, *dW, *dO, *dOut;
    hipMalloc(&dM, 8 * ne); hipMalloc(&dW, 8 * 4 * s); hipMalloc(&dO, 8 * ne); hipMalloc(&dOut, 8 * 64);
    hipMemcpy(dM, A, 8 * ne, hipMemcpyHostToDevice); hipMemcpy(dO, A, 8 * ne, hipMemcpyHostToDevice);
    hipMemset(dW, 0, 8 * 4 * s);
    double out[64], w[128], M[600];
    if (s > QD_EIG_REG) {
        k_mem<<<1, 1>>>(s, dM, dW, dO, dOut);
        hipMemcpy(out, dOut, 8 * 3, hipMemcpyDeviceToHost); hipMemcpy(w, dW, 8 * 4 * s, hipMemcpyDeviceToHost); hipMemcpy(M, dM, 8 * ne, hipMemcpyDeviceToHost);
        printf("device: lam %.17g res %.3e its %g\n", out[0], out[1], out[2]);
        printf("al:"); for (int i = 0; i < s; ++i) printf(" %.6e", w[i]); printf("\nbe:"); for (int i = 0; i < s; ++i) printf(" %.6e", w[s + i]);
        printf("\ntau:"); for (int i = 0; i < s; ++i) printf(" %.6e", w[2 * s + i]); printf("\ny:"); for (int i = 0; i < s; ++i) printf(" %.6e", w[3 * s + i]); printf("\n");
        double* Mh = (double*)malloc(8 * ne); memcpy(Mh, A, 8 * ne); double wh[128]; double lam, res; int its;
        qd_eig_lowest_mem(s, Mh, wh, A, lam, res, &its);
        printf("host:   lam %.17g res %.3e its %d\n", lam, res, its);
        printf("al:"); for (int i = 0; i < s; ++i) printf(" %.6e", wh[i]); printf("\nbe:"); for (int i = 0; i < s; ++i) printf(" %.6e", wh[s + i]); printf("\n");
    }
    else {
        switch (s) {
#define C(n) case n: k_reg<n><<<1, 64>>>(dO, dOut); break;
            C(2) C(3) C(4) C(5) C(6) C(7) C(8)
#undef C
        }
        hipMemcpy(out, dOut, 8 * (3 + s), hipMemcpyDeviceToHost);
        printf("device: lam %.17g res %.3e its %g\nx:", out[0], out[1], out[2]);
        for (int i = 0; i < s; ++i) printf(" %.6e", out[3 + i]);
        double lam, res, x[8]; int its;
        switch (s) {
#define C(n) case n: qd_eig_lowest<n, true>(A, lam, x, res, &its); break;
            C(2) C(3) C(4) C(5) C(6) C(7) C(8)
#undef C
        }
        printf("\nhost:   lam %.17g res %.3e its %d\nx:", lam, res, its);
        for (int i = 0; i < s; ++i) printf(" %.6e", x[i]);
        printf("\n");
    }
    return 0;
}

// raw accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 and after one / two Newton steps (decides how many steps qd_rcp / qd_sqrt_rsqrt need)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = fma(fma(-v, r0, 1.0), r0, r0);
    double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    double y0 = __builtin_amdgcn_rsq(v);
    double y1 = y0 * fma(-0.5 * v * y0, y0, 1.5);
    double y2 = y1 * fma(-0.5 * v * y1, y1, 1.5);
    out[6 * i + 0] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2; out[6 * i + 3] = y0; out[6 * i + 4] = y1; out[6 * i + 5] = y2;
}
int main() {
    const int n = 1 << 20; double *hx = new double[n], *ho = new double[6 * n], *dx, *dout;
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (double)(s >> 11) / 9007199254740992.0; hx[i] = ldexp(1.0 + u, (int)(s % 200) - 100); }
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8); hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n); hipMemcpy(ho, dout, 6 * n * 8, hipMemcpyDeviceToHost);
    double e[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        long double v = hx[i], r = 1.0L / v, y = 1.0L / sqrtl(v);
        for (int j = 0; j < 3; ++j) { double d = (double)fabsl((ho[6 * i + j] - r) / r); if (d > e[j]) e[j] = d; }
        for (int j = 3; j < 6; ++j) { double d = (double)fabsl((ho[6 * i + j] - y) / y); if (d > e[j]) e[j] = d; }
    }
    printf("max rel err rcp raw %.3e, 1 Newton %.3e, 2 Newton %.3e | rsq raw %.3e, 1 Newton %.3e, 2 Newton %.3e\n", e[0], e[1], e[2], e[3], e[4], e[5]);
    return 0;
}

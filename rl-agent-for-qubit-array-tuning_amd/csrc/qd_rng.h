// qd_rng.h -- counter-based random numbers for the stochastic stages (a16):
// Philox4x32-10 (Salmon et al., SC'11), uniform and Box-Muller normal variates.
// Counter-based so every (env, observation, channel, pixel, purpose) tuple has its
// own reproducible stream with no state to store.  __host__ __device__ so the
// known-answer vectors are checked on the CPU too.
#pragma once
#include <math.h>
#include "qd_common.h"

struct QdPhilox { uint32_t v[4]; };

QD_HD void qd_mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    const unsigned long long p = (unsigned long long)a * (unsigned long long)b;
    hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
}

QD_HD QdPhilox qd_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        qd_mulhilo(0xD2511F53u, c0, hi0, lo0);
        qd_mulhilo(0xCD9E8D57u, c2, hi1, lo1);
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    QdPhilox o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// uniform in (0,1): 53 random bits, never 0
QD_HD double qd_u01(uint32_t a, uint32_t b) {
    const unsigned long long x = (((unsigned long long)a << 32) | b) >> 11;
    return ((double)x + 0.5) * (1.0 / 9007199254740992.0);
}

// two independent standard normals from one Philox block
QD_HD void qd_normal2(const QdPhilox& p, double& z0, double& z1) {
    const double u1 = qd_u01(p.v[0], p.v[1]), u2 = qd_u01(p.v[2], p.v[3]);
    const double r = sqrt(-2.0 * log(u1));
    const double t = 6.283185307179586476925 * u2;
    z0 = r * cos(t); z1 = r * sin(t);
}

// purposes (counter word 3)
#define QD_RNG_WHITE 1u
#define QD_RNG_RADIAL 2u
#define QD_RNG_TELEGRAPH 3u
#define QD_RNG_LATCH 4u

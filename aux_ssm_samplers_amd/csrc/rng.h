// rng.h -- Threefry-2x32-20 (Salmon et al., Random123; the block function behind jax.random) and the
// bits -> U[0,1) / N(0,1) maps of the fill kernels.  Restated in oracle/rng_np.py.
#pragma once
#include "smallmat.h"

namespace ax {

AX_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// in/out: counter words (x0, x1); key (k0, k1)
AX_HD void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
    const uint32_t ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
    constexpr int ROT[8] = {13, 15, 26, 6, 17, 29, 16, 24};
    x0 += ks[0];
    x1 += ks[1];
#pragma unroll
    for (int r = 0; r < 20; ++r) {
        x0 += x1;
        x1 = rotl32(x1, ROT[r % 8]);
        x1 ^= x0;
        if (r % 4 == 3) {
            const int j = r / 4 + 1;
            x0 += ks[j % 3];
            x1 += ks[(j + 1) % 3] + (uint32_t)j;
        }
    }
}

// U[0,1): the top 24 (fp32) / 32 (fp64) bits scaled -- never 1.0
template <typename R> AX_HD R bits_to_uniform(uint32_t b);
template <> AX_HD float bits_to_uniform<float>(uint32_t b) { return (float)(b >> 8) * 5.9604644775390625e-8f; }
template <> AX_HD double bits_to_uniform<double>(uint32_t b) { return (double)b * 2.3283064365386963e-10; }

// Box-Muller on two 32-bit words.  fp64: u = (b + 0.5) 2^-32 in (0,1), double math.  fp32: u = ((b >> 8) + 0.5) 2^-24,
// float math throughout (the cSMC kernels draw N of these per time step; fp64 log/cos there cost more than the step).
template <typename R> AX_HD R bits_to_normal(uint32_t b0, uint32_t b1);
template <> AX_HD double bits_to_normal<double>(uint32_t b0, uint32_t b1) {
    const double u1 = ((double)b0 + 0.5) * 2.3283064365386963e-10;
    const double u2 = ((double)b1 + 0.5) * 2.3283064365386963e-10;
    const double r = sqrt(-2.0 * log(u1));
    return r * cos(6.283185307179586476925286766559 * u2);
}
template <> AX_HD float bits_to_normal<float>(uint32_t b0, uint32_t b1) {
    const float u1 = ((float)(b0 >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float u2 = ((float)(b1 >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float r = sqrtf(-2.0f * logf(u1));
    return r * cosf(6.283185307179586f * u2);
}

}  // namespace ax

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

// Box-Muller on one Threefry block (two 32-bit words) -> TWO normals (cos and sin branch).  Normal number `idx` of a stream
// is branch (idx & 1) of the block with counter idx >> 1.  fp64: u = (b + 0.5) 2^-32 in (0,1), double math.  fp32:
// u = ((b >> 8) + 0.5) 2^-24, float math throughout (the cSMC kernels draw N of these per time step).
template <typename R> AX_HD void bits_to_normal2(uint32_t b0, uint32_t b1, R& z0, R& z1);
template <> AX_HD void bits_to_normal2<double>(uint32_t b0, uint32_t b1, double& z0, double& z1) {
    const double u1 = ((double)b0 + 0.5) * 2.3283064365386963e-10;
    const double u2 = ((double)b1 + 0.5) * 2.3283064365386963e-10;
    const double r = sqrt(-2.0 * log(u1));
    const double a = 6.283185307179586476925286766559 * u2;
    z0 = r * cos(a);
    z1 = r * sin(a);
}
template <> AX_HD void bits_to_normal2<float>(uint32_t b0, uint32_t b1, float& z0, float& z1) {
    const float u1 = ((float)(b0 >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float u2 = ((float)(b1 >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float r = sqrtf(-2.0f * logf(u1));
    const float a = 6.283185307179586f * u2;
    z0 = r * cosf(a);
    z1 = r * sinf(a);
}
// counter words of block `blk` of a stream
AX_HD void stream_counter(uint32_t stream, unsigned long long blk, uint32_t& x0, uint32_t& x1) {
    x0 = (uint32_t)(blk & 0xffffffffull);
    x1 = stream ^ (uint32_t)((blk >> 32) << 16);
}
// normal number idx of (key, stream)
template <typename R> AX_HD R stream_normal(uint32_t k0, uint32_t k1, uint32_t stream, unsigned long long idx) {
    uint32_t x0, x1;
    stream_counter(stream, idx >> 1, x0, x1);
    threefry2x32(k0, k1, x0, x1);
    R z0, z1;
    bits_to_normal2<R>(x0, x1, z0, z1);
    return (idx & 1) ? z1 : z0;
}
template <typename R> AX_HD R stream_uniform(uint32_t k0, uint32_t k1, uint32_t stream, unsigned long long idx) {
    uint32_t x0, x1;
    stream_counter(stream, idx, x0, x1);
    threefry2x32(k0, k1, x0, x1);
    return bits_to_uniform<R>(x0);
}

}  // namespace ax

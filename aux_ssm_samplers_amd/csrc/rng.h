// rng.h -- Threefry-2x32-20 (Salmon et al., Random123; the block function behind jax.random) and the
// bits -> U[0,1) / N(0,1) maps of the fill kernels.  Restated in oracle/rng_np.py.
#pragma once
#include "smallmat.h"
#include "det_math.h"

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

// cos(2 pi u), sin(2 pi u) for u in [0, 1): q = rint(4u), f = u - q/4 (exact, |f| <= 1/8), theta = 2 pi f in
// [-pi/4, pi/4], the fdlibm k_sin/k_cos polynomials on theta, then the quadrant rotation.  Explicit fma everywhere so that
// the result does not depend on the contraction setting of the translation unit.  (The libm sin/cos carry their
// large-argument reduction inline on the GPU -- ~100 instructions that an angle in [0, 2 pi) never needs.)
AX_HD float rng_fma(float a, float b, float c) { return fmaf(a, b, c); }
AX_HD double rng_fma(double a, double b, double c) { return fma(a, b, c); }
template <typename R> struct SinCosPoly;
template <> struct SinCosPoly<float> {
    static constexpr int NS = 4, NC = 4;
    static constexpr float S[4] = {-1.6666667163e-01f, 8.3333337680e-03f, -1.9841270114e-04f, 2.7557314297e-06f};
    static constexpr float Cc[4] = {4.1666667908e-02f, -1.3888889225e-03f, 2.4801587642e-05f, -2.7557314297e-07f};
};
template <> struct SinCosPoly<double> {
    static constexpr int NS = 6, NC = 6;
    static constexpr double S[6] = {-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                                    2.75573137070700676789e-06,  -2.50507602534068634195e-08, 1.58969099521155010221e-10};
    static constexpr double Cc[6] = {4.16666666666666019037e-02,  -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                                     -2.75573143513906633035e-07, 2.08757232129817482790e-09,  -1.13596475577881948265e-11};
};
template <typename R> AX_HD void sincos_2pi(R u, R& c, R& s) {
    using P = SinCosPoly<R>;
    const R q = rint((R)4 * u);  // 0..4
    const R f = rng_fma((R)-0.25, q, u);
    const R th = (R)6.283185307179586476925286766559 * f;
    const R z = th * th;
    R ps = P::S[P::NS - 1], pc = P::Cc[P::NC - 1];
#pragma unroll
    for (int k = P::NS - 2; k >= 0; --k) ps = rng_fma(ps, z, P::S[k]);
#pragma unroll
    for (int k = P::NC - 2; k >= 0; --k) pc = rng_fma(pc, z, P::Cc[k]);
    const R sn = rng_fma(th * z, ps, th);
    const R cs = rng_fma(z * z, pc, rng_fma((R)-0.5, z, (R)1));
    const int qi = (int)q & 3;
    const R a = (qi & 1) ? sn : cs, b = (qi & 1) ? cs : sn;  // cos = +-a, sin = +-b
    c = (qi == 1 || qi == 2) ? -a : a;
    s = (qi >= 2) ? -b : b;
}

// CONTRACT of the normal transform (restated in oracle/rng_np.py, to rounding): the BITS are pinned (Threefry-2x32-20, Random123 KATs; uniforms are
// bit-exact against the oracle); the normals are pinned to a TOLERANCE -- fp64: det_log<true> + sqrt + sincos_2pi with explicit fma, within 1e-12 of
// the oracle's libm / no-fma restatement; fp32 on the DEVICE: the hardware v_log_f32 / v_sqrt_f32 (1 ulp each), within 2e-5 relative / 2e-6 absolute
// of the oracle (tests/test_rng.py), so device fp32 normals are reproducible on the device only: every bit-exact cSMC / PIT check that uses keyed
// noise draws it on the device first (csmc/_device.py::key_noise) and hands the arrays to the oracle.
// Box-Muller on one Threefry block (two 32-bit words) -> TWO normals (cos and sin branch).  Normal number `idx` of a stream
// is branch (idx & 1) of the block with counter idx >> 1.  fp64: u = (b + 0.5) 2^-32 in (0,1), double math.  fp32:
// u = ((b >> 8) + 0.5) 2^-24, float math throughout (the cSMC kernels draw N of these per time step).
template <typename R> AX_HD void bits_to_normal2(uint32_t b0, uint32_t b1, R& z0, R& z1);
template <> AX_HD void bits_to_normal2<double>(uint32_t b0, uint32_t b1, double& z0, double& z1) {
    const double u1 = ((double)b0 + 0.5) * 2.3283064365386963e-10;
    const double u2 = ((double)b1 + 0.5) * 2.3283064365386963e-10;
    // det_log: the fdlibm-style sequence of det_math.h (about thirty fp64 operations; libm's log is about sixty on the device, and the
    // chain-shared scans draw their noise inside memory-bound passes where every fp64 instruction shows).  1-2 ulp, as before.
    const double r = sqrt(-2.0 * det_log<true>(u1));
    double c, s;
    sincos_2pi<double>(u2, c, s);
    z0 = r * c;
    z1 = r * s;
}
template <> AX_HD void bits_to_normal2<float>(uint32_t b0, uint32_t b1, float& z0, float& z1) {
    const float u1 = ((float)(b0 >> 8) + 0.5f) * 5.9604644775390625e-8f;
    const float u2 = ((float)(b1 >> 8) + 0.5f) * 5.9604644775390625e-8f;
    // device: the hardware log2 / sqrt (1 ulp each) instead of libm's correctly-rounded sequences (~35 instructions): a normal deviate does
    // not need them, and the cSMC forward pass draws N of these per time step.  (The host build keeps libm; the two agree to ~1e-7.)
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln 2 log2(u1)
#else
    const float r = sqrtf(-2.0f * logf(u1));
#endif
    float c, s;
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cos_f32 / v_sin_f32 take their argument in REVOLUTIONS (cos(2 pi u), sin(2 pi u)), u in [0, 1): two instructions instead of the quadrant split
    // and two polynomials (device fp32 normals are reproducible on the device only, see the contract above)
    c = __builtin_amdgcn_cosf(u2);
    s = __builtin_amdgcn_sinf(u2);
#else
    sincos_2pi<float>(u2, c, s);
#endif
    z0 = r * c;
    z1 = r * s;
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
// uniform number idx of (key, stream): word (idx & 1) of block idx >> 1
template <typename R> AX_HD R stream_uniform(uint32_t k0, uint32_t k1, uint32_t stream, unsigned long long idx) {
    uint32_t x0, x1;
    stream_counter(stream, idx >> 1, x0, x1);
    threefry2x32(k0, k1, x0, x1);
    return bits_to_uniform<R>((idx & 1) ? x1 : x0);
}
// both numbers of block blk
template <typename R> AX_HD void stream_normal2(uint32_t k0, uint32_t k1, uint32_t stream, unsigned long long blk, R& z0, R& z1) {
    uint32_t x0, x1;
    stream_counter(stream, blk, x0, x1);
    threefry2x32(k0, k1, x0, x1);
    bits_to_normal2<R>(x0, x1, z0, z1);
}
template <typename R> AX_HD void stream_uniform2(uint32_t k0, uint32_t k1, uint32_t stream, unsigned long long blk, R& u0, R& u1) {
    uint32_t x0, x1;
    stream_counter(stream, blk, x0, x1);
    threefry2x32(k0, k1, x0, x1);
    u0 = bits_to_uniform<R>(x0);
    u1 = bits_to_uniform<R>(x1);
}


// ---- noise generated INSIDE its first consumer (chain-minor layout; auxssm_kalman_sweep_keyed) ----------------------------------------
// The D normals of chain c at one time step sit at flat indices base + k stride (k < D) of a (T, D, C) array, base = (t D) C + c, stride = C:
// chains c (even) and c + 1 share every Threefry block.  The lane pair splits the blocks (the even lane computes those of the even
// components, the odd lane those of the odd ones, each both outputs) and swaps the halves with one DPP move, so a lane pays for D / 2 blocks
// and gets the values auxssm_rng_normal(key, stream 0) puts at those indices, bit for bit.  Needs C even and both lanes of a pair active.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float swap_neighbour(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ double swap_neighbour(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
#endif
template <typename R, int D> AX_HD void normals_cm(uint32_t k0, uint32_t k1, long long base, long long stride, R* out) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool odd = base & 1;
    const long long be = base - (odd ? 1 : 0);  // the even partner's index
#pragma unroll
    for (int k = 0; k + 1 < D; k += 2) {
        const int kk = odd ? k + 1 : k;
        R z0, z1;
        stream_normal2<R>(k0, k1, 0, (unsigned long long)((be + kk * stride) >> 1), z0, z1);
        const R recv = swap_neighbour(odd ? z0 : z1);
        out[kk] = odd ? z1 : z0;
        out[odd ? k : k + 1] = recv;
    }
    if (D & 1) {
        R z0, z1;
        stream_normal2<R>(k0, k1, 0, (unsigned long long)((be + (D - 1) * stride) >> 1), z0, z1);
        out[D - 1] = odd ? z1 : z0;
    }
#else
#pragma unroll
    for (int k = 0; k < D; ++k) out[k] = stream_normal<R>(k0, k1, 0, (unsigned long long)(base + k * stride));
#endif
}

}  // namespace ax

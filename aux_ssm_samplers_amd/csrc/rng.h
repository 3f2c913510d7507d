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
// bit-exact against the oracle); the normals are pinned to a TOLERANCE -- fp64: the table-driven transform below, within 1e-12 (relative) / 1e-13
// (absolute) of the oracle's libm restatement (its own error is ~1e-15); fp32 on the DEVICE: the hardware v_log_f32 / v_sqrt_f32 (1 ulp each), within
// 2e-5 relative / 2e-6 absolute of the oracle (tests/test_rng.py), so device fp32 normals are reproducible on the device only: every bit-exact cSMC /
// PIT check that uses keyed noise draws it on the device first (csmc/_device.py::key_noise) and hands the arrays to the oracle.
// Box-Muller on one Threefry block (two 32-bit words) -> TWO normals (cos and sin branch).  Normal number `idx` of a stream
// is branch (idx & 1) of the block with counter idx >> 1.  fp64: u = (b + 0.5) 2^-32 in (0,1), double math.  fp32:
// u = ((b >> 8) + 0.5) 2^-24, float math throughout (the cSMC kernels draw N of these per time step).
//
// fp64 transform (round 4).  The chain-shared sweep draws four fp64 normals per chain and time step INSIDE its streaming passes, which are bound by
// vector-instruction issue, not by HBM: the fdlibm-style log (a division + a degree-7 polynomial), the compiler's range-checked sqrt and the quadrant
// split + two degree-6 polynomials of sincos cost ~108 instructions per Box-Muller pair next to the 72 of its Threefry block.  Both transcendental
// arguments are 32-bit integers, so a small table does the range reduction and short polynomials the rest (~48 instructions, same accuracy):
//   radius   u1 = m 2^e, m in [1/2, 1); F_j = (256 + j)/512 the nearest table point, t = (m - F_j)/F_j in [-2^-9, 2^-9];
//            -2 ln u1 = e (-2 ln 2) + (-2 ln F_j) - 2 log1p(t), log1p by its degree-5 Taylor polynomial (next term 1e-17); near u1 = 1 (e = 0, j = 256)
//            the first two terms vanish EXACTLY, so small radii keep their relative accuracy; sqrt by v_rsq_f64 + one Goldschmidt + one Newton step
//            (the argument is in [4.6e-10, 46]: no range scaling);
//   angle    b1 = 2^24 j + s with s in [-2^23, 2^23): (cos, sin)(2 pi (b1 + 1/2) 2^-32) = rotation of the table point (C_j, S_j) = (cos, sin)(2 pi j/256) by
//            theta = (s + 1/2) 2 pi 2^-32, |theta| <= pi/256: degree-5 / degree-6 Taylor polynomials (next terms 1e-17 / 1e-20).
// The tables (rng_tables.h, generated: tools/gen_rng_tables.py; 8 KB) are read through a TABLE PROVIDER: NormTabGlobal gathers from global memory (every kernel
// that is not bound by its draws), NormTabLds from a copy the workgroup staged in LDS (the fused passes).  Same operations either way: same bits.
#include "rng_tables.h"
static const double RNG_LOG_TAB_HOST[2 * 257] = AX_RNG_LOG_TAB_INIT;
static const double RNG_TURN_TAB_HOST[2 * 256] = AX_RNG_TURN_TAB_INIT;
#if defined(__HIPCC__)
static __device__ const double __attribute__((aligned(16))) RNG_LOG_TAB_DEV[2 * 257] = AX_RNG_LOG_TAB_INIT;
static __device__ const double __attribute__((aligned(16))) RNG_TURN_TAB_DEV[2 * 256] = AX_RNG_TURN_TAB_INIT;
#endif
constexpr int RNG_TAB_DOUBLES = 2 * 257 + 2 * 256 + 2;  // log table, two doubles of padding (16-byte alignment of the turn table), turn table
constexpr int RNG_TAB_TURN_OFF = 2 * 257 + 2;
struct NormTabGlobal {
    AX_HD void log_entry(int j, double& inv, double& L) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const double2 e = reinterpret_cast<const double2*>(RNG_LOG_TAB_DEV)[j];
        inv = e.x, L = e.y;
#else
        inv = RNG_LOG_TAB_HOST[2 * j], L = RNG_LOG_TAB_HOST[2 * j + 1];
#endif
    }
    AX_HD void turn_entry(int j, double& c, double& s) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const double2 e = reinterpret_cast<const double2*>(RNG_TURN_TAB_DEV)[j];
        c = e.x, s = e.y;
#else
        c = RNG_TURN_TAB_HOST[2 * j], s = RNG_TURN_TAB_HOST[2 * j + 1];
#endif
    }
};
#if defined(__HIPCC__)
// the workgroup's LDS copy: RNG_TAB_DOUBLES doubles at a 16-byte aligned address, filled by stage() (ends with a barrier)
struct NormTabLds {
    const double* p;
    __device__ __forceinline__ static void stage(double* lds, bool sync = true) {
        // (all of a lane's loads first, then its LDS writes: the two tables are 513 16-byte entries, i.e. up to nine trips of a one-wave workgroup)
        const double2* s0 = reinterpret_cast<const double2*>(RNG_LOG_TAB_DEV);
        const double2* s1 = reinterpret_cast<const double2*>(RNG_TURN_TAB_DEV);
        double2* d0 = reinterpret_cast<double2*>(lds);
        double2* d1 = reinterpret_cast<double2*>(lds + RNG_TAB_TURN_OFF);
        const int B = blockDim.x;
        for (int i = threadIdx.x; i < 257; i += 4 * B) {
            const bool p1 = i + B < 257, p2 = i + 2 * B < 257, p3 = i + 3 * B < 257;
            const bool q0 = i < 256, q1 = i + B < 256, q2 = i + 2 * B < 256, q3 = i + 3 * B < 256;
            const double2 z{0, 0};
            const double2 a0 = s0[i], a1 = p1 ? s0[i + B] : z, a2 = p2 ? s0[i + 2 * B] : z, a3 = p3 ? s0[i + 3 * B] : z;
            const double2 b0 = q0 ? s1[i] : z, b1 = q1 ? s1[i + B] : z, b2 = q2 ? s1[i + 2 * B] : z, b3 = q3 ? s1[i + 3 * B] : z;
            d0[i] = a0;
            if (p1) d0[i + B] = a1;
            if (p2) d0[i + 2 * B] = a2;
            if (p3) d0[i + 3 * B] = a3;
            if (q0) d1[i] = b0;
            if (q1) d1[i + B] = b1;
            if (q2) d1[i + 2 * B] = b2;
            if (q3) d1[i + 3 * B] = b3;
        }
        if (sync) __syncthreads();
    }
    __device__ __forceinline__ void log_entry(int j, double& inv, double& L) const {
        const double2 e = reinterpret_cast<const double2*>(p)[j];
        inv = e.x, L = e.y;
    }
    __device__ __forceinline__ void turn_entry(int j, double& c, double& s) const {
        const double2 e = reinterpret_cast<const double2*>(p + RNG_TAB_TURN_OFF)[j];
        c = e.x, s = e.y;
    }
};
#endif
// sqrt of x in [2^-40, 2^6]: no range scaling, no special cases (the compiler's sqrt carries both)
AX_HD double sqrt_pos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
#else
    return sqrt(x);
#endif
}
template <typename TAB> AX_HD void bm_fp64(uint32_t b0, uint32_t b1, const TAB& tab, double& z0, double& z1) {
    // radius: r = sqrt(-2 ln u1), u1 = (b0 + 1/2) 2^-32
    const double v = (double)b0 + 0.5;
#if defined(__HIP_DEVICE_COMPILE__)
    const double mant = __builtin_amdgcn_frexp_mant(v);
    const int ex = __builtin_amdgcn_frexp_exp(v);
#else
    int ex;
    const double mant = frexp(v, &ex);
#endif
    const double jd = rint(fma(mant, 512.0, -256.0));  // 0 .. 256
    double inv, Lj;
    tab.log_entry((int)jd, inv, Lj);
    const double t = fma(jd, -0.001953125, mant - 0.5) * inv;  // (m - F_j) / F_j: the difference is exact
    const double dk = (double)(ex - 32);
    double p = fma(t, -0.4, 0.5);  // -2 log1p(t) = -2 t + t^2 (1 - 2/3 t + 1/2 t^2 - 2/5 t^3)
    p = fma(p, t, -0.66666666666666663);
    p = fma(p, t, 1.0);
    double L = fma(dk, AX_RNG_NEG2LN2, Lj);
    L = fma(t, -2.0, L);
    L = fma(p, t * t, L);
    const double r = sqrt_pos(L);
    // angle
    const int jj = (int)((b1 + 0x800000u) >> 24);  // nearest table point (sector 256 = sector 0: the sum wraps)
    const int sl = (int)(b1 << 8) >> 8;            // b1 = 2^24 jj + sl (mod 2^32), sl in [-2^23, 2^23)
    double C, S;
    tab.turn_entry(jj, C, S);
    const double th = fma((double)sl, AX_RNG_TURN_SCALE, 0.5 * AX_RNG_TURN_SCALE);
    const double z = th * th;
    const double sf = fma(th * z, fma(z, 8.3333333333333332e-3, -1.6666666666666666e-1), th);
    double cf = fma(z, -1.3888888888888889e-3, 4.1666666666666664e-2);
    cf = fma(cf, z, -0.5);
    cf = fma(cf, z, 1.0);
    z0 = r * fma(-S, sf, C * cf);
    z1 = r * fma(C, sf, S * cf);
}
template <typename R> AX_HD void bits_to_normal2(uint32_t b0, uint32_t b1, R& z0, R& z1);
template <> AX_HD void bits_to_normal2<double>(uint32_t b0, uint32_t b1, double& z0, double& z1) { bm_fp64(b0, b1, NormTabGlobal{}, z0, z1); }
// the same through a table provider (fp32 draws use the hardware functions and no table)
template <typename R, typename TAB> AX_HD void bits_to_normal2_t(uint32_t b0, uint32_t b1, const TAB& tab, R& z0, R& z1) {
    if constexpr (sizeof(R) == 8) bm_fp64(b0, b1, tab, z0, z1);
    else bits_to_normal2<R>(b0, b1, z0, z1);
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
template <typename R, typename TAB = NormTabGlobal>
AX_HD void stream_normal2(uint32_t k0, uint32_t k1, uint32_t stream, unsigned long long blk, R& z0, R& z1, const TAB& tab = TAB{}) {
    uint32_t x0, x1;
    stream_counter(stream, blk, x0, x1);
    threefry2x32(k0, k1, x0, x1);
    bits_to_normal2_t<R>(x0, x1, tab, z0, z1);
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
// fp64: the odd lane turns its angle back by a quarter (b1 - 2^30: table point j - 64, whose entry is (S_j, -C_j) EXACTLY, rng_tables.h), so its transform returns
// (r sin, -r cos) by the very operations that give the even lane (r cos, r sin): both parities keep the first output and send the second, and no value is selected
// by parity before the exchange (selects on doubles are two instructions each, and indexing `out` by a lane-dependent component cost a compare-select chain per slot).
template <typename R, int D, typename TAB = NormTabGlobal>
AX_HD void normals_cm(uint32_t k0, uint32_t k1, long long base, long long stride, R* out, const TAB& tab = TAB{}) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool odd = base & 1;
    const long long be = base - (odd ? 1 : 0);  // the even partner's index
    const uint32_t quarter = odd ? 0x40000000u : 0u, sgn = odd ? 0x80000000u : 0u;
    auto pair = [&](int kk, R& keep, R& send) {  // the lane's block: keep = its own normal, send = its partner's
        uint32_t x0, x1;
        stream_counter(0, (unsigned long long)((be + kk * stride) >> 1), x0, x1);
        threefry2x32(k0, k1, x0, x1);
        if constexpr (sizeof(R) == 8) {
            double kp, sd;
            bm_fp64(x0, x1 - quarter, tab, kp, sd);
            keep = kp;
            send = __hiloint2double(__double2hiint(sd) ^ (int)sgn, __double2loint(sd));
        } else {
            R z0, z1;
            bits_to_normal2<R>(x0, x1, z0, z1);
            keep = odd ? z1 : z0;
            send = odd ? z0 : z1;
        }
    };
#pragma unroll
    for (int k = 0; k + 1 < D; k += 2) {
        R keep, send;
        pair(odd ? k + 1 : k, keep, send);
        const R recv = swap_neighbour(send);
        out[k] = odd ? recv : keep;
        out[k + 1] = odd ? keep : recv;
    }
    if (D & 1) {
        R keep, send;
        pair(D - 1, keep, send);
        out[D - 1] = keep;
    }
#else
#pragma unroll
    for (int k = 0; k < D; ++k) out[k] = stream_normal<R>(k0, k1, 0, (unsigned long long)(base + k * stride));
#endif
}

}  // namespace ax

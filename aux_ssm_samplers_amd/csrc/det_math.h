// det_math.h -- bit-reproducible exp / log for the conditional-SMC kernels.
//
// Ancestor indices must be bit-exact against the CPU oracle (BASELINE north_star).  They are the result of
// searchsorted(cumsum(exp(lw - logsumexp(lw))), r), so exp and log themselves have to return identical bits on the
// GPU and on the host.  libm / ocml do not promise that, hence these fixed operation sequences: only IEEE-754
// +, -, *, /, fma, rint and integer bit manipulation, each correctly rounded on gfx950 and on x86-64.
// Files using this header are compiled with -ffp-contract=off so that no other fusion happens.
// Accuracy: ~1 ulp (fp32), ~1-2 ulp (fp64); algorithms after fdlibm's e_expf/e_logf/e_exp/e_log.
// oracle/csmc_ref.c carries its own independent restatement of the same sequences.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AXD_HD __host__ __device__ __forceinline__
#else
#define AXD_HD inline
#endif

namespace ax {

AXD_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
AXD_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
AXD_HD double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
AXD_HD uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }

// ---- exp, fp32: x = k ln2 + r, |r| <= ln2/2, degree-7 Taylor in Horner form with fma ----------------
AXD_HD float det_exp(float x) {
    // no early returns: the special cases are selected at the end, so that the GPU runs one straight instruction stream (three nested
    // exec-mask branches per call otherwise); inside the range the operation sequence and its bits are unchanged
    const bool isnan = !(x == x), big = x > 88.72f, small = x < -87.3f;  // results below the normal range are flushed to +0 (documented)
    const float xs = (isnan || big || small) ? 0.0f : x;
    const float kf = rintf(xs * 1.44269504088896341f);
    float r = fmaf(-kf, 6.93145751953125e-1f, xs);     // ln2 hi (few mantissa bits -> kf*hi exact)
    r = fmaf(-kf, 1.42860682030941723212e-6f, r);      // ln2 lo
    float p = 1.9841270114e-4f;                         // 1/5040
    p = fmaf(p, r, 1.3888889225e-3f);                   // 1/720
    p = fmaf(p, r, 8.3333337670e-3f);                   // 1/120
    p = fmaf(p, r, 4.1666667908e-2f);                   // 1/24
    p = fmaf(p, r, 1.6666667163e-1f);                   // 1/6
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    const int k = (int)kf;                              // in [-126, 128]
#if defined(__HIP_DEVICE_COMPILE__)
    // p in [0.7, 1.5] and the result is a normal number (or overflows to +inf exactly as the two-step product does): ldexp is the same exact scaling,
    // one instruction (v_ldexp_f32) instead of seven
    float res = __builtin_ldexpf(p, k);
#else
    // scale by 2^k in two exact steps (k may be 128)
    const int k1 = k / 2, k2 = k - k1;
    float res = p * u2f((uint32_t)(k1 + 127) << 23) * u2f((uint32_t)(k2 + 127) << 23);
#endif
    res = small ? 0.0f : res;
    res = big ? INFINITY : res;
    return isnan ? x : res;
}

// ---- log, fp32 (fdlibm e_logf): x = 2^e * m, m in [sqrt(1/2), sqrt(2)); f = m - 1; s = f/(2+f) ------------------
AXD_HD float det_log(float x) {
    if (!(x == x)) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    uint32_t ix = f2u(x);
    int e = 0;
    if (ix < 0x00800000u) {  // subnormal: scale up by 2^25
        x = x * 33554432.0f;
        ix = f2u(x);
        e = -25;
    }
    e += (int)(ix >> 23) - 127;
    ix &= 0x007fffffu;
    const uint32_t i = (ix + (0x95f64u << 3)) & 0x800000u;   // m >= sqrt(2) -> halve
    const float m = u2f(ix | (i ^ 0x3f800000u));
    e += (int)(i >> 23);
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * fmaf(w, 0.24279078841f, 0.40000972152f);
    const float t2 = z * fmaf(w, 0.28498786688f, 0.66666662693f);
    const float R = t2 + t1;
    const float hfsq = 0.5f * f * f;
    const float dk = (float)e;
    // log(x) = dk*ln2_hi - ((hfsq - (s*(hfsq+R) + dk*ln2_lo)) - f)
    return fmaf(dk, 6.9313812256e-01f, -((hfsq - fmaf(s, hfsq + R, dk * 9.0580006145e-06f)) - f));
}

// ---- exp, fp64 (fdlibm e_exp structure with fma) -------------------------------------------------------------------
AXD_HD double det_exp(double x) {
    if (!(x == x)) return x;
    if (x > 709.78) return INFINITY;
    if (x < -708.0) return 0.0;              // flushed below the normal range (documented)
    const double kf = rint(x * 1.44269504088896338700e+00);
    const double hi = fma(-kf, 6.93147180369123816490e-01, x);
    const double lo = kf * 1.90821492927058770002e-10;
    const double r = hi - lo;
    const double t = r * r;
    double c = 4.13813679705723846039e-08;
    c = fma(c, t, -1.65339022054652515390e-06);
    c = fma(c, t, 6.61375632143793436117e-05);
    c = fma(c, t, -2.77777777770155933842e-03);
    c = fma(c, t, 1.66666666666666019037e-01);
    c = r - t * c;
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const int k = (int)kf;
    const int k1 = k / 2, k2 = k - k1;
    return y * u2d((uint64_t)(k1 + 1023) << 52) * u2d((uint64_t)(k2 + 1023) << 52);
}

// ---- log, fp64 (fdlibm e_log) ----------------------------------------------------------------------------------------
// NORMAL: the caller guarantees a positive, finite, normal x (the Box-Muller uniforms (b + 0.5) 2^-32 are in [2^-33, 1)): the same value bit for bit
// without the special-case branches, which split the noise-drawing scan passes into a dozen basic blocks.
template <bool NORMAL = false> AXD_HD double det_log(double x) {
    uint64_t ix = d2u(x);
    int e = 0;
    if constexpr (!NORMAL) {
        if (!(x == x)) return x;
        if (x < 0.0) return NAN;
        if (x == 0.0) return -INFINITY;
        if (x == INFINITY) return x;
        if (ix < 0x0010000000000000ull) {
            x = x * 18014398509481984.0;  // 2^54
            ix = d2u(x);
            e = -54;
        }
    }
    uint32_t hx = (uint32_t)(ix >> 32);
    e += (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    const uint32_t i = (hx + 0x95f64u) & 0x100000u;
    ix = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (ix & 0xffffffffull);
    e += (int)(i >> 20);
    const double m = u2d(ix);
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                              6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

}  // namespace ax

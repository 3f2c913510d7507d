/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (never linked or loaded by the product).
 *
 * Plain-C, single-threaded restatement of the reference's conditional SMC sweep:
 *   aux_samplers/_primitives/csmc/csmc.py      _csmc :69-107, _backward_scanning_pass :110-124,
 *                                              _backward_sampling_pass :127-149, kernel :52-59
 *   aux_samplers/_primitives/csmc/resamplings.py  multinomial :14-37   (-> jax.random.choice(p=w): cumsum, r = c[-1](1-u),
 *                                              searchsorted -- third-party JAX, restated from its published algorithm)
 *   aux_samplers/_primitives/math/utils.py     normalize :23-39 (exp(lw - logsumexp(lw)))
 *   aux_samplers/csmc/generic.py               kernel :56-72  (u = x + sqrt(delta/2) eps)
 *   aux_samplers/csmc/independent.py           AuxiliaryM0 :143-158, AuxiliaryG0 :163-169, AuxiliaryMtDynamics :192-198,
 *                                              AuxiliaryGt :238-248 (classical, non-gradient branch)
 * for the closed Feynman-Kac family of include/auxssm.h (linear-Gaussian transition; flat / Gaussian / SV potential).
 *
 * "Parity unpinned" against JAX bits: JAX is not installed, and XLA's summation order / exp / log are unknowable here,
 * so the float-level contract is fixed HERE and the HIP kernels must reproduce it bit for bit:
 *   - exp/log: the fdlibm-derived fixed operation sequences below (IEEE +,-,*,/,fma,rint only);
 *   - standalone primitives (normalize / multinomial / systematic) and the parallel-in-time sweep:
 *       cumsum : Kogge-Stone scan inside each group of 64 consecutive particles, group totals added left to right,
 *                c_i = (t_0 + ... + t_{g-1}) + local_i;
 *       sum    : balanced binary tree inside each group of 64, then left to right over groups;  max: exact;
 *   - the sequential sweep (csmc_ref_sweep), "sweep contract" of csrc/csmc_dev.h: the weights carried between steps are
 *     exp(lw - max lw), NOT divided by their sum -- resampling is searchsorted(cumsum(w), c[-1] (1 - u)) (resamplings.py:35-36 ->
 *     jax.random.choice), invariant to the scale of w, so normalize()'s logsumexp (math/utils.py:38-39) is never formed;
 *       cumsum : inside each group of 64 the scan order of the GPU's DPP network: Kogge-Stone with offsets 1, 2, 4, 8 inside each
 *                row of 16, then row 1 += last of row 0 and row 3 += last of row 2, then rows 2, 3 += last of row 1; the (up to 16) group
 *                totals, padded with +0, are prefix-summed by the same Kogge-Stone network on ONE row of 16 lanes (contract v3, round 3:
 *                every wave reads the totals into lanes 0..15 and scans them with four DPP adds; its own base is one readlane);
 *       search : branch-free lower bound by descent over the whole array (contract v3): pos = 0; for s = S0, S0/2, .., 1 (S0 the largest
 *                power of two below N): if (pos + s - 1 < N and c[pos + s - 1] < r) pos += s; clipped to N - 1 (== searchsorted on a
 *                non-decreasing c);   single draw of the backward pass: B = 64 g + #{l < 64 : c_{64 g + l} < r}, g = #{k < ng - 1 : P[k] < r},
 *                clipped to N - 1 (every wave finds the group and counts inside it: one barrier per backward step);
 *       densities : Gaussian log-densities multiply by the RECIPROCAL diagonal of the Cholesky factor, computed once per factor in the
 *                working precision (contract v3: no division per particle and step);
 *   - every multiply-add that is fused is written as fma(); compile with -ffp-contract=off.
 * What IS pinned: the statistical known answers of the reference's tests (test_csmc.py::test_flat_potential :18-69,
 * test_resamplings.py::test_multinomial_resampling :11-24) -- see tests/test_oracle_csmc.py -- and, since round 3, the LITERAL
 * restatement of the reference's own arithmetic order, oracle/csmc_np.py (normalised weights, plain cumsum, searchsorted, generic
 * Python model objects): tests/test_oracle_csmc_literal.py demands the same ancestors / backward indices / trajectories from both in
 * fp64 for every member of the closed family (and over the 50 000 sweeps of the reference's flat-potential protocol), and bounds the
 * fp32 tie rate of the two orders (1.4e-4 per draw at N = 1024, always a neighbouring particle).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC csmc_ref.c -o _build/libcsmc_ref.so -lm   (oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXD 32 /* the register kernels cover dx <= 4, csrc/csmc_wide.hip dx <= 32 */
#define PIT_SC 8 /* sub-chunks per chunk in the stitch of the parallel-in-time sweep (csrc/pit.hip) */

/* ------------------------------------------------------------------------------------------------ */
/* real-type generic code via the preprocessor: this file includes itself twice                          */
/* ------------------------------------------------------------------------------------------------ */
#ifndef CSMC_REF_BODY

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
static uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }

static float exp_f32(float x) {
    if (x != x) return x;
    if (x > 88.72f) return INFINITY;
    if (x < -87.3f) return 0.0f;
    const float kf = rintf(x * 1.44269504088896341f);
    float r = fmaf(-kf, 6.93145751953125e-1f, x);
    r = fmaf(-kf, 1.42860682030941723212e-6f, r);
    float p = 1.9841270114e-4f;
    p = fmaf(p, r, 1.3888889225e-3f);
    p = fmaf(p, r, 8.3333337670e-3f);
    p = fmaf(p, r, 4.1666667908e-2f);
    p = fmaf(p, r, 1.6666667163e-1f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    const int k = (int)kf;
    const int k1 = k / 2, k2 = k - k1;
    return p * u2f((uint32_t)(k1 + 127) << 23) * u2f((uint32_t)(k2 + 127) << 23);
}
static float log_f32(float x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    uint32_t ix = f2u(x);
    int e = 0;
    if (ix < 0x00800000u) { x = x * 33554432.0f; ix = f2u(x); e = -25; }
    e += (int)(ix >> 23) - 127;
    ix &= 0x007fffffu;
    const uint32_t i = (ix + (0x95f64u << 3)) & 0x800000u;
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
    return fmaf(dk, 6.9313812256e-01f, -((hfsq - fmaf(s, hfsq + R, dk * 9.0580006145e-06f)) - f));
}
static double exp_f64(double x) {
    if (x != x) return x;
    if (x > 709.78) return INFINITY;
    if (x < -708.0) return 0.0;
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
static double log_f64(double x) {
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    uint64_t ix = d2u(x);
    int e = 0;
    if (ix < 0x0010000000000000ull) { x = x * 18014398509481984.0; ix = d2u(x); e = -54; }
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

float csmc_ref_expf(float x) { return exp_f32(x); }
float csmc_ref_logf(float x) { return log_f32(x); }
double csmc_ref_exp(double x) { return exp_f64(x); }
double csmc_ref_log(double x) { return log_f64(x); }

typedef struct {
    int proposal, potential, D, backward;
    const double *m0, *LP0, *F, *b, *LQ; /* (D), (D,D) lower, (D,D), (D), (D,D) lower */
    double sig_y;
    int transition; /* 0 linear (F, b); 1 Lorenz-63 Euler-Maruyama: theta = F[0..2], dt = b[0] (examples/lorenz/model.py:10-25) */
    /* time-varying linear transitions (csmc.py:103 scans Mt.params): row t = transition t -> t+1; NULL = invariant F / b / LQ */
    const double *F_t, *b_t, *LQ_t; /* (T-1,D,D), (T-1,D), (T-1,D,D) lower */
    int gradient; /* 0 none; 1 reference (correction at t = 0 only: independent.py:265-266 sums it over all particles for t >= 1); 2 exact */
} fk_model;

#define CSMC_REF_BODY
#define REAL float
#define SUF(n) n##_f32
#define EXP exp_f32
#define LOG log_f32
#define FMA fmaf
#include "csmc_ref.c"
#undef REAL
#undef SUF
#undef EXP
#undef LOG
#undef FMA
#define REAL double
#define SUF(n) n##_f64
#define EXP exp_f64
#define LOG log_f64
#define FMA fma
#include "csmc_ref.c"

#else /* CSMC_REF_BODY: the generic part, compiled once per real type */

typedef struct {
    int proposal, potential, D, transition;
    REAL m0[MAXD], LP0[MAXD * MAXD], F[MAXD * MAXD], b[MAXD], LQ[MAXD * MAXD];
    REAL c_init, c_trans, c_obs, inv_sig_y;
    REAL iLP0[MAXD], iLQ[MAXD]; /* reciprocal diagonals of the two Cholesky factors: the log-densities multiply by them (contract v3: no division per particle) */
} SUF(fk);

static void SUF(fk_fill)(SUF(fk) * m, const fk_model* g) {
    const int D = g->D;
    memset(m, 0, sizeof(*m));
    m->proposal = g->proposal; m->potential = g->potential; m->D = D; m->transition = g->transition;
    for (int k = 0; k < D; ++k) { m->m0[k] = (REAL)g->m0[k]; m->b[k] = (REAL)g->b[k]; }
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            m->LP0[i * MAXD + j] = (REAL)g->LP0[i * D + j];
            m->F[i * MAXD + j] = (REAL)g->F[i * D + j];
            m->LQ[i * MAXD + j] = (REAL)g->LQ[i * D + j];
        }
    REAL ci = 0, ct = 0;
    for (int k = 0; k < D; ++k) { ci -= LOG(m->LP0[k * MAXD + k]); ct -= LOG(m->LQ[k * MAXD + k]); }
    for (int k = 0; k < D; ++k) { m->iLP0[k] = (REAL)1 / m->LP0[k * MAXD + k]; m->iLQ[k] = (REAL)1 / m->LQ[k * MAXD + k]; }
    const REAL hl2pi = (REAL)0.91893853320467274178;
    m->c_init = ci - (REAL)D * hl2pi;
    m->c_trans = ct - (REAL)D * hl2pi;
    if (g->potential == 1) {
        m->inv_sig_y = (REAL)1 / (REAL)g->sig_y;
        m->c_obs = -(REAL)D * LOG((REAL)g->sig_y) - (REAL)D * hl2pi;
    } else if (g->potential == 3) { /* masked Gaussian observations: constant per observed component */
        m->inv_sig_y = (REAL)1 / (REAL)g->sig_y;
        m->c_obs = -LOG((REAL)g->sig_y) - hl2pi;
    } else {
        m->inv_sig_y = 0;
        m->c_obs = -hl2pi;
    }
}

/* log N(x; mean, L L^T) with the additive constant precomputed; iL = the reciprocal diagonal of L (contract v3) */
static REAL SUF(gauss)(int D, const REAL* x, const REAL* mean, const REAL* L, const REAL* iL, REAL cst) {
    REAL z[MAXD], q = 0;
    for (int k = 0; k < D; ++k) {
        REAL acc = x[k] - mean[k];
        for (int j = 0; j < k; ++j) acc = FMA(-L[k * MAXD + j], z[j], acc);
        z[k] = acc * iL[k];
        q = FMA(z[k], z[k], q);
    }
    return FMA((REAL)-0.5, q, cst);
}
static void SUF(tmean)(const SUF(fk) * m, const REAL* xp, REAL* mu) {
    if (m->transition == 1) { /* x + dt (phi_0(x) + theta * phi(x)), examples/lorenz/model.py:10-25; fixed operation order */
        const REAL th1 = m->F[0], th2 = m->F[1], th3 = m->F[2], dt = m->b[0];
        const REAL f1 = th1 * (xp[1] - xp[0]);
        const REAL f2 = FMA(-xp[0], xp[2], FMA(th2, xp[0], -xp[1]));
        const REAL f3 = FMA(xp[0], xp[1], -(th3 * xp[2]));
        mu[0] = FMA(dt, f1, xp[0]);
        mu[1] = FMA(dt, f2, xp[1]);
        mu[2] = FMA(dt, f3, xp[2]);
        return;
    }
    for (int k = 0; k < m->D; ++k) {
        REAL acc = m->b[k];
        for (int j = 0; j < m->D; ++j) acc = FMA(m->F[k * MAXD + j], xp[j], acc);
        mu[k] = acc;
    }
}
static REAL SUF(pot)(const SUF(fk) * m, const REAL* x, const REAL* y) {
    const int D = m->D;
    if (m->potential == 0) return (REAL)0;
    if (m->potential == 1) {
        REAL q = 0;
        for (int k = 0; k < D; ++k) { const REAL z = (y[k] - x[k]) * m->inv_sig_y; q = FMA(z, z, q); }
        return FMA((REAL)-0.5, q, m->c_obs);
    }
    if (m->potential == 3) { /* y_k ~ N(x_k, sig_y^2) for the finite y_k only */
        REAL q = 0;
        int nobs = 0;
        for (int k = 0; k < D; ++k)
            if (y[k] - y[k] == 0) { const REAL z = (y[k] - x[k]) * m->inv_sig_y; q = FMA(z, z, q); ++nobs; }
        return FMA((REAL)-0.5, q, (REAL)nobs * m->c_obs);
    }
    REAL acc = 0;
    for (int k = 0; k < D; ++k) {
        const REAL e = EXP(-x[k]);
        const REAL s = FMA(y[k] * y[k], e, x[k]);
        const REAL v = FMA((REAL)-0.5, s, m->c_obs);
        acc += (v == v) ? v : (REAL)0;
    }
    return acc;
}

/* ---- the fixed reduction orders ---- */
static REAL SUF(tree64)(const REAL* v, int n) { /* balanced binary tree over 64 slots, missing slots are +0 */
    REAL t[64];
    for (int i = 0; i < 64; ++i) t[i] = i < n ? v[i] : (REAL)0;
    for (int off = 1; off < 64; off <<= 1)
        for (int i = 0; i < 64; i += 2 * off) t[i] = t[i] + t[i + off];
    return t[0];
}
static REAL SUF(sum)(const REAL* v, int N) {
    REAL s = 0;
    for (int g = 0; g * 64 < N; ++g) {
        const int n = N - g * 64 < 64 ? N - g * 64 : 64;
        const REAL t = SUF(tree64)(v + g * 64, n);
        s = g == 0 ? t : s + t;
    }
    return s;
}
static void SUF(cumsum)(const REAL* w, int N, REAL* c) {
    REAL pre = 0;
    for (int g = 0; g * 64 < N; ++g) {
        REAL t[64], o[64];
        const int n = N - g * 64 < 64 ? N - g * 64 : 64;
        for (int i = 0; i < 64; ++i) t[i] = i < n ? w[g * 64 + i] : (REAL)0;
        for (int off = 1; off < 64; off <<= 1) { /* Kogge-Stone */
            for (int i = 0; i < 64; ++i) o[i] = i >= off ? t[i] + t[i - off] : t[i];
            memcpy(t, o, sizeof t);
        }
        for (int i = 0; i < n; ++i) c[g * 64 + i] = g == 0 ? t[i] : pre + t[i];
        pre = g == 0 ? t[63] : pre + t[63];
    }
}
/* normalize (math/utils.py:23-39) */
static void SUF(normalize)(const REAL* lw, int N, REAL* w, REAL* tmp) {
    REAL m = lw[0];
    for (int i = 1; i < N; ++i) m = m > lw[i] ? m : lw[i];
    if (!(m - m == 0)) m = 0;
    for (int i = 0; i < N; ++i) tmp[i] = EXP(lw[i] - m);
    const REAL lse = LOG(SUF(sum)(tmp, N)) + m;
    for (int i = 0; i < N; ++i) w[i] = EXP(lw[i] - lse);
}
static int SUF(lower_bound)(const REAL* c, int n, REAL r) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (c[mid] < r) lo = mid + 1; else hi = mid; }
    return lo;
}
/* jax.random.choice(key, N, p=w, shape=()) given the uniform draw */
static int SUF(choice)(const REAL* c, int N, REAL un) {
    const REAL r = c[N - 1] * ((REAL)1 - un);
    const int i = SUF(lower_bound)(c, N, r);
    return i < N - 1 ? i : N - 1;
}

/* sup_x G(x) of the potential given y (csrc/csmc.hip::k_csmc_potbound); +inf where the potential is unbounded */
static REAL SUF(pot_bound)(const SUF(fk) * m, const REAL* y) {
    const int D = m->D;
    if (m->potential == 0) return (REAL)0;
    if (m->potential == 1) return m->c_obs;
    if (m->potential == 3) {
        int nobs = 0;
        for (int k = 0; k < D; ++k) nobs += (y[k] - y[k] == 0) ? 1 : 0;
        return (REAL)nobs * m->c_obs;
    }
    REAL b = 0;
    for (int k = 0; k < D; ++k) {
        const REAL y2 = y[k] * y[k];
        REAL v = 0;
        if (y2 - y2 == 0) v = y2 > 0 ? FMA((REAL)-0.5, (REAL)1 + LOG(y2), m->c_obs) : (REAL)INFINITY;
        b += v > 0 ? v : (REAL)0;
    }
    return b;
}

/* ---- sweep contract (csrc/csmc_dev.h) ---- */
static REAL SUF(expmax)(const REAL* lw, int N, REAL* w) {
    REAL m = lw[0];
    for (int i = 1; i < N; ++i) m = m > lw[i] ? m : lw[i];
    if (!(m - m == 0)) m = 0;
    for (int i = 0; i < N; ++i) w[i] = EXP(lw[i] - m);
    return m;
}
static void SUF(cumsum_dpp_p)(const REAL* w, int N, REAL* c, REAL* Pout) {
    /* sweep contract v3: inside each group of 64 the DPP scan order of the wave; the (up to 16) group totals, padded with +0, are prefix-summed by
     * the same Kogge-Stone network on one row of 16 lanes (offsets 1, 2, 4, 8; lanes without a source add +0); c_i = P[g - 1] + local_i (g > 0) */
    REAL loc[1024], tot[16], o16[16];
    const int ng = (N + 63) / 64;
    for (int k = 0; k < 16; ++k) tot[k] = 0;
    for (int g = 0; g < ng; ++g) {
        REAL t[64], o[64];
        const int n = N - g * 64 < 64 ? N - g * 64 : 64;
        for (int i = 0; i < 64; ++i) t[i] = i < n ? w[g * 64 + i] : (REAL)0;
        for (int off = 1; off < 16; off <<= 1) { /* Kogge-Stone inside every row of 16 (row_shr:off; lanes without a source add +0) */
            for (int i = 0; i < 64; ++i) o[i] = t[i] + ((i & 15) >= off ? t[i - off] : (REAL)0);
            memcpy(t, o, sizeof t);
        }
        for (int i = 0; i < 64; ++i) o[i] = t[i] + (((i >> 4) & 1) ? t[(i & ~15) - 1] : (REAL)0); /* row_bcast:15, rows 1 and 3 */
        memcpy(t, o, sizeof t);
        for (int i = 0; i < 64; ++i) o[i] = t[i] + (i >= 32 ? t[31] : (REAL)0);                   /* row_bcast:31, rows 2 and 3 */
        memcpy(t, o, sizeof t);
        for (int i = 0; i < n; ++i) loc[g * 64 + i] = t[i];
        tot[g] = t[63];
    }
    for (int off = 1; off < 16; off <<= 1) {
        for (int k = 0; k < 16; ++k) o16[k] = tot[k] + (k >= off ? tot[k - off] : (REAL)0);
        memcpy(tot, o16, sizeof tot);
    }
    for (int i = 0; i < N; ++i) c[i] = i < 64 ? loc[i] : tot[i / 64 - 1] + loc[i];
    if (Pout) memcpy(Pout, tot, sizeof tot);
}
static void SUF(cumsum_dpp)(const REAL* w, int N, REAL* c) { SUF(cumsum_dpp_p)(w, N, c, NULL); }
/* conditional-multinomial ancestor of one particle: branch-free lower bound by descent over the whole cumulative-weight array (contract v3) --
 * pos = 0; for s = S0, S0/2, ..., 1 (S0 = the largest power of two below N): if (pos + s - 1 < N and c[pos + s - 1] < r) pos += s; clipped to N - 1.
 * On a non-decreasing c this IS searchsorted(c, r, side='left') of resamplings.py:35-36 -> jax.random.choice. */
static int SUF(choice2)(const REAL* c, int N, REAL un) {
    const REAL r = c[N - 1] * ((REAL)1 - un);
    int s0 = 1;
    while (s0 * 2 < N) s0 *= 2;
    int pos = 0;
    for (int s = s0; s > 0; s >>= 1)
        if (pos + s - 1 < N && c[pos + s - 1] < r) pos += s;
    return pos < N - 1 ? pos : N - 1;
}
/* the single draw of the backward pass (contract v3): B = 64 g + #{l < 64 : c_{64 g + l} < r} with g = #{k < ng - 1 : P[k] < r} (P the prefix of the
 * group totals); on a non-decreasing c this is #{j : c_j < r} = searchsorted(c, r) */
static int SUF(choice_count)(const REAL* c, const REAL* P, int N, REAL un) {
    const REAL r = c[N - 1] * ((REAL)1 - un);
    const int ng = (N + 63) / 64;
    int g = 0;
    for (int k = 0; k < ng - 1; ++k) g += P[k] < r;
    int B = 64 * g;
    for (int j = 64 * g; j < 64 * g + 64 && j < N; ++j) B += c[j] < r;
    return B < N - 1 ? B : N - 1;
}

/* ---- time-varying transitions and gradient-informed proposals (csmc/independent.py:57-75 gradient=True, :121-134, :173-190, :252-268) ---- */
typedef struct { REAL F[MAXD * MAXD], b[MAXD], LQ[MAXD * MAXD], iLQ[MAXD], c_trans; } SUF(trans);
static void SUF(trans_at)(const SUF(fk) * m, const fk_model* g, long t, SUF(trans) * tr) {
    const int D = m->D;
    if (!g->F_t) {
        memcpy(tr->F, m->F, sizeof tr->F); memcpy(tr->b, m->b, sizeof tr->b); memcpy(tr->LQ, m->LQ, sizeof tr->LQ);
        memcpy(tr->iLQ, m->iLQ, sizeof tr->iLQ);
        tr->c_trans = m->c_trans;
        return;
    }
    memset(tr, 0, sizeof *tr);
    REAL c = 0;
    for (int i = 0; i < D; ++i) {
        tr->b[i] = (REAL)g->b_t[t * D + i];
        for (int j = 0; j < D; ++j) {
            tr->F[i * MAXD + j] = (REAL)g->F_t[(t * D + i) * D + j];
            tr->LQ[i * MAXD + j] = (REAL)g->LQ_t[(t * D + i) * D + j];
        }
    }
    for (int k = 0; k < D; ++k) { c -= LOG(tr->LQ[k * MAXD + k]); tr->iLQ[k] = (REAL)1 / tr->LQ[k * MAXD + k]; }
    tr->c_trans = c - (REAL)D * (REAL)0.91893853320467274178;
}
static void SUF(tmean_t)(const SUF(fk) * m, const SUF(trans) * tr, const REAL* xp, REAL* mu) {
    if (m->transition == 1) { SUF(tmean)(m, xp, mu); return; }
    for (int k = 0; k < m->D; ++k) {
        REAL acc = tr->b[k];
        for (int j = 0; j < m->D; ++j) acc = FMA(tr->F[k * MAXD + j], xp[j], acc);
        mu[k] = acc;
    }
}
static void SUF(cho_solve_)(int D, const REAL* L, const REAL* r, REAL* w) {
    REAL z[MAXD];
    for (int k = 0; k < D; ++k) {
        REAL acc = r[k];
        for (int j = 0; j < k; ++j) acc = FMA(-L[k * MAXD + j], z[j], acc);
        z[k] = acc / L[k * MAXD + k];
    }
    for (int k = D - 1; k >= 0; --k) {
        REAL acc = z[k];
        for (int j = k + 1; j < D; ++j) acc = FMA(-L[j * MAXD + k], w[j], acc);
        w[k] = acc / L[k * MAXD + k];
    }
}
/* gradient at u (T, D) of log M0(u_0) + G0(u_0) + sum_t [log Mt(u_{t+1} | u_t) + Gt(u_{t+1})] (independent.py:121-134) */
static void SUF(grad_logpi)(const SUF(fk) * m, const fk_model* g, int T, const REAL* u, const REAL* y, REAL* grad) {
    const int D = m->D;
    for (long t = 0; t < T; ++t) {
        const REAL* ut = u + t * D;
        REAL gr[MAXD], r[MAXD], w[MAXD], mu[MAXD];
        for (int k = 0; k < D; ++k) {
            const REAL yy = y ? y[t * D + k] : (REAL)0;
            REAL v = 0;
            if (m->potential == 1 || (m->potential == 3 && yy - yy == 0)) v = ((yy - ut[k]) * m->inv_sig_y) * m->inv_sig_y;
            else if (m->potential == 2) {
                const REAL e = EXP(-ut[k]);
                v = (REAL)0.5 * FMA(yy * yy, e, (REAL)-1);
                v = (v == v) ? v : (REAL)0;
            }
            gr[k] = v;
        }
        SUF(trans) tr;
        if (t == 0) {
            for (int k = 0; k < D; ++k) r[k] = ut[k] - m->m0[k];
            SUF(cho_solve_)(D, m->LP0, r, w);
        } else {
            SUF(trans_at)(m, g, t - 1, &tr);
            SUF(tmean_t)(m, &tr, ut - D, mu);
            for (int k = 0; k < D; ++k) r[k] = ut[k] - mu[k];
            SUF(cho_solve_)(D, tr.LQ, r, w);
        }
        for (int k = 0; k < D; ++k) gr[k] = gr[k] - w[k];
        if (t + 1 < T) {
            SUF(trans_at)(m, g, t, &tr);
            SUF(tmean_t)(m, &tr, ut, mu);
            for (int k = 0; k < D; ++k) r[k] = ut[D + k] - mu[k];
            SUF(cho_solve_)(D, tr.LQ, r, w);
            if (m->transition == 1) {
                const REAL th1 = m->F[0], th2 = m->F[1], th3 = m->F[2], dt = m->b[0];
                const REAL J[9] = {-th1, th1, (REAL)0, th2 - ut[2], (REAL)-1, -ut[0], ut[1], ut[0], -th3};
                for (int k = 0; k < 3; ++k) {
                    REAL acc = 0;
                    for (int j = 0; j < 3; ++j) acc = FMA(J[j * 3 + k], w[j], acc);
                    gr[k] = gr[k] + FMA(dt, acc, w[k]);
                }
            } else {
                for (int k = 0; k < D; ++k) {
                    REAL acc = 0;
                    for (int j = 0; j < D; ++j) acc = FMA(tr.F[j * MAXD + k], w[j], acc);
                    gr[k] = gr[k] + acc;
                }
            }
        }
        for (int k = 0; k < D; ++k) grad[t * D + k] = gr[k];
    }
}
static REAL SUF(grad_corr)(int D, const REAL* x, const REAL* u, const REAL* pm, REAL s) {
    REAL acc = 0;
    for (int k = 0; k < D; ++k) {
        const REAL d1 = x[k] - u[k], d2 = x[k] - pm[k];
        acc = FMA(d2, d2, acc);
        acc = FMA(-d1, d1, acc);
    }
    return acc * ((REAL)0.5 / (s * s));
}

/* One sweep of one chain.  x (T,D) in/out; y (T,D) or NULL; shd (T) or NULL; eps_aux (T,D) or NULL;
 * eps_prop (T,N,D); u_res (T-1,N); u_bwd (T); outputs anc (T), xs (T,N,D), lws (T,N), As (T-1,N) [all required]. */
int SUF(csmc_ref_sweep)(const fk_model* g, int T, int N, REAL* x, const REAL* y, const REAL* shd, const REAL* eps_aux,
                        const REAL* eps_prop, const REAL* u_res, const REAL* u_bwd, int32_t* anc, REAL* xs, REAL* lws,
                        int32_t* As) {
    SUF(fk) m;
    SUF(fk_fill)(&m, g);
    const int D = m.D;
    REAL* u = (REAL*)malloc(sizeof(REAL) * (size_t)T * D);
    REAL* w = (REAL*)malloc(sizeof(REAL) * N);
    REAL* c = (REAL*)malloc(sizeof(REAL) * N);
    REAL* tmp = (REAL*)malloc(sizeof(REAL) * N);
    REAL* lw = (REAL*)malloc(sizeof(REAL) * N);
    REAL zero[MAXD] = {0};
    REAL* grad = (REAL*)malloc(sizeof(REAL) * (size_t)T * D);
    if (m.proposal == 1) /* csmc/generic.py:67 */
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < D; ++k) u[t * D + k] = FMA(shd[t], eps_aux[t * D + k], x[t * D + k]);
    if (m.proposal == 1 && g->gradient) SUF(grad_logpi)(&m, g, T, u, y, grad);
    SUF(trans) tr;
    REAL pm[MAXD];
    /* t = 0 (csmc.py:74-80) */
    for (int i = 0; i < N; ++i) {
        REAL* xi = xs + (size_t)i * D;
        const REAL* e = eps_prop + (size_t)i * D;
        if (m.proposal == 0) {
            for (int k = 0; k < D; ++k) {
                REAL acc = m.m0[k];
                for (int j = 0; j <= k; ++j) acc = FMA(m.LP0[k * MAXD + j], e[j], acc);
                xi[k] = acc;
            }
        } else {
            for (int k = 0; k < D; ++k) {
                pm[k] = g->gradient ? FMA(shd[0] * shd[0], grad[k], u[k]) : u[k];
                xi[k] = FMA(shd[0], e[k], pm[k]);
            }
        }
        if (i == 0) for (int k = 0; k < D; ++k) xi[k] = x[k];
        REAL gq = SUF(pot)(&m, xi, y ? y : zero);
        if (m.proposal == 1) {
            gq = gq + SUF(gauss)(D, xi, m.m0, m.LP0, m.iLP0, m.c_init);
            if (g->gradient) gq = gq + SUF(grad_corr)(D, xi, u, pm, shd[0]);
        }
        lws[i] = gq;
    }
    REAL* fmax = (REAL*)malloc(sizeof(REAL) * (size_t)T); /* block maximum of log_ws[t] (non-finite -> 0): the backward pass shifts by it */
    fmax[0] = SUF(expmax)(lws, N, w);
    /* sweep contract, shifts: gb[t] = sup_x G_t(x) (+inf: none), the reduction-free part of the forward shift (k_csmc_potbound) */
    const int bmode = (m.potential == 0 || y) && g->gradient != 2;
    REAL* gb = (REAL*)malloc(sizeof(REAL) * (size_t)T);
    for (int t = 0; t < T; ++t) gb[t] = bmode ? SUF(pot_bound)(&m, y ? y + (size_t)t * D : zero) : (REAL)0;
    int used_bound = 0;
    for (int t = 1; t < T; ++t) {
        const REAL* xprev = xs + (size_t)(t - 1) * N * D;
        REAL* xcur = xs + (size_t)t * N * D;
        const REAL* yt = y ? y + (size_t)t * D : zero;
        SUF(cumsum_dpp)(w, N, c);
        if (used_bound && !(c[N - 1] > 0)) { /* every weight of step t - 1 underflowed under its bound: the exact maximum after all */
            fmax[t - 1] = SUF(expmax)(lw, N, w);
            SUF(cumsum_dpp)(w, N, c);
        }
        SUF(trans_at)(&m, g, t - 1, &tr);
        for (int i = 0; i < N; ++i) {
            int idx = 0;
            if (i > 0) idx = SUF(choice2)(c, N, u_res[(size_t)(t - 1) * N + i]); /* resamplings.py:35-36 */
            As[(size_t)(t - 1) * N + i] = idx;
            const REAL* xp = xprev + (size_t)idx * D;
            const REAL* e = eps_prop + ((size_t)t * N + i) * D;
            REAL* xi = xcur + (size_t)i * D;
            if (m.proposal == 0) {
                REAL mu0[MAXD];
                SUF(tmean_t)(&m, &tr, xp, mu0);
                for (int k = 0; k < D; ++k) {
                    REAL acc = mu0[k];
                    for (int j = 0; j <= k; ++j) acc = FMA(tr.LQ[k * MAXD + j], e[j], acc);
                    xi[k] = acc;
                }
            } else {
                for (int k = 0; k < D; ++k) {
                    pm[k] = g->gradient ? FMA(shd[t] * shd[t], grad[t * D + k], u[t * D + k]) : u[t * D + k];
                    xi[k] = FMA(shd[t], e[k], pm[k]);
                }
            }
            if (i == 0) for (int k = 0; k < D; ++k) xi[k] = x[t * D + k];
            REAL gq = SUF(pot)(&m, xi, yt);
            if (m.proposal == 1) {
                REAL mu[MAXD];
                SUF(tmean_t)(&m, &tr, xp, mu);
                gq = SUF(gauss)(D, xi, mu, tr.LQ, tr.iLQ, tr.c_trans) + gq;
                if (g->gradient == 2) gq = gq + SUF(grad_corr)(D, xi, u + t * D, pm, shd[t]);
            }
            lw[i] = gq;
        }
        memcpy(lws + (size_t)t * N, lw, sizeof(REAL) * N);
        {
            const REAL Mb = gb[t] + (m.proposal == 1 ? tr.c_trans : (REAL)0);
            used_bound = bmode && t < T - 1 && (Mb - Mb == 0);
            if (used_bound) {
                for (int i = 0; i < N; ++i) w[i] = EXP(lw[i] - Mb);
                fmax[t] = Mb;
            } else {
                fmax[t] = SUF(expmax)(lw, N, w);
            }
        }
    }
    free(gb);
    /* backward (csmc.py:110-149) */
    REAL Pt[16];
    SUF(cumsum_dpp_p)(w, N, c, Pt);
    int B = SUF(choice_count)(c, Pt, N, u_bwd[T - 1]);
    anc[T - 1] = B;
    REAL xn[MAXD];
    for (int k = 0; k < D; ++k) xn[k] = x[(T - 1) * D + k] = xs[((size_t)(T - 1) * N + B) * D + k];
    for (int t = T - 2; t >= 0; --t) {
        if (!g->backward) {
            B = As[(size_t)t * N + B];
        } else {
            SUF(trans_at)(&m, g, t, &tr); /* Pt.logpdf(x_{t+1}, xs_t, params_t) (csmc.py:136) */
            for (int i = 0; i < N; ++i) {
                REAL mu[MAXD];
                SUF(tmean_t)(&m, &tr, xs + ((size_t)t * N + i) * D, mu);
                lw[i] = SUF(gauss)(D, xn, mu, tr.LQ, tr.iLQ, tr.c_trans) + lws[(size_t)t * N + i];
            }
            /* sweep contract: weights shifted by a bound of their maximum (forward block maximum + log-normaliser of the transition density);
               the exact maximum only when every weight underflowed */
            REAL Mb = fmax[t] + tr.c_trans;
            if (!(Mb - Mb == 0)) Mb = 0;
            for (int i = 0; i < N; ++i) w[i] = EXP(lw[i] - Mb);
            SUF(cumsum_dpp_p)(w, N, c, Pt);
            if (!(c[N - 1] > 0)) {
                SUF(expmax)(lw, N, w);
                SUF(cumsum_dpp_p)(w, N, c, Pt);
            }
            B = SUF(choice_count)(c, Pt, N, u_bwd[t]);
        }
        anc[t] = B;
        for (int k = 0; k < D; ++k) xn[k] = x[t * D + k] = xs[((size_t)t * N + B) * D + k];
    }
    free(u); free(w); free(c); free(tmp); free(lw); free(grad); free(fmax);
    return 0;
}

/* ---- parallel-in-time conditional SMC (conditional dSMC): aux_samplers/_primitives/csmc/pit/csmc.py:69-114, operator.py:74-149,
 * dc_map.py:70-121, driven as csmc/independent.py:78-118 does (classical branch: proposals N(u_t, delta_t/2 I)).  The tree is
 * evaluated node by node with FULL block gathers, exactly as the reference's operator does (trajectories, origins and weights of a
 * whole block are re-indexed at every stitch); the HIP kernels (csrc/pit.hip) keep only boundary indices and must give the same
 * trajectory and origins.  Arithmetic contract: see the header of csrc/pit.hip. */
/* x (T,D) in/out; y (T,D) or NULL; shd (T); eps_aux (T,D); eps_prop (T,N,D); u_res (T,N); outputs anc (T) and, if non-NULL,
 * xs (T,N,D) the leaf particles.  T >= 2. */
int SUF(csmc_ref_pit_sweep)(const fk_model* g, int T, int N, REAL* x, const REAL* y, const REAL* shd, const REAL* eps_aux,
                            const REAL* eps_prop, const REAL* u_res, int32_t* anc, REAL* xs_out) {
    SUF(fk) m;
    SUF(fk_fill)(&m, g);
    const int D = m.D;
    const size_t TN = (size_t)T * N;
    REAL* xs = (REAL*)malloc(sizeof(REAL) * TN * D);   /* current block trajectories, (T, N, D) */
    REAL* xt = (REAL*)malloc(sizeof(REAL) * TN * D);
    REAL* lw = (REAL*)malloc(sizeof(REAL) * TN);       /* per (t, slot) log-weights */
    int32_t* org = (int32_t*)malloc(sizeof(int32_t) * TN);
    int32_t* ot = (int32_t*)malloc(sizeof(int32_t) * TN);
    REAL zero[MAXD] = {0};
    const REAL nln = -LOG((REAL)N);
    /* leaves (pit/csmc.py:77-96).  Gradient-informed proposals (csmc/independent.py:81-84): mt = N(u + delta/2 grad, delta/2 I) proposes, qt = N(u, delta/2 I)
     * is the target's factor, so every leaf carries log_wts = qt.logpdf - mt.logpdf (pit/csmc.py:83-88) -- per particle, slot 0 included. */
    REAL* uall = (REAL*)malloc(sizeof(REAL) * (size_t)T * D);
    REAL* grad = (REAL*)malloc(sizeof(REAL) * (size_t)T * D);
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < D; ++k) uall[t * D + k] = FMA(shd[t], eps_aux[t * D + k], x[t * D + k]);
    if (g->gradient) SUF(grad_logpi)(&m, g, T, uall, y, grad);
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < N; ++n) {
            REAL pm[MAXD];
            for (int k = 0; k < D; ++k) {
                pm[k] = g->gradient ? FMA(shd[t] * shd[t], grad[t * D + k], uall[t * D + k]) : uall[t * D + k];
                xs[((size_t)t * N + n) * D + k] = n == 0 ? x[t * D + k] : FMA(shd[t], eps_prop[((size_t)t * N + n) * D + k], pm[k]);
            }
            org[(size_t)t * N + n] = n;
            lw[(size_t)t * N + n] = g->gradient ? SUF(grad_corr)(D, xs + ((size_t)t * N + n) * D, uall + t * D, pm, shd[t]) : nln;
        }
    if (xs_out) memcpy(xs_out, xs, sizeof(REAL) * TN * D);
    {
        REAL* g0 = (REAL*)malloc(sizeof(REAL) * N);
        REAL* tmp = (REAL*)malloc(sizeof(REAL) * N);
        for (int t = 0; t < (g->gradient ? T : 1); ++t) { /* normalise per time step (:91); without gradients the rows t >= 1 are -log N already */
            REAL mx;
            for (int n = 0; n < N; ++n) {
                g0[n] = g->gradient ? lw[(size_t)t * N + n] : (REAL)0;
                if (t == 0) {
                    REAL gq = SUF(pot)(&m, xs + (size_t)n * D, y ? y : zero);
                    gq = gq + SUF(gauss)(D, xs + (size_t)n * D, m.m0, m.LP0, m.iLP0, m.c_init);
                    g0[n] = g->gradient ? g0[n] + gq : gq;
                }
            }
            mx = g0[0];
            for (int n = 1; n < N; ++n) mx = mx > g0[n] ? mx : g0[n];
            if (!(mx - mx == 0)) mx = 0;
            for (int n = 0; n < N; ++n) tmp[n] = EXP(g0[n] - mx);
            const REAL lse = LOG(SUF(sum)(tmp, N)) + mx;
            for (int n = 0; n < N; ++n) lw[(size_t)t * N + n] = g0[n] - lse;
        }
        free(g0); free(tmp);
    }
    free(uall); free(grad);
    int K = 0;
    while ((1 << K) < T) ++K;
    const long long NN = (long long)N * N;
    const int NCH = N <= 32 ? 64 : (N <= 128 ? 256 : 1024);
    const int Lc = (int)((NN + NCH - 1) / NCH);
    const int Ls = (Lc + PIT_SC - 1) / PIT_SC;
    REAL* sub = (REAL*)malloc(sizeof(REAL) * PIT_SC * 1024);
    REAL* mu = (REAL*)malloc(sizeof(REAL) * N * D);
    REAL* pg = (REAL*)malloc(sizeof(REAL) * N);
    REAL* cs = (REAL*)malloc(sizeof(REAL) * 1024);
    REAL* ss = (REAL*)malloc(sizeof(REAL) * 1024);
    int* li = (int*)malloc(sizeof(int) * N);
    int* ri = (int*)malloc(sizeof(int) * N);
    for (int k = 0; k < K; ++k) {
        const long long bs = 1ll << k;
        for (long long s0 = 0; s0 < T; s0 += 2 * bs) {
            const long long mid = s0 + bs;
            if (mid >= T) continue; /* passthrough (dc_map.py:93-105) */
            const long long e1 = s0 + 2 * bs < T ? s0 + 2 * bs : T;
            const int root = k == K - 1;
            const REAL* xa = xs + (size_t)(mid - 1) * N * D;
            const REAL* xb = xs + (size_t)mid * N * D;
            const REAL* yv = y ? y + (size_t)mid * D : zero;
            SUF(trans) tr;
            SUF(trans_at)(&m, g, mid - 1, &tr); /* the transition across the boundary (time-varying: row mid - 1) */
            for (int n = 0; n < N; ++n) {
                SUF(tmean_t)(&m, &tr, xa + (size_t)n * D, mu + (size_t)n * D);
                pg[n] = SUF(pot)(&m, xb + (size_t)n * D, yv) + lw[(size_t)mid * N + n];
            }
            const REAL* ha = lw + (size_t)(mid - 1) * N;
#define PIT_V(i, j) ((SUF(gauss)(D, xb + (size_t)(j) * D, mu + (size_t)(i) * D, tr.LQ, tr.iLQ, tr.c_trans) + pg[j]) + ha[i])
            REAL vmax = -INFINITY;
            for (long long p = 0; p < NN; ++p) { const REAL v = PIT_V(p / N, p % N); vmax = v > vmax ? v : vmax; }
            if (!(vmax - vmax == 0)) vmax = 0;
            for (int c = 0; c < NCH; ++c) { /* chunk = SC sub-chunks; the chunk sum is the left-to-right sum of the sub-chunk sums */
                const long long p0 = (long long)c * Lc, p1 = p0 + Lc < NN ? p0 + Lc : NN;
                REAL s = 0;
                long long p = p0;
                for (int b = 0; b < PIT_SC; ++b) {
                    const long long pe = p0 + (long long)(b + 1) * Ls < p1 ? p0 + (long long)(b + 1) * Ls : p1;
                    REAL sb = 0;
                    for (; p < pe; ++p) sb = sb + EXP(PIT_V(p / N, p % N) - vmax);
                    sub[(size_t)b * NCH + c] = sb;
                    s = s + sb;
                }
                ss[c] = s;
            }
            SUF(cumsum)(ss, NCH, cs);
            const int last_chunk = (int)((NN - 1) / Lc);
            const int ndraw = root ? 1 : N;
            for (int n = 0; n < ndraw; ++n) {
                int il = 0, jr = 0;
                if (root || n > 0) {
                    const REAL r = cs[NCH - 1] * ((REAL)1 - u_res[(size_t)mid * N + n]);
                    int ts = SUF(lower_bound)(cs, NCH, r);
                    ts = ts < last_chunk ? ts : last_chunk;
                    const REAL pre = ts > 0 ? cs[ts - 1] : (REAL)0;
                    const long long c0 = (long long)ts * Lc, c1 = c0 + Lc < NN ? c0 + Lc : NN;
                    const int nsub = (int)((c1 - c0 + Ls - 1) / Ls);
                    int bsel = nsub - 1;
                    REAL acc = 0;
                    for (int b = 0; b < nsub; ++b) {
                        const REAL nacc = acc + sub[(size_t)b * NCH + ts];
                        const REAL cvb = ts > 0 ? pre + nacc : nacc;
                        if (cvb >= r || b == nsub - 1) { bsel = b; break; }
                        acc = nacc;
                    }
                    const long long q0 = c0 + (long long)bsel * Ls, q1 = q0 + Ls < c1 ? q0 + Ls : c1;
                    long long psel = q1 - 1;
                    for (long long p = q0; p < q1; ++p) {
                        acc = acc + EXP(PIT_V(p / N, p % N) - vmax);
                        const REAL cv = ts > 0 ? pre + acc : acc;
                        if (cv >= r) { psel = p; break; }
                    }
                    il = (int)(psel / N);
                    jr = (int)(psel % N);
                }
                li[n] = il; ri[n] = jr;
            }
#undef PIT_V
            /* _gather_results (operator.py:87-110): re-index both blocks, reset their weights to -log N; the root keeps ONE trajectory,
             * stored in slot 0 */
            for (long long t = s0; t < e1; ++t)
                for (int n = 0; n < ndraw; ++n) {
                    const int src = t < mid ? li[n] : ri[n];
                    for (int q = 0; q < D; ++q) xt[((size_t)t * N + n) * D + q] = xs[((size_t)t * N + src) * D + q];
                    ot[(size_t)t * N + n] = org[(size_t)t * N + src];
                }
            for (long long t = s0; t < e1; ++t)
                for (int n = 0; n < ndraw; ++n) {
                    for (int q = 0; q < D; ++q) xs[((size_t)t * N + n) * D + q] = xt[((size_t)t * N + n) * D + q];
                    org[(size_t)t * N + n] = ot[(size_t)t * N + n];
                    lw[(size_t)t * N + n] = nln;
                }
        }
    }
    for (int t = 0; t < T; ++t) {
        anc[t] = org[(size_t)t * N];
        for (int q = 0; q < D; ++q) x[t * D + q] = xs[((size_t)t * N) * D + q];
    }
    free(xs); free(xt); free(lw); free(org); free(ot); free(mu); free(pg); free(cs); free(ss); free(li); free(ri); free(sub);
    return 0;
}

/* the closed-form gradient alone (tests pin it against finite differences of the joint log-density) */
void SUF(csmc_ref_grad)(const fk_model* g, int T, const REAL* u, const REAL* y, REAL* grad) {
    SUF(fk) m;
    SUF(fk_fill)(&m, g);
    SUF(grad_logpi)(&m, g, T, u, y, grad);
}

/* conditional multinomial resampling alone (resamplings.py:14-37), for the reference's statistical test */
void SUF(csmc_ref_multinomial)(const REAL* w, int N, const REAL* un, int32_t* idx) {
    REAL* c = (REAL*)malloc(sizeof(REAL) * N);
    SUF(cumsum)(w, N, c);
    for (int i = 0; i < N; ++i) idx[i] = i == 0 ? 0 : SUF(choice)(c, N, un[i]);
    free(c);
}
/* conditional systematic resampling (resamplings.py:40-86), given the three uniforms (U, V, W) */
void SUF(csmc_ref_systematic)(const REAL* w, int M, int N, const REAL* uvw, int32_t* out) {
    REAL* c = (REAL*)malloc(sizeof(REAL) * M);
    int* idx = (int*)malloc(sizeof(int) * N);
    SUF(cumsum)(w, M, c);
    const REAL U = uvw[0], V = uvw[1], W = uvw[2];
    const REAL tmp = (REAL)N * w[0];
    const REAL fl = (REAL)floor((double)tmp);
    REAL uni;
    if (tmp <= (REAL)1) uni = tmp * U;
    else {
        const REAL rem = tmp - fl;
        const REAL p_cond = rem * (fl + (REAL)1) / tmp;
        uni = V < p_cond ? rem * U : rem + ((REAL)1 - rem) * U;
    }
    int nz = 0;
    for (int n = 0; n < N; ++n) {
        const REAL pos = ((REAL)n + uni) / (REAL)N;
        idx[n] = SUF(lower_bound)(c, M, pos);
        nz += idx[n] == 0;
    }
    for (int n = 0; n < N; ++n) {
        int o = idx[n];
        if (nz != 1) {
            const int roll_idx = (int)floor((double)((REAL)nz * W));
            const int shift = roll_idx < nz ? roll_idx : -1; /* zero_loc[k] = k for k < nz (idx is non-decreasing), fill value -1 */
            int src = (n + shift) % N;
            if (src < 0) src += N;
            o = idx[src];
        }
        out[n] = o < 0 ? 0 : (o > M - 1 ? M - 1 : o);
    }
    free(c); free(idx);
}
void SUF(csmc_ref_normalize)(const REAL* lw, int N, REAL* w) {
    REAL* tmp = (REAL*)malloc(sizeof(REAL) * N);
    SUF(normalize)(lw, N, w, tmp);
    free(tmp);
}

#endif /* CSMC_REF_BODY */

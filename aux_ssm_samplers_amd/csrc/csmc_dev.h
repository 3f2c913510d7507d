// csmc_dev.h -- what the conditional-SMC units share (csmc.hip: the sequential sweep; pit.hip: the parallel-in-time sweep): the
// device-side Feynman-Kac model, its densities in fixed operation order, and the workgroup reductions whose orders the C oracle
// (oracle/csmc_ref.c) restates.  Units including this are compiled with -ffp-contract=off.
#pragma once
#include <cstring>
#include "ctx.h"
#include "det_math.h"
#include "rng.h"

namespace ax {

constexpr int CS_MAXD = 4;

template <typename R> struct FkDev {
    int proposal, potential, D, transition;  // transition: 0 = linear-Gaussian (F, b); 1 = Lorenz-63 Euler-Maruyama (theta = F[0][0..2], dt = b[0])
    R m0[CS_MAXD], LP0[CS_MAXD * CS_MAXD], F[CS_MAXD * CS_MAXD], b[CS_MAXD], LQ[CS_MAXD * CS_MAXD];
    R c_init, c_trans, c_obs, inv_sig_y;  // additive constants: -sum log L_kk - D/2 log 2pi, etc.
    R iLP0[CS_MAXD], iLQ[CS_MAXD];        // reciprocal diagonals of LP0 / LQ: the log-densities multiply by them (sweep contract v3)
    // time-varying linear transitions (device arrays, row t = transition t -> t+1; null: the invariant F / b / LQ above)
    const R* Ft;   // (T-1, D, D)
    const R* bt;   // (T-1, D)
    const R* LQt;  // (T-1, D, D) lower
    const R* ctt;  // (T-1) additive constants of the transition densities (k_csmc_ctrans)
    const R* idt;  // (T-1, D) reciprocal diagonals of LQt (k_csmc_ctrans)
    int gradient;  // AUXSSM_GRAD_*
};
// the transition t -> t+1 of the model: matrices through pointers (wave-uniform loads when time-varying)
template <typename R> struct TransT {
    const R* F;
    const R* b;
    const R* LQ;
    int ld;  // leading dimension of F / LQ: CS_MAXD for the struct arrays, D for the device rows
    R c_trans;
    const R* iL;  // reciprocal diagonal of LQ
};
template <typename R, int D> __device__ __forceinline__ TransT<R> trans_at(const FkDev<R>& m, long long t) {
    if (m.Ft) return TransT<R>{m.Ft + t * D * D, m.bt + t * D, m.LQt + t * D * D, D, m.ctt[t], m.idt + t * D};
    return TransT<R>{m.F, m.b, m.LQ, CS_MAXD, m.c_trans, m.iLQ};
}
// the same with the choice made at compile time (the persistent sweep kernels: no branch on the model kind inside the time loop)
template <typename R, int D, bool TV> __device__ __forceinline__ TransT<R> trans_at_c(const FkDev<R>& m, long long t) {
    if constexpr (TV) return TransT<R>{m.Ft + t * D * D, m.bt + t * D, m.LQt + t * D * D, D, m.ctt[t], m.idt + t * D};
    else return TransT<R>{m.F, m.b, m.LQ, CS_MAXD, m.c_trans, m.iLQ};
}

struct CsmcArgs {
    int C, T, N, backward;
    const void* y;       // (T, D) shared by chains (may be null for the flat potential)
    const void* shd;     // (T) sqrt(delta_t / 2), AUX proposal only
    void* x;             // (C, T, D) reference trajectory in, new trajectory out
    void* u;             // (C, T, D) auxiliary variables (workspace), AUX only
    void* grad;          // (C, T, D) gradient of the model's joint log-density at u (workspace), gradient proposals only
    void* xs;            // (C, T, N, D)
    void* lws;           // (C, T, N)
    int32_t* As;         // (C, T-1, N) or null
    void* wT;            // (C, N)
    void* fmax;          // (C, T) the shift the forward pass used for the weights of step t (an upper bound of max_i log_ws[t][i], or that maximum;
                         // non-finite -> 0): the backward pass builds its own bound from it (sweep contract)
    const void* gb;      // (T) upper bound of the potential G_t over x (a function of y_t only; +inf where there is none); null: exact maxima only
    int32_t* anc;        // (C, T)
    int noise_mode;      // 0 explicit arrays, 1 Threefry
    int pregen = 0;      // Threefry mode with the forward pass's draws generated into eps_prop / u_res BEFORE the pass (k_csmc_pregen: a sweep with fewer chains than
                         // CUs leaves most of the chip idle while every step of its few workgroups waits for a Threefry block and a Box-Muller pair)
    uint32_t key0, key1;
    const void* eps_aux;   // (C, T, D)
    const void* eps_prop;  // (C, T, N, D)
    const void* u_res;     // (C, T-1, N)
    const void* u_bwd;     // (C, T); with in-kernel draws the sweep fills its own array first (k_csmc_ubwd): the backward kernels always read it
    // chain batching (csmc.hip::auxssm_csmc_sweep): the particle system of one chain is T N (D + 1) reals -- 537 MB at C3 -- so a sweep over more chains
    // than the device holds runs the forward + backward pair batch by batch, [c0, c0 + C) per launch.  Every array above is indexed by the GLOBAL
    // chain c0 + blockIdx.x (so are the random streams: a batched sweep is bit for bit the unbatched one); the workspace-owned xs / lws / As of a
    // batch are passed with their base moved back by c0 records.
    int c0 = 0;
    int cb = 0;                               // host side: chains per batch
    size_t xs_rec = 0, lws_rec = 0, As_rec = 0;  // host side: bytes per chain of the workspace-owned arrays (0: caller-owned, indexed globally anyway)
};

enum { STREAM_EPS_AUX = 1, STREAM_EPS_PROP = 2, STREAM_U_RES = 3, STREAM_U_BWD = 4 };

template <typename R> __device__ __forceinline__ R noise_normal(const CsmcArgs& a, const void* arr, uint32_t stream, long long idx) {
    if (a.noise_mode == 0) return ((const R*)arr)[idx];
    return stream_normal<R>(a.key0, a.key1, stream, (unsigned long long)idx);
}
template <typename R> __device__ __forceinline__ R noise_uniform(const CsmcArgs& a, const void* arr, uint32_t stream, long long idx) {
    if (a.noise_mode == 0) return ((const R*)arr)[idx];
    return stream_uniform<R>(a.key0, a.key1, stream, (unsigned long long)idx);
}

AXD_HD float fma_(float a, float b, float c) { return fmaf(a, b, c); }
AXD_HD double fma_(double a, double b, double c) { return fma(a, b, c); }

// log N(x; mean, L L^T) = cst - 0.5 |L^-1 (x - mean)|^2, forward substitution in a fixed order; iL = the reciprocal diagonal of L, computed once
// per factor (sweep contract v3: a multiplication instead of an IEEE division -- ten instructions -- per particle, component and step)
template <typename R, int D> AXD_HD R gauss_chol_logpdf(const R* x, const R* mean, const R* L, const R* iL, R cst, int ld = CS_MAXD) {
    R z[D];
    R q = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        R acc = x[k] - mean[k];
#pragma unroll
        for (int j = 0; j < k; ++j) acc = fma_(-L[k * ld + j], z[j], acc);
        z[k] = acc * iL[k];
        q = fma_(z[k], z[k], q);
    }
    return fma_((R)-0.5, q, cst);
}
template <typename R, int D> AXD_HD void trans_mean(const FkDev<R>& m, const R* xp, R* mu);
// mean of the transition tr applied to xp (linear, or the Lorenz-63 Euler-Maruyama step of the invariant model)
template <typename R, int D> __device__ __forceinline__ void trans_mean_t(const FkDev<R>& m, const TransT<R>& tr, const R* xp, R* mu) {
    if (m.transition == 1) {
        trans_mean<R, D>(m, xp, mu);
        return;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
        R acc = tr.b[k];
#pragma unroll
        for (int j = 0; j < D; ++j) acc = fma_(tr.F[k * tr.ld + j], xp[j], acc);
        mu[k] = acc;
    }
}
template <typename R, int D> AXD_HD void trans_mean(const FkDev<R>& m, const R* xp, R* mu) {
    if constexpr (D == 3) {
        if (m.transition == 1) {  // x + dt (phi_0(x) + theta * phi(x)), examples/lorenz/model.py:10-25; fixed operation order
            const R th1 = m.F[0], th2 = m.F[1], th3 = m.F[2], dt = m.b[0];
            const R f1 = th1 * (xp[1] - xp[0]);
            const R f2 = fma_(-xp[0], xp[2], fma_(th2, xp[0], -xp[1]));
            const R f3 = fma_(xp[0], xp[1], -(th3 * xp[2]));
            mu[0] = fma_(dt, f1, xp[0]);
            mu[1] = fma_(dt, f2, xp[1]);
            mu[2] = fma_(dt, f3, xp[2]);
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
        R acc = m.b[k];
#pragma unroll
        for (int j = 0; j < D; ++j) acc = fma_(m.F[k * CS_MAXD + j], xp[j], acc);
        mu[k] = acc;
    }
}
// potential g_t(x_t) (csmc test fixtures test_csmc/common.py:52-75; SV examples/stochastic_volatility/auxiliary_csmc.py:40-46)
template <typename R, int D> AXD_HD R potential(const FkDev<R>& m, const R* x, const R* y) {
    if (m.potential == 0) return (R)0;
    if (m.potential == 1) {  // y ~ N(x, sig_y^2 I)
        R q = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const R z = (y[k] - x[k]) * m.inv_sig_y;
            q = fma_(z, z, q);
        }
        return fma_((R)-0.5, q, m.c_obs);
    }
    if (m.potential == 3) {  // y_k ~ N(x_k, sig_y^2) for the finite y_k only (missing components / whole missing steps are skipped)
        R q = 0;
        int nobs = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (y[k] - y[k] == 0) {
                const R z = (y[k] - x[k]) * m.inv_sig_y;
                q = fma_(z, z, q);
                ++nobs;
            }
        }
        return fma_((R)-0.5, q, (R)nobs * m.c_obs);
    }
    // stochastic volatility: y_k ~ N(0, exp(x_k)):  -0.5 (y^2 e^{-x} + x) - 0.5 log 2pi, NaN terms -> 0
    R acc = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R e = det_exp(-x[k]);
        const R s = fma_(y[k] * y[k], e, x[k]);
        const R v = fma_((R)-0.5, s, m.c_obs);
        acc += (v == v) ? v : (R)0;
    }
    return acc;
}

// ---- block primitives (TB threads = NW waves) ---------------------------------------------------------------------
template <typename R> __device__ __forceinline__ R wave_max(R v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const R o = __shfl_xor(v, off, 64);
        v = v > o ? v : o;  // NaN-agnostic: weights are never NaN for valid models
    }
    return v;
}
template <typename R> __device__ __forceinline__ R wave_sum_tree(R v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <typename R> __device__ __forceinline__ R wave_scan_ks(R v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const R o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// the up-to-16 per-wave partials of a block reduction, fetched with wide LDS reads into registers (same values, same
// left-to-right combination order as a scalar loop over red[]; only the dependent LDS round trips disappear)
template <typename R> __device__ __forceinline__ void load16(const R* red, R* t) {
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = red[k];
}

// normalize (math/utils.py:23-39): w = exp(lw - logsumexp(lw)); logsumexp = log(sum(exp(lw - max))) + max
// red: 48 slots (max in [0,16), sum in [16,32), scan totals in [32,48)); slots of unused waves are never read.
template <typename R> __device__ __forceinline__ R block_normalize(R lw, R* red, int tid, int nw) {
    const int lane = tid & 63, wv = tid >> 6;
    R m = wave_max(lw);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    R t[16];
    load16<R>(red, t);
    m = t[0];
    if (nw == 16) {  // full workgroup: no per-slot masks
#pragma unroll
        for (int k = 1; k < 16; ++k) m = t[k] > m ? t[k] : m;
    } else {
#pragma unroll
        for (int k = 1; k < 16; ++k) m = (k < nw && t[k] > m) ? t[k] : m;
    }
    if (!(m - m == 0)) m = 0;  // non-finite max -> 0 (jax logsumexp)
    const R e = det_exp(lw - m);
    R s = wave_sum_tree(e);
    if (lane == 0) red[16 + wv] = s;
    __syncthreads();
    load16<R>(red + 16, t);
    s = t[0];
    if (nw == 16) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s = s + t[k];
    } else {
#pragma unroll
        for (int k = 1; k < 16; ++k) s = k < nw ? s + t[k] : s;
    }
    const R lse = det_log(s) + m;
    return det_exp(lw - lse);
}

// inclusive cumsum of w into c[] (slots [32,48) of red hold the wave totals); c[] valid after the trailing barrier
template <typename R> __device__ __forceinline__ void block_cumsum(R w, R* c, R* red, int tid, int nw) {
    const int lane = tid & 63, wv = tid >> 6;
    const R v = wave_scan_ks(w, lane);
    if (lane == 63) red[32 + wv] = v;
    __syncthreads();
    R t[16];
    load16<R>(red + 32, t);
    R pre = t[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) pre = k < wv ? pre + t[k] : pre;
    c[tid] = wv > 0 ? pre + v : v;
    __syncthreads();
}

// first index j in [0, n) with c[j] >= r  (jnp.searchsorted side='left'); n if none
template <typename R> __device__ __forceinline__ int lower_bound(const R* c, int n, R r) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c[mid] < r) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// ---- sweep contract (k_csmc_fwd / k_csmc_bwd of csmc.hip; restated by oracle/csmc_ref.c::csmc_ref_sweep) ---------------------------
// The sequential sweep carries UNNORMALISED weights e_i = exp(lw_i - max lw): conditional multinomial resampling only ever uses
// searchsorted(cumsum(w), c[-1] (1 - u)) (resamplings.py:35-36 -> jax.random.choice), which is invariant to the scale of w, so the
// normaliser of normalize() (math/utils.py:38-39: one block sum, one log and one more exp per particle and step) is never formed.
//   cumsum : inside each group of 64 consecutive particles the DPP scan of the hardware -- Kogge-Stone with offsets 1, 2, 4, 8 inside
//            every row of 16 lanes, then row 1 += last of row 0 and row 3 += last of row 2, then rows 2 and 3 += last of row 1; the (up to 16)
//            group totals, padded with +0, prefix-summed by the same Kogge-Stone network on one row of 16 lanes (v3): c_i = P[g - 1] + local_i.
//   search : branch-free lower bound by descent over the whole array (v3): pos = 0; for s = S0, S0 / 2, ..., 1 (S0 the largest power of two
//            below N): if (pos + s - 1 < N and c[pos + s - 1] < r) pos += s; clipped to N - 1.  On a non-decreasing c this IS
//            searchsorted(c, r, side='left').
//   densities : Gaussian log-densities multiply by the reciprocal diagonal of the Cholesky factor, computed once per factor (v3).
//   single draw (backward pass): B = 64 g + #{l < 64 : c_{64 g + l} < r}, g = #{k < ng - 1 : P[k] < r}, clipped to N - 1 (v3; again searchsorted
//            on a non-decreasing c): every wave finds g and counts inside the group by ballot -- one barrier per backward step.
//   shifts   : the weights of a step are e_i = exp(lw_i - M) with M an upper bound of max_i lw_i that needs NO reduction where one exists, the
//            exact maximum otherwise, and the exact maximum after all whenever every e_i underflowed (cumulative total not > 0: detected where
//            the total is formed -- one step later in the forward pass, in the same step in the backward pass).  Scale-invariant as above.
//            Forward, 1 <= t < T - 1, not the exact-gradient proposals: M_t = gb_t (+ c_t for the auxiliary proposals), gb_t = sup_x G_t(x)
//            (k_csmc_potbound: 0 | c_obs | nobs c_obs | sum_k max(0, c_obs - (1 + log y_k^2) / 2); +inf -> no bound), c_t the log-normaliser
//            of the transition density; t = 0 and t = T - 1 use the exact maximum.  fmax[t] = the shift finally used.
//            Backward: lw_i = log_ws[t][i] + log p(x_{t+1} | x_t^i) <= M := fmax[t] + c_t.
//   max    : exact, any order.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_mov(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
// inclusive scan of the wave in the order stated above (lanes without a source add +0)
template <typename R> __device__ __forceinline__ R wave_scan_dpp(R v) {
    v = v + dpp_mov<DPP_ROW_SHR1, 0xf>((R)0, v);
    v = v + dpp_mov<DPP_ROW_SHR2, 0xf>((R)0, v);
    v = v + dpp_mov<DPP_ROW_SHR4, 0xf>((R)0, v);
    v = v + dpp_mov<DPP_ROW_SHR8, 0xf>((R)0, v);
    v = v + dpp_mov<DPP_ROW_BCAST15, 0xa>((R)0, v);
    v = v + dpp_mov<DPP_ROW_BCAST31, 0xc>((R)0, v);
    return v;
}
// max of the wave, in every lane (exact: the order is immaterial)
template <typename R> __device__ __forceinline__ R wave_max_dpp(R v) {
    R o;
    o = dpp_mov<DPP_ROW_SHR1, 0xf>(v, v); v = v > o ? v : o;
    o = dpp_mov<DPP_ROW_SHR2, 0xf>(v, v); v = v > o ? v : o;
    o = dpp_mov<DPP_ROW_SHR4, 0xf>(v, v); v = v > o ? v : o;
    o = dpp_mov<DPP_ROW_SHR8, 0xf>(v, v); v = v > o ? v : o;
    o = dpp_mov<DPP_ROW_BCAST15, 0xa>(v, v); v = v > o ? v : o;
    o = dpp_mov<DPP_ROW_BCAST31, 0xc>(v, v); v = v > o ? v : o;
    return __shfl(v, 63, 64);  // lane 63 holds the maximum of the wave
}
// e_i = exp(lw_i - max lw) (non-finite max -> 0, as jax's logsumexp); red slots [0, 16).
// NW = 8 / 16: the workgroup is exactly NW full waves (N = 64 NW particles) -- no per-group bounds selects, the NW wave maxima are reduced by
// one more DPP pass instead of fifteen compare / select pairs per lane.
template <typename R, int NW = 0> __device__ __forceinline__ R block_expmax(R lw, R* red, int tid, int nw, R* m_out = nullptr) {
    const int lane = tid & 63, wv = tid >> 6;
    R m = wave_max_dpp(lw);
    if constexpr (NW > 0) {
        if (lane == 0) red[wv] = m;
        __syncthreads();
        m = wave_max_dpp(red[lane & (NW - 1)]);
    } else if (nw > 1) {
        if (lane == 0) red[wv] = m;
        __syncthreads();
        R t[16];
        load16<R>(red, t);
        m = t[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) m = (k < nw && t[k] > m) ? t[k] : m;
    }
    if (!(m - m == 0)) m = 0;
    if (m_out) *m_out = m;
    return det_exp(lw - m);
}
// lane `l` (wave-uniform) of v
__device__ __forceinline__ float readlane_(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double readlane_(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Sweep contract v3 (round 3), the prefix over the (up to 16) group totals: every wave reads the totals into lanes 0..15 (all four rows alike; slots of
// absent groups hold +0 -- the kernels zero red[32 .. 48) once) and scans them with the Kogge-Stone network of one DPP row (offsets 1, 2, 4, 8).  Lane k then
// holds P[k]; a wave's own base is ONE readlane -- no dependent left-to-right adds, no branch on the wave id (31 branches per wave and step before).
//   base = P[wv - 1] (0 for the first wave);   tot = P[last - 1] + t[last] = c[N - 1] bit for bit (the last live particle's cumulative weight)
template <typename R> __device__ __forceinline__ void totals_prefix(const R* red, int lane, int wv, int last, R& pre, R& tot, R* Pv = nullptr) {
    const R tv = red[32 + (lane & 15)];
    R v = tv;
    v = v + dpp_mov<DPP_ROW_SHR1, 0xf>((R)0, v);
    v = v + dpp_mov<DPP_ROW_SHR2, 0xf>((R)0, v);
    v = v + dpp_mov<DPP_ROW_SHR4, 0xf>((R)0, v);
    v = v + dpp_mov<DPP_ROW_SHR8, 0xf>((R)0, v);
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    pre = wvu > 0 ? readlane_(v, wvu > 0 ? wvu - 1 : 0) : (R)0;
    tot = last > 0 ? readlane_(v, last > 0 ? last - 1 : 0) + readlane_(tv, last) : readlane_(tv, 0);
    if (Pv) *Pv = v;
}
// The single draw of the backward pass (sweep contract v3): B = 64 g + #{l < 64 : c_{64 g + l} < r}, g = #{k < ng - 1 : P[k] < r} -- on a non-decreasing c
// exactly #{j : c_j < r} = searchsorted(c, r).  Every wave finds g from the totals' prefix it holds in lanes 0..15 (one ballot) and counts inside group g from
// the group's LOCAL scan values, which each wave left in LDS before the one barrier of the step: no second barrier, no exchange of per-wave counts.
//   vloc: this step's image of the local scan values (64 per group); Pv: totals_prefix's lane vector; ng groups; returns B clipped to N - 1
template <typename R> __device__ __forceinline__ int draw_two_level(const R* vloc, R Pv, int lane, int ng, int N, R r) {
    const unsigned long long below = __ballot(Pv < r);
    const unsigned long long mask = ng > 1 ? ((1ull << (ng - 1)) - 1ull) : 0ull;
    const int g = __popcll(below & mask);  // (wave-uniform)
    const int gu = __builtin_amdgcn_readfirstlane(g);
    const R base = gu > 0 ? readlane_(Pv, gu > 0 ? gu - 1 : 0) : (R)0;
    const int j = 64 * gu + lane;
    const R vl = vloc[j];
    const R cg = gu > 0 ? base + vl : vl;
    const int cntg = __popcll(__ballot(j < N && cg < r));
    const int B = 64 * gu + cntg;
    return B < N - 1 ? B : N - 1;
}
// inclusive cumsum of w into c[] in the sweep contract's order; tot = c[N - 1]; c[] valid after the trailing barrier.  red slots [32, 48).
// PAD: c[] is stored with one spare slot per 32 entries (index cpad(i) = i + (i >> 5)): the probes of the search below sit at strides of 512 .. 1
// entries, which without the padding fall into one or two LDS banks (a 16- to 32-way conflict on the later probes of every lane).
__host__ __device__ __forceinline__ constexpr int cpad(int i) { return i + (i >> 5); }
template <typename R, int NW = 0, bool PAD = false> __device__ __forceinline__ void block_cumsum_dpp(R w, R* c, R* red, int tid, int nw, R& tot) {
    const int lane = tid & 63, wv = tid >> 6;
    const R v = wave_scan_dpp(w);
    if (lane == 63) red[32 + wv] = v;
    __syncthreads();
    R pre;
    totals_prefix<R>(red, lane, wv, (NW > 0 ? NW : nw) - 1, pre, tot);
    c[PAD ? cpad(tid) : tid] = wv > 0 ? pre + v : v;
    __syncthreads();
}
// the ancestor search of the sweep contract (v3): branch-free lower bound by descent over the whole cumulative-weight array.  On the padded image of a full
// workgroup every probe is one LDS read at (running padded position + constant): while pos stays a multiple of 2 s, cpad(pos + s - 1) =
// cpad(pos) + (s - 1) + ((s - 1) >> 5), and taking the step adds s + (s >> 5).
template <typename R, int NW = 0, bool PAD = false> __device__ __forceinline__ int search2(const R* c, int N, R r) {
    if constexpr (NW > 0 && PAD) {
        // TWO levels of the descent per LDS round trip: the probe of step s and BOTH candidate probes of step s / 2 (after a step not taken / taken) are three
        // reads at constant offsets from the same position, issued together; the comparisons are exactly those of the one-level descent, in its order
        // (ten dependent LDS latencies per search were the longest chain of the forward step)
        int ppos = 0;
        constexpr int S0 = NW * 32;
        static_assert((S0 & (S0 - 1)) == 0, "the two-level descent needs a power-of-two particle count");
        constexpr int LEVELS = 32 - __builtin_clz((unsigned)S0);  // steps S0, S0 / 2, ..., 1
        constexpr bool ODD = (LEVELS & 1) != 0;  // an odd number of levels (9 at N = 512, 7 at N = 128): the first one alone, the rest in pairs
        if constexpr (ODD) ppos += c[ppos + (S0 - 1) + ((S0 - 1) >> 5)] < r ? S0 + (S0 >> 5) : 0;
#pragma unroll
        for (int s = ODD ? S0 / 2 : S0; s >= 2; s >>= 2) {
            const int h = s >> 1;
            const int ks = (s - 1) + ((s - 1) >> 5), kh = (h - 1) + ((h - 1) >> 5), ps = s + (s >> 5), ph = h + (h >> 5);
            const R A = c[ppos + ks], B0 = c[ppos + kh], B1 = c[ppos + ps + kh];
            const bool a = A < r;
            const bool b = (a ? B1 : B0) < r;
            ppos += (a ? ps : 0) + (b ? ph : 0);
        }
        const int pos = ppos - ((ppos * 1986) >> 16);  // ppos = 33 (pos >> 5) + (pos & 31)
        return pos < N - 1 ? pos : N - 1;
    } else {
        int s0 = 1;
        while (s0 * 2 < N) s0 *= 2;
        int pos = 0;
        for (int s = s0; s > 0; s >>= 1) {
            const int q = pos + s - 1;
            pos += (q < N && c[PAD ? cpad(q < N ? q : N - 1) : (q < N ? q : N - 1)] < r) ? s : 0;
        }
        return pos < N - 1 ? pos : N - 1;
    }
}

// ---- kernels both sweeps launch (csmc.hip: sequential; pit.hip: parallel in time) ------------------------------------------------------------
// the backward pass's uniforms, one per (chain, time step), drawn ONCE into an array (uniform c T + t of stream 4): inside the pass a single lane
// needed a single number per step, and the whole wave ran a Threefry block for it -- three quarters of the pass's vector instructions
template <typename R> __global__ void k_csmc_ubwd(long long n, uint32_t key0, uint32_t key1, R* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i >= n) return;
    R u0, u1;
    stream_uniform2<R>(key0, key1, STREAM_U_BWD, (unsigned long long)i, u0, u1);
    out[2 * i] = u0;
    if (2 * i + 1 < n) out[2 * i + 1] = u1;
}
// The forward pass's in-kernel draws (csmc.hip::k_csmc_fwd: one Threefry block serves two consecutive time steps of a particle), written out as the explicit
// arrays the same kernel reads in explicit-noise mode -- value for value what it would have drawn itself:
//   eps_prop[ch][t][n][k] = normal  2 (((ch T2 + (t >> 1)) N + n) D + k) + (t & 1) of stream 2,   u_res[ch][s][n] = uniform 2 ((ch T2 + (s >> 1)) N + n) + (s & 1) of stream 3
// grid (C T2, ceil(N D / 256)): blockIdx.x = ch T2 + h is the pair of time steps (2 h, 2 h + 1) of chain ch -- no 64-bit division per thread
template <typename R> __global__ void __launch_bounds__(256) k_csmc_pregen(int T, int N, int D, uint32_t key0, uint32_t key1, R* __restrict__ eps, R* __restrict__ ures) {
    const int T2 = (T + 1) >> 1, row = blockIdx.x, ch = row / T2, h = row - ch * T2;
    const int j = blockIdx.y * 256 + threadIdx.x, ND = N * D;
    if (j < ND) {
        R z0, z1;
        stream_normal2<R>(key0, key1, STREAM_EPS_PROP, (unsigned long long)row * ND + j, z0, z1);
        R* e = eps + ((long long)ch * T + 2 * h) * ND + j;
        e[0] = z0;
        if (2 * h + 1 < T) e[ND] = z1;
    }
    if (j < N) {
        R u0, u1;
        stream_uniform2<R>(key0, key1, STREAM_U_RES, (unsigned long long)row * N + j, u0, u1);
        R* u = ures + ((long long)ch * (T - 1) + 2 * h) * N + j;
        if (2 * h < T - 1) u[0] = u0;
        if (2 * h + 1 < T - 1) u[N] = u1;
    }
}
// ---- prologue: u = x + sqrt(delta_t/2) eps   (csmc/generic.py:67) ------------------------------------------------------
template <typename R> __global__ void k_csmc_aux(CsmcArgs a, int D) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.C * a.T * D;
    if (g >= total) return;
    const long long t = (g / D) % a.T;
    const R e = noise_normal<R>(a, a.eps_aux, STREAM_EPS_AUX, g);
    ((R*)a.u)[g] = fma_(((const R*)a.shd)[t], e, ((const R*)a.x)[g]);
}

// additive constants of time-varying transition densities: ct[t] = -sum_k log LQ_t[k][k] - D/2 log 2 pi
template <typename R, int D> __global__ void k_csmc_ctrans(int n, const R* __restrict__ LQt, R* __restrict__ ct, R* __restrict__ idt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    R c = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R l = LQt[((long long)t * D + k) * D + k];
        c -= det_log(l);
        idt[(long long)t * D + k] = (R)1 / l;  // reciprocal diagonal (sweep contract v3)
    }
    ct[t] = c - (R)D * (R)0.91893853320467274178;
}
// w <- (L L^T)^-1 r, L lower with leading dimension ld; fixed operation order (restated by oracle/csmc_ref.c::cho_solve_)
template <typename R, int D> __device__ __forceinline__ void cho_solve_fixed(const R* L, int ld, const R* r, R* w) {
    R z[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        R acc = r[k];
#pragma unroll
        for (int j = 0; j < k; ++j) acc = fma_(-L[k * ld + j], z[j], acc);
        z[k] = acc / L[k * ld + k];
    }
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        R acc = z[k];
#pragma unroll
        for (int j = k + 1; j < D; ++j) acc = fma_(-L[j * ld + k], w[j], acc);
        w[k] = acc / L[k * ld + k];
    }
}
// gradient at u of  log M0(u_0) + G0(u_0) + sum_t [log Mt(u_{t+1} | u_t) + Gt(u_{t+1})]  (csmc/independent.py:121-134, jax.grad there),
// closed form for the model family: one thread per (chain, time step)
template <typename R, int D> __global__ void k_csmc_grad(CsmcArgs a, FkDev<R> m) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)a.C * a.T) return;
    const long long t = g % a.T;
    const R* u = (const R*)a.u + g * D;
    R ut[D], gr[D], r[D], w[D], mu[D];
#pragma unroll
    for (int k = 0; k < D; ++k) ut[k] = u[k];
    const R* yv = (const R*)a.y;
    // potential
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R y = yv ? yv[t * D + k] : (R)0;
        R v = 0;
        if (m.potential == 1 || (m.potential == 3 && y - y == 0)) v = ((y - ut[k]) * m.inv_sig_y) * m.inv_sig_y;
        else if (m.potential == 2) {
            const R e = det_exp(-ut[k]);
            v = (R)0.5 * fma_(y * y, e, (R)-1);
            v = (v == v) ? v : (R)0;
        }
        gr[k] = v;
    }
    // density of u_t given the past
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) r[k] = ut[k] - m.m0[k];
        cho_solve_fixed<R, D>(m.LP0, CS_MAXD, r, w);
    } else {
        const TransT<R> tr = trans_at<R, D>(m, t - 1);
        trans_mean_t<R, D>(m, tr, u - D, mu);
#pragma unroll
        for (int k = 0; k < D; ++k) r[k] = ut[k] - mu[k];
        cho_solve_fixed<R, D>(tr.LQ, tr.ld, r, w);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) gr[k] = gr[k] - w[k];
    // density of u_{t+1} given u_t:  J(u_t)^T Q^-1 (u_{t+1} - mean(u_t))
    if (t + 1 < a.T) {
        const TransT<R> tr = trans_at<R, D>(m, t);
        trans_mean_t<R, D>(m, tr, ut, mu);
#pragma unroll
        for (int k = 0; k < D; ++k) r[k] = u[D + k] - mu[k];
        cho_solve_fixed<R, D>(tr.LQ, tr.ld, r, w);
        if constexpr (D == 3) {
            if (m.transition == 1) {  // Lorenz-63: J = I + dt dphi/dx (examples/lorenz/model.py:10-25)
                const R th1 = m.F[0], th2 = m.F[1], th3 = m.F[2], dt = m.b[0];
                const R J[9] = {-th1, th1, (R)0, th2 - ut[2], (R)-1, -ut[0], ut[1], ut[0], -th3};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    R acc = 0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc = fma_(J[j * 3 + k], w[j], acc);
                    gr[k] = gr[k] + fma_(dt, acc, w[k]);
                }
                goto done;
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            R acc = 0;
#pragma unroll
            for (int j = 0; j < D; ++j) acc = fma_(tr.F[j * tr.ld + k], w[j], acc);
            gr[k] = gr[k] + acc;
        }
    }
done:
#pragma unroll
    for (int k = 0; k < D; ++k) ((R*)a.grad)[g * D + k] = gr[k];
}

// sum_k [log N(x_k; u_k, s) - log N(x_k; pm_k, s)] = sum_k ((x_k - pm_k)^2 - (x_k - u_k)^2) / (2 s^2)   (independent.py:184-189)
template <typename R, int D> __device__ __forceinline__ R grad_correction(const R* x, const R* u, const R* pm, R s) {
    R acc = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R d1 = x[k] - u[k], d2 = x[k] - pm[k];
        acc = fma_(d2, d2, acc);
        acc = fma_(-d1, d1, acc);
    }
    return acc * ((R)0.5 / (s * s));
}

template <typename R> static void fill_model(FkDev<R>& m, const auxssm_fk_model* fk, const double* host) {
    // host = [m0 (D) | chol_P0 (D*D) | F (D*D) | b (D) | chol_Q (D*D)] as doubles
    const int D = fk->dx;
    memset(&m, 0, sizeof(m));  // (also: no time-varying arrays, no gradient; the sweep entry point sets them)
    m.proposal = fk->proposal;
    m.potential = fk->potential;
    m.D = D;
    m.transition = fk->transition;
    const double* p = host;
    for (int k = 0; k < D; ++k) m.m0[k] = (R)p[k];
    p += D;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) m.LP0[i * CS_MAXD + j] = (R)p[i * D + j];
    p += D * D;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) m.F[i * CS_MAXD + j] = (R)p[i * D + j];
    p += D * D;
    for (int k = 0; k < D; ++k) m.b[k] = (R)p[k];
    p += D;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) m.LQ[i * CS_MAXD + j] = (R)p[i * D + j];
    // additive constants, computed once on the host in precision R (they enter both the GPU and the oracle as data)
    R ci = 0, ct = 0;
    for (int k = 0; k < D; ++k) {
        ci -= det_log(m.LP0[k * CS_MAXD + k]);
        ct -= det_log(m.LQ[k * CS_MAXD + k]);
    }
    for (int k = 0; k < D; ++k) {
        m.iLP0[k] = (R)1 / m.LP0[k * CS_MAXD + k];
        m.iLQ[k] = (R)1 / m.LQ[k * CS_MAXD + k];
    }
    const R half_log_2pi = (R)0.91893853320467274178;
    m.c_init = ci - (R)D * half_log_2pi;
    m.c_trans = ct - (R)D * half_log_2pi;
    if (fk->potential == 1) {
        m.inv_sig_y = (R)1 / (R)fk->sig_y;
        m.c_obs = -(R)D * det_log((R)fk->sig_y) - (R)D * half_log_2pi;
    } else if (fk->potential == 3) {  // per observed component
        m.inv_sig_y = (R)1 / (R)fk->sig_y;
        m.c_obs = -det_log((R)fk->sig_y) - half_log_2pi;
    } else {
        m.inv_sig_y = 0;
        m.c_obs = -half_log_2pi;
    }
}

}  // namespace ax

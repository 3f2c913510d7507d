// csmc.hip -- conditional SMC (particle Gibbs) sweep: reference aux_samplers/_primitives/csmc/csmc.py,
// resamplings.py::multinomial, math/utils.py::normalize, and the auxiliary wrappers csmc/generic.py and
// csmc/independent.py (classical, non-gradient branch).  Compiled with -ffp-contract=off; every multiply-add that
// is meant to be fused is an explicit fma so that the CPU oracle reproduces the arithmetic bit for bit.
//
// Execution model: ONE workgroup per chain, one lane per particle (N <= 1024), a persistent loop over the T
// time steps (the recursion is sequential in t; throughput comes from running >= 256 chains side by side).
// Per step: block inclusive scan of the normalised weights (wave-level Kogge-Stone with shuffles + ordered wave
// totals through LDS) -> N binary searches in LDS (conditional multinomial resampling, index 0 pinned) -> gather
// parents from LDS -> propagate -> pin particle 0 to the reference trajectory -> log-weights -> block max / sum
// -> normalise.  xs, log_ws, As stream to HBM with the particle index fastest (coalesced).
//
// Reduction orders (the contract the oracle restates, SURVEY 7 "bit-exact ancestors"): the sweep kernels follow the "sweep contract"
// of csmc_dev.h (unnormalised weights exp(lw - max), the hardware's DPP scan order, two-level search, ballot-counted single draw);
// the standalone primitives (normalize / multinomial / systematic) and the parallel-in-time sweep keep the Kogge-Stone-in-64 cumsum,
// the balanced-tree-in-64 sum and the plain binary search of csmc_dev.h.
#include "csmc_dev.h"

namespace ax {

// gb[t] = sup_x G_t(x): the reduction-free part of the forward weights' shift (sweep contract, csmc_dev.h); +inf where the potential is unbounded
template <typename R, int D> __global__ void k_csmc_potbound(int T, FkDev<R> m, const R* __restrict__ y, R* __restrict__ gb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    R b = 0;
    if (m.potential == 1) b = m.c_obs;
    else if (m.potential == 3) {
        int nobs = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) nobs += (y[(long long)t * D + k] - y[(long long)t * D + k] == 0) ? 1 : 0;
        b = (R)nobs * m.c_obs;
    } else if (m.potential == 2) {  // sum_k [c_obs - (x + y^2 e^-x) / 2] <= sum_k max(0, c_obs - (1 + log y^2) / 2)  (a NaN term counts 0)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const R yk = y[(long long)t * D + k], y2 = yk * yk;
            R v = (R)0;
            if (y2 - y2 == 0) v = y2 > (R)0 ? fma_((R)-0.5, (R)1 + det_log(y2), m.c_obs) : (R)INFINITY;
            b += v > (R)0 ? v : (R)0;
        }
    }
    gb[t] = b;
}
// Diagnostic builds only (tools/csmc_ablate.sh): -DAUXSSM_CSMC_ABLATE=<mask> removes one phase of the forward step at a time (wrong results, right
// shape) to attribute its time: 1 search, 2 in-kernel draws, 4 potential / transition log-density, 8 max + exp, 16 cumsum; 32: max + exp of the
// BACKWARD pass.  0 in the product.
#ifndef AUXSSM_CSMC_ABLATE
#define AUXSSM_CSMC_ABLATE 0
#endif
constexpr int CSMC_ABL = AUXSSM_CSMC_ABLATE;

// ---- forward pass (_csmc, csmc.py:69-107) -------------------------------------------------------------------------------
// NW = 8 / 16: exactly NW full waves (N = blockDim = 64 NW: the C4 / C3 shapes): no liveness / group-bound selects (csmc_dev.h); NW = 0: any N
// SP = 1: the instantiation of config C3's shape -- auxiliary independent proposals, the stochastic-volatility potential, a time-invariant linear transition, draws
// generated in the kernel, no ancestor trace (backward sampling): the run-time switches on the model kind are folded at compile time (they are wave-uniform
// branches, two dozen per time step); same operations on the same operands, bit for bit (tests/test_gpu_csmc.py runs both instantiations on C3's model)
template <typename R, int D, bool TV, bool GRAD, int NW, int SP = 0> __global__ void __launch_bounds__(1024) k_csmc_fwd(CsmcArgs a, FkDev<R> m) {
    if constexpr (SP == 1) {
        m.proposal = 1;
        m.potential = 2;
        m.transition = 0;
        a.As = nullptr;
        a.noise_mode = 1;
        a.pregen = 0;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TB = blockDim.x, nw = TB >> 6, tid = threadIdx.x, N = a.N, T = a.T;
    // two images of (c, xprev), alternated by time-step parity: readers of step t never race writers of step t+1,
    // which removes the end-of-step barrier (4 barriers per step: max, sum, wave totals, publish)
    const int CP = cpad(TB);            // the cumsum image is padded against LDS bank conflicts of the search (csmc_dev.h::cpad)
    R* cbuf = (R*)smem;                 // [2][CP]
    R* xbuf = cbuf + 2 * CP;            // [2][TB][D]
    R* red = xbuf + 2 * TB * D;         // [48]
    const int ch = a.c0 + blockIdx.x;
    const bool live = NW > 0 ? true : tid < N;
    if (tid < 16) red[32 + tid] = 0;  // totals of absent groups: +0 (totals_prefix reads all 16 slots; ordered by the first barrier below)
    const R* xstar = (const R*)a.x + (long long)ch * T * D;
    const R* uaux = (const R*)a.u + (long long)ch * T * D;
    const R* gaux = GRAD ? (const R*)a.grad + (long long)ch * T * D : uaux;
    const R* yv = (const R*)a.y;
    R* xs = (R*)a.xs + (long long)ch * T * N * D;
    R* lws = (R*)a.lws + (long long)ch * T * N;
    int32_t* As = a.As ? a.As + (long long)ch * (T - 1) * N : nullptr;
    const long long eps_base = (long long)ch * T * N * D;
    const long long ures_base = (long long)ch * (T - 1) * N;
    const R ninf = -INFINITY;

    // t = 0  (csmc.py:74-80)
    // In-kernel draws (THREEFRY): one Threefry block serves TWO consecutive time steps of a particle, so each step pays for
    // one block (normals on even t, uniforms on odd t) instead of two.  With T2 = ceil(T/2):
    //   eps_prop[ch][t][n][k] = normal  2 * (((ch T2 + (t >> 1)) N + n) D + k) + (t & 1)  of stream 2
    //   u_res[ch][s][n]       = uniform 2 * ((ch T2 + (s >> 1)) N + n) + (s & 1)          of stream 3
    // (flat auxssm_rng_* indices; csmc/_device.py::key_noise builds the equivalent explicit arrays).
    const bool gen = a.noise_mode != 0 && !a.pregen;
    const long long T2 = (T + 1) >> 1;
    R x[D], eps[D], eps_nx[D], ycur[D], pm[D], un_nx = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        eps_nx[k] = 0;
        if (gen) {
            R z0, z1;
            stream_normal2<R>(a.key0, a.key1, STREAM_EPS_PROP, (unsigned long long)(((long long)ch * T2 * N + tid) * D + k), z0, z1);
            eps[k] = live ? z0 : (R)0;
            eps_nx[k] = live ? z1 : (R)0;
        } else {
            eps[k] = live ? ((const R*)a.eps_prop)[eps_base + (long long)tid * D + k] : (R)0;
        }
        ycur[k] = yv ? yv[k] : (R)0;
    }
    if (m.proposal == 0) {  // M0 = N(m0, P0)
#pragma unroll
        for (int k = 0; k < D; ++k) {
            R acc = m.m0[k];
#pragma unroll
            for (int j = 0; j <= k; ++j) acc = fma_(m.LP0[k * CS_MAXD + j], eps[j], acc);
            x[k] = acc;
        }
    } else {  // AuxiliaryM0: N(u_0 [+ delta_0/2 grad_0], delta_0/2 I)  (independent.py:143-158)
        const R s0 = ((const R*)a.shd)[0];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            pm[k] = GRAD ? fma_(s0 * s0, gaux[k], uaux[k]) : uaux[k];
            x[k] = fma_(s0, eps[k], pm[k]);
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = xstar[k];
    }
    R lw;
    {
        R g = potential<R, D>(m, x, ycur);
        if (m.proposal == 1) {
            g = g + gauss_chol_logpdf<R, D>(x, m.m0, m.LP0, m.iLP0, m.c_init);  // AuxiliaryG0 (independent.py:163-169)
            if constexpr (GRAD) g = g + grad_correction<R, D>(x, uaux, pm, ((const R*)a.shd)[0]);  // GradientAuxiliaryG0 (:173-190)
        }
        lw = live ? g : ninf;
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < D; ++k) xs[(long long)tid * D + k] = x[k];
        lws[tid] = lw;
    }
    R* fmax = a.fmax ? (R*)a.fmax + (long long)ch * T : nullptr;
    R mstep;
    R w = block_expmax<R, NW>(lw, red, tid, nw, &mstep);
    if (fmax && tid == 0) fmax[0] = mstep;
    const R* gbp = (const R*)a.gb;
    const bool bmode = gbp != nullptr && !(GRAD && m.gradient == 2);  // (the exact-gradient correction is unbounded in x)
    bool used_bound = false;

    for (int t = 1; t < T; ++t) {
        // issue this step's independent loads first
        R un = 0;
        const R gbt = bmode ? gbp[t] : (R)0;
#pragma unroll
        for (int k = 0; k < D; ++k) ycur[k] = yv ? yv[(long long)t * D + k] : (R)0;
        if (!gen) {
#pragma unroll
            for (int k = 0; k < D; ++k) eps[k] = live ? ((const R*)a.eps_prop)[eps_base + ((long long)t * N + tid) * D + k] : (R)0;
            if (live) un = ((const R*)a.u_res)[ures_base + (long long)(t - 1) * N + tid];
        } else if (CSMC_ABL & 2) {
#pragma unroll
            for (int k = 0; k < D; ++k) eps[k] = (R)0.25;
            un = (R)0.37;
        } else if (t & 1) {  // normals cached by step t - 1; uniforms of steps t and t + 1
#pragma unroll
            for (int k = 0; k < D; ++k) eps[k] = eps_nx[k];
            stream_uniform2<R>(a.key0, a.key1, STREAM_U_RES, (unsigned long long)(((long long)ch * T2 + ((t - 1) >> 1)) * N + tid), un, un_nx);
        } else {  // uniform cached by step t - 1; normals of steps t and t + 1
            un = un_nx;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                R z0, z1;
                stream_normal2<R>(a.key0, a.key1, STREAM_EPS_PROP, (unsigned long long)((((long long)ch * T2 + (t >> 1)) * N + tid) * D + k), z0, z1);
                eps[k] = live ? z0 : (R)0;
                eps_nx[k] = live ? z1 : (R)0;
            }
        }
        // conditional multinomial resampling (resamplings.py:14-37 -> jax.random.choice: cumsum, r = c[-1] (1-u), searchsorted)
        R* c = cbuf + (t & 1) * CP;
        R* xprev = xbuf + (t & 1) * TB * D;
#pragma unroll
        for (int k = 0; k < D; ++k) xprev[tid * D + k] = x[k];
        R tot;
        if (CSMC_ABL & 16) {
            tot = (R)N;
            c[cpad(tid)] = w;
            __syncthreads();
        } else
            block_cumsum_dpp<R, NW, true>(w, c, red, tid, nw, tot);  // trailing barrier also publishes xprev; tot = c[N - 1]
        if (used_bound && !(tot > (R)0)) {  // every weight of step t - 1 underflowed under its bound: the exact maximum after all (uniform)
            __syncthreads();                // (every lane is past its reads of the step's images before they are rewritten)
            w = block_expmax<R, NW>(lw, red, tid, nw, &mstep);
            if (fmax && tid == 0) fmax[t - 1] = mstep;
            block_cumsum_dpp<R, NW, true>(w, c, red, tid, nw, tot);
        }
        int idx = 0;
        if (CSMC_ABL & 1) idx = (tid * 7) & (N - 1);
        else if (live && tid > 0) idx = search2<R, NW, true>(c, N, tot * ((R)1 - un));
        R xp[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xp[k] = xprev[idx * D + k];
        // propagate (csmc.py:91-92); the transition t - 1 -> t (time-varying: row t - 1 of the device arrays)
        const TransT<R> tr = trans_at_c<R, D, TV>(m, t - 1);
        if (m.proposal == 0) {
            R mu[D];
            trans_mean_t<R, D>(m, tr, xp, mu);
#pragma unroll
            for (int k = 0; k < D; ++k) {
                R acc = mu[k];
#pragma unroll
                for (int j = 0; j <= k; ++j) acc = fma_(tr.LQ[k * tr.ld + j], eps[j], acc);
                x[k] = acc;
            }
        } else {  // AuxiliaryMtDynamics: N(u_t [+ delta_t/2 grad_t], delta_t/2 I), independent of the parent (independent.py:192-198)
            const R st = ((const R*)a.shd)[t];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                pm[k] = GRAD ? fma_(st * st, gaux[(long long)t * D + k], uaux[(long long)t * D + k]) : uaux[(long long)t * D + k];
                x[k] = fma_(st, eps[k], pm[k]);
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) x[k] = xstar[(long long)t * D + k];
        }
        // weights (csmc.py:95-96)
        if (CSMC_ABL & 4) lw = x[0] * (R)0.01;
        else {
            R g = potential<R, D>(m, x, ycur);
            if (m.proposal == 1) {  // AuxiliaryGt = Mt.logpdf + Gt (independent.py:238-248)
                R mu[D];
                trans_mean_t<R, D>(m, tr, xp, mu);
                g = gauss_chol_logpdf<R, D>(x, mu, tr.LQ, tr.iL, tr.c_trans, tr.ld) + g;
                // GradientAuxiliaryGt (:252-268): in the reference the correction is summed over all particles, i.e. a constant of the
                // step (AUXSSM_GRAD_REFERENCE: nothing to add); AUXSSM_GRAD_EXACT applies it per particle
                if constexpr (GRAD) {
                    if (m.gradient == 2) g = g + grad_correction<R, D>(x, uaux + (long long)t * D, pm, ((const R*)a.shd)[t]);
                }
            }
            lw = live ? g : ninf;
        }
        if (live) {
            const long long o = (long long)t * N + tid;
#pragma unroll
            for (int k = 0; k < D; ++k) xs[o * D + k] = x[k];
            lws[o] = lw;
            if (As) As[(long long)(t - 1) * N + tid] = idx;
        }
        if (CSMC_ABL & 8) w = lw * (R)0.001 + (R)1;
        else {
            // the shift of this step's weights (sweep contract): a reduction-free bound where there is one, else the block maximum
            R Mb = gbt + (m.proposal == 1 ? tr.c_trans : (R)0);
            used_bound = bmode && t < T - 1 && (Mb - Mb == 0);
            if (used_bound) {
                w = det_exp(lw - Mb);
                mstep = Mb;
            } else {
                w = block_expmax<R, NW>(lw, red, tid, nw, &mstep);
            }
            if (fmax && tid == 0) fmax[t] = mstep;
        }
    }
    if (live) ((R*)a.wT)[(long long)ch * N + tid] = w;
}

// ---- backward passes (csmc.py:110-149) ------------------------------------------------------------------------------------
// One draw per step: B = #{j : c_j < r} by ballot + per-wave counts (no serial search), the candidate particles of the step are
// published to LDS before the first barrier so that x_t^B is an LDS read, and the next step's rows (xs, log_ws) and uniform are
// fetched one step ahead: no global-memory latency on the dependent chain.  4 barriers per step (max, wave totals, publish, counts).
template <typename R, int D, bool TV, int NW> __global__ void __launch_bounds__(1024) k_csmc_bwd(CsmcArgs a, FkDev<R> m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TB = blockDim.x, nw = TB >> 6, tid = threadIdx.x, N = a.N, T = a.T;
    R* c = (R*)smem;                 // [2][TB] the step's cumulative weights (generic) / local scan values (full workgroups), by step parity
    R* red = c + 2 * TB;             // [48] + [16]: a second set of wave totals (slots [48, 64)) for the odd steps of the one-barrier loop
    R* xpub = red + 64;              // [2][TB][D] candidate particles of the step, by step parity
    R* ubuf = xpub + 2 * TB * D;     // [2] the step's uniform, by step parity
    const int ch = a.c0 + blockIdx.x;
    const bool live = NW > 0 ? true : tid < N;
    if (tid < 16) red[32 + tid] = 0, red[48 + tid] = 0;  // totals of absent groups: +0 (csmc_dev.h::totals_prefix)
    const R* xs = (const R*)a.xs + (long long)ch * T * N * D;
    const R* lws = (const R*)a.lws + (long long)ch * T * N;
    const int32_t* As = a.As ? a.As + (long long)ch * (T - 1) * N : nullptr;
    R* xout = (R*)a.x + (long long)ch * T * D;
    int32_t* anc = a.anc + (long long)ch * T;
    const long long ub_base = (long long)ch * T;
    const R ninf = -INFINITY;

    // B_T ~ choice(w_T)   (csmc.py:111 / :131); w_T are the forward pass's unnormalised weights
    R w = live ? ((const R*)a.wT)[(long long)ch * N + tid] : (R)0;
    if (tid == 0) ubuf[1] = ((const R*)a.u_bwd)[ub_base + (T - 1)];
    {
        R xi[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xi[k] = live ? xs[((long long)(T - 1) * N + tid) * D + k] : (R)0;
#pragma unroll
        for (int k = 0; k < D; ++k) xpub[(TB + tid) * D + k] = xi[k];
    }
    R tot;
    int B;
    {   // (sweep contract: the two-level single draw; the generic cumsum image c[] holds base + local, so the local values are recovered per group)
        const int lane = tid & 63, wv = tid >> 6;
        const R v = wave_scan_dpp(w);
        c[TB + tid] = v;
        if (lane == 63) red[32 + wv] = v;
        __syncthreads();
        R pre, Pv;
        totals_prefix<R>(red, lane, wv, nw - 1, pre, tot, &Pv);
        B = draw_two_level<R>(c + TB, Pv, lane, nw, N, tot * ((R)1 - ubuf[1]));
    }
    R xn[D];
#pragma unroll
    for (int k = 0; k < D; ++k) xn[k] = xpub[(TB + B) * D + k];
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < D; ++k) xout[(long long)(T - 1) * D + k] = xn[k];
        anc[T - 1] = B;
    }
    if (!a.backward) {
        // ancestor tracing: B_{t-1} = A_t[B_t]  (csmc.py:114-121); a dependent pointer chase, one lane
        if (tid == 0) {
            for (int t = T - 1; t >= 1; --t) {
                B = As[(long long)(t - 1) * N + B];
#pragma unroll
                for (int k = 0; k < D; ++k) xout[(long long)(t - 1) * D + k] = xs[((long long)(t - 1) * N + B) * D + k];
                anc[t - 1] = B;
            }
        }
        return;
    }
    // backward sampling (Whiteley), csmc.py:134-146
    __syncthreads();  // the parity-1 slots of the first draw are free again
    R xi_nx[D], lw_nx = ninf, un_nx = 0, fm_nx = 0;
    const R* fmax = (const R*)a.fmax + (long long)ch * T;
    if (T >= 2) {
#pragma unroll
        for (int k = 0; k < D; ++k) xi_nx[k] = live ? xs[((long long)(T - 2) * N + tid) * D + k] : (R)0;
        lw_nx = live ? lws[(long long)(T - 2) * N + tid] : ninf;
        fm_nx = fmax[T - 2];
        if (tid == 0) un_nx = ((const R*)a.u_bwd)[ub_base + (T - 2)];
    }
    for (int t = T - 2; t >= 0; --t) {
        const int par = t & 1;
        R xi[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xi[k] = xi_nx[k];
        const R lwi = lw_nx, un_t = un_nx, fm_t = fm_nx;
        if (t > 0) {  // the rows of step t - 1: independent of this step's draw
#pragma unroll
            for (int k = 0; k < D; ++k) xi_nx[k] = live ? xs[((long long)(t - 1) * N + tid) * D + k] : (R)0;
            lw_nx = live ? lws[(long long)(t - 1) * N + tid] : ninf;
            fm_nx = fmax[t - 1];
            if (tid == 0) un_nx = ((const R*)a.u_bwd)[ub_base + (t - 1)];
        }
        R lw = ninf;
        const TransT<R> tr = trans_at_c<R, D, TV>(m, t);  // Pt.logpdf(x_{t+1}, xs_t, params_t) (csmc.py:136)
        if (live) {
            R mu[D];
            trans_mean_t<R, D>(m, tr, xi, mu);
            lw = gauss_chol_logpdf<R, D>(xn, mu, tr.LQ, tr.iL, tr.c_trans, tr.ld) + lwi;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) xpub[(par * TB + tid) * D + k] = xi[k];
        if (tid == 0) ubuf[par] = un_t;
        // weights shifted by a bound of their maximum that needs no reduction (sweep contract): the forward pass's block maximum of
        // log_ws[t] plus the transition's log-normaliser; the exact maximum only if everything underflowed
        R Mb = fm_t + tr.c_trans;
        if (!(Mb - Mb == 0)) Mb = 0;
        if (CSMC_ABL & 32) w = lw * (R)0.001 + (R)1;  // (diagnostic build: the backward pass without its exp)
        else w = det_exp(lw - Mb);
        {   // ONE barrier per step: local scan values and wave totals of this parity are published together with xpub / ubuf; every wave then finds the
            // group and counts inside it on its own (csmc_dev.h::draw_two_level)
            const int lane = tid & 63, wv = tid >> 6;
            R* vloc = c + par * TB;
            R* tl = red + 32 + par * 16;
            R v = wave_scan_dpp(w);
            vloc[tid] = v;
            if (lane == 63) tl[wv] = v;
            __syncthreads();
            R pre, Pv;
            totals_prefix<R>(tl - 32, lane, wv, nw - 1, pre, tot, &Pv);
            if (!(tot > (R)0)) {  // (uniform) every weight underflowed under its bound: the exact maximum after all
                __syncthreads();  // (every wave has read this parity's totals before they are rewritten)
                w = block_expmax<R, NW>(lw, red, tid, nw);
                v = wave_scan_dpp(w);
                vloc[tid] = v;
                if (lane == 63) tl[wv] = v;
                __syncthreads();
                totals_prefix<R>(tl - 32, lane, wv, nw - 1, pre, tot, &Pv);
            }
            B = draw_two_level<R>(vloc, Pv, lane, nw, N, tot * ((R)1 - ubuf[par]));
        }
#pragma unroll
        for (int k = 0; k < D; ++k) xn[k] = xpub[(par * TB + B) * D + k];
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < D; ++k) xout[(long long)t * D + k] = xn[k];
            anc[t] = B;
        }
    }
}

// ---- standalone primitives: normalize (math/utils.py:23-39) and conditional multinomial resampling (resamplings.py:14-37),
// one workgroup per row, exactly the block primitives of the forward pass
template <typename R> __global__ void __launch_bounds__(1024) k_normalize_resample(int N, const R* lw, const R* w_in, const R* un, R* w_out, int32_t* idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TB = blockDim.x, nw = TB >> 6, tid = threadIdx.x;
    R* c = (R*)smem;
    R* red = c + TB;
    const long long row = (long long)blockIdx.x * N;
    const bool live = tid < N;
    R w;
    if (lw) {
        w = block_normalize<R>(live ? lw[row + tid] : (R)-INFINITY, red, tid, nw);
        if (live && w_out) w_out[row + tid] = w;
    } else {
        w = live ? w_in[row + tid] : (R)0;
    }
    if (!idx) return;
    block_cumsum<R>(w, c, red, tid, nw);
    if (live) {
        int i = 0;
        if (tid > 0) {
            const R r = c[N - 1] * ((R)1 - un[row + tid]);
            i = lower_bound<R>(c, N, r);
            i = i < N - 1 ? i : N - 1;
        }
        idx[row + tid] = i;
    }
}

// conditional systematic resampling (resamplings.py:40-86; Chopin & Singh, Algorithm 4): M normalised weights -> N indices, index 0 kept at
// position 0.  One workgroup per row; (U, V, W) ~ U[0,1)^3 given per row.  Same block cumsum as the multinomial path.
template <typename R>
__global__ void __launch_bounds__(1024) k_systematic(int M, int N, const R* __restrict__ w_in, const R* __restrict__ uvw, int32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TB = blockDim.x, nw = TB >> 6, tid = threadIdx.x;
    R* c = (R*)smem;
    R* red = c + TB;
    int* idx = (int*)(red + 48);
    __shared__ int nzero;
    const long long row = blockIdx.x;
    const R w = tid < M ? w_in[row * M + tid] : (R)0;
    if (tid == 0) nzero = 0;
    block_cumsum<R>(w, c, red, tid, nw);
    const R U = uvw[row * 3], V = uvw[row * 3 + 1], W = uvw[row * 3 + 2];
    const R tmp = (R)N * w_in[row * M];
    const R fl = floor(tmp);
    R uni;
    if (tmp <= (R)1) {
        uni = tmp * U;
    } else {
        const R rem = tmp - fl;
        const R p_cond = rem * (fl + (R)1) / tmp;
        uni = V < p_cond ? rem * U : rem + ((R)1 - rem) * U;
    }
    int i = 0;
    if (tid < N) {
        const R pos = ((R)tid + uni) / (R)N;
        i = lower_bound<R>(c, M, pos);
        idx[tid] = i;
        if (i == 0) atomicAdd(&nzero, 1);
    }
    __syncthreads();
    if (tid < N) {
        const int nz = nzero;
        int o = i;
        if (nz != 1) {
            // idx is non-decreasing, so its zeros are the first nz positions: zero_loc[k] = k for k < nz, the fill value -1 otherwise
            const int roll_idx = (int)floor((R)nz * W);
            const int shift = roll_idx < nz ? roll_idx : -1;
            int src = (tid + shift) % N;
            if (src < 0) src += N;
            o = idx[src];
        }
        o = o < 0 ? 0 : (o > M - 1 ? M - 1 : o);
        out[row * N + tid] = o;
    }
}

template <typename R, int D>
static int run_csmc(auxssm_ctx* h, const auxssm_fk_model* fk, const double* host_model, CsmcArgs& a, void* ctt) {
    FkDev<R> m;
    fill_model<R>(m, fk, host_model);
    m.gradient = fk->gradient;
    const int TB = (a.N + 63) / 64 * 64;
    if (fk->F_t && a.T > 1) {
        m.Ft = (const R*)fk->F_t;
        m.bt = (const R*)fk->b_t;
        m.LQt = (const R*)fk->chol_Q_t;
        m.ctt = (const R*)ctt;
        m.idt = (const R*)ctt + (a.T - 1);  // (the caller sizes ctt for (T - 1) (1 + D) reals)
        hipLaunchKernelGGL((k_csmc_ctrans<R, D>), dim3((a.T - 1 + 255) / 256), dim3(256), 0, h->stream, a.T - 1, m.LQt, (R*)ctt, (R*)ctt + (a.T - 1));
    }
    if (a.gb) hipLaunchKernelGGL((k_csmc_potbound<R, D>), dim3((a.T + 255) / 256), dim3(256), 0, h->stream, a.T, m, (const R*)a.y, (R*)a.gb);
    if (fk->proposal == 1) {
        const long long total = (long long)a.C * a.T * D;
        hipLaunchKernelGGL((k_csmc_aux<R>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, a, D);
        if (fk->gradient) {
            const long long tot = (long long)a.C * a.T;
            hipLaunchKernelGGL((k_csmc_grad<R, D>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, a, m);
        }
    }
    // forward + backward pass, batch of chains by batch (CsmcArgs::c0; one batch unless the particle systems of all chains do not fit the device)
    const int cb = a.cb > 0 ? a.cb : a.C;
    for (int c0 = 0; c0 < a.C; c0 += cb) {
    CsmcArgs ab = a;
    ab.c0 = c0;
    ab.C = a.C - c0 < cb ? a.C - c0 : cb;
    ab.xs = (char*)a.xs - (size_t)c0 * a.xs_rec;
    ab.lws = (char*)a.lws - (size_t)c0 * a.lws_rec;
    if (a.As) ab.As = (int32_t*)((char*)a.As - (size_t)c0 * a.As_rec);
    {
        ProfScope ps(h, AUXSSM_K_CSMC_FWD);
        const size_t lds = (size_t)2 * (cpad(TB) + TB * D) * sizeof(R) + 48 * sizeof(R) + 64;
        const bool tv = m.Ft != nullptr, gr = m.gradient != 0;
        const int fullw = (TB == a.N && (a.N == 1024 || a.N == 512)) ? a.N / 64 : 0;
#define AX_FWD1(TVv, GRv, NWv)                                                                                                                                  \
    do {                                                                                                                                                        \
        if (lds > 48 * 1024) AX_HIP(hipFuncSetAttribute((const void*)k_csmc_fwd<R, D, TVv, GRv, NWv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_csmc_fwd<R, D, TVv, GRv, NWv>), dim3(ab.C), dim3(TB), lds, h->stream, ab, m);                                                   \
    } while (0)
#define AX_FWD(TVv, GRv)                \
    do {                                \
        if (fullw == 16) AX_FWD1(TVv, GRv, 16); \
        else if (fullw == 8) AX_FWD1(TVv, GRv, 8); \
        else AX_FWD1(TVv, GRv, 0);      \
    } while (0)
        static const bool spec_on = [] { const char* e = getenv("AUXSSM_CSMC_SPEC"); return !(e && atoi(e) == 0); }();   // 0: the generic instantiation (measurement, tests)
        const bool c3_shape = spec_on && D == 1 && sizeof(R) == 4 && fullw == 16 && !tv && !gr && fk->proposal == 1 && fk->potential == 2 && m.transition == 0 &&
                              ab.As == nullptr && ab.noise_mode != 0 && !ab.pregen;
        if (c3_shape) {
            if constexpr (D == 1 && sizeof(R) == 4) {
                if (lds > 48 * 1024) AX_HIP(hipFuncSetAttribute((const void*)k_csmc_fwd<R, D, false, false, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_csmc_fwd<R, D, false, false, 16, 1>), dim3(ab.C), dim3(TB), lds, h->stream, ab, m);
            }
        } else if (tv && gr) AX_FWD(true, true);
        else if (tv) AX_FWD(true, false);
        else if (gr) AX_FWD(false, true);
        else AX_FWD(false, false);
#undef AX_FWD
#undef AX_FWD1
    }
    {
        ProfScope ps(h, AUXSSM_K_CSMC_BWD);
        const size_t lds = (size_t)2 * TB * sizeof(R) + 64 * sizeof(R) + (size_t)2 * TB * D * sizeof(R) + 2 * sizeof(R) + 64;
        const int fullw = (TB == a.N && (a.N == 1024 || a.N == 512)) ? a.N / 64 : 0;
#define AX_BWD(TVv, NWv)                                                                                                                                 \
    do {                                                                                                                                                 \
        if (lds > 48 * 1024) AX_HIP(hipFuncSetAttribute((const void*)k_csmc_bwd<R, D, TVv, NWv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_csmc_bwd<R, D, TVv, NWv>), dim3(ab.C), dim3(TB), lds, h->stream, ab, m);                                                   \
    } while (0)
        if (m.Ft) {
            if (fullw == 16) AX_BWD(true, 16);
            else if (fullw == 8) AX_BWD(true, 8);
            else AX_BWD(true, 0);
        } else {
            if (fullw == 16) AX_BWD(false, 16);
            else if (fullw == 8) AX_BWD(false, 8);
            else AX_BWD(false, 0);
        }
#undef AX_BWD
    }
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

}  // namespace ax

namespace ax {
int run_csmc_wide(auxssm_ctx* h, int dtype, const auxssm_fk_model* fk, CsmcArgs& a, void* ctt);  // csmc_wide.hip
}
using namespace ax;

extern "C" int auxssm_normalize_resample(auxssm_handle h, int dtype, int32_t rows, int32_t N, const void* log_weights,
                                         const void* weights, const void* uniforms, void* weights_out, int32_t* indices) {
    if (!h) {
        set_error("handle is NULL");
        return AUXSSM_ERR_ARG;
    }
    AX_HIP(hipSetDevice(h->device));
    ++h->api_calls;
    if (dtype != AUXSSM_F32 && dtype != AUXSSM_F64) {
        set_error("dtype must be 0 (f32) or 1 (f64)");
        return AUXSSM_ERR_ARG;
    }
    if (rows < 1 || N < 1 || N > 1024) {
        set_error("need rows >= 1 and 1 <= N <= 1024");
        return AUXSSM_ERR_ARG;
    }
    if ((!log_weights) == (!weights)) {
        set_error("give exactly one of log_weights / weights");
        return AUXSSM_ERR_ARG;
    }
    if (indices && !uniforms) {
        set_error("indices need uniforms");
        return AUXSSM_ERR_ARG;
    }
    const int TB = (N + 63) / 64 * 64;
    if (dtype == AUXSSM_F32)
        hipLaunchKernelGGL((k_normalize_resample<float>), dim3(rows), dim3(TB), (size_t)TB * 4 + 48 * 4 + 64, h->stream, N,
                           (const float*)log_weights, (const float*)weights, (const float*)uniforms, (float*)weights_out, indices);
    else
        hipLaunchKernelGGL((k_normalize_resample<double>), dim3(rows), dim3(TB), (size_t)TB * 8 + 48 * 8 + 64, h->stream, N,
                           (const double*)log_weights, (const double*)weights, (const double*)uniforms, (double*)weights_out, indices);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

extern "C" int auxssm_systematic_resample(auxssm_handle h, int dtype, int32_t rows, int32_t M, int32_t N, const void* weights, const void* uvw,
                                          int32_t* indices) {
    if (!h) {
        set_error("handle is NULL");
        return AUXSSM_ERR_ARG;
    }
    AX_HIP(hipSetDevice(h->device));
    ++h->api_calls;
    if (dtype != AUXSSM_F32 && dtype != AUXSSM_F64) {
        set_error("dtype must be 0 (f32) or 1 (f64)");
        return AUXSSM_ERR_ARG;
    }
    if (rows < 1 || M < 1 || M > 1024 || N < 1 || N > 1024) {
        set_error("need rows >= 1, 1 <= M <= 1024 weights and 1 <= N <= 1024 draws");
        return AUXSSM_ERR_ARG;
    }
    if (!weights || !uvw || !indices) {
        set_error("weights/uvw/indices must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    const int TB = ((M > N ? M : N) + 63) / 64 * 64;
    const size_t sR = dtype == AUXSSM_F32 ? 4 : 8;
    const size_t lds = (size_t)TB * sR + 48 * sR + (size_t)TB * 4 + 64;
    if (dtype == AUXSSM_F32)
        hipLaunchKernelGGL((k_systematic<float>), dim3(rows), dim3(TB), lds, h->stream, M, N, (const float*)weights, (const float*)uvw, indices);
    else
        hipLaunchKernelGGL((k_systematic<double>), dim3(rows), dim3(TB), lds, h->stream, M, N, (const double*)weights, (const double*)uvw, indices);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

extern "C" int auxssm_csmc_sweep(auxssm_handle h, int dtype, const auxssm_fk_model* fk, int32_t C, int32_t T, int32_t N,
                                 int32_t backward, const void* sqrt_half_delta, void* x, const auxssm_csmc_noise* noise,
                                 int32_t* ancestors, void* xs_out, void* log_ws_out, int32_t* As_out) {
    if (!h) {
        set_error("handle is NULL");
        return AUXSSM_ERR_ARG;
    }
    AX_HIP(hipSetDevice(h->device));
    ++h->api_calls;
    if (dtype != AUXSSM_F32 && dtype != AUXSSM_F64) {
        set_error("dtype must be 0 (f32) or 1 (f64)");
        return AUXSSM_ERR_ARG;
    }
    if (!fk || !x || !noise || !ancestors) {
        set_error("model/x/noise/ancestors must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    if (C < 1 || T < 1 || N < 2 || N > 1024) {
        set_error("need C >= 1, T >= 1, 2 <= N <= 1024 (got C=%d T=%d N=%d)", C, T, N);
        return AUXSSM_ERR_ARG;
    }
    const int D = fk->dx;
    const bool wide = D > CS_MAXD;  // csmc_wide.hip: one wave per chain, particles' components in LDS rows
    if (D < 1 || D > 32) {
        set_error("dx=%d: the cSMC kernels cover 1 <= dx <= 32", D);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    if (wide && (N > 64 || fk->transition != AUXSSM_TRANS_LINEAR)) {
        set_error("dx=%d runs the wide-state cSMC kernels: N <= 64 particles, linear-Gaussian transitions", D);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    if (fk->proposal != AUXSSM_PROP_BOOTSTRAP_LG && fk->proposal != AUXSSM_PROP_AUX_INDEPENDENT) {
        set_error("unknown proposal kind %d", fk->proposal);
        return AUXSSM_ERR_ARG;
    }
    if (fk->potential < AUXSSM_POT_FLAT || fk->potential > AUXSSM_POT_GAUSS_OBS_MASKED) {
        set_error("unknown potential kind %d", fk->potential);
        return AUXSSM_ERR_ARG;
    }
    if (!fk->m0 || !fk->chol_P0 || !fk->F || !fk->b || !fk->chol_Q) {
        set_error("model has a NULL m0/chol_P0/F/b/chol_Q host pointer");
        return AUXSSM_ERR_ARG;
    }
    if (fk->potential != AUXSSM_POT_FLAT && !fk->y) {
        set_error("potential needs observations y");
        return AUXSSM_ERR_ARG;
    }
    if (fk->potential == AUXSSM_POT_GAUSS_OBS && !(fk->sig_y > 0)) {
        set_error("sig_y must be > 0");
        return AUXSSM_ERR_ARG;
    }
    if (fk->proposal == AUXSSM_PROP_AUX_INDEPENDENT && !sqrt_half_delta) {
        set_error("the auxiliary proposal needs sqrt_half_delta (T)");
        return AUXSSM_ERR_ARG;
    }
    {
        const int ntv = (fk->F_t != nullptr) + (fk->b_t != nullptr) + (fk->chol_Q_t != nullptr);
        if (ntv != 0 && ntv != 3) {
            set_error("time-varying transitions need F_t, b_t and chol_Q_t together");
            return AUXSSM_ERR_ARG;
        }
        if (ntv && fk->transition != AUXSSM_TRANS_LINEAR) {
            set_error("time-varying parameters are for the linear transition only");
            return AUXSSM_ERR_ARG;
        }
    }
    if (fk->gradient != AUXSSM_GRAD_NONE && fk->gradient != AUXSSM_GRAD_REFERENCE && fk->gradient != AUXSSM_GRAD_EXACT) {
        set_error("unknown gradient mode %d", fk->gradient);
        return AUXSSM_ERR_ARG;
    }
    if (fk->gradient != AUXSSM_GRAD_NONE && fk->proposal != AUXSSM_PROP_AUX_INDEPENDENT) {
        set_error("gradient-informed proposals belong to AUXSSM_PROP_AUX_INDEPENDENT");
        return AUXSSM_ERR_ARG;
    }
    if (noise->mode == AUXSSM_NOISE_EXPLICIT) {
        if (!noise->eps_prop || !noise->u_bwd || (T > 1 && !noise->u_res) ||
            (fk->proposal == AUXSSM_PROP_AUX_INDEPENDENT && !noise->eps_aux)) {
            set_error("explicit noise needs eps_prop, u_res, u_bwd (and eps_aux for the auxiliary proposal)");
            return AUXSSM_ERR_ARG;
        }
    } else if (noise->mode != AUXSSM_NOISE_THREEFRY) {
        set_error("unknown noise mode %d", noise->mode);
        return AUXSSM_ERR_ARG;
    }
    // host-side model parameters (doubles): m0 | chol_P0 | F | b | chol_Q
    std::vector<double> hm((size_t)2 * D + 3 * D * D);
    {
        double* p = hm.data();
        memcpy(p, fk->m0, D * sizeof(double)); p += D;
        memcpy(p, fk->chol_P0, D * D * sizeof(double)); p += D * D;
        memcpy(p, fk->F, D * D * sizeof(double)); p += D * D;
        memcpy(p, fk->b, D * sizeof(double)); p += D;
        memcpy(p, fk->chol_Q, D * D * sizeof(double));
    }
    const size_t sR = dtype == AUXSSM_F32 ? 4 : 8;
    const size_t CT = (size_t)C * T;
    // the particle systems (xs, lws, As) dominate: T N (D + 1) reals per chain.  When those of all C chains do not fit what the device has free
    // (counting the handle's current workspace, which a larger reservation replaces), the sweep runs in batches of cb chains (CsmcArgs::c0)
    const size_t xs_rec = xs_out ? 0 : (size_t)T * N * D * sR, lws_rec = log_ws_out ? 0 : (size_t)T * N * sR;
    const size_t As_rec = (!backward && !As_out) ? (size_t)(T > 1 ? T - 1 : 1) * N * 4 : 0;
    const size_t big = xs_rec + lws_rec + As_rec;
    int cb = C;
    if (big) {
        const size_t small = 4096 + 8 * 256 + (size_t)C * N * sR + CT * sR + (size_t)T * (2 + D) * sR + 2 * CT * D * sR;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            const double budget = 0.8 * (double)(fr + h->ws_bytes);  // (ws_reserve adds an eighth)
            if ((double)small + (double)C * (double)big > budget) {
                const double fit = (budget - (double)small) / (double)big;
                if (fit < 1.0) {
                    set_error("one chain's particle system (%zu bytes) does not fit the device (%zu free)", big, fr + h->ws_bytes);
                    return AUXSSM_ERR_NOMEM;
                }
                if (fit < (double)cb) cb = (int)fit;
                if (cb > h->num_cu) cb -= cb % h->num_cu;  // one workgroup per chain and CU: whole rounds of the chip per batch
            }
        }
        if (const char* ev = getenv("AUXSSM_CSMC_BATCH")) {  // tests: force small batches
            const int v = atoi(ev);
            if (v >= 1 && v < cb) cb = v;
        }
    }
    const size_t CBT = (size_t)cb * T;
    size_t need = 4096 + ((size_t)3 * D * D + 4 * D + 8) * sR + 256;  // (+ the wide kernels' model block)
    if (!xs_out) need += CBT * N * D * sR + 256;
    if (!log_ws_out) need += CBT * N * sR + 256;
    if (!backward && !As_out) need += (size_t)cb * (T > 1 ? T - 1 : 1) * N * 4 + 256;
    need += (size_t)C * N * sR + 256;
    need += 2 * (CT * sR + 256);  // fmax, the backward pass's uniforms
    need += (size_t)T * sR + 256;  // gb
    need += 2 * (CT * D * sR + 256) + (size_t)T * (1 + D) * sR + 256;
    // fewer chains than CUs: the forward pass's draws are generated up front by the whole chip (csmc_dev.h::k_csmc_pregen) when the two arrays fit
    const size_t pre_eps = CT * N * D * sR + 256, pre_u = (size_t)C * (T > 1 ? T - 1 : 1) * N * sR + 256;
    bool pregen = noise->mode == AUXSSM_NOISE_THREEFRY && !wide && T > 1 && C < h->num_cu && cb == C && !getenv("AUXSSM_CSMC_NO_PREGEN");
    if (pregen) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || (double)(need + pre_eps + pre_u) > 0.7 * (double)(fr + h->ws_bytes)) pregen = false;
    }
    if (pregen) need += pre_eps + pre_u;
    int rc = ws_reserve(h, need);
    if (rc) return rc;
    CsmcArgs a;
    a.C = C; a.T = T; a.N = N; a.backward = backward ? 1 : 0;
    a.y = fk->y;
    a.shd = sqrt_half_delta;
    a.x = x;
    a.u = ws_take(h, CT * D * sR);
    a.grad = fk->gradient ? ws_take(h, CT * D * sR) : nullptr;
    void* ctt = fk->F_t ? ws_take(h, (size_t)T * (1 + D) * sR) : nullptr;  // constants + reciprocal diagonals of the T - 1 transitions
    a.cb = cb; a.xs_rec = xs_rec; a.lws_rec = lws_rec; a.As_rec = As_rec;
    a.xs = xs_out ? xs_out : ws_take(h, CBT * N * D * sR);
    a.lws = log_ws_out ? log_ws_out : ws_take(h, CBT * N * sR);
    a.As = As_out ? As_out : (!backward ? (int32_t*)ws_take(h, (size_t)cb * (T > 1 ? T - 1 : 1) * N * 4) : nullptr);
    a.wT = ws_take(h, (size_t)C * N * sR);
    a.fmax = ws_take(h, (size_t)C * T * sR);
    static const bool bound_on = !getenv("AUXSSM_CSMC_NO_BOUND");
    a.gb = (bound_on && (fk->potential == 0 || fk->y)) ? ws_take(h, (size_t)T * sR) : nullptr;
    a.anc = ancestors;
    a.noise_mode = noise->mode;
    a.key0 = noise->key0; a.key1 = noise->key1;
    a.eps_aux = noise->eps_aux; a.eps_prop = noise->eps_prop; a.u_res = noise->u_res; a.u_bwd = noise->u_bwd;
    if (noise->mode == AUXSSM_NOISE_THREEFRY) {  // the backward pass's uniforms, drawn once (csmc_dev.h::k_csmc_ubwd)
        void* ub = ws_take(h, CT * sR);
        if (!ub) return AUXSSM_ERR_NOMEM;
        const long long n = (long long)CT;
        if (dtype == AUXSSM_F32) hipLaunchKernelGGL((k_csmc_ubwd<float>), dim3((unsigned)(((n + 1) / 2 + 255) / 256)), dim3(256), 0, h->stream, n, noise->key0, noise->key1, (float*)ub);
        else hipLaunchKernelGGL((k_csmc_ubwd<double>), dim3((unsigned)(((n + 1) / 2 + 255) / 256)), dim3(256), 0, h->stream, n, noise->key0, noise->key1, (double*)ub);
        a.u_bwd = ub;
    }
    if (!a.u || !a.xs || !a.lws || !a.wT || !a.fmax || (!backward && !a.As) || (fk->gradient && !a.grad) || (fk->F_t && !ctt)) return AUXSSM_ERR_NOMEM;
    if (pregen) {
        void* pe = ws_take(h, pre_eps - 256);
        void* pu = ws_take(h, pre_u - 256);
        if (!pe || !pu) return AUXSSM_ERR_NOMEM;
        const int T2 = (T + 1) >> 1;
        const dim3 grid((unsigned)C * T2, (unsigned)((N * D + 255) / 256));
        ProfScope ps(h, AUXSSM_K_RNG);
        if (dtype == AUXSSM_F32) hipLaunchKernelGGL((k_csmc_pregen<float>), grid, dim3(256), 0, h->stream, T, N, D, noise->key0, noise->key1, (float*)pe, (float*)pu);
        else hipLaunchKernelGGL((k_csmc_pregen<double>), grid, dim3(256), 0, h->stream, T, N, D, noise->key0, noise->key1, (double*)pe, (double*)pu);
        a.pregen = 1;
        a.eps_prop = pe;
        a.u_res = pu;
    }
    if (wide) return run_csmc_wide(h, dtype, fk, a, ctt);
#define AX_CSMC_D(R)                                                        \
    switch (D) {                                                            \
        case 1: return run_csmc<R, 1>(h, fk, hm.data(), a, ctt);                 \
        case 2: return run_csmc<R, 2>(h, fk, hm.data(), a, ctt);                 \
        case 3: return run_csmc<R, 3>(h, fk, hm.data(), a, ctt);                 \
        default: return run_csmc<R, 4>(h, fk, hm.data(), a, ctt);                \
    }
    if (dtype == AUXSSM_F32) { AX_CSMC_D(float) } else { AX_CSMC_D(double) }
#undef AX_CSMC_D
}

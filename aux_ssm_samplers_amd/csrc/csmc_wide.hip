// csmc_wide.hip -- the conditional-SMC sweep for WIDE states with FEW particles: 4 < dx <= 32, N <= 64 (the reference's own timed stochastic-volatility
// protocol is D = 30, N = 25, T = 250: examples/stochastic_volatility/experiment.sh:1-10, experiment.py:38-55, auxiliary_csmc.py:14-46).
//
// Same algorithm, same sweep contract (csmc_dev.h: unnormalised weights shifted by a bound or the exact maximum, DPP-order cumsum, descent search,
// ballot-counted single draw, reciprocal Cholesky diagonals, det_exp / det_log, explicit fma) and the same oracle (oracle/csmc_ref.c, MAXD = 32) as the
// register kernels of csmc.hip -- what changes is where a particle lives: ONE WAVE per chain, one lane per particle, the particle's dx components in an
// LDS row of odd stride (the register kernels keep them in registers, which stops at dx = 4: R x[D], eps[D], mu[D], z[D] ... and the model BY VALUE in the
// kernel arguments).  The model's matrices (F, chol Q: dx x dx) are staged in LDS once per workgroup and read at wave-uniform addresses (broadcasts).
// Per step and particle: the parent's mean is a dx x dx mat-vec, the transition density a forward substitution -- O(dx^2) fused multiply-adds in the
// contract's fixed order.  A single wave needs no workgroup barrier at all: the cumulative weights come out of one DPP scan, the group total is a
// readlane, the backward draw a ballot.  Linear-Gaussian transitions, every potential / proposal of the family; no time-varying rows, no gradient
// proposals (AUXSSM_ERR_UNSUPPORTED).
//
// In-kernel draws (AUXSSM_NOISE_THREEFRY) use the NATURAL flat indices of the explicit arrays -- eps_prop[c][t][n][k] = normal ((c T + t) N + n) dx + k of
// stream 2, u_res[c][s][n] = uniform (c (T-1) + s) N + n of stream 3, u_bwd[c][t] = uniform c T + t of stream 4 -- not the two-steps-per-block packing
// of the register kernels (csmc/_device.py::key_noise(wide=True) builds the equivalent arrays).
#include "csmc_dev.h"

namespace ax {

constexpr int CSW_MAXD = 32;

template <typename R> struct FkW {
    int proposal, potential, D;
    const R *m0, *LP0, *iLP0, *F, *b, *LQ, *iLQ;  // device arrays, matrices row-major with leading dimension D
    R c_init, c_trans, c_obs, inv_sig_y;
};

// g_t(x) for a particle row in LDS (csmc_dev.h::potential with a runtime dimension, same operations)
template <typename R> __device__ __forceinline__ R potential_w(const FkW<R>& m, const R* x, const R* y) {
    const int D = m.D;
    if (m.potential == 0) return (R)0;
    if (m.potential == 1) {
        R q = 0;
        for (int k = 0; k < D; ++k) {
            const R z = (y[k] - x[k]) * m.inv_sig_y;
            q = fma_(z, z, q);
        }
        return fma_((R)-0.5, q, m.c_obs);
    }
    if (m.potential == 3) {
        R q = 0;
        int nobs = 0;
        for (int k = 0; k < D; ++k) {
            if (y[k] - y[k] == 0) {
                const R z = (y[k] - x[k]) * m.inv_sig_y;
                q = fma_(z, z, q);
                ++nobs;
            }
        }
        return fma_((R)-0.5, q, (R)nobs * m.c_obs);
    }
    R acc = 0;
    for (int k = 0; k < D; ++k) {
        const R e = det_exp(-x[k]);
        const R s = fma_(y[k] * y[k], e, x[k]);
        const R v = fma_((R)-0.5, s, m.c_obs);
        acc += (v == v) ? v : (R)0;
    }
    return acc;
}
// mu = F xp + b into an LDS row (csmc_dev.h::trans_mean, linear)
template <typename R> __device__ __forceinline__ void trans_mean_w(int D, const R* F, const R* b, const R* xp, R* mu) {
    for (int k = 0; k < D; ++k) {
        R acc = b[k];
        for (int j = 0; j < D; ++j) acc = fma_(F[k * D + j], xp[j], acc);
        mu[k] = acc;
    }
}
// log N(x; mean, L L^T) by forward substitution, z kept in an LDS row (csmc_dev.h::gauss_chol_logpdf)
template <typename R> __device__ __forceinline__ R gauss_w(int D, const R* x, const R* mean, const R* L, const R* iL, R cst, R* z) {
    R q = 0;
    for (int k = 0; k < D; ++k) {
        R acc = x[k] - mean[k];
        for (int j = 0; j < k; ++j) acc = fma_(-L[k * D + j], z[j], acc);
        const R zk = acc * iL[k];
        z[k] = zk;
        q = fma_(zk, zk, q);
    }
    return fma_((R)-0.5, q, cst);
}
// e_i = exp(lw_i - max lw) over ONE wave (csmc_dev.h::block_expmax)
template <typename R> __device__ __forceinline__ R wave_expmax(R lw, R* m_out) {
    R m = wave_max_dpp(lw);
    if (!(m - m == 0)) m = 0;
    if (m_out) *m_out = m;
    return det_exp(lw - m);
}
// the descent search of the sweep contract on a wave's cumulative weights held in LDS (unpadded: at most 64 entries)
template <typename R> __device__ __forceinline__ int search_w(const R* c, int N, R r) {
    int s0 = 1;
    while (s0 * 2 < N) s0 *= 2;
    int pos = 0;
    for (int s = s0; s > 0; s >>= 1) {
        const int q = pos + s - 1;
        pos += (q < N && c[q < N ? q : N - 1] < r) ? s : 0;
    }
    return pos < N - 1 ? pos : N - 1;
}

template <typename R> __global__ void k_cw_potbound(int T, FkW<R> m, const R* __restrict__ y, R* __restrict__ gb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int D = m.D;
    R b = 0;
    if (m.potential == 1) b = m.c_obs;
    else if (m.potential == 3) {
        int nobs = 0;
        for (int k = 0; k < D; ++k) nobs += (y[(long long)t * D + k] - y[(long long)t * D + k] == 0) ? 1 : 0;
        b = (R)nobs * m.c_obs;
    } else if (m.potential == 2) {
        for (int k = 0; k < D; ++k) {
            const R yk = y[(long long)t * D + k], y2 = yk * yk;
            R v = (R)0;
            if (y2 - y2 == 0) v = y2 > (R)0 ? fma_((R)-0.5, (R)1 + det_log(y2), m.c_obs) : (R)INFINITY;
            b += v > (R)0 ? v : (R)0;
        }
    }
    gb[t] = b;
}

// LDS plan of both passes: [F D*D | LQ D*D | b D | iLQ D | c 64 | rows ...], rows of stride S = D | 1 (odd) per lane
template <typename R> struct CwLds {
    int D, S;
    R *F, *LQ, *b, *iL, *c, *r0, *r1, *r2, *r3, *sh;
    __device__ CwLds(char* smem, int D_) : D(D_), S(D_ | 1) {
        F = (R*)smem;
        LQ = F + D * D;
        b = LQ + D * D;
        iL = b + D;
        c = iL + D;
        r0 = c + 64;
        r1 = r0 + 64 * S;
        r2 = r1 + 64 * S;
        r3 = r2 + 64 * S;
        sh = r3 + 64 * S;  // [D] one shared vector (backward pass: x_{t+1})
    }
    static size_t bytes(int D) { return ((size_t)2 * D * D + 2 * D + 64 + (size_t)4 * 64 * (D | 1) + D) * sizeof(R) + 64; }
};
template <typename R> __device__ __forceinline__ void cw_stage(const FkW<R>& m, CwLds<R>& L, int tid) {
    const int D = m.D;
    for (int i = tid; i < D * D; i += 64) L.F[i] = m.F[i], L.LQ[i] = m.LQ[i];
    for (int i = tid; i < D; i += 64) L.b[i] = m.b[i], L.iL[i] = m.iLQ[i];
    __syncthreads();
}

// ---- forward pass (_csmc, csmc.py:69-107) ---------------------------------------------------------------------------------------------------
template <typename R> __global__ void __launch_bounds__(64) k_cw_fwd(CsmcArgs a, FkW<R> m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, N = a.N, T = a.T, D = m.D;
    CwLds<R> L(smem, D);
    for (int k = tid; k < D; k += 64) L.sh[k] = 0;  // an all-zero observation row for the flat potential (published by the staging barrier)
    cw_stage<R>(m, L, tid);
    const int S = L.S;
    const int ch = a.c0 + blockIdx.x;
    const bool live = tid < N;
    const R* xstar = (const R*)a.x + (long long)ch * T * D;
    const R* uaux = (const R*)a.u + (long long)ch * T * D;
    const R* yv = (const R*)a.y;
    R* xs = (R*)a.xs + (long long)ch * T * N * D;
    R* lws = (R*)a.lws + (long long)ch * T * N;
    int32_t* As = a.As ? a.As + (long long)ch * (T - 1) * N : nullptr;
    R* fmax = a.fmax ? (R*)a.fmax + (long long)ch * T : nullptr;
    const R ninf = -INFINITY;
    R* xc = L.r0 + tid * S;    // this step's particle
    R* xpb = L.r1;             // last step's particles (all lanes), row i at xpb + i S
    R* wk = L.r2 + tid * S;    // noise of the step, then the substitution's z
    R* mu = L.r3 + tid * S;    // the parent's transition mean
    auto yrow = [&](int t) -> const R* { return yv ? yv + (long long)t * D : L.sh; };
    auto draw_eps = [&](int t) {
        for (int k = 0; k < D; ++k) {
            const long long idx = (((long long)ch * T + t) * N + tid) * D + k;
            wk[k] = live ? noise_normal<R>(a, a.eps_prop, STREAM_EPS_PROP, idx) : (R)0;
        }
    };
    // t = 0 (csmc.py:74-80)
    draw_eps(0);
    if (m.proposal == 0) {
        for (int k = 0; k < D; ++k) {
            R acc = m.m0[k];
            for (int j = 0; j <= k; ++j) acc = fma_(m.LP0[k * D + j], wk[j], acc);
            xc[k] = acc;
        }
    } else {
        const R s0 = ((const R*)a.shd)[0];
        for (int k = 0; k < D; ++k) xc[k] = fma_(s0, wk[k], uaux[k]);
    }
    if (tid == 0)
        for (int k = 0; k < D; ++k) xc[k] = xstar[k];
    R lw;
    {
        R g = potential_w<R>(m, xc, yrow(0));
        if (m.proposal == 1) g = g + gauss_w<R>(D, xc, m.m0, m.LP0, m.iLP0, m.c_init, wk);
        lw = live ? g : ninf;
    }
    if (live) {
        for (int k = 0; k < D; ++k) xs[(long long)tid * D + k] = xc[k];
        lws[tid] = lw;
    }
    R mstep;
    R w = wave_expmax<R>(lw, &mstep);
    if (fmax && tid == 0) fmax[0] = mstep;
    const R* gbp = (const R*)a.gb;
    const bool bmode = gbp != nullptr;
    bool used_bound = false;
    for (int t = 1; t < T; ++t) {
        // publish the last step's particles, draw this step's noise
        for (int k = 0; k < D; ++k) xpb[tid * S + k] = xc[k];
        const R un = (live && tid > 0) ? noise_uniform<R>(a, a.u_res, STREAM_U_RES, ((long long)ch * (T - 1) + (t - 1)) * N + tid) : (R)0;
        draw_eps(t);
        // conditional multinomial resampling (resamplings.py:14-37): one wave = one group of the contract's cumsum
        R cv = wave_scan_dpp(w);
        R tot = readlane_(cv, 63);
        if (used_bound && !(tot > (R)0)) {  // every weight underflowed under its bound: the exact maximum after all
            w = wave_expmax<R>(lw, &mstep);
            if (fmax && tid == 0) fmax[t - 1] = mstep;
            cv = wave_scan_dpp(w);
            tot = readlane_(cv, 63);
        }
        L.c[tid] = cv;
        __syncthreads();  // (one wave: orders the LDS writes above before the reads below)
        int idx = 0;
        if (live && tid > 0) idx = search_w<R>(L.c, N, tot * ((R)1 - un));
        const R* xp = xpb + idx * S;
        // propagate (csmc.py:91-92)
        if (m.proposal == 0) {
            trans_mean_w<R>(D, L.F, L.b, xp, mu);
            for (int k = 0; k < D; ++k) {
                R acc = mu[k];
                for (int j = 0; j <= k; ++j) acc = fma_(L.LQ[k * D + j], wk[j], acc);
                xc[k] = acc;
            }
        } else {
            const R st = ((const R*)a.shd)[t];
            for (int k = 0; k < D; ++k) xc[k] = fma_(st, wk[k], uaux[(long long)t * D + k]);
        }
        if (tid == 0)
            for (int k = 0; k < D; ++k) xc[k] = xstar[(long long)t * D + k];
        // weights (csmc.py:95-96)
        R g = potential_w<R>(m, xc, yrow(t));
        if (m.proposal == 1) {  // AuxiliaryGt = Mt.logpdf + Gt (independent.py:238-248)
            trans_mean_w<R>(D, L.F, L.b, xp, mu);
            g = gauss_w<R>(D, xc, mu, L.LQ, L.iL, m.c_trans, wk) + g;
        }
        lw = live ? g : ninf;
        if (live) {
            const long long o = (long long)t * N + tid;
            for (int k = 0; k < D; ++k) xs[o * D + k] = xc[k];
            lws[o] = lw;
            if (As) As[(long long)(t - 1) * N + tid] = idx;
        }
        const R Mb = (bmode ? gbp[t] : (R)0) + (m.proposal == 1 ? m.c_trans : (R)0);
        used_bound = bmode && t < T - 1 && (Mb - Mb == 0);
        if (used_bound) {
            w = det_exp(lw - Mb);
            mstep = Mb;
        } else {
            w = wave_expmax<R>(lw, &mstep);
        }
        if (fmax && tid == 0) fmax[t] = mstep;
        __syncthreads();  // (every lane is past its reads of xpb / c before the next step rewrites them)
    }
    if (live) ((R*)a.wT)[(long long)ch * N + tid] = w;
}

// ---- backward passes (csmc.py:110-149) --------------------------------------------------------------------------------------------------------
template <typename R> __global__ void __launch_bounds__(64) k_cw_bwd(CsmcArgs a, FkW<R> m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, N = a.N, T = a.T, D = m.D;
    CwLds<R> L(smem, D);
    cw_stage<R>(m, L, tid);
    const int S = L.S;
    const int ch = a.c0 + blockIdx.x;
    const bool live = tid < N;
    const R* xs = (const R*)a.xs + (long long)ch * T * N * D;
    const R* lws = (const R*)a.lws + (long long)ch * T * N;
    const int32_t* As = a.As ? a.As + (long long)ch * (T - 1) * N : nullptr;
    R* xout = (R*)a.x + (long long)ch * T * D;
    int32_t* anc = a.anc + (long long)ch * T;
    const R* fmax = (const R*)a.fmax + (long long)ch * T;
    const R ninf = -INFINITY;
    R* xi = L.r0 + tid * S;
    R* zr = L.r1 + tid * S;
    R* mu = L.r2 + tid * S;
    R* xn = L.sh;  // x_{t+1}, shared by the wave
    auto count_below = [&](R cv, R r) -> int {
        const unsigned long long bal = __ballot(live && cv < r);
        const int B = __popcll(bal);
        return B < N - 1 ? B : N - 1;
    };
    // B_T ~ choice(w_T)
    int B;
    {
        const R w = live ? ((const R*)a.wT)[(long long)ch * N + tid] : (R)0;
        const R cv = wave_scan_dpp(w);
        const R tot = readlane_(cv, 63);
        const R un = ((const R*)a.u_bwd)[(long long)ch * T + (T - 1)];
        B = count_below(cv, tot * ((R)1 - un));
    }
    for (int k = tid; k < D; k += 64) {
        const R v = xs[((long long)(T - 1) * N + B) * D + k];
        xn[k] = v;
        xout[(long long)(T - 1) * D + k] = v;
    }
    if (tid == 0) anc[T - 1] = B;
    if (!a.backward) {
        if (tid == 0) {
            for (int t = T - 1; t >= 1; --t) {
                B = As[(long long)(t - 1) * N + B];
                for (int k = 0; k < D; ++k) xout[(long long)(t - 1) * D + k] = xs[((long long)(t - 1) * N + B) * D + k];
                anc[t - 1] = B;
            }
        }
        return;
    }
    __syncthreads();
    for (int t = T - 2; t >= 0; --t) {
        R lw = ninf;
        if (live) {
            for (int k = 0; k < D; ++k) xi[k] = xs[((long long)t * N + tid) * D + k];
            trans_mean_w<R>(D, L.F, L.b, xi, mu);
            lw = gauss_w<R>(D, xn, mu, L.LQ, L.iL, m.c_trans, zr) + lws[(long long)t * N + tid];  // Pt.logpdf(x_{t+1}, xs_t) + log_ws_t (csmc.py:136)
        }
        R Mb = fmax[t] + m.c_trans;
        if (!(Mb - Mb == 0)) Mb = 0;
        R w = det_exp(lw - Mb);
        R cv = wave_scan_dpp(w);
        R tot = readlane_(cv, 63);
        if (!(tot > (R)0)) {
            w = wave_expmax<R>(lw, nullptr);
            cv = wave_scan_dpp(w);
            tot = readlane_(cv, 63);
        }
        const R un = ((const R*)a.u_bwd)[(long long)ch * T + t];
        B = count_below(cv, tot * ((R)1 - un));
        __syncthreads();  // (every lane has read x_{t+1} before it is replaced)
        for (int k = tid; k < D; k += 64) {
            const R v = xs[((long long)t * N + B) * D + k];
            xn[k] = v;
            xout[(long long)t * D + k] = v;
        }
        if (tid == 0) anc[t] = B;
        __syncthreads();
    }
}

// host: the model as one device block [m0 | LP0 | iLP0 | F | b | LQ | iLQ], constants as csmc_dev.h::fill_model computes them
template <typename R> static int run_cw(auxssm_ctx* h, const auxssm_fk_model* fk, CsmcArgs& a, void* dev_block, R* host_block) {
    const int D = fk->dx;
    FkW<R> m;
    memset(&m, 0, sizeof(m));
    m.proposal = fk->proposal; m.potential = fk->potential; m.D = D;
    R* p = host_block;
    R* hm0 = p; p += D;
    R* hLP0 = p; p += D * D;
    R* hiLP0 = p; p += D;
    R* hF = p; p += D * D;
    R* hb = p; p += D;
    R* hLQ = p; p += D * D;
    R* hiLQ = p; p += D;
    for (int k = 0; k < D; ++k) hm0[k] = (R)fk->m0[k], hb[k] = (R)fk->b[k];
    for (int i = 0; i < D * D; ++i) hLP0[i] = (R)fk->chol_P0[i], hF[i] = (R)fk->F[i], hLQ[i] = (R)fk->chol_Q[i];
    R ci = 0, ct = 0;
    for (int k = 0; k < D; ++k) {
        ci -= det_log(hLP0[k * D + k]);
        ct -= det_log(hLQ[k * D + k]);
        hiLP0[k] = (R)1 / hLP0[k * D + k];
        hiLQ[k] = (R)1 / hLQ[k * D + k];
    }
    const R half_log_2pi = (R)0.91893853320467274178;
    m.c_init = ci - (R)D * half_log_2pi;
    m.c_trans = ct - (R)D * half_log_2pi;
    if (fk->potential == 1) {
        m.inv_sig_y = (R)1 / (R)fk->sig_y;
        m.c_obs = -(R)D * det_log((R)fk->sig_y) - (R)D * half_log_2pi;
    } else if (fk->potential == 3) {
        m.inv_sig_y = (R)1 / (R)fk->sig_y;
        m.c_obs = -det_log((R)fk->sig_y) - half_log_2pi;
    } else {
        m.inv_sig_y = 0;
        m.c_obs = -half_log_2pi;
    }
    const size_t nb = (size_t)(p - host_block) * sizeof(R);
    AX_HIP(hipMemcpyAsync(dev_block, host_block, nb, hipMemcpyHostToDevice, h->stream));
    AX_HIP(hipStreamSynchronize(h->stream));  // (the host block is the caller's stack vector)
    const R* d = (const R*)dev_block;
    m.m0 = d; d += D;
    m.LP0 = d; d += D * D;
    m.iLP0 = d; d += D;
    m.F = d; d += D * D;
    m.b = d; d += D;
    m.LQ = d; d += D * D;
    m.iLQ = d;
    if (a.gb) hipLaunchKernelGGL((k_cw_potbound<R>), dim3((a.T + 255) / 256), dim3(256), 0, h->stream, a.T, m, (const R*)a.y, (R*)a.gb);
    if (fk->proposal == 1) {
        const long long total = (long long)a.C * a.T * D;
        hipLaunchKernelGGL((k_csmc_aux<R>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, a, D);
    }
    const size_t lds = CwLds<R>::bytes(D);
    if (lds > 48 * 1024) {
        AX_HIP(hipFuncSetAttribute((const void*)k_cw_fwd<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AX_HIP(hipFuncSetAttribute((const void*)k_cw_bwd<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
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
            hipLaunchKernelGGL((k_cw_fwd<R>), dim3(ab.C), dim3(64), lds, h->stream, ab, m);
        }
        {
            ProfScope ps(h, AUXSSM_K_CSMC_BWD);
            hipLaunchKernelGGL((k_cw_bwd<R>), dim3(ab.C), dim3(64), lds, h->stream, ab, m);
        }
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// called by auxssm_csmc_sweep (csmc.hip) for dx > CS_MAXD; dev_block / host_block hold 2 dx^2 + ... reals (csmc_wide_block_reals)
int run_csmc_wide(auxssm_ctx* h, int dtype, const auxssm_fk_model* fk, CsmcArgs& a, void* dev_block) {
    const int D = fk->dx;
    std::vector<double> host((size_t)3 * D * D + 4 * D + 8);
    if (dtype == AUXSSM_F32) return run_cw<float>(h, fk, a, dev_block, (float*)host.data());
    return run_cw<double>(h, fk, a, dev_block, host.data());
}

}  // namespace ax

// pit.hip -- parallel-in-time conditional SMC (conditional dSMC, Corenflos et al. 2022): reference
// aux_samplers/_primitives/csmc/pit/csmc.py (_csmc :69-114), operator.py (stitching weights :125-149, resampling :74-84,
// _gather_results :87-110), dc_map.py (the divide-and-conquer tree :70-121), as used by csmc/independent.py:78-118
// (`get_kernel(..., parallel=True)`, classical branch: proposals N(u_t, delta_t/2 I) independent across time).
// Compiled with -ffp-contract=off; fused multiply-adds are explicit, so oracle/csmc_ref.c restates the arithmetic bit for bit.
//
// The reference gathers whole blocks of trajectories at every level of the tree (T N d values moved per level).  Here a tree
// node keeps only what later levels read -- for each of its N slots the LEAF particle index at the node's first and at its last
// time step -- plus the (left, right) slot pair each slot was stitched from.  The up-sweep is one launch per level, one workgroup
// per stitch; the single output trajectory (the last level draws one pair, operator.py:76-79) is then read off by walking the
// stored pairs down from the root, one thread per time step.  Memory O(T N), stitching work (T-1) N^2 as in the reference.
//
// Tree (dc_map.py): level k = 0..K-1 (2^K >= T) has nodes j covering [j 2^(k+1), (j+1) 2^(k+1)); the node stitches its children
// at mid = j 2^(k+1) + 2^k when mid < T, else it IS its left child (passthrough :93-105) and stores nothing.
// One stitch (operator.py): log W[i][j] = Gt(x_b[j], x_a[i]) + lw_a[i] + lw_b[j] over the N x N pairs of a's last-step and b's first-step
// particles, Gt = log transition + potential (independent.py:238-248); N conditional multinomial draws (pair 0 pinned to (0, 0),
// resamplings.py:36) from the N^2 flattened weights, or ONE unconditional draw at the root.
// Arithmetic contract (oracle/csmc_ref.c restates it):
//   v[p], p = i N + j:  (gauss_r(x_b[j]; mean(x_a[i])) + (potential_j + lw_b)) + lw_a[i];  gauss_r = the Cholesky log-density with the
//                       reciprocal diagonal of chol(Q) as a multiplier;
//   M = max v (exact); e[p] = exp(v[p] - M);  the N^2 values are cut into NCH consecutive chunks of Lc = ceil(N^2/NCH), NCH = 64 (N <= 32),
//   256 (N <= 128) or 1024 = the lanes of the stitch's workgroup; chunk sums are serial left to right, the cumsum over the NCH chunk sums is
//   the block cumsum of csmc_dev.h.  A chunk is cut again into SC = 8 consecutive sub-chunks of Ls = ceil(Lc/8): the chunk sum is the
//   left-to-right sum of its sub-chunk sums, each a serial sum from 0.  A draw r = total (1 - u) picks the first chunk with cumsum >= r,
//   inside it the first sub-chunk b with (previous chunks' cumsum + (S_0 + .. + S_b)) >= r (the last non-empty one if none), inside that
//   the first p with (previous chunks' cumsum + ((S_0 + .. + S_{b-1}) + e .. + e_p)) >= r (its last p if none): a draw recomputes one
//   sub-chunk, not the chunk.
#include "csmc_dev.h"

namespace ax {

struct PitArgs {
    int C, T, N, K;
    const void* y;    // (T, D) or null
    const void* shd;  // (T)
    void* x;          // (C, T, D) reference trajectory in, new trajectory out
    void* xs;         // (C, T, N, D) leaf particles
    void* lw0;        // (C, N) normalised log-weights of the leaf at t = 0
    // gradient-informed proposals (csmc/independent.py:81-84: mt = N(u + delta/2 grad, delta/2 I), qt = N(u, delta/2 I); pit/csmc.py:83-88: the leaf
    // weights are qt.logpdf - mt.logpdf, per particle): u, grad (C, T, D) and the normalised leaf log-weights of EVERY time step, lwt (C, T, N); null otherwise
    const void* u;
    const void* grad;
    void* lwt;
    uint16_t* Ls;     // (C, tot, N) left slot of each stitched pair, nodes of all levels back to back (off[k] = first node of level k)
    uint16_t* Rs;     // (C, tot, N) right slot
    uint16_t* Fi;     // (C, tot, N) leaf particle index at the node's first time step
    uint16_t* La;     // (C, tot, N) leaf particle index at the node's last time step
    int32_t* anc;     // (C, T)
    long long tot;
    long long off[32];
    double neg_log_n;
    int noise_mode;
    uint32_t key0, key1;
    const void* eps_aux;   // (C, T, D)
    const void* eps_prop;  // (C, T, N, D)
    const void* u_res;     // (C, T, N): row t feeds the stitch at the boundary (t-1 | t); row 0 is never read
};

template <typename R> __device__ __forceinline__ R pit_normal(const PitArgs& a, const void* arr, uint32_t stream, long long idx) {
    if (a.noise_mode == 0) return ((const R*)arr)[idx];
    return stream_normal<R>(a.key0, a.key1, stream, (unsigned long long)idx);
}
template <typename R> __device__ __forceinline__ R pit_uniform(const PitArgs& a, const void* arr, uint32_t stream, long long idx) {
    if (a.noise_mode == 0) return ((const R*)arr)[idx];
    return stream_uniform<R>(a.key0, a.key1, stream, (unsigned long long)idx);
}

// log(w / sum w) of a block's log-weights, the reductions of block_normalize
template <typename R> __device__ __forceinline__ R block_lognormalize(R lw, R* red, int tid, int nw) {
    const int lane = tid & 63, wv = tid >> 6;
    R m = wave_max(lw);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    R t[16];
    load16<R>(red, t);
    m = t[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = (k < nw && t[k] > m) ? t[k] : m;
    if (!(m - m == 0)) m = 0;
    const R e = det_exp(lw - m);
    R s = wave_sum_tree(e);
    if (lane == 0) red[16 + wv] = s;
    __syncthreads();
    load16<R>(red + 16, t);
    s = t[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) s = k < nw ? s + t[k] : s;
    return lw - (det_log(s) + m);
}

// leaves: u = x + sqrt(delta/2) eps_aux (csmc/independent.py:101-102); particles ~ N(u_t, delta_t/2 I), slot 0 = x (pit/csmc.py:77-80);
// weights 0 except G0 at t = 0, normalised per time step (:85-91).  One workgroup per (t, chain), one lane per particle.
template <typename R, int D, bool GRAD> __global__ void __launch_bounds__(1024) k_pit_leaves(PitArgs a, FkDev<R> m) {
    __shared__ R red[48];
    const int t = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, N = a.N, T = a.T;
    const bool live = tid < N;
    const long long ct = (long long)c * T + t;
    const R sh = ((const R*)a.shd)[t];
    R x[D], uu[D], pm[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R xr = ((const R*)a.x)[ct * D + k];
        if constexpr (GRAD) {  // u was formed by k_csmc_aux (the gradient kernel reads it), the proposal mean is shifted by delta_t / 2 grad_t
            uu[k] = ((const R*)a.u)[ct * D + k];
            pm[k] = fma_(sh * sh, ((const R*)a.grad)[ct * D + k], uu[k]);
        } else {
            uu[k] = fma_(sh, pit_normal<R>(a, a.eps_aux, STREAM_EPS_AUX, ct * D + k), xr);
            pm[k] = uu[k];
        }
        const R e = live ? pit_normal<R>(a, a.eps_prop, STREAM_EPS_PROP, (ct * N + tid) * D + k) : (R)0;
        x[k] = tid == 0 ? xr : fma_(sh, e, pm[k]);
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < D; ++k) ((R*)a.xs)[(ct * N + tid) * D + k] = x[k];
    }
    if (!GRAD && t != 0) return;
    R g = 0;
    if constexpr (GRAD) g = grad_correction<R, D>(x, uu, pm, sh);  // qt.logpdf(x) - mt.logpdf(x) (pit/csmc.py:84-85), slot 0 included
    if (t == 0) {
        R y0[D];
#pragma unroll
        for (int k = 0; k < D; ++k) y0[k] = a.y ? ((const R*)a.y)[k] : (R)0;
        R g0 = potential<R, D>(m, x, y0);
        g0 = g0 + gauss_chol_logpdf<R, D>(x, m.m0, m.LP0, m.iLP0, m.c_init);  // AuxiliaryG0 (independent.py:163-169)
        g = GRAD ? g + g0 : g0;                                               // log_wts.at[0].add(log_w0) (pit/csmc.py:90-91)
    }
    const R lw = block_lognormalize<R>(live ? g : (R)-INFINITY, red, tid, (N + 63) >> 6);
    if (live) {
        if constexpr (GRAD) ((R*)a.lwt)[ct * N + tid] = lw;
        else ((R*)a.lw0)[(long long)c * N + tid] = lw;
    }
}

constexpr int PIT_SC = 8;  // sub-chunks per chunk (arithmetic contract, see the header)

// the stitch of node j at level k (see the header); grid (nodes of the level, chains), NCH lanes
template <typename R, int D> __global__ void __launch_bounds__(1024) k_pit_stitch(PitArgs a, FkDev<R> m, int k) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int j = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, N = a.N, T = a.T;
    const long long s0 = (long long)j << (k + 1), mid = s0 + (1ll << k);
    if (mid >= T) return;  // passthrough node (uniform per workgroup)
    R* xb = (R*)smem_raw;     // [N][D] first-step particles of the right block
    R* mu = xb + N * D;       // [N][D] transition means of the last-step particles of the left block
    R* pg = mu + N * D;       // [N] potential + lw_b
    R* hh = pg + N;           // [N] lw_a
    R* cs = hh + N;           // [NCH] cumsum of the chunk sums
    const int NCH = blockDim.x, nw = NCH >> 6;
    R* red = cs + NCH;        // [48]
    R* sub = red + 48;        // [PIT_SC][NCH] sub-chunk sums
    const bool root = k == a.K - 1;
    const long long chain_nodes = (long long)c * a.tot;
    // children: left = (k-1, 2j), complete; right = (k-1, 2j+1), resolved through passthrough nodes down to a stitched node or a leaf
    int kb = k - 1;
    long long jb = 2ll * j + 1;
    while (kb >= 0 && (jb << (kb + 1)) + (1ll << kb) >= T) {
        jb <<= 1;
        --kb;
    }
    const uint16_t* la_left = k > 0 ? a.La + (chain_nodes + a.off[k - 1] + 2ll * j) * N : nullptr;
    const uint16_t* fi_left = k > 0 ? a.Fi + (chain_nodes + a.off[k - 1] + 2ll * j) * N : nullptr;
    const uint16_t* fi_right = kb >= 0 ? a.Fi + (chain_nodes + a.off[kb] + jb) * N : nullptr;
    const uint16_t* la_right = kb >= 0 ? a.La + (chain_nodes + a.off[kb] + jb) * N : nullptr;
    const R nln = (R)a.neg_log_n;
    const TransT<R> tr = trans_at<R, D>(m, mid - 1);  // the transition across the boundary (time-varying: row mid - 1 of the device arrays)
    if (tid < N) {
        const int ia = la_left ? la_left[tid] : tid;
        const int ib = fi_right ? fi_right[tid] : tid;
        const R* xa = (const R*)a.xs + (((long long)c * T + mid - 1) * N + ia) * D;
        const R* xv = (const R*)a.xs + (((long long)c * T + mid) * N + ib) * D;
        R xr[D], xl[D], mm[D], yv[D];
#pragma unroll
        for (int q = 0; q < D; ++q) {
            xl[q] = xa[q];
            xr[q] = xv[q];
            yv[q] = a.y ? ((const R*)a.y)[mid * D + q] : (R)0;
        }
        trans_mean_t<R, D>(m, tr, xl, mm);
#pragma unroll
        for (int q = 0; q < D; ++q) {
            mu[tid * D + q] = mm[q];
            xb[tid * D + q] = xr[q];
        }
        // the log-weights the two blocks bring: -log N once a block has been stitched (operator.py:106-108), the LEAF's own while it is a single time step --
        // the left block at level 0, the right one when no level below stitched it (kb < 0)
        R wl = nln, wr = nln;
        if (a.lwt) {
            if (k == 0) wl = ((const R*)a.lwt)[((long long)c * T + mid - 1) * N + ia];
            if (kb < 0) wr = ((const R*)a.lwt)[((long long)c * T + mid) * N + ib];
        } else if (mid == 1) {
            wl = ((const R*)a.lw0)[(long long)c * N + ia];
        }
        pg[tid] = potential<R, D>(m, xr, yv) + wr;
        hh[tid] = wl;
    }
    __syncthreads();
    const long long NN = (long long)N * N;
    const int Lc = (int)((NN + NCH - 1) / NCH);
    const long long p0 = (long long)tid * Lc;
    const long long p1 = p0 + Lc < NN ? p0 + Lc : NN;
    auto value = [&](int i, int jj) -> R {
        R xj[D], mi[D];
#pragma unroll
        for (int q = 0; q < D; ++q) {
            xj[q] = xb[jj * D + q];
            mi[q] = mu[i * D + q];
        }
        return (gauss_chol_logpdf<R, D>(xj, mi, tr.LQ, tr.iL, tr.c_trans, tr.ld) + pg[jj]) + hh[i];
    };
    // pass 1: max
    R vmax = -INFINITY;
    {
        int i = (int)(p0 / N), jj = (int)(p0 - (long long)i * N);
        for (long long p = p0; p < p1; ++p) {
            const R v = value(i, jj);
            vmax = v > vmax ? v : vmax;
            if (++jj == N) jj = 0, ++i;
        }
    }
    {
        const int lane = tid & 63, wv = tid >> 6;
        const R wm = wave_max(vmax);
        if (lane == 0) red[wv] = wm;
        __syncthreads();
        R t16[16];
        load16<R>(red, t16);
        vmax = t16[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) vmax = (q < nw && t16[q] > vmax) ? t16[q] : vmax;
        if (!(vmax - vmax == 0)) vmax = 0;
    }
    // pass 2: sub-chunk sums of exp(v - M), chunk sums, block cumsum
    const int Ls = (Lc + PIT_SC - 1) / PIT_SC;
    R s = 0;
    {
        int i = (int)(p0 / N), jj = (int)(p0 - (long long)i * N);
        long long p = p0;
#pragma unroll 1
        for (int b = 0; b < PIT_SC; ++b) {
            const long long pe = p0 + (long long)(b + 1) * Ls < p1 ? p0 + (long long)(b + 1) * Ls : p1;
            R sb = 0;
            for (; p < pe; ++p) {
                sb = sb + det_exp(value(i, jj) - vmax);
                if (++jj == N) jj = 0, ++i;
            }
            sub[b * NCH + tid] = sb;
            s = s + sb;
        }
    }
    block_cumsum<R>(s, cs, red, tid, nw);
    // pass 3: the draws
    uint16_t* Lo = a.Ls + (chain_nodes + a.off[k] + j) * N;
    uint16_t* Ro = a.Rs + (chain_nodes + a.off[k] + j) * N;
    uint16_t* Fo = a.Fi + (chain_nodes + a.off[k] + j) * N;
    uint16_t* Ao = a.La + (chain_nodes + a.off[k] + j) * N;
    if (tid < N && (!root || tid == 0)) {
        int il = 0, jr = 0;
        if (root || tid > 0) {
            const R un = pit_uniform<R>(a, a.u_res, STREAM_U_RES, ((long long)c * T + mid) * N + tid);
            const R r = cs[NCH - 1] * ((R)1 - un);
            int ts = lower_bound<R>(cs, NCH, r);
            const int last_chunk = (int)((NN - 1) / Lc);
            ts = ts < last_chunk ? ts : last_chunk;
            const R pre = ts > 0 ? cs[ts - 1] : (R)0;
            const long long c0 = (long long)ts * Lc;
            const long long c1 = c0 + Lc < NN ? c0 + Lc : NN;
            // the sub-chunk: first b whose running sum reaches r, else the last non-empty one
            const int nsub = (int)((c1 - c0 + Ls - 1) / Ls);
            int bsel = nsub - 1;
            R acc = 0;
            for (int b = 0; b < nsub; ++b) {
                const R nacc = acc + sub[b * NCH + ts];
                const R cvb = ts > 0 ? pre + nacc : nacc;
                if (cvb >= r || b == nsub - 1) {
                    bsel = b;
                    break;
                }
                acc = nacc;
            }
            const long long q0 = c0 + (long long)bsel * Ls;
            const long long q1 = q0 + Ls < c1 ? q0 + Ls : c1;
            long long psel = q1 - 1;
            int i = (int)(q0 / N), jj = (int)(q0 - (long long)i * N);
            for (long long p = q0; p < q1; ++p) {
                acc = acc + det_exp(value(i, jj) - vmax);
                const R cv = ts > 0 ? pre + acc : acc;
                if (cv >= r) {
                    psel = p;
                    break;
                }
                if (++jj == N) jj = 0, ++i;
            }
            il = (int)(psel / N);
            jr = (int)(psel - (long long)il * N);
        }
        Lo[tid] = (uint16_t)il;
        Ro[tid] = (uint16_t)jr;
        Fo[tid] = fi_left ? fi_left[il] : (uint16_t)il;
        Ao[tid] = la_right ? la_right[jr] : (uint16_t)jr;
    }
}

// read the selected trajectory off the tree: one thread per (chain, time step) walks the stored pairs down from the root
template <typename R> __global__ void k_pit_trace(PitArgs a, int D) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)a.C * a.T) return;
    const int c = (int)(g / a.T);
    const long long t = g % a.T;
    int q = 0;
    for (int k = a.K - 1; k >= 0; --k) {
        const long long j = t >> (k + 1);
        if ((j << (k + 1)) + (1ll << k) >= a.T) continue;  // passthrough: same slot in the left child
        const long long o = ((long long)c * a.tot + a.off[k] + j) * a.N + q;
        q = ((t >> k) & 1) ? a.Rs[o] : a.Ls[o];
    }
    a.anc[g] = q;
    for (int kk = 0; kk < D; ++kk) ((R*)a.x)[g * D + kk] = ((const R*)a.xs)[(g * a.N + q) * D + kk];
}

template <typename R, int D> static int run_pit(auxssm_ctx* h, const auxssm_fk_model* fk, const double* host_model, PitArgs& a, void* ctt) {
    FkDev<R> m;
    fill_model<R>(m, fk, host_model);
    m.gradient = fk->gradient;
    const int TB = (a.N + 63) / 64 * 64;
    if (fk->F_t) {  // time-varying transitions (csmc.py:103 scans Mt.params; here AuxiliaryGt's Mt, independent.py:238-248)
        m.Ft = (const R*)fk->F_t;
        m.bt = (const R*)fk->b_t;
        m.LQt = (const R*)fk->chol_Q_t;
        m.ctt = (const R*)ctt;
        m.idt = (const R*)ctt + (a.T - 1);
        hipLaunchKernelGGL((k_csmc_ctrans<R, D>), dim3((a.T - 1 + 255) / 256), dim3(256), 0, h->stream, a.T - 1, m.LQt, (R*)ctt, (R*)ctt + (a.T - 1));
    }
    if (fk->gradient) {  // u and the gradient of the model's joint log-density at u (independent.py:82, :121-134): the sequential sweep's kernels
        CsmcArgs ca{};
        ca.C = a.C; ca.T = a.T; ca.N = a.N;
        ca.y = a.y; ca.shd = a.shd; ca.x = a.x; ca.u = const_cast<void*>(a.u); ca.grad = const_cast<void*>(a.grad);
        ca.noise_mode = a.noise_mode; ca.key0 = a.key0; ca.key1 = a.key1; ca.eps_aux = a.eps_aux;
        const long long total = (long long)a.C * a.T * D, tot = (long long)a.C * a.T;
        hipLaunchKernelGGL((k_csmc_aux<R>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, ca, D);
        hipLaunchKernelGGL((k_csmc_grad<R, D>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, ca, m);
        hipLaunchKernelGGL((k_pit_leaves<R, D, true>), dim3(a.T, a.C), dim3(TB), 0, h->stream, a, m);
    } else {
        hipLaunchKernelGGL((k_pit_leaves<R, D, false>), dim3(a.T, a.C), dim3(TB), 0, h->stream, a, m);
    }
    const int NCH = a.N <= 32 ? 64 : (a.N <= 128 ? 256 : 1024);  // part of the arithmetic contract (header)
    const size_t lds = ((size_t)a.N * (2 * D + 2) + NCH + 48 + (size_t)PIT_SC * NCH) * sizeof(R) + 64;
    if (lds > 48 * 1024) AX_HIP(hipFuncSetAttribute((const void*)k_pit_stitch<R, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
        ProfScope ps(h, AUXSSM_K_PIT_STITCH);
        for (int k = 0; k < a.K; ++k) {
            const long long nodes = ((long long)a.T + (2ll << k) - 1) >> (k + 1);
            hipLaunchKernelGGL((k_pit_stitch<R, D>), dim3((unsigned)nodes, a.C), dim3(NCH), lds, h->stream, a, m, k);
        }
    }
    const long long total = (long long)a.C * a.T;
    hipLaunchKernelGGL((k_pit_trace<R>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, a, D);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

}  // namespace ax

using namespace ax;

extern "C" int auxssm_csmc_pit_sweep(auxssm_handle h, int dtype, const auxssm_fk_model* fk, int32_t C, int32_t T, int32_t N,
                                     const void* sqrt_half_delta, void* x, const auxssm_csmc_noise* noise, int32_t* ancestors) {
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
    if (!fk || !x || !noise || !ancestors || !sqrt_half_delta) {
        set_error("model/x/noise/ancestors/sqrt_half_delta must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    if (C < 1 || C > 65535 || T < 2 || N < 2 || N > 1024) {  // the chain index is grid.y of the leaf / stitch launches
        set_error("need 1 <= C <= 65535, T >= 2, 2 <= N <= 1024 (got C=%d T=%d N=%d)", C, T, N);
        return AUXSSM_ERR_ARG;
    }
    const int D = fk->dx;
    if (D < 1 || D > CS_MAXD) {
        set_error("dx=%d not instantiated (1..%d)", D, CS_MAXD);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    if (fk->proposal != AUXSSM_PROP_AUX_INDEPENDENT) {
        set_error("the parallel-in-time sweep needs proposals that are independent across time: AUXSSM_PROP_AUX_INDEPENDENT");
        return AUXSSM_ERR_ARG;
    }
    if ((fk->F_t || fk->b_t || fk->chol_Q_t) && !(fk->F_t && fk->b_t && fk->chol_Q_t)) {
        set_error("time-varying transitions need all of F_t, b_t, chol_Q_t (device arrays with T - 1 rows)");
        return AUXSSM_ERR_ARG;
    }
    if (fk->F_t && fk->transition != AUXSSM_TRANS_LINEAR) {
        set_error("time-varying transitions are linear-Gaussian");
        return AUXSSM_ERR_ARG;
    }
    if (fk->gradient != AUXSSM_GRAD_NONE && fk->gradient != AUXSSM_GRAD_REFERENCE && fk->gradient != AUXSSM_GRAD_EXACT) {
        set_error("unknown gradient mode %d", fk->gradient);
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
    if (noise->mode == AUXSSM_NOISE_EXPLICIT) {
        if (!noise->eps_prop || !noise->u_res || !noise->eps_aux) {
            set_error("explicit noise needs eps_aux (C,T,dx), eps_prop (C,T,N,dx) and u_res (C,T,N)");
            return AUXSSM_ERR_ARG;
        }
    } else if (noise->mode != AUXSSM_NOISE_THREEFRY) {
        set_error("unknown noise mode %d", noise->mode);
        return AUXSSM_ERR_ARG;
    }
    std::vector<double> hm((size_t)2 * D + 3 * D * D);
    {
        double* p = hm.data();
        memcpy(p, fk->m0, D * sizeof(double)); p += D;
        memcpy(p, fk->chol_P0, D * D * sizeof(double)); p += D * D;
        memcpy(p, fk->F, D * D * sizeof(double)); p += D * D;
        memcpy(p, fk->b, D * sizeof(double)); p += D;
        memcpy(p, fk->chol_Q, D * D * sizeof(double));
    }
    PitArgs a;
    memset(&a, 0, sizeof(a));
    a.C = C; a.T = T; a.N = N;
    int K = 0;
    while ((1ll << K) < T) ++K;
    a.K = K;
    long long tot = 0;
    for (int k = 0; k < K; ++k) {
        a.off[k] = tot;
        tot += ((long long)T + (2ll << k) - 1) >> (k + 1);
    }
    a.tot = tot;
    a.neg_log_n = dtype == AUXSSM_F32 ? (double)(-det_log((float)N)) : -det_log((double)N);
    const size_t sR = dtype == AUXSSM_F32 ? 4 : 8;
    const size_t CT = (size_t)C * T;
    const size_t tree = (size_t)C * tot * N * sizeof(uint16_t);
    size_t need = 8192 + CT * N * D * sR + (size_t)C * N * sR + 4 * (tree + 256);
    if (fk->gradient) need += 2 * (CT * D * sR + 256) + CT * N * sR + 256;  // u, grad, the leaf log-weights of every time step
    if (fk->F_t) need += (size_t)T * (1 + D) * sR + 256;
    int rc = ws_reserve(h, need);
    if (rc) return rc;
    a.y = fk->y;
    a.shd = sqrt_half_delta;
    a.x = x;
    a.xs = ws_take(h, CT * N * D * sR);
    a.lw0 = ws_take(h, (size_t)C * N * sR);
    if (fk->gradient) {  // (in the parallel kernel the correction is per particle in either mode: pit/csmc.py:83-88 has no summed variant)
        a.u = ws_take(h, CT * D * sR);
        a.grad = ws_take(h, CT * D * sR);
        a.lwt = ws_take(h, CT * N * sR);
        if (!a.u || !a.grad || !a.lwt) return AUXSSM_ERR_NOMEM;
    }
    void* ctt = fk->F_t ? ws_take(h, (size_t)T * (1 + D) * sR) : nullptr;
    if (fk->F_t && !ctt) return AUXSSM_ERR_NOMEM;
    a.Ls = (uint16_t*)ws_take(h, tree);
    a.Rs = (uint16_t*)ws_take(h, tree);
    a.Fi = (uint16_t*)ws_take(h, tree);
    a.La = (uint16_t*)ws_take(h, tree);
    a.anc = ancestors;
    a.noise_mode = noise->mode;
    a.key0 = noise->key0; a.key1 = noise->key1;
    a.eps_aux = noise->eps_aux; a.eps_prop = noise->eps_prop; a.u_res = noise->u_res;
    if (!a.xs || !a.lw0 || !a.Ls || !a.Rs || !a.Fi || !a.La) return AUXSSM_ERR_NOMEM;
#define AX_PIT_D(R)                                                    \
    switch (D) {                                                       \
        case 1: return run_pit<R, 1>(h, fk, hm.data(), a, ctt);             \
        case 2: return run_pit<R, 2>(h, fk, hm.data(), a, ctt);             \
        case 3: return run_pit<R, 3>(h, fk, hm.data(), a, ctt);             \
        default: return run_pit<R, 4>(h, fk, hm.data(), a, ctt);            \
    }
    if (dtype == AUXSSM_F32) { AX_PIT_D(float) } else { AX_PIT_D(double) }
#undef AX_PIT_D
}

// kernels.hip.h -- __global__ wrappers, the three-launch chunked associative scan, and the templated
// launchers that the instantiation units (inst_*.hip) expose through the tables in ctx.h.
//
// Scan structure (both the filter's (A,b,C,eta,J) scan and the sampler's (G,e) scan):
//   1. k_scan_reduce : one lane per chunk of E consecutive elements, sequential combine in registers
//                      -> one aggregate per chunk.
//   2. k_scan_aggs   : one workgroup per sequence scans the chunk aggregates (per-lane sequential +
//                      Kogge-Stone across the workgroup through LDS) and writes, per chunk, only the
//                      reduced exclusive prefix the final pass needs ((b,C) resp. e).
//   3. k_scan_down   : one lane per chunk re-walks its elements with the cheap "apply" form of the combine
//                      and writes the outputs in the caller's dense layout.
// No inter-workgroup communication inside a launch, so no spin-waits and a deterministic combination tree.
#pragma once
#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "ctx.h"
#include "affine_shared.h"

namespace ax {

constexpr int TB_ELEM = 128;  // elementwise kernels: threads per workgroup
constexpr int TB_SCAN = 64;   // chunk kernels: one wave per workgroup
constexpr int TB_AGGS = 256;  // aggregate scan: threads per sequence

// ---- block-wide deterministic sum (fixed tree) ------------------------------------------------------------
template <typename R, int TB> __device__ __forceinline__ R block_sum(R v, R* sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
#pragma unroll
    for (int off = TB / 2; off > 0; off >>= 1) {
        if (tid < off) sh[tid] += sh[tid + off];
        __syncthreads();
    }
    const R r = sh[0];
    __syncthreads();
    return r;
}

// ---- wave-cooperative record loads staged through LDS --------------------------------------------------------------------
// The wave needs record r(l) of N reals for each lane l, and those 64 records are contiguous in memory (ascending or
// descending with the lane).  Copy them as one contiguous stream (lane l reads element j*64 + l: 512 B / 1 KiB per
// instruction, 4-8 cache lines instead of 64), scatter into an LDS image with an ODD record stride (conflict-free per-lane
// reads), then every lane reads its own record.  Anything else (broadcast stride 0, batch-strided) falls back to a direct read.
// Workgroups using this are exactly one wave, so __syncthreads() is only a compiler/LDS fence.
template <typename R> struct WaveIO {
    R* lds;
    int lane;
    int nvalid;  // valid lanes are 0 .. nvalid-1
    // phase 1: buf[j] = element j*64 + lane of the wave's contiguous 64-record stream (fully coalesced), or the lane's own
    // record for strides that are not record-dense.  No barrier: every array's fetch is issued before any finish.
    template <typename R2, int N> __device__ __forceinline__ void fetch(const R2* lane_ptr, long long stride, long long se, bool valid, R2* buf) const {
        static_assert(sizeof(R2) == sizeof(R), "one real type per kernel");
        if (se == 1 && (stride == N || stride == -N)) {
            const R2* p0 = stride < 0 ? lane_ptr + (long long)(lane - (nvalid - 1)) * N : lane_ptr - (long long)lane * N;
            const int total = nvalid * N;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int q = j * 64 + lane;
                buf[j] = q < total ? p0[q] : (R2)0;
            }
        } else if (valid) {
            lds_<R2, N>(lane_ptr, se, buf);
        } else {
#pragma unroll
            for (int e = 0; e < N; ++e) buf[e] = 0;
        }
    }
    template <typename R2, int P> __device__ __forceinline__ void fetch_upper(const R2* lane_ptr, long long se, bool valid, R2* buf) const {
        if (valid) lds_upper_<R2, P>(lane_ptr, se, buf);
    }
    // phase 2: scatter the stream into an LDS image with an odd record stride, then read this lane's record back.
    template <typename R2, int N> __device__ __forceinline__ void finish(long long stride, long long se, bool valid, R2* buf) const {
        if (se == 1 && (stride == N || stride == -N)) {
            constexpr int RS = N | 1;
            R2* L = (R2*)lds;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int q = j * 64 + lane;
                const int r = q / N;
                if (r < 64) L[r * RS + (q - r * N)] = buf[j];
            }
            __syncthreads();
            const int myr = stride < 0 ? nvalid - 1 - lane : lane;
#pragma unroll
            for (int e = 0; e < N; ++e) buf[e] = valid ? L[myr * RS + e] : (R2)0;
            __syncthreads();
        }
    }
};
template <typename R> constexpr size_t stage_bytes(int maxrec) { return (size_t)64 * (maxrec | 1) * sizeof(R) + 16; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- XCD-aware (tile, sequence) decode for the elementwise kernels ------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share one; MI355X_MICROARCH.md), each XCD with a private L2.
// Chain-shared model parameters of a time tile should therefore be touched by ONE XCD: put tile % 8 in the low 3 bits of the
// block id, so that every sequence's block for that tile lands on the same XCD, and consecutive block ids sweep
// 8 tiles x all sequences (co-resident in time).  Purely a speed choice: any placement gives the same results.
// grid = ntile8 * 8 * S blocks with ntile8 = ceil(ntile / 8); out-of-range tiles exit.
__device__ __forceinline__ void decode_tile_seq(int S, int& tile, int& s) {
    const unsigned b = blockIdx.x;
    const unsigned lo = b & 7u, rest = b >> 3;
    s = (int)(rest % (unsigned)S);
    tile = (int)((rest / (unsigned)S) * 8u + lo);
}
inline unsigned grid_tile_seq(int ntile, int S) { return (unsigned)(((ntile + 7) / 8) * 8) * (unsigned)S; }

// ---- Kalman elementwise kernels -----------------------------------------------------------------------------
template <typename R, int D, int P> __global__ void __launch_bounds__(TB_ELEM) k_filter_t0(FilterArgs a) {
    if (memo_skip(a)) return;
    const int s = blockIdx.x * TB_ELEM + threadIdx.x;
    if (s < a.d.S()) body_filter_t0<R, D, P>(a, s);
}

template <typename R, int D> __global__ void __launch_bounds__(TB_ELEM) k_filter_t0_sv(FilterArgs a) {
    resolve_step(a);
    const int s = blockIdx.x * TB_ELEM + threadIdx.x;
    if (s < a.d.S()) body_filter_t0_sv<R, D>(a, s);
}

// grid = ntile * S, sequence index fastest so that workgroups of different chains touching the same time
// tile (hence the same chain-shared model parameters) are co-scheduled and share them through L2.
template <typename R, int D, int P> __global__ void __launch_bounds__(TB_ELEM) k_filter_init(FilterArgs a) {
    if (memo_skip(a)) return;
    resolve_step(a);
    int tile, s;
    decode_tile_seq(a.d.S(), tile, s);
    const int i = tile * TB_ELEM + threadIdx.x;
    if (i >= a.d.n()) return;
    DirectIO io;
    body_filter_init<R, D, P>(a, io, s, i, true);
}


// The reference's own second pass for the marginal log-likelihood (filtering.py:60-62): increment t from the filtered moments of t - 1 -- predict, then the
// innovation's log-density around the PREDICTED mean.  fp64 runs read ell off the scan's log-scale instead (FiltElem::z: no pass); in fp32 that scale is the small
// difference of element scales ~ -|y - H b|^2 / 2S around the dynamics offset (-2.6e5 per step at Lorenz-63 scale) and came back +-20 off over C4's 16 384 steps
// (round 4, tools/c4_fp32_diag.py), so fp32 runs of the element path take this pass: one lane per step, tile sums in fp64, part[s * ntile + tile].
template <typename R, int D, int P>
__global__ void __launch_bounds__(TB_ELEM) k_ell_pass(FilterArgs a, R* __restrict__ part, int ntile) {
    __shared__ Acc sh[TB_ELEM];
    int tile, s;
    decode_tile_seq(a.d.S(), tile, s);
    if (tile >= ntile) return;
    const int i = tile * TB_ELEM + threadIdx.x;
    Acc v = 0;
    if (i < a.d.n()) {
        const int c = s / a.d.B, b = s % a.d.B;
        const long long t = (long long)i + 1;
        R m[D], Pd[D * D], F[D * D], bd[D], Q[D * D], H[P * D], cv[P], y[P], Rm[P * P];
        rd<R, D>(a.ms, c, i, b, m);
        rd_cov<R, D>(a.Ps, c, i, b, a.ps_packed, Pd);
        rd<R, D * D>(a.Fs, c, i, b, F);
        rd<R, D>(a.bs, c, i, b, bd);
        rd<R, D * D>(a.Qs, c, i, b, Q);
        rd<R, P * D>(a.Hs, c, t, b, H);
        rd<R, P>(a.cs, c, t, b, cv);
        rd<R, P>(a.ys, c, t, b, y);
        rd_upper<R, P>(a.Rs, c, t, b, Rm);
        kalman_predict<R, D>(m, Pd, F, bd, Q);
        v = (Acc)kalman_update<R, D, P>(m, Pd, H, cv, Rm, y);
    }
    const Acc tot = block_sum<Acc, TB_ELEM>(v, sh);
    if (threadIdx.x == 0) part[(long long)s * ntile + tile] = (R)tot;
}

// lanes over i = t - 1; the t = 0 terms are added by lane 0 of tile 0
template <typename R, int D, int P>
__global__ void __launch_bounds__(TB_ELEM) k_joint_logpdf(LogpdfArgs a, R* __restrict__ part, int ntile) {
    __shared__ R sh[TB_ELEM];
    int tile, s;
    decode_tile_seq(a.d.S(), tile, s);
    if (tile >= ntile) return;
    const int n = a.d.T - 1;
    const int i = tile * TB_ELEM + threadIdx.x;
    DirectIO io;
    R v = body_joint_logpdf<R, D, P>(a, io, s, i, i < n);
    if (tile == 0 && threadIdx.x == 0) v += body_joint_logpdf_head<R, D, P>(a, s);
    const R tot = block_sum<R, TB_ELEM>(v, sh);
    if (threadIdx.x == 0) part[(long long)s * ntile + tile] = tot;
}

// five per-chain sums of the fused sweep log-density pass; part layout [5][C][ntile]
template <typename R, int D, int PO>
__global__ void __launch_bounds__(TB_ELEM) k_sweep_logpdf(SweepLogpdfArgs a, Acc* __restrict__ part, int ntile) {
    resolve_step(a);
    __shared__ Acc sh[TB_ELEM];
    const int C = a.d.C;
    int tile, c;
    decode_tile_seq(C, tile, c);
    if (tile >= ntile) return;
    const int n = a.d.T - 1;
    const int i = tile * TB_ELEM + threadIdx.x;
    R v[5] = {0, 0, 0, 0, 0};
    DirectIO io;
    body_sweep_logpdf<R, D, PO>(a, io, c, i, i < n, v);
    if (tile == 0 && threadIdx.x == 0) {
        R h[5];
        body_sweep_logpdf_head<R, D, PO>(a, c, h);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += h[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const Acc tot = block_sum<Acc, TB_ELEM>((Acc)v[k], sh);
        if (threadIdx.x == 0) part[((long long)k * C + c) * ntile + tile] = tot;
    }
}

// out[c] = sum_b ( add0[c*B+b] + sum_tile part[(c*B+b)*ntile + tile] ), fixed order.  One workgroup per chain.
template <typename R>
__global__ void __launch_bounds__(TB_ELEM) k_reduce_rows(const R* __restrict__ part, const R* __restrict__ add0,
                                                          int B, int ntile, R* __restrict__ out) {
    __shared__ Acc sh[TB_ELEM];  // (summed in fp64 whatever R: the partial sums of an fp32 log-likelihood add up to ~1e5 at C4's size)
    const int c = blockIdx.x;
    const long long base = (long long)c * B * ntile;
    const long long tot_n = (long long)B * ntile;
    Acc v = 0;
    for (long long k = threadIdx.x; k < tot_n; k += TB_ELEM) v += (Acc)part[base + k];
    if (add0)
        for (int b = threadIdx.x; b < B; b += TB_ELEM) v += (Acc)add0[(long long)c * B + b];
    const Acc tot = block_sum<Acc, TB_ELEM>(v, sh);
    if (threadIdx.x == 0) out[c] = (R)tot;
}

template <typename R, int D> __global__ void __launch_bounds__(TB_ELEM) k_sample_init(SampleArgs a) {
    int tile, s;
    decode_tile_seq(a.d.S(), tile, s);
    const int n = a.d.T - 1;
    if (tile == 0 && threadIdx.x == 0) body_sample_last<R, D>(a, s);
    const int jp = tile * TB_ELEM + threadIdx.x;
    if (jp >= n) return;
    DirectIO io;
    body_sample_init<R, D>(a, io, s, jp, true);
}

// ---- chain-minor mode: lanes <-> sequences -----------------------------------------------------------------------------
// Used by the fused sweep when many chains run side by side.  Per-chain buffers are [t][e][s] (cm_arr), so every component
// load/store of a wave is one contiguous 64-real run, and chain-shared model parameters are the same address in every lane
// (one cache line per instruction).  No transposition between the elementwise and the scan kernels, no LDS staging.
// A workgroup = one wave = 64 sequences x a tile of TI consecutive time steps; reductions over time are per-lane sums.
constexpr int TB_CM = 64;
// Wave-uniform loop index made opaque to loop-strength-reduction: with runtime strides LSR otherwise keeps one 64-bit
// induction pointer PER LOAD of the body in VGPRs (a log-likelihood tile loop at fp64 d=4 p=8: 512 registers + scratch vs 316 with this).
__device__ __forceinline__ int opaque_uniform(int i) {
    int ii = __builtin_amdgcn_readfirstlane(i);
    asm volatile("" : "+s"(ii));
    return ii;
}
struct CmTile {
    int s, i0, i1, tt;
    bool live;
};
__device__ __forceinline__ CmTile decode_cm(int S, int n, int TI) {
    const int stiles = (S + TB_CM - 1) / TB_CM;
    CmTile c;
    c.tt = blockIdx.x / stiles;
    c.s = (blockIdx.x % stiles) * TB_CM + threadIdx.x;
    c.i0 = c.tt * TI;
    c.i1 = min(n, c.i0 + TI);
    c.live = c.s < S;
    return c;
}
inline unsigned grid_cm(int S, int n, int TI) { return (unsigned)((S + TB_CM - 1) / TB_CM) * (unsigned)((n + TI - 1) / TI > 0 ? (n + TI - 1) / TI : 1); }

template <typename R, int D, int P, int P1> __global__ void __launch_bounds__(TB_CM) k_filter_init_cm(FilterArgs a, int TI) {
    resolve_step(a);
    const CmTile c = decode_cm(a.d.S(), a.d.n(), TI);
    if (!c.live) return;
    DirectIO io;
#pragma unroll 1
    for (int i = c.i0; i < c.i1; ++i) {
        body_filter_init<R, D, P, DirectIO, P1>(a, io, c.s, opaque_uniform(i), true);
    }
}
template <typename R, int D> __global__ void __launch_bounds__(TB_ELEM) k_sample_shared_tab(SampleArgs a) {
    if (memo_skip(a)) return;
    const int t = blockIdx.x * TB_ELEM + threadIdx.x;
    if (t < a.d.T) body_sample_shared_tab<R, D>(a, t);
}
template <typename R, int D> __global__ void __launch_bounds__(TB_CM) k_sample_init_cm(SampleArgs a, int TI) {
    const CmTile c = decode_cm(a.d.S(), a.d.T - 1, TI);
    if (!c.live) return;
    DirectIO io;
    if (c.tt == 0) body_sample_last<R, D>(a, c.s);
#pragma unroll 1
    for (int jp = c.i0; jp < c.i1; ++jp) {
        body_sample_init<R, D>(a, io, c.s, opaque_uniform(jp), true);
    }
}
template <typename R, int D, int PO>
__global__ void __launch_bounds__(TB_CM) k_sweep_logpdf_cm(SweepLogpdfArgs a, Acc* __restrict__ part, int ntile, int TI) {
    resolve_step(a);
    const CmTile c = decode_cm(a.d.C, a.d.T - 1, TI);
    if (!c.live) return;
    Acc v[5] = {0, 0, 0, 0, 0};
    if (c.tt == 0) {
        R h[5];
        body_sweep_logpdf_head<R, D, PO>(a, c.s, h);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = (Acc)h[k];
    }
    // stream the chain's values: (x, xp) of the previous step stay in registers, the next step's reads fly during this step's arithmetic
    const Arr& ua = a.u_fly ? a.eps_aux : a.u;
    R xq[D], xpq[D], xn[D], xpn[D], un[D];
    rd<R, D>(a.x, c.s, c.i0, 0, xq);
    rd<R, D>(a.xp, c.s, c.i0, 0, xpq);
    rd<R, D>(a.x, c.s, (long long)c.i0 + 1, 0, xn);
    rd<R, D>(a.xp, c.s, (long long)c.i0 + 1, 0, xpn);
    rd<R, D>(ua, c.s, (long long)c.i0 + 1, 0, un);
    LogProd<R> lp[4];   // the tile's determinants (Q_t and R_t per step and chain): one logarithm per sum and tile instead of D + PO per step
#pragma unroll 1
    for (int i = c.i0; i < c.i1; ++i) {
        R xc[D], xpc[D], uc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xc[k] = xn[k], xpc[k] = xpn[k], uc[k] = un[k];
        const int iu = opaque_uniform(i);
        const long long t = (long long)iu + 1;
        if (i + 1 < c.i1) {
            rd<R, D>(a.x, c.s, t + 1, 0, xn);
            rd<R, D>(a.xp, c.s, t + 1, 0, xpn);
            rd<R, D>(ua, c.s, t + 1, 0, un);
        }
        R H[PO * D], cv[PO], y[PO], Rm[PO * PO], F[D * D], bd[D], Q[D * D], w[5];
        rd<R, PO * D>(a.Hs, c.s, t, 0, H);
        rd<R, PO>(a.cs, c.s, t, 0, cv);
        rd<R, PO>(a.ys, c.s, t, 0, y);
        rd_upper<R, PO>(a.Rs, c.s, t, 0, Rm);
        rd<R, D * D>(a.Fs, c.s, iu, 0, F);
        rd<R, D>(a.bs, c.s, iu, 0, bd);
        rd<R, D * D>(a.Qs, c.s, iu, 0, Q);
        R f[4];
        sweep_logpdf_core<R, D, PO>(a, xc, xpc, uc, xq, xpq, H, cv, y, Rm, F, bd, Q, w, f);
#pragma unroll
        for (int k = 0; k < D; ++k) xq[k] = xc[k], xpq[k] = xpc[k];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += (Acc)w[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) lp[k].mul(f[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += (Acc)((R)0.5 * lp[k].log());
#pragma unroll
    for (int k = 0; k < 5; ++k) part[((long long)k * a.d.C + c.s) * ntile + c.tt] = v[k];
}

template <typename R, int D, int PO> __global__ void __launch_bounds__(TB_ELEM) k_sweep_logpdf_tab(SweepLogpdfArgs a) {
    if (memo_skip(a)) return;
    const int i = blockIdx.x * TB_ELEM + threadIdx.x;
    if (i < a.d.T - 1) body_sweep_logpdf_tab<R, D, PO>(a, i);
}
template <typename R, int D, int PO>
__global__ void __launch_bounds__(TB_CM) k_sweep_logpdf_cm_shared(SweepLogpdfArgs a, Acc* __restrict__ part, int ntile, int TI) {
    resolve_step(a);
    const CmTile c = decode_cm(a.d.C, a.d.T - 1, TI);
    if (!c.live) return;
    Acc v[5] = {0, 0, 0, 0, 0};
    if (c.tt == 0) {
        R h[5];
        body_sweep_logpdf_head<R, D, PO>(a, c.s, h);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = (Acc)h[k];
    }
    // stream the chain's values: (x, xp) of the previous step stay in registers, the next step's reads fly during this step's arithmetic
    const Arr& ua = a.u_fly ? a.eps_aux : a.u;
    R xq[D], xpq[D], xn[D], xpn[D], un[D];
    rd<R, D>(a.x, c.s, c.i0, 0, xq);
    rd<R, D>(a.xp, c.s, c.i0, 0, xpq);
    rd<R, D>(a.x, c.s, (long long)c.i0 + 1, 0, xn);
    rd<R, D>(a.xp, c.s, (long long)c.i0 + 1, 0, xpn);
    rd<R, D>(ua, c.s, (long long)c.i0 + 1, 0, un);
#pragma unroll 1
    for (int i = c.i0; i < c.i1; ++i) {
        R xc[D], xpc[D], uc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xc[k] = xn[k], xpc[k] = xpn[k], uc[k] = un[k];
        if (i + 1 < c.i1) {
            const long long tn = (long long)opaque_uniform(i + 1) + 1;
            rd<R, D>(a.x, c.s, tn, 0, xn);
            rd<R, D>(a.xp, c.s, tn, 0, xpn);
            rd<R, D>(ua, c.s, tn, 0, un);
        }
        R w[5];
        body_sweep_logpdf_shared<R, D, PO>(a, opaque_uniform(i), xc, xpc, uc, xq, xpq, w);
#pragma unroll
        for (int k = 0; k < D; ++k) xq[k] = xc[k], xpq[k] = xpc[k];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += (Acc)w[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) part[((long long)k * a.d.C + c.s) * ntile + c.tt] = v[k];
}

// SEMI-shared per-chain passes (kalman_bodies.h): the table's whitening rows, the chain's own transition mean.  LORENZ: the Lorenz-63 step instead of (F, b).
template <typename R, int D, int PO, bool LORENZ>
__global__ void __launch_bounds__(TB_CM) k_sweep_logpdf_cm_semi(SweepLogpdfArgs a, Acc* __restrict__ part, int ntile, int TI) {
    resolve_step(a);
    const CmTile c = decode_cm(a.d.C, a.d.T - 1, TI);
    if (!c.live) return;
    Acc v[5] = {0, 0, 0, 0, 0};
    if (c.tt == 0) {
        R h[5];
        body_sweep_logpdf_head<R, D, PO>(a, c.s, h);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = (Acc)h[k];
    }
    const Arr& ua = a.u_fly ? a.eps_aux : a.u;
    R xq[D], xpq[D], xn[D], xpn[D], un[D];
    rd<R, D>(a.x, c.s, c.i0, 0, xq);
    rd<R, D>(a.xp, c.s, c.i0, 0, xpq);
    rd<R, D>(a.x, c.s, (long long)c.i0 + 1, 0, xn);
    rd<R, D>(a.xp, c.s, (long long)c.i0 + 1, 0, xpn);
    rd<R, D>(ua, c.s, (long long)c.i0 + 1, 0, un);
#pragma unroll 1
    for (int i = c.i0; i < c.i1; ++i) {
        R xc[D], xpc[D], uc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xc[k] = xn[k], xpc[k] = xpn[k], uc[k] = un[k];
        const int iu = opaque_uniform(i);
        if (i + 1 < c.i1) {
            const long long tn = (long long)iu + 2;
            rd<R, D>(a.x, c.s, tn, 0, xn);
            rd<R, D>(a.xp, c.s, tn, 0, xpn);
            rd<R, D>(ua, c.s, tn, 0, un);
        }
        R w[5];
        if constexpr (LORENZ) {
            body_lorenz_logpdf_semi<R, PO>(a, c.s, iu, xc, xpc, uc, xq, xpq, w);
        } else {
            R F[D * D], bd[D];
            rd<R, D * D>(a.Fs, c.s, iu, 0, F);
            rd<R, D>(a.bs, c.s, iu, 0, bd);
            body_sweep_logpdf_semi<R, D, PO>(a, iu, xc, xpc, uc, xq, xpq, F, bd, w);
        }
#pragma unroll
        for (int k = 0; k < D; ++k) xq[k] = xc[k], xpq[k] = xpc[k];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += (Acc)w[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) part[((long long)k * a.d.C + c.s) * ntile + c.tt] = v[k];
}

// the Lorenz sweep's five per-chain sums (kalman_bodies.h::body_lorenz_logpdf); part layout [5][C][ntile]
template <typename R, int PO> __global__ void __launch_bounds__(TB_ELEM) k_lorenz_logpdf(SweepLogpdfArgs a, Acc* __restrict__ part, int ntile) {
    resolve_step(a);
    __shared__ Acc sh[TB_ELEM];
    const int C = a.d.C;
    int tile, c;
    decode_tile_seq(C, tile, c);
    if (tile >= ntile) return;
    const int n = a.d.T - 1;
    const int i = tile * TB_ELEM + threadIdx.x;
    R v[5];
    body_lorenz_logpdf<R, PO>(a, c, i, i < n, v);
    if (tile == 0 && threadIdx.x == 0) {
        R h[5];
        body_lorenz_logpdf_head<R, PO>(a, c, h);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += h[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const Acc tot = block_sum<Acc, TB_ELEM>((Acc)v[k], sh);
        if (threadIdx.x == 0) part[((long long)k * C + c) * ntile + tile] = tot;
    }
}
template <typename R, int PO> __global__ void __launch_bounds__(TB_CM) k_lorenz_logpdf_cm(SweepLogpdfArgs a, Acc* __restrict__ part, int ntile, int TI) {
    resolve_step(a);
    const CmTile c = decode_cm(a.d.C, a.d.T - 1, TI);
    if (!c.live) return;
    Acc v[5] = {0, 0, 0, 0, 0};
    if (c.tt == 0) {
        R h[5];
        body_lorenz_logpdf_head<R, PO>(a, c.s, h);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = (Acc)h[k];
    }
    LogProd<R> lp[4];
#pragma unroll 1
    for (int i = c.i0; i < c.i1; ++i) {
        R w[5], f[4];
        body_lorenz_logpdf<R, PO>(a, c.s, opaque_uniform(i), true, w, f);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += (Acc)w[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) lp[k].mul(f[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += (Acc)((R)0.5 * lp[k].log());
#pragma unroll
    for (int k = 0; k < 5; ++k) part[((long long)k * a.d.C + c.s) * ntile + c.tt] = v[k];
}

// the SV sweep's five per-chain sums (kalman_bodies.h::SvLogpdfArgs); part layout [5][C][ntile]
template <typename R, int D> __global__ void __launch_bounds__(TB_ELEM) k_sv_logpdf(SvLogpdfArgs a, Acc* __restrict__ part, int ntile) {
    resolve_step(a);
    __shared__ Acc sh[TB_ELEM];
    const int C = a.d.C;
    int tile, c;
    decode_tile_seq(C, tile, c);
    if (tile >= ntile) return;
    const int n = a.d.T - 1;
    const int i = tile * TB_ELEM + threadIdx.x;
    R v[5], f[4];
    body_sv_logpdf<R, D>(a, c, i, i < n, v, f);
    if (tile == 0 && threadIdx.x == 0) {
        R h[5], fh[4];
        body_sv_logpdf_head<R, D>(a, c, h, fh);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += h[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += (R)0.5 * log_(fh[k]);
    }
    if (i < n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += (R)0.5 * log_(f[k]);   // (one step per lane: nothing to multiply up)
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const Acc tot = block_sum<Acc, TB_ELEM>((Acc)v[k], sh);
        if (threadIdx.x == 0) part[((long long)k * C + c) * ntile + tile] = tot;
    }
}
template <typename R, int D> __global__ void __launch_bounds__(TB_CM) k_sv_logpdf_cm(SvLogpdfArgs a, Acc* __restrict__ part, int ntile, int TI) {
    resolve_step(a);
    const CmTile c = decode_cm(a.d.C, a.d.T - 1, TI);
    if (!c.live) return;
    Acc v[5] = {0, 0, 0, 0, 0};
    LogProd<R> lp[4];   // the tile's determinants: one logarithm per sum and tile instead of three per step
    if (c.tt == 0) {
        R h[5], fh[4];
        body_sv_logpdf_head<R, D>(a, c.s, h, fh);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = (Acc)h[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) lp[k].mul(fh[k]);
    }
#pragma unroll 1
    for (int i = c.i0; i < c.i1; ++i) {
        R w[5], f[4];
        body_sv_logpdf<R, D>(a, c.s, opaque_uniform(i), true, w, f);
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += (Acc)w[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) lp[k].mul(f[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += (Acc)((R)0.5 * lp[k].log());
#pragma unroll
    for (int k = 0; k < 5; ++k) part[((long long)k * a.d.C + c.s) * ntile + c.tt] = v[k];
}

// scan passes: lane <-> sequence, workgroup <-> (chunk, 64 sequences)
template <class Op>
__global__ void __launch_bounds__(TB_CM) k_scan_reduce_cm(typename Op::Args a, ScanBufs sb, int S, int n) {
    resolve_step(a);
    using R = typename Op::R;
    using Full = typename Op::Full;
    const ScanLayout lay = Op::layout(a);
    const int stiles = (S + TB_CM - 1) / TB_CM;
    const int ch = blockIdx.x / stiles;
    const int s = (blockIdx.x % stiles) * TB_CM + threadIdx.x;
    if (s >= S) return;
    const int i0 = ch * lay.E, i1 = min(n, i0 + lay.E);
    using Raw = typename Op::Raw;
    Full acc;
    Raw nxt;
    if constexpr (Op::kFold) {
        // the operator folds raw steps onto the prefix itself (no element is formed); the next step's reads fly during the fold
        Op::init_acc(a, s, ch, acc);
        Op::load_raw(a, s, i0, nxt);
        typename Op::Carry cy;
        for (int i = i0; i < i1; ++i) {
            const Raw cur = nxt;
            if (i + 1 < i1) Op::load_raw(a, s, opaque_uniform(i + 1), nxt);
            Op::fold(a, s, opaque_uniform(i), cur, acc, cy);
        }
        Op::carry_flush(cy, acc.z);
    } else {
        Op::load_elem(a, s, i0, acc);
        if (i0 + 1 < i1) Op::load_raw(a, s, i0 + 1, nxt);
        for (int i = i0 + 1; i < i1; ++i) {
            const Raw cur = nxt;
            if (i + 1 < i1) Op::load_raw(a, s, opaque_uniform(i + 1), nxt);  // next element's reads fly during the build + combine
            Full e, o;
            Op::build(a, s, opaque_uniform(i), cur, e);
            Op::combine(acc, e, o);
            acc = o;
        }
    }
    Op::store_rec((R*)sb.agg + ((long long)s * lay.nchunk + ch) * Full::NPAD, acc);
}
template <class Op, typename = void> struct CarryOf { struct type {}; };
template <class Op> struct CarryOf<Op, std::void_t<typename Op::Carry>> { using type = typename Op::Carry; };
// (operators may ask for two waves per SIMD in the down pass -- Op::kDownWaves -- when their walk is within a few registers of that budget)
template <class Op, typename = void> struct DownWaves { static constexpr int value = 1; };
template <class Op> struct DownWaves<Op, decltype((void)Op::kDownWaves)> { static constexpr int value = Op::kDownWaves; };
template <class Op>
__global__ void __launch_bounds__(TB_CM, DownWaves<Op>::value) k_scan_down_cm(typename Op::Args a, ScanBufs sb, int S, int n) {
    resolve_step(a);
    using R = typename Op::R;
    using Full = typename Op::Full;
    using Pre = typename Op::Pre;
    const ScanLayout lay = Op::layout(a);
    const int stiles = (S + TB_CM - 1) / TB_CM;
    const int ch = blockIdx.x / stiles;
    const int s = (blockIdx.x % stiles) * TB_CM + threadIdx.x;
    if (s >= S) return;
    const int i0 = ch * lay.E, i1 = min(n, i0 + lay.E);
    Pre p;
    if (lay.nchunk > 1 && !(Op::kFold && ch == 0)) {
        Op::load_pre((const R*)sb.pre + ((long long)s * lay.nchunk + ch) * Pre::NPAD, p);
    } else {
        if constexpr (Op::kFold) {
            Op::init_pre(a, s, p);
        } else {
            Full id;
            Op::identity(id);
            Op::to_pre(id, p);
        }
    }
    // Folding operators return the log-likelihood as per-chunk sums of the walk's own increments, NOT as the log-scale of the scanned prefix: a chunk aggregate
    // is a quadratic form in the chunk's start state AROUND THE ORIGIN, its scale is ~ -|x|^2 J / 2 (-2e8 at Lorenz-63 scale with delta = 1e-5), and in fp32 the
    // aggregate scan returned the total to +-ulp(2e8) = 16 per combine (round 4, tools/c4_accept_probe.py); the increments themselves are O(dim) each.
    if constexpr (Op::kFold) p.z = 0;
    typename CarryOf<Op>::type cy;
    {
        using Raw = typename Op::Raw;
        Raw nxt;
        if (i0 < i1) Op::load_raw(a, s, i0, nxt);
        for (int i = i0; i < i1; ++i) {
            const Raw cur = nxt;
            if (i + 1 < i1) Op::load_raw(a, s, opaque_uniform(i + 1), nxt);
            if constexpr (Op::kFold) {
                Op::walk(a, s, opaque_uniform(i), cur, p, cy);
            } else {
                Full e;
                Pre o;
                Op::build(a, s, opaque_uniform(i), cur, e);
                Op::apply(p, e, o);
                p = o;
            }
            Op::write_out(a, s, opaque_uniform(i), p);
        }
    }
    if constexpr (Op::kFold) {
        Op::carry_flush(cy, p.z);
        Op::write_zpart(a, s, ch, lay.nchunk, p.z);
    }
}

// ---- generic chunked scan ---------------------------------------------------------------------------------------
// one wave = one group of 64 chunks of one sequence; row k of the group is 64 contiguous records (ScanLayout)
template <class Op>
__global__ void __launch_bounds__(TB_SCAN) k_scan_reduce(typename Op::Args a, ScanBufs sb, int n) {
    if (memo_skip(a)) return;
    resolve_step(a);
    using R = typename Op::R;
    using Full = typename Op::Full;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ScanLayout lay = Op::layout(a);
    const int s = blockIdx.x / lay.ngrp, grp = blockIdx.x % lay.ngrp;
    const int lane = threadIdx.x;
    const int ch = grp * lay.W + lane;
    const bool inrow = lane < lay.W;
    const bool live = inrow && ch < lay.nchunk;
    const int len = live ? min(lay.E, n - ch * lay.E) : 0;
    WaveIO<R> io{(R*)smem, lane, lay.W};
    R rec[Full::NPAD], nxt[Full::NPAD];
    Full acc;
    io.template fetch<R, Full::NPAD>(Op::row_ptr(a, s, grp, 0) + (long long)lane * Full::NPAD, Full::NPAD, 1, inrow, rec);
    io.template finish<R, Full::NPAD>(Full::NPAD, 1, inrow, rec);
    Op::unpack(rec, acc);
    if (lay.E > 1) io.template fetch<R, Full::NPAD>(Op::row_ptr(a, s, grp, 1) + (long long)lane * Full::NPAD, Full::NPAD, 1, inrow, nxt);
    for (int k = 1; k < lay.E; ++k) {
#pragma unroll
        for (int e = 0; e < Full::NPAD; ++e) rec[e] = nxt[e];
        io.template finish<R, Full::NPAD>(Full::NPAD, 1, inrow, rec);
        // the next row's global reads fly while this row is combined
        if (k + 1 < lay.E) io.template fetch<R, Full::NPAD>(Op::row_ptr(a, s, grp, k + 1) + (long long)lane * Full::NPAD, Full::NPAD, 1, inrow, nxt);
        if (k < len) {
            Full cur, o;
            Op::unpack(rec, cur);
            Op::combine(acc, cur, o);
            acc = o;
        }
    }
    if (live) Op::store_rec((R*)sb.agg + ((long long)s * lay.nchunk + ch) * Full::NPAD, acc);
}

template <class Op> __global__ void __launch_bounds__(TB_AGGS) k_scan_aggs(ScanBufs sb, int nchunk) {
    if (memo_skip(sb)) return;
    using R = typename Op::R;
    using Full = typename Op::Full;
    using Pre = typename Op::Pre;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    R* lds = (R*)smem;
    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    const int E2 = (nchunk + TB_AGGS - 1) / TB_AGGS;
    const int j0 = min(nchunk, tid * E2), j1 = min(nchunk, j0 + E2);
    const R* agg = (const R*)sb.agg + (long long)s * nchunk * Full::NPAD;
    R* pre = (R*)sb.pre + (long long)s * nchunk * Pre::NPAD;
    Full acc;
    Op::identity(acc);
    if (j0 < j1) Op::load_rec(agg + (long long)j0 * Full::NPAD, acc);
    for (int j = j0 + 1; j < j1; ++j) {
        Full e, o;
        Op::load_rec(agg + (long long)j * Full::NPAD, e);
        Op::combine(acc, e, o);
        acc = o;
    }
    // Kogge-Stone inclusive scan of the per-lane totals
    for (int off = 1; off < TB_AGGS; off <<= 1) {
        Op::store_rec(lds + tid * Full::NPAD, acc);
        __syncthreads();
        Full left;
        if (tid >= off) Op::load_rec(lds + (tid - off) * Full::NPAD, left);
        __syncthreads();
        if (tid >= off) {
            Full o;
            Op::combine(left, acc, o);
            acc = o;
        }
    }
    Op::store_rec(lds + tid * Full::NPAD, acc);
    __syncthreads();
    Full ex;
    Op::identity(ex);
    if (tid > 0) Op::load_rec(lds + (tid - 1) * Full::NPAD, ex);
    // the exclusive prefixes inside this lane's run: only the reduced part (what the final pass reads) is carried, with the cheap
    // "apply" form of the combine
    Pre p;
    Op::to_pre(ex, p);
    for (int j = j0; j < j1; ++j) {
        Op::store_pre(pre + (long long)j * Pre::NPAD, p);
        if (j + 1 < j1) {
            Full e;
            Pre o;
            Op::load_rec(agg + (long long)j * Full::NPAD, e);
            Op::apply(p, e, o);
            p = o;
        }
    }
}

template <class Op>
__global__ void __launch_bounds__(TB_SCAN) k_scan_down(typename Op::Args a, ScanBufs sb, int n) {
    if (memo_skip(a)) return;
    resolve_step(a);
    using R = typename Op::R;
    using Full = typename Op::Full;
    using Pre = typename Op::Pre;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ScanLayout lay = Op::layout(a);
    const int s = blockIdx.x / lay.ngrp, grp = blockIdx.x % lay.ngrp;
    const int lane = threadIdx.x;
    const int ch = grp * lay.W + lane;
    const bool inrow = lane < lay.W;
    const bool live = inrow && ch < lay.nchunk;
    const int len = live ? min(lay.E, n - ch * lay.E) : 0;
    Pre p;
    if (lay.nchunk > 1 && live) {
        Op::load_pre((const R*)sb.pre + ((long long)s * lay.nchunk + ch) * Pre::NPAD, p);
    } else {
        Full id;
        Op::identity(id);
        Op::to_pre(id, p);
    }
    WaveIO<R> io{(R*)smem, lane, lay.W};
    R rec[Full::NPAD], nxt[Full::NPAD];
    io.template fetch<R, Full::NPAD>(Op::row_ptr(a, s, grp, 0) + (long long)lane * Full::NPAD, Full::NPAD, 1, inrow, nxt);
    for (int k = 0; k < lay.E; ++k) {
#pragma unroll
        for (int e = 0; e < Full::NPAD; ++e) rec[e] = nxt[e];
        io.template finish<R, Full::NPAD>(Full::NPAD, 1, inrow, rec);
        if (k + 1 < lay.E) io.template fetch<R, Full::NPAD>(Op::row_ptr(a, s, grp, k + 1) + (long long)lane * Full::NPAD, Full::NPAD, 1, inrow, nxt);
        if (k < len) {
            Full cur;
            Pre o;
            Op::unpack(rec, cur);
            Op::apply(p, cur, o);
            p = o;
            Op::write_out(a, s, ch * lay.E + k, p);
        }
    }
}

// ---- chain-shared model parameters: per-chain affine scans (affine_shared.h) -------------------------------------------------------
// grid of the per-chain passes: workgroup = one wave = 64 chains x one chunk of E positions.  Block id -> (chunk, chain tile) with
// chunk % 8 in the low three bits, so the waves that read one chunk's table rows sit on one XCD (one L2 / scalar-cache fill per row).
__device__ __forceinline__ bool decode_aff(int S, int nchunk, int& ch, int& s) {
    const int stiles = (S + TB_CM - 1) / TB_CM;
    const unsigned b = blockIdx.x, lo = b & 7u, rest = b >> 3;
    s = (int)(rest % (unsigned)stiles) * TB_CM + threadIdx.x;
    ch = (int)(rest / (unsigned)stiles) * 8 + (int)lo;
    return ch < nchunk && s < S;
}
inline unsigned grid_aff(int S, int nchunk) { return (unsigned)(((nchunk + 7) / 8) * 8) * (unsigned)((S + TB_CM - 1) / TB_CM); }

template <typename R, int D, int P> __global__ void __launch_bounds__(TB_ELEM) k_gain_tab(FilterArgs a, const R* __restrict__ Ps1) {
    if (memo_skip(a)) return;
    const int i = blockIdx.x * TB_ELEM + threadIdx.x;
    if (i < a.d.n()) body_gain_tab<R, D, P>(a, Ps1, i);
}
template <typename R, int D, int P> __global__ void __launch_bounds__(TB_ELEM) k_obs_info_tab(FilterArgs a) {
    resolve_step(a);
    const int i = blockIdx.x * TB_ELEM + threadIdx.x;
    if constexpr (P > D) {
        if (i < a.d.n()) body_obs_info_tab<R, D, P>(a, i);
    }
}
// the mask carrier of the matrix filter when the concatenated observations are built on the fly: [0 ; yobs_t]
template <typename R> __global__ void k_mask_obs(int T, int D, int P, Arr yobs, R* __restrict__ out, const int* memo = nullptr) {
    if (memo_skip_p(memo)) return;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)T * P) return;
    const long long t = g / P;
    const int k = (int)(g % P);
    out[g] = k < D ? (R)0 : at<R>(yobs, 0, t, 0)[k - D];
}
// dense (T, D, D) covariances of the matrix filter -> chain 0's slot of the caller's (strided) covariance buffer
template <typename R, int D> __global__ void k_copy_cov(int T, const R* __restrict__ src, Arr dst, const int* memo = nullptr) {
    if (memo_skip_p(memo)) return;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)T * D * D) return;
    const long long t = g / (D * D);
    const int e = (int)(g % (D * D));
    const_cast<R*>(at<R>(dst, 0, t, 0))[(long long)e * dst.se] = src[g];
}
// product of the chunk's matrices (scan order), one lane per chunk: the G of the chunk aggregates
template <class Op, int D> __global__ void __launch_bounds__(TB_CM) k_aff_chunkprod(typename Op::Args a, typename Op::R* __restrict__ cprod, int N, AffPlan pl) {
    using R = typename Op::R;
    const int ch = blockIdx.x * TB_CM + threadIdx.x;
    if (ch >= pl.nchunk) return;
    const int j0 = ch * pl.E, j1 = min(N, j0 + pl.E);
    R M[D * D];
#pragma unroll
    for (int k = 0; k < D * D; ++k) M[k] = (k / D == k % D) ? (R)1 : (R)0;
    for (int j = j0; j < j1; ++j) {
        R G[D * D], o[D * D];
        Op::mat(a, j, G);
        mm<R, D, D, D>(G, M, o);
#pragma unroll
        for (int k = 0; k < D * D; ++k) M[k] = o[k];
    }
#pragma unroll
    for (int k = 0; k < D * D; ++k) cprod[(long long)ch * D * D + k] = M[k];
}
template <class Op, int D>
__global__ void __launch_bounds__(TB_CM) k_aff_reduce(typename Op::Args a, ScanBufs sb, int S, int N, AffPlan pl) {
    resolve_step(a);
    using R = typename Op::R;
    using Full = SampElem<R, D>;
    int ch, s;
    if (!decode_aff(S, pl.nchunk, ch, s)) return;
    const int j0 = ch * pl.E, j1 = min(N, j0 + pl.E);
    Full agg;
    if (ch == 0) Op::init(a, s, agg.e);
    else {
#pragma unroll
        for (int k = 0; k < D; ++k) agg.e[k] = 0;
    }
#pragma unroll 1
    for (int j = j0; j < j1; ++j) Op::fold(a, s, opaque_uniform(j), agg.e);
    // the aggregate of (chain, chunk) is the affine map (G_chunk, e): G_chunk is the chain-shared chunk product (k_aff_chunkprod's table, read by
    // k_aff_aggs), only the offset is the chain's -- D reals per record instead of D^2 + D
    stv<R, D>((R*)sb.agg + ((long long)s * pl.nchunk + ch) * SampPre<R, D>::NPAD, agg.e);
}
// k_scan_aggs<SampleOp> for these aggregates: one workgroup per chain scans (G_j, e_j), j < nchunk, with G_j from the shared table
template <typename R, int D> __global__ void __launch_bounds__(TB_AGGS) k_aff_aggs(ScanBufs sb, const R* __restrict__ cprod, int nchunk) {
    using Op = SampleOp<R, D>;
    using Full = SampElem<R, D>;
    using Pre = SampPre<R, D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    R* lds = (R*)smem;
    const int s = blockIdx.x;
    const int tid = threadIdx.x;
    const int E2 = (nchunk + TB_AGGS - 1) / TB_AGGS;
    const int j0 = min(nchunk, tid * E2), j1 = min(nchunk, j0 + E2);
    const R* agg = (const R*)sb.agg + (long long)s * nchunk * Pre::NPAD;
    R* pre = (R*)sb.pre + (long long)s * nchunk * Pre::NPAD;
    auto load = [&](int j, Full& e) {
        ldv<R, D * D>(cprod + (long long)j * D * D, e.G);
        ldv<R, D>(agg + (long long)j * Pre::NPAD, e.e);
    };
    Full acc;
    Op::identity(acc);
    if (j0 < j1) load(j0, acc);
    for (int j = j0 + 1; j < j1; ++j) {
        Full e, o;
        load(j, e);
        Op::combine(acc, e, o);
        acc = o;
    }
    for (int off = 1; off < TB_AGGS; off <<= 1) {  // Kogge-Stone inclusive scan of the per-lane totals
        Op::store_rec(lds + tid * Full::NPAD, acc);
        __syncthreads();
        Full left;
        if (tid >= off) Op::load_rec(lds + (tid - off) * Full::NPAD, left);
        __syncthreads();
        if (tid >= off) {
            Full o;
            Op::combine(left, acc, o);
            acc = o;
        }
    }
    Op::store_rec(lds + tid * Full::NPAD, acc);
    __syncthreads();
    Full ex;
    Op::identity(ex);
    if (tid > 0) Op::load_rec(lds + (tid - 1) * Full::NPAD, ex);
    Pre p;
    Op::to_pre(ex, p);
    for (int j = j0; j < j1; ++j) {
        Op::store_pre(pre + (long long)j * Pre::NPAD, p);
        if (j + 1 < j1) {
            Full e;
            Pre o;
            load(j, e);
            Op::apply(p, e, o);
            p = o;
        }
    }
}
template <class Op, int D>
__global__ void __launch_bounds__(TB_CM) k_aff_down(typename Op::Args a, ScanBufs sb, typename Op::R* __restrict__ part, int S, int N, AffPlan pl) {
    resolve_step(a);
    using R = typename Op::R;
    int ch, s;
    if (!decode_aff(S, pl.nchunk, ch, s)) return;
    const int j0 = ch * pl.E, j1 = min(N, j0 + pl.E);
    R h[D];
    if (ch == 0) Op::init(a, s, h);
    else ldv<R, D>((const R*)sb.pre + ((long long)s * pl.nchunk + ch) * SampPre<R, D>::NPAD, h);
    R acc = 0;
#pragma unroll 1
    for (int j = j0; j < j1; ++j) Op::walk(a, s, opaque_uniform(j), h, acc);
    if (part) part[(long long)s * pl.nchunk + ch] = acc;
}
// chunks per SIMD lane of an affine recursion (api.hip::plan_aff): the operator's own figure, else 8
template <class Op, typename = void> struct AffWaves { static constexpr int value = 8; };
template <class Op> struct AffWaves<Op, decltype((void)Op::kAffWaves)> { static constexpr int value = Op::kAffWaves; };
constexpr int kAffWavesMax = 11;
template <typename R, int D> size_t aff_ws_bytes(const auxssm_ctx* h, int S, int N, int parallel) {
    const AffPlan pl = plan_aff(h, S, N, parallel, kAffWavesMax);  // (the workspace of the operator with the most chunks)
    return (size_t)S * pl.nchunk * (SampElem<R, D>::NPAD + SampPre<R, D>::NPAD + 1) * sizeof(R) + (size_t)pl.nchunk * D * D * sizeof(R) + 1024;
}
// part: [S][nchunk] accumulators of the down pass (may be null); returns the chunk count through *nchunk_out
template <class Op, int D> int run_affine(auxssm_ctx* h, const typename Op::Args& a, int S, int N, int parallel, typename Op::R** part_out, int* nchunk_out) {
    using R = typename Op::R;
    static_assert(AffWaves<Op>::value <= kAffWavesMax, "aff_ws_bytes sizes for kAffWavesMax");
    const AffPlan pl = plan_aff(h, S, N, parallel, AffWaves<Op>::value);
    ScanBufs sb{nullptr, nullptr};
    R* part = part_out ? (R*)ws_take(h, (size_t)S * pl.nchunk * sizeof(R)) : nullptr;
    if (part_out) *part_out = part;
    if (nchunk_out) *nchunk_out = pl.nchunk;
    if (pl.nchunk > 1) {
        R* cprod;
        {
            SideScope side(h);  // the chunk products of the chain-shared matrices belong to the model stage (ctx.h::SideStage) when one is open
            cprod = (R*)ws_take(h, (size_t)pl.nchunk * D * D * sizeof(R));
            if (!cprod) return AUXSSM_ERR_NOMEM;
            hipLaunchKernelGGL((k_aff_chunkprod<Op, D>), dim3((pl.nchunk + TB_CM - 1) / TB_CM), dim3(TB_CM), 0, h->stream, a, cprod, N, pl);
        }
        {
            const int rc = side_close(h);
            if (rc) return rc;
        }
        sb.agg = ws_take(h, (size_t)S * pl.nchunk * SampPre<R, D>::NPAD * sizeof(R));
        sb.pre = ws_take(h, (size_t)S * pl.nchunk * SampPre<R, D>::NPAD * sizeof(R));
        if (!sb.agg || !sb.pre || (part_out && !part)) return AUXSSM_ERR_NOMEM;
        hipLaunchKernelGGL((k_aff_reduce<Op, D>), dim3(grid_aff(S, pl.nchunk)), dim3(TB_CM), 0, h->stream, a, sb, S, N, pl);
        const size_t lds = (size_t)TB_AGGS * SampElem<R, D>::NPAD * sizeof(R);
        hipLaunchKernelGGL((k_aff_aggs<R, D>), dim3(S), dim3(TB_AGGS), lds, h->stream, sb, (const R*)cprod, pl.nchunk);
    }
    hipLaunchKernelGGL((k_aff_down<Op, D>), dim3(grid_aff(S, pl.nchunk)), dim3(TB_CM), 0, h->stream, a, sb, part, S, N, pl);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// ---- few sequences: the scan as TILES OF 256 ELEMENTS, Kogge-Stone inside a tile (round 4) -----------------------------------------------------------------
// The chunked scan above is work-efficient (two combines per element) but DEEP: at one C2 sequence (n = 65535, fp64 d = 4) its three launches walk 11 + (21 + 8 + 21) + 12
// dependent combines of ~3.5 us each -- 280 us during which one to sixteen waves of a 1024-SIMD chip work.  When the whole problem is at most two tiles per CU, depth is
// what costs, not work: every element gets a lane, a workgroup scans its 256 elements in eight Kogge-Stone levels (LDS exchange, as k_scan_aggs), the tile totals are
// scanned by k_scan_aggs (eight more levels, one workgroup per sequence) and a third launch applies each tile's exclusive prefix to the elements' local prefixes:
// 8 + 8 + 1 combines deep, n log2(256) combines of work.  Same operator, another (fixed, deterministic) association order: results agree with the chunked scan to rounding.
constexpr int TB_KS = 256;
// `weight`: the operator's combine relative to the fp32 d = 3 filter combine (measured: that one still gains at four tiles per CU -- C4 at 16 chains 42.6k -> 55.8k
// sweeps/s -- the fp64 d = 4 one, 4.2 times the record and twice the width, at one tile per CU (C2 at one chain 2.05k -> 4.3k) but not at 16: 14.0k -> 12.1k)
inline bool use_ks_scan(const auxssm_ctx* h, int S, int n, int parallel, double weight) {
    static const int mode = [] { const char* e = getenv("AUXSSM_KS_SCAN"); return e ? atoi(e) : 1; }();  // 0 off, 1 auto, 2 always (tests)
    if (!parallel || mode == 0 || n < 2 * TB_KS) return false;
    const long long tiles = (long long)S * ((n + TB_KS - 1) / TB_KS);
    return mode == 2 || (double)tiles * weight <= 10.0 * h->num_cu;
}
template <class Op> inline bool use_ks(const auxssm_ctx* h, int S, int n, int parallel) {
    const double r = (double)Op::Full::NPAD / 28.0;
    return use_ks_scan(h, S, n, parallel, (double)sizeof(typename Op::R) / 4.0 * r * sqrt(r));
}
template <class Op>
__global__ void __launch_bounds__(TB_KS) k_ks_tile(typename Op::Args a, typename Op::R* __restrict__ incl, typename Op::R* __restrict__ tagg, int n, int ntile) {
    if (memo_skip(a)) return;
    resolve_step(a);
    using R = typename Op::R;
    using Full = typename Op::Full;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    R* lds = (R*)smem;
    const int s = blockIdx.x / ntile, tile = blockIdx.x % ntile;
    const int tid = threadIdx.x;
    const int i = tile * TB_KS + tid;
    const bool live = i < n;
    Full acc;
    Op::identity(acc);
    if (live) Op::load_elem(a, s, i, acc);
    for (int off = 1; off < TB_KS; off <<= 1) {
        Op::store_rec(lds + tid * Full::NPAD, acc);
        __syncthreads();
        Full left;
        if (tid >= off) Op::load_rec(lds + (tid - off) * Full::NPAD, left);
        __syncthreads();
        if (tid >= off) {
            Full o;
            Op::combine(left, acc, o);
            acc = o;
        }
    }
    if (live) Op::store_rec(incl + ((long long)s * n + i) * Full::NPAD, acc);
    if (tid == TB_KS - 1) Op::store_rec(tagg + ((long long)s * ntile + tile) * Full::NPAD, acc);  // (dead lanes hold the identity: the last tile's total is its last live prefix)
}
template <class Op>
__global__ void __launch_bounds__(TB_KS) k_ks_down(typename Op::Args a, const typename Op::R* __restrict__ incl, const typename Op::R* __restrict__ tpre, int n, int ntile) {
    if (memo_skip(a)) return;
    resolve_step(a);
    using R = typename Op::R;
    using Full = typename Op::Full;
    using Pre = typename Op::Pre;
    const int s = blockIdx.x / ntile, tile = blockIdx.x % ntile;
    const int i = tile * TB_KS + threadIdx.x;
    if (i >= n) return;
    Pre p, o;
    Full e;
    Op::load_pre(tpre + ((long long)s * ntile + tile) * Pre::NPAD, p);
    Op::load_rec(incl + ((long long)s * n + i) * Full::NPAD, e);
    Op::apply(p, e, o);
    Op::write_out(a, s, i, o);
}

// the tile scan's filter operator builds its elements itself (filtering.py:188-250 inside k_ks_tile: no k_filter_init launch, no element buffer written and read back)
template <typename R_, int D, int P, int P1> struct FilterOpBuild : FilterOp<R_, D> {
    using Full = typename FilterOp<R_, D>::Full;
    static __device__ __forceinline__ void load_elem(const FilterArgs& a, int s, int i, Full& e) {
        DirectIO io;
        filter_build_elem<R_, D, P, DirectIO, P1>(a, io, s, i, true, e);
    }
};

// waves per SIMD of the chain-minor scan passes by element size (plan_scan): measured on the SV second-order sweep (fp64 d = 1: 256 chains 130k -> 161k sweeps/s
// with chunks of 64 instead of 256 steps; 1024 chains the same from 64 to 256) and on the Lorenz sweep (fp32 d = 3, chunk length swept at 64 / 256 / 1024 chains:
// best 12 / 16 / 32 steps = 4 waves, +15 / +17 / +28 % over the one-wave plan); the fp64 d >= 3 operators (264 .. 352 registers) keep the one-wave plan
template <typename R, int D> constexpr int scan_waves() { return sizeof(R) == 4 ? (D <= 3 ? 4 : 2) : (D == 1 ? 4 : D == 2 ? 2 : 1); }
template <class Op> size_t scan_ws_bytes(const auxssm_ctx* h, int S, int n, int parallel) {
    const ScanPlan pl = plan_scan(h, S, n, parallel, SCAN_WAVES_MAX);  // (the most chunks any plan of this shape has)
    size_t b = 0;
    if (use_ks<Op>(h, S, n, parallel)) {
        const size_t ntile = (size_t)(n + TB_KS - 1) / TB_KS;
        b += ((size_t)S * n * Op::Full::NPAD + (size_t)S * ntile * (Op::Full::NPAD + Op::Pre::NPAD)) * sizeof(typename Op::R) + 1024;
    }
    if (pl.nchunk <= 1) return b;
    return b + (size_t)S * pl.nchunk * (Op::Full::NPAD + Op::Pre::NPAD) * sizeof(typename Op::R) + 512;
}
inline ScanLayout make_layout(const ScanPlan& pl, int cm, int S) {
    const int W = pl.nchunk < 64 ? pl.nchunk : 64;
    return ScanLayout{pl.E, pl.nchunk, (pl.nchunk + W - 1) / W, W, cm, S};
}
// time steps per wave in the chain-minor elementwise kernels (AUXSSM_TI overrides, for tuning)
inline int ti_cm() {
    static int v = [] { const char* e = getenv("AUXSSM_TI"); const int t = e ? atoi(e) : 16; return t >= 1 && t <= 1024 ? t : 16; }();
    return v;
}
#define TI_CM ti_cm()
// ... of the per-chain log-density passes: longer runs once the grid holds ~32 waves per SIMD anyway (their per-tile costs -- the t - 1 reads, one logarithm per
// sum -- amortise; SV second order, 1024 chains x 65536 steps: 244k -> 267k sweeps/s at 64 steps, Lorenz 256 chains x 16384: best at 16); never below TI_CM,
// which sizes the partial-sum buffers
inline int ti_cm_for(const auxssm_ctx* h, int C, int n) {
    static const bool forced = getenv("AUXSSM_TI") != nullptr;
    if (forced) return TI_CM;
    const long long t = (long long)C * n / ((long long)64 * h->num_cu * 4 * 32);
    return (int)std::min<long long>(64, std::max<long long>(TI_CM, t));
}
// ... and of the chain-shared log-density pass (run_sweep_logpdf; AUXSSM_TI_SHARED overrides)
inline int ti_shared() {
    static int v = [] { const char* e = getenv("AUXSSM_TI_SHARED"); const int t = e ? atoi(e) : 64; return t >= 1 && t <= 1024 ? t : 64; }();
    return v;
}

// `a` must already carry the layout the element buffer was written with (make_layout(plan_scan(...)))
// DownOp: the operator of the final pass (same element/prefix types as Op; e.g. SampleOpFly, which rebuilds its elements)
template <class Op, class DownOp = Op, class ReduceOp = Op>
int run_scan(auxssm_ctx* h, const typename Op::Args& a, int S, int n) {
    using R = typename Op::R;
    if (n <= 0 || S <= 0) return AUXSSM_OK;
    const ScanLayout lay = Op::layout(a);
    if (!lay.cm && std::is_same<Op, DownOp>::value && std::is_same<Op, ReduceOp>::value && use_ks<Op>(h, S, n, lay.nchunk > 1)) {
        // few sequences: tiles of 256 elements, Kogge-Stone inside a tile (above)
        const int ntile = (n + TB_KS - 1) / TB_KS;
        R* incl = (R*)ws_take(h, (size_t)S * n * Op::Full::NPAD * sizeof(R));
        ScanBufs tb{ws_take(h, (size_t)S * ntile * Op::Full::NPAD * sizeof(R)), ws_take(h, (size_t)S * ntile * Op::Pre::NPAD * sizeof(R)), a.memo};
        if (!incl || !tb.agg || !tb.pre) return AUXSSM_ERR_NOMEM;
        const size_t lds = (size_t)TB_KS * Op::Full::NPAD * sizeof(R);
        static_assert(TB_KS == TB_AGGS, "k_scan_aggs shares the tile's LDS plan");
        hipLaunchKernelGGL((k_ks_tile<Op>), dim3((unsigned)S * ntile), dim3(TB_KS), lds, h->stream, a, incl, (R*)tb.agg, n, ntile);
        hipLaunchKernelGGL((k_scan_aggs<Op>), dim3(S), dim3(TB_AGGS), lds, h->stream, tb, ntile);
        hipLaunchKernelGGL((k_ks_down<Op>), dim3((unsigned)S * ntile), dim3(TB_KS), 0, h->stream, a, (const R*)incl, (const R*)tb.pre, n, ntile);
        AX_HIP(hipGetLastError());
        return AUXSSM_OK;
    }
    ScanBufs sb{nullptr, nullptr, a.memo};
    const unsigned grid = lay.cm ? (unsigned)((S + TB_CM - 1) / TB_CM) * lay.nchunk : (unsigned)S * lay.ngrp;
    const size_t stage = stage_bytes<R>(Op::Full::NPAD);
    if (lay.nchunk > 1) {
        sb.agg = ws_take(h, (size_t)S * lay.nchunk * Op::Full::NPAD * sizeof(R));
        sb.pre = ws_take(h, (size_t)S * lay.nchunk * Op::Pre::NPAD * sizeof(R));
        if (lay.cm) hipLaunchKernelGGL((k_scan_reduce_cm<ReduceOp>), dim3(grid), dim3(TB_CM), 0, h->stream, a, sb, S, n);
        else hipLaunchKernelGGL((k_scan_reduce<Op>), dim3(grid), dim3(TB_SCAN), stage, h->stream, a, sb, n);
        const size_t lds = (size_t)TB_AGGS * Op::Full::NPAD * sizeof(R);
        hipLaunchKernelGGL((k_scan_aggs<Op>), dim3(S), dim3(TB_AGGS), lds, h->stream, sb, lay.nchunk);
    }
    if (lay.cm) hipLaunchKernelGGL((k_scan_down_cm<DownOp>), dim3(grid), dim3(TB_CM), 0, h->stream, a, sb, S, n);
    else hipLaunchKernelGGL((k_scan_down<Op>), dim3(grid), dim3(TB_SCAN), stage, h->stream, a, sb, n);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// ---- launchers ---------------------------------------------------------------------------------------------------
inline int ntiles(int n) { return (n + TB_ELEM - 1) / TB_ELEM; }

template <typename R, int D, int P> size_t filter_ws(const auxssm_ctx* h, const KDims& d, int parallel) {
    const int S = d.S(), n = d.n();
    const ScanLayout lay = make_layout(plan_scan(h, S, n, parallel), 0, S);
    size_t b = 0;
    // (the time-minor layout is never smaller than the chain-minor one)
    b += (size_t)S * lay.seq_records() * FiltElem<R, D>::NPAD * sizeof(R) + 256;
    b += (size_t)S * sizeof(R) + 256;                           // ell0
    const int nchunk_cm = make_layout(plan_scan(h, S, n, parallel, SCAN_WAVES_MAX), 1, S).nchunk;
    b += (size_t)S * (std::max(ntiles(n), std::max(lay.nchunk, nchunk_cm)) + 1) * sizeof(R) + 256;  // ell partials (per tile, or per chunk of either layout)
    b += scan_ws_bytes<FilterOp<R, D>>(h, S, n, parallel);
    // chain-shared parameters: the one-sequence matrix filter (elements, scan buffers, moments, mask carrier), the gain table and the
    // per-chain affine scan
    const ScanLayout l1 = make_layout(plan_scan(h, 1, n, 1), 0, 1);
    b += (size_t)l1.seq_records() * FiltElem<R, D>::NPAD * sizeof(R) + scan_ws_bytes<FilterOp<R, D>>(h, 1, n, 1) + 1024;
    b += (size_t)d.T * (D + D * D + P) * sizeof(R) + 1024;
    b += (size_t)(n > 0 ? n : 1) * GainRow<R, D, P>::NPAD * sizeof(R) + 256;
    b += aff_ws_bytes<R, D>(h, S, n, parallel);
    b += (size_t)(n > 0 ? n : 1) * ObsInfoRow<R, D>::NPAD * sizeof(R) + 256;
    return b;
}

// the chain-shared half of the filter: the matrix filter on ONE sequence and the gain table derived from it (a.tab).  Inside a sweep that
// opened a side stage (ctx.h::SideStage) all of it -- launches, scratch and the table -- goes to the side stream and its slab.
template <typename R, int D, int P> int build_gain_table(auxssm_ctx* h, FilterArgs& a) {
    const int n = a.d.n(), T = a.d.T;
    // the covariances are chain-independent: the caller lays them out once, (T, D, D) dense with chain stride 0 (ctx.h::chain_shared_mode),
    // and the matrix filter writes them in place; any other layout gets chain 0's slot filled from a scratch copy
    const bool ps_once = a.Ps.sc == 0 && a.Ps.se == 1 && a.Ps.st == (long long)D * D;
    if (a.tab && a.tab_ready) return AUXSSM_OK;
    // the matrix filter: the parallel filter on ONE sequence, always the parallel plan (its means are not used: the observation
    // values are the mask carrier's).  Dense (time-minor) layout.
    SideScope side(h);
    if (side.on && !ps_once) {
        set_error("internal: the side stage needs the shared covariance layout");
        return AUXSSM_ERR_ARG;
    }
    R* ms1 = (R*)ws_take(h, (size_t)T * D * sizeof(R));
    R* Ps1 = ps_once ? const_cast<R*>((const R*)a.Ps.ptr) : (R*)ws_take(h, (size_t)T * D * D * sizeof(R));
    R* sc1 = (R*)ws_take(h, 4 * sizeof(R));
    if (!ms1 || !Ps1 || !sc1) return AUXSSM_ERR_NOMEM;
    ProfScope ps(h, AUXSSM_K_FILTER_TAB);
    FilterArgs am = a;
    am.d = KDims{1, T, 1};
    am.aux_on = 0;
    am.tab = nullptr;
    am.pc = nullptr;
    am.ms = dense_arr(ms1, am.d, D);
    am.Ps = dense_arr(Ps1, am.d, (long long)D * D);
    am.ell0 = sc1;
    am.ellz = sc1 + 1;
    if (a.aux_on) {
        R* ym = (R*)ws_take(h, (size_t)T * P * sizeof(R));
        if (!ym) return AUXSSM_ERR_NOMEM;
        const long long tot = (long long)T * P;
        hipLaunchKernelGGL((k_mask_obs<R>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, T, D, P, a.aux_yobs, ym, a.memo);
        am.ys = dense_arr(ym, am.d, P);
    } else if (a.mask_ys.ptr) {
        am.ys = a.mask_ys;  // a chain-independent mask carrier
    } else {
        am.ys = Arr{a.ys.ptr, 0, a.ys.st, 0, a.ys.se};  // chain 0's observations
    }
    am.lay = make_layout(plan_scan(h, 1, n, 1), 0, 1);
    const bool ks1 = n > 0 && use_ks<FilterOp<R, D>>(h, 1, n, am.lay.nchunk > 1);  // (one sequence: the tile scan builds its own elements)
    am.elem = ks1 ? nullptr : ws_take(h, (size_t)am.lay.total_reals(n, 1, FiltElem<R, D>::NPAD) * sizeof(R));
    if (!ks1 && !am.elem) return AUXSSM_ERR_NOMEM;
    hipLaunchKernelGGL((k_filter_t0<R, D, P>), dim3(1), dim3(TB_ELEM), 0, h->stream, am);
    if (!ks1) hipLaunchKernelGGL((k_filter_init<R, D, P>), dim3(grid_tile_seq(ntiles(n), 1)), dim3(TB_ELEM), 0, h->stream, am);
    const int rc = ks1 ? run_scan<FilterOpBuild<R, D, P, 0>>(h, am, 1, n) : run_scan<FilterOp<R, D>>(h, am, 1, n);
    if (rc) return rc;
    a.tab = ws_take(h, (size_t)n * GainRow<R, D, P>::NPAD * sizeof(R));
    if (!a.tab) return AUXSSM_ERR_NOMEM;
    hipLaunchKernelGGL((k_gain_tab<R, D, P>), dim3((n + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, a, (const R*)Ps1);
    if (side.on) h->side.last_tab = a.tab;
    if (!ps_once)  // chain 0's slot of the caller's buffer (the sampler's table reads that one)
        hipLaunchKernelGGL((k_copy_cov<R, D>), dim3((unsigned)(((long long)T * D * D + 255) / 256)), dim3(256), 0, h->stream, T, (const R*)Ps1, a.Ps, a.memo);
    return AUXSSM_OK;
}

// Chain-shared model parameters (affine_shared.h): the d x d block-affine scan runs once, on one sequence (the matrix filter), the
// chains carry an affine recursion with chain-shared matrices.  `a` holds the chain-minor views of ys / ms / Ps; ell0 is filled.
template <typename R, int D, int P> int run_filter_shared(auxssm_ctx* h, FilterArgs& a, int parallel, void* ell_out) {
    const int S = a.d.S(), n = a.d.n();
    const bool ps_once = a.Ps.sc == 0 && a.Ps.se == 1 && a.Ps.st == (long long)D * D;
    {
        const int rc = build_gain_table<R, D, P>(h, a);
        if (rc) return rc;
    }
    {
        const int rc = side_close(h);  // (no-op without a side stage) `stream` waits for the model stage before the first kernel that reads its products
        if (rc) return rc;
    }
    // t = 0 update of every chain (after the join: it reads the concatenated model).  The one shared Ps[0] slot is the matrix filter's: the chains do
    // not rewrite it (same value; and the sampler's table may be reading it on the side stream by now)
    a.t0_keep_ps = ps_once ? 1 : 0;
    hipLaunchKernelGGL((k_filter_t0<R, D, P>), dim3((S + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, a);
    R* part = nullptr;
    int nchunk = 1;
    {
        ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
        const int rc = run_affine<FilterMeanOp<R, D, P>, D>(h, a, S, n, parallel, &part, &nchunk);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((k_reduce_rows<R>), dim3(a.d.C), dim3(TB_ELEM), 0, h->stream, (const R*)part, (const R*)a.ell0, a.d.B, nchunk, (R*)ell_out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

template <typename R, int D, int P> int run_filter(auxssm_ctx* h, const FilterArgs& a_in, int parallel, void* ell_out) {
    FilterArgs a = a_in;
    const int S = a.d.S(), n = a.d.n();
    const int cm = a_in.lay.cm;  // the caller chose the layout of ms / Ps; the element buffer follows it
    // block-diagonal R with a leading dx x dx block (the concatenated auxiliary observations): information form
    static const bool blk_on = [] { const char* e = getenv("AUXSSM_INFO_BLOCKS"); return e ? atoi(e) != 0 : true; }();
    const bool blk = blk_on && P > D && a_in.pblk == D;
    a.lay = make_layout(plan_scan(h, S, n, parallel, cm ? scan_waves<R, D>() : 1), cm, S);
    // chain-shared model parameters (the factories of a linear-Gaussian model): what jax.vmap leaves unbatched in the reference
    const bool shared_on = h->share_model != 0;
    const bool shared = shared_on && cm && n > 0 && a.d.B == 1 && a.d.C > 1 && a.Fs.sc == 0 && a.Qs.sc == 0 && a.bs.sc == 0 && a.Hs.sc == 0 &&
                        a.Rs.sc == 0 && a.cs.sc == 0 && a.P0.sc == 0;
    a.ell0 = ws_take(h, (size_t)S * sizeof(R));
    if (!a.ell0) return AUXSSM_ERR_NOMEM;
    if (a_in.sv_order != 0) {
        // SV factories, chain-minor, per-chain observation model: no observation arrays and no elements -- both scan passes fold the steps in information
        // form from (x_lin, u, y) (kalman_bodies.h::FilterOpFlySV); the down pass of the proposal filter writes u and the filtered moments, the reverse
        // filter's (no_moments) only the log-likelihood increments
        if constexpr (P == D) {
            if (!cm || a.d.B != 1) {
                set_error("internal: the array-free SV filter needs the chain-minor layout");
                return AUXSSM_ERR_ARG;
            }
            hipLaunchKernelGGL((k_filter_t0_sv<R, D>), dim3((S + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, a);
            if (n > 0) {
                a.elem = nullptr;
                a.ellz = ws_take(h, (size_t)S * a.lay.nchunk * sizeof(R));
                if (!a.ellz) return AUXSSM_ERR_NOMEM;
                ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
                const int rc = a.aux_eps.ptr ? run_scan<FilterOp<R, D>, FilterOpFlySV<R, D, true>, FilterOpFlySV<R, D, false>>(h, a, S, n)
                                             : run_scan<FilterOp<R, D>, FilterOpFlySV<R, D, false>, FilterOpFlySV<R, D, false>>(h, a, S, n);
                if (rc) return rc;
            }
            hipLaunchKernelGGL((k_reduce_rows<R>), dim3(a.d.C), dim3(TB_ELEM), 0, h->stream, (const R*)a.ellz, (const R*)a.ell0, a.d.B, n > 0 ? a.lay.nchunk : 0, (R*)ell_out);
            AX_HIP(hipGetLastError());
            return AUXSSM_OK;
        } else {
            set_error("internal: the SV factories observe every state component (dy == dx)");
            return AUXSSM_ERR_ARG;
        }
    }
    if (shared) return run_filter_shared<R, D, P>(h, a, parallel, ell_out);
    hipLaunchKernelGGL((k_filter_t0<R, D, P>), dim3((S + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, a);
    // general chain-minor path on the concatenated auxiliary model: elements are built on the fly inside both scan passes
    // (FilterArgs::aux_on is the caller's promise that ys holds row t = 0 only and that the model is the concatenated one)
    const bool fly = a_in.aux_on != 0 && n > 0;
    if (fly && !(cm && P > D && a_in.pblk == D && a.d.B == 1 && a.Hs.sc == 0 && a.Rs.sc == 0 && a.cs.sc == 0 && a_in.aux_yobs.sc == 0)) {
        set_error("internal: on-the-fly auxiliary observations need the chain-minor layout and a chain-shared concatenated observation model");
        return AUXSSM_ERR_ARG;
    }
    if (fly) {
        if constexpr (P > D) {
            a.elem = nullptr;
            a.obs_tab = ws_take(h, (size_t)n * ObsInfoRow<R, D>::NPAD * sizeof(R));
            a.ellz = ws_take(h, (size_t)S * a.lay.nchunk * sizeof(R));  // [sequence][chunk]: the down pass's per-chunk sums of the log-likelihood increments
            if (!a.obs_tab || !a.ellz) return AUXSSM_ERR_NOMEM;
            {
                ProfScope ps(h, AUXSSM_K_FILTER_TAB);
                hipLaunchKernelGGL((k_obs_info_tab<R, D, P>), dim3((n + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, a);
            }
            {
                ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
                const int rc = run_scan<FilterOp<R, D>, FilterOpFly<R, D, P, false>, FilterOpFly<R, D, P, false>>(h, a, S, n);
                if (rc) return rc;
            }
            hipLaunchKernelGGL((k_reduce_rows<R>), dim3(a.d.C), dim3(TB_ELEM), 0, h->stream, (const R*)a.ellz, (const R*)a.ell0, a.d.B, a.lay.nchunk, (R*)ell_out);
            AX_HIP(hipGetLastError());
            return AUXSSM_OK;
        }
    }
    // lanes <-> sequences on any model (the caller's arrays through their strides): no element buffer -- composites from elements built in registers, then the
    // sequential recursion from each chunk's prefix (kalman_bodies.h::FilterOpBuildCm / FilterOpSeqWalk).  AUXSSM_CM_ELEM=1: the materialised-element passes.
    static const bool cm_elem = [] { const char* e = getenv("AUXSSM_CM_ELEM"); return e && atoi(e) != 0; }();
    if (cm && n > 0 && !cm_elem && !a.ps_packed) {
        a.elem = nullptr;
        a.ellz = ws_take(h, (size_t)S * a.lay.nchunk * sizeof(R));
        if (!a.ellz) return AUXSSM_ERR_NOMEM;
        {
            ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
            const int rc = blk ? run_scan<FilterOp<R, D>, FilterOpSeqWalk<R, D, P>, FilterOpBuildCm<R, D, P, (P > D ? D : 0)>>(h, a, S, n)
                               : run_scan<FilterOp<R, D>, FilterOpSeqWalk<R, D, P>, FilterOpBuildCm<R, D, P, 0>>(h, a, S, n);
            if (rc) return rc;
        }
        hipLaunchKernelGGL((k_reduce_rows<R>), dim3(a.d.C), dim3(TB_ELEM), 0, h->stream, (const R*)a.ellz, (const R*)a.ell0, a.d.B, a.lay.nchunk, (R*)ell_out);
        AX_HIP(hipGetLastError());
        return AUXSSM_OK;
    }
    const bool ks = !cm && n > 0 && use_ks<FilterOp<R, D>>(h, S, n, a.lay.nchunk > 1);  // few sequences: the tile scan, which builds its own elements (FilterOpBuild)
    a.elem = ks ? nullptr : ws_take(h, (size_t)a.lay.total_reals(n, S, FiltElem<R, D>::NPAD) * sizeof(R));
    // the marginal log-likelihood of t = 1..T-1 is the log-scale of the scan's total product (kalman_math.h::FiltElem::z): the final
    // pass writes it per sequence; the reference's second pass over the filtered moments (filtering.py:60-62) does not exist here
    a.ellz = ws_take(h, (size_t)S * sizeof(R));
    const int nt = cm ? a.lay.nchunk : ntiles(n);
    if (ks) {
        ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
        const int rc = blk ? run_scan<FilterOpBuild<R, D, P, (P > D ? D : 0)>>(h, a, S, n) : run_scan<FilterOpBuild<R, D, P, 0>>(h, a, S, n);
        if (rc) return rc;
    } else if (n > 0) {
        {
            ProfScope ps(h, AUXSSM_K_FILTER_INIT);
            if (cm && blk) hipLaunchKernelGGL((k_filter_init_cm<R, D, P, (P > D ? D : 0)>), dim3(grid_cm(S, n, TI_CM)), dim3(TB_CM), 0, h->stream, a, TI_CM);
            else if (cm) hipLaunchKernelGGL((k_filter_init_cm<R, D, P, 0>), dim3(grid_cm(S, n, TI_CM)), dim3(TB_CM), 0, h->stream, a, TI_CM);
            else hipLaunchKernelGGL((k_filter_init<R, D, P>), dim3(grid_tile_seq(nt, S)), dim3(TB_ELEM), 0, h->stream, a);
        }
        {
            ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
            const int rc = run_scan<FilterOp<R, D>>(h, a, S, n);
            if (rc) return rc;
        }
    }
    if (sizeof(R) == 4 && !cm && n > 0 && !a.ps_packed) {  // fp32, time-minor element path: ell by the reference's second pass (k_ell_pass) instead of the scan's log-scale
        const int ntl = ntiles(n);
        R* part = (R*)ws_take(h, (size_t)S * ntl * sizeof(R));
        if (!part) return AUXSSM_ERR_NOMEM;
        // (tried: this pass on a fork stream beside the sampler / the next filter -- nothing reads ell before the accept step.  The two event hops per filter cost
        // more than the 16 us they hide: C4 at 8 chains 37.9k -> 35.0k sweeps/s.  Not kept.)
        hipLaunchKernelGGL((k_ell_pass<R, D, P>), dim3(grid_tile_seq(ntl, S)), dim3(TB_ELEM), 0, h->stream, a, part, ntl);
        hipLaunchKernelGGL((k_reduce_rows<R>), dim3(a.d.C), dim3(TB_ELEM), 0, h->stream, (const R*)part, (const R*)a.ell0, a.d.B, ntl, (R*)ell_out);
        AX_HIP(hipGetLastError());
        return AUXSSM_OK;
    }
    hipLaunchKernelGGL((k_reduce_rows<R>), dim3(a.d.C), dim3(TB_ELEM), 0, h->stream, (const R*)a.ellz, (const R*)a.ell0, a.d.B, n > 0 ? 1 : 0,
                       (R*)ell_out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

template <typename R, int D> size_t sample_ws(const auxssm_ctx* h, const KDims& d, int parallel) {
    const int S = d.S();
    const ScanLayout lay = make_layout(plan_scan(h, S, d.T, parallel), 0, S);
    return (size_t)S * lay.seq_records() * SampElem<R, D>::NPAD * sizeof(R) + 256 + scan_ws_bytes<SampleOp<R, D>>(h, S, d.T, parallel) +
           (size_t)d.T * SampShared<R, D>::NPAD * sizeof(R) + 256 + aff_ws_bytes<R, D>(h, S, d.T, parallel);
}

template <typename R, int D> int run_sample(auxssm_ctx* h, const SampleArgs& a_in, int parallel) {
    SampleArgs a = a_in;
    const int S = a.d.S(), T = a.d.T;
    const int cm = a_in.lay.cm;
    a.lay = make_layout(plan_scan(h, S, T, parallel, cm ? scan_waves<R, D>() : 1), cm, S);
    static const bool fly = [] { const char* e = getenv("AUXSSM_SAMPLE_FLY"); return e ? atoi(e) != 0 : true; }();
    const bool shared_on = h->share_model != 0;
    if (cm && shared_on && a.ps_shared && a.d.B == 1 && a.Fs.sc == 0 && a.Qs.sc == 0 && a.bs.sc == 0) {
        // chain-shared covariances: gains and Cholesky factors once per time step, a chain's element is two small mat-vecs
        a.elem = nullptr;
        {
            SideScope side(h);  // (model stage, when the sweep opened one: the table reads the shared covariances and the model only)
            a.tab = ws_take(h, (size_t)T * SampShared<R, D>::NPAD * sizeof(R));
            if (!a.tab) return AUXSSM_ERR_NOMEM;
            ProfScope ps(h, AUXSSM_K_SAMPLE_INIT);
            hipLaunchKernelGGL((k_sample_shared_tab<R, D>), dim3((T + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, a);
        }
        ProfScope ps(h, AUXSSM_K_SAMPLE_SCAN);
        const int rc = run_affine<SampleAffOp<R, D>, D>(h, a, S, T, parallel, nullptr, nullptr);
        if (rc) return rc;
        AX_HIP(hipGetLastError());
        return AUXSSM_OK;
    }
    if (a.ps_packed && !(cm && fly)) {
        set_error("internal: packed covariance records are read by the on-the-fly chain-minor sampler only");
        return AUXSSM_ERR_UNSUPPORTED;
    }
    if (cm && fly) {
        // chain-minor: elements are recomputed on the fly by both scan passes (SampleOpFly), nothing to initialise
        a.elem = nullptr;
        ProfScope ps(h, AUXSSM_K_SAMPLE_SCAN);
        const int rc = run_scan<SampleOp<R, D>, SampleOpFly<R, D>, SampleOpFly<R, D>>(h, a, S, T);
        if (rc) return rc;
        AX_HIP(hipGetLastError());
        return AUXSSM_OK;
    }
    if (!cm && !a.ps_packed && use_ks<SampleOp<R, D>>(h, S, T, a.lay.nchunk > 1)) {
        // few sequences: the tile scan builds its own elements (SampleOpFly::load_elem): no k_sample_init launch, no element buffer
        a.elem = nullptr;
        ProfScope ps(h, AUXSSM_K_SAMPLE_SCAN);
        const int rc = run_scan<SampleOpFly<R, D>>(h, a, S, T);
        if (rc) return rc;
        AX_HIP(hipGetLastError());
        return AUXSSM_OK;
    }
    a.elem = ws_take(h, (size_t)a.lay.total_reals(T, S, SampElem<R, D>::NPAD) * sizeof(R));
    {
        ProfScope ps(h, AUXSSM_K_SAMPLE_INIT);
        const int nt = ntiles(T - 1) > 0 ? ntiles(T - 1) : 1;
        if (cm) hipLaunchKernelGGL((k_sample_init_cm<R, D>), dim3(grid_cm(S, T - 1, TI_CM)), dim3(TB_CM), 0, h->stream, a, TI_CM);
        else hipLaunchKernelGGL((k_sample_init<R, D>), dim3(grid_tile_seq(nt, S)), dim3(TB_ELEM), 0, h->stream, a);
    }
    {
        ProfScope ps(h, AUXSSM_K_SAMPLE_SCAN);
        const int rc = run_scan<SampleOp<R, D>>(h, a, S, T);
        if (rc) return rc;
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

template <typename R, int D, int P> size_t logpdf_ws(const auxssm_ctx*, const KDims& d) {
    return (size_t)d.S() * (ntiles(d.T) + 1) * sizeof(R) + 256;
}

template <typename R, int D, int P> int run_logpdf(auxssm_ctx* h, const LogpdfArgs& a, void* out) {
    const int S = a.d.S(), T = a.d.T;
    const int nt = ntiles(T - 1) > 0 ? ntiles(T - 1) : 1;
    R* part = (R*)ws_take(h, (size_t)S * nt * sizeof(R));
    ProfScope ps(h, AUXSSM_K_LOGPDF);
    hipLaunchKernelGGL((k_joint_logpdf<R, D, P>), dim3(grid_tile_seq(nt, S)), dim3(TB_ELEM), 0, h->stream, a, part, nt);
    hipLaunchKernelGGL((k_reduce_rows<R>), dim3(a.d.C), dim3(TB_ELEM), 0, h->stream, (const R*)part, (const R*)nullptr,
                       a.d.B, nt, (R*)out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

template <typename R, int D, int PO> size_t sweep_logpdf_ws(const auxssm_ctx*, const KDims& d) {
    const int ti = std::min(TI_CM, ti_shared());
    return (size_t)5 * d.C * (std::max(ntiles(d.T), (d.T + ti - 1) / ti) + 1) * sizeof(Acc) + 256 +
           (size_t)d.T * LogShared<R, D, PO>::NPAD * sizeof(R) + 256;
}
// out: [5][C] of Acc
template <typename R, int D, int PO> int run_sweep_logpdf(auxssm_ctx* h, const SweepLogpdfArgs& a, void* out) {
    const bool cm = a.xp.se != 1;  // chain-minor proposal buffer -> lanes over chains
    const int n = a.d.T - 1;
    const bool shared_on = h->share_model != 0;
    const bool shared = shared_on && cm && n > 0 && a.d.C > 1 && a.Fs.sc == 0 && a.Qs.sc == 0 && a.bs.sc == 0 && a.Hs.sc == 0 && a.Rs.sc == 0 &&
                        a.cs.sc == 0 && a.ys.sc == 0;
    // time steps per lane: the streamed shared pass (52 scalars per step from its table) likes longer runs than the per-chain pass -- measured at C2,
    // two runs each: 0.41-0.44 / 0.39 / 0.38-0.41 / 0.38-0.40 ms at 16 / 32 / 48 / 64 steps; the per-chain pass 1.28 / 1.31 / 1.37 / 1.34 ms
    // the SEMI-shared form: covariances, observation model and data common to the chains (chain stride 0), whatever the transition's F, b (AUXSSM_LOGPDF_SEMI=0: off)
    static const bool semi_on = [] { const char* e = getenv("AUXSSM_LOGPDF_SEMI"); return e ? atoi(e) != 0 : true; }();
    const bool semi = semi_on && cm && !shared && n > 0 && a.d.C > 1 && a.Qs.sc == 0 && a.Hs.sc == 0 && a.Rs.sc == 0 && a.cs.sc == 0 && a.ys.sc == 0;
    const int TI = shared ? ti_shared() : ti_cm_for(h, a.d.C, n);
    const int C = a.d.C, nt = cm ? ((n + TI - 1) / TI > 0 ? (n + TI - 1) / TI : 1) : (ntiles(n) > 0 ? ntiles(n) : 1);
    Acc* part = (Acc*)ws_take(h, (size_t)5 * C * nt * sizeof(Acc));
    ProfScope ps(h, AUXSSM_K_LOGPDF);
    if (shared) {  // chain-shared parameters: factor Q_{t-1} and Robs_t once per time step
        SweepLogpdfArgs as = a;
        {
            SideScope side(h);  // (model stage)
            as.tab = ws_take(h, (size_t)n * LogShared<R, D, PO>::NPAD * sizeof(R));
            if (!as.tab) return AUXSSM_ERR_NOMEM;
            hipLaunchKernelGGL((k_sweep_logpdf_tab<R, D, PO>), dim3((n + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, as);
        }
        {
            const int rc = side_close(h);
            if (rc) return rc;
        }
        hipLaunchKernelGGL((k_sweep_logpdf_cm_shared<R, D, PO>), dim3(grid_cm(C, n, TI)), dim3(TB_CM), 0, h->stream, as, part, nt, TI);
    } else if (cm && semi) {  // the chains' common covariances / observation model: their whitening rows once per time step, the transition mean per chain
        SweepLogpdfArgs as = a;
        as.tab_semi = 1;
        as.tab = ws_take(h, (size_t)n * LogShared<R, D, PO>::NPAD * sizeof(R));
        if (!as.tab) return AUXSSM_ERR_NOMEM;
        hipLaunchKernelGGL((k_sweep_logpdf_tab<R, D, PO>), dim3((n + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, as);
        hipLaunchKernelGGL((k_sweep_logpdf_cm_semi<R, D, PO, false>), dim3(grid_cm(C, n, TI)), dim3(TB_CM), 0, h->stream, as, part, nt, TI);
    } else if (cm) hipLaunchKernelGGL((k_sweep_logpdf_cm<R, D, PO>), dim3(grid_cm(C, n, TI)), dim3(TB_CM), 0, h->stream, a, part, nt, TI);
    else hipLaunchKernelGGL((k_sweep_logpdf<R, D, PO>), dim3(grid_tile_seq(nt, C)), dim3(TB_ELEM), 0, h->stream, a, part, nt);
    hipLaunchKernelGGL((k_reduce_rows<Acc>), dim3(5 * C), dim3(TB_ELEM), 0, h->stream, (const Acc*)part, (const Acc*)nullptr, 1, nt, (Acc*)out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

template <typename R, int D> size_t sv_logpdf_ws(const auxssm_ctx*, const KDims& d) {
    return (size_t)5 * d.C * (std::max(ntiles(d.T), (d.T + TI_CM - 1) / TI_CM) + 1) * sizeof(Acc) + 256;
}
// out: [5][C] of Acc
template <typename R, int D> int run_sv_logpdf(auxssm_ctx* h, const SvLogpdfArgs& a, void* out) {
    const bool cm = a.xp.se != 1;
    const int n = a.d.T - 1, C = a.d.C;
    const int TI = ti_cm_for(h, C, n);
    const int nt = cm ? ((n + TI - 1) / TI > 0 ? (n + TI - 1) / TI : 1) : (ntiles(n) > 0 ? ntiles(n) : 1);
    Acc* part = (Acc*)ws_take(h, (size_t)5 * C * nt * sizeof(Acc));
    ProfScope ps(h, AUXSSM_K_LOGPDF);
    if (cm) hipLaunchKernelGGL((k_sv_logpdf_cm<R, D>), dim3(grid_cm(C, n, TI)), dim3(TB_CM), 0, h->stream, a, part, nt, TI);
    else hipLaunchKernelGGL((k_sv_logpdf<R, D>), dim3(grid_tile_seq(nt, C)), dim3(TB_ELEM), 0, h->stream, a, part, nt);
    hipLaunchKernelGGL((k_reduce_rows<Acc>), dim3(5 * C), dim3(TB_ELEM), 0, h->stream, (const Acc*)part, (const Acc*)nullptr, 1, nt, (Acc*)out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// out: [5][C] of Acc; instantiated for D = 3 only (null entry otherwise)
template <typename R, int PO> int run_lorenz_logpdf(auxssm_ctx* h, const SweepLogpdfArgs& a, void* out) {
    const bool cm = a.xp.se != 1;
    const int n = a.d.T - 1, C = a.d.C;
    const int TI = ti_cm_for(h, C, n);
    const int nt = cm ? ((n + TI - 1) / TI > 0 ? (n + TI - 1) / TI : 1) : (ntiles(n) > 0 ? ntiles(n) : 1);
    Acc* part = (Acc*)ws_take(h, (size_t)5 * C * nt * sizeof(Acc));
    ProfScope ps(h, AUXSSM_K_LOGPDF);
    static const bool semi_on = [] { const char* e = getenv("AUXSSM_LOGPDF_SEMI"); return e ? atoi(e) != 0 : true; }();
    const bool semi = semi_on && cm && n > 0 && C > 1 && a.Qs.sc == 0 && a.Hs.sc == 0 && a.Rs.sc == 0 && a.cs.sc == 0 && a.ys.sc == 0;
    if (semi) {
        SweepLogpdfArgs as = a;
        as.tab_semi = 1;
        as.tab = ws_take(h, (size_t)n * LogShared<R, 3, PO>::NPAD * sizeof(R));
        if (!as.tab) return AUXSSM_ERR_NOMEM;
        hipLaunchKernelGGL((k_sweep_logpdf_tab<R, 3, PO>), dim3((n + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, as);
        hipLaunchKernelGGL((k_sweep_logpdf_cm_semi<R, 3, PO, true>), dim3(grid_cm(C, n, TI)), dim3(TB_CM), 0, h->stream, as, part, nt, TI);
    } else if (cm) hipLaunchKernelGGL((k_lorenz_logpdf_cm<R, PO>), dim3(grid_cm(C, n, TI)), dim3(TB_CM), 0, h->stream, a, part, nt, TI);
    else hipLaunchKernelGGL((k_lorenz_logpdf<R, PO>), dim3(grid_tile_seq(nt, C)), dim3(TB_ELEM), 0, h->stream, a, part, nt);
    hipLaunchKernelGGL((k_reduce_rows<Acc>), dim3(5 * C), dim3(TB_ELEM), 0, h->stream, (const Acc*)part, (const Acc*)nullptr, 1, nt, (Acc*)out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}
template <typename R, int D, int PO> constexpr sweep_logpdf_fn lorenz_logpdf_entry() {
    if constexpr (D == 3 && PO <= 3) return &run_lorenz_logpdf<R, PO>;
    else return nullptr;
}

#include "fused_shared.h"

// one instantiation unit = one (dtype, D): all P for the filter / logpdf, plus the sampler
#define AX_KALMAN_ENTRY(R, D, P) \
    { &run_filter<R, D, P>, &filter_ws<R, D, P>, &run_logpdf<R, D, P>, &logpdf_ws<R, D, P> }

#define AX_DEFINE_UNIT(NAME, R, D)                                                                   \
    namespace ax {                                                                                   \
    const KalmanEntry* kalman_unit_##NAME(int P) {                                                   \
        static const KalmanEntry tab[MAX_P] = {AX_KALMAN_ENTRY(R, D, 1), AX_KALMAN_ENTRY(R, D, 2),  \
                                               AX_KALMAN_ENTRY(R, D, 3), AX_KALMAN_ENTRY(R, D, 4),  \
                                               AX_KALMAN_ENTRY(R, D, 5), AX_KALMAN_ENTRY(R, D, 6),  \
                                               AX_KALMAN_ENTRY(R, D, 7), AX_KALMAN_ENTRY(R, D, 8)}; \
        return (P >= 1 && P <= MAX_P) ? &tab[P - 1] : nullptr;                                       \
    }                                                                                                \
    const SweepLogpdfEntry* sweep_logpdf_unit_##NAME(int PO) {                                       \
        static const SweepLogpdfEntry tab[4] = {{&run_sweep_logpdf<R, D, 1>, &sweep_logpdf_ws<R, D, 1>, lorenz_logpdf_entry<R, D, 1>(), &run_fused_shared<R, D, 1>, &fused_ws<R, D, 1>}, \
                                                {&run_sweep_logpdf<R, D, 2>, &sweep_logpdf_ws<R, D, 2>, lorenz_logpdf_entry<R, D, 2>(), &run_fused_shared<R, D, 2>, &fused_ws<R, D, 2>}, \
                                                {&run_sweep_logpdf<R, D, 3>, &sweep_logpdf_ws<R, D, 3>, lorenz_logpdf_entry<R, D, 3>(), &run_fused_shared<R, D, 3>, &fused_ws<R, D, 3>}, \
                                                {&run_sweep_logpdf<R, D, 4>, &sweep_logpdf_ws<R, D, 4>, lorenz_logpdf_entry<R, D, 4>(), &run_fused_shared<R, D, 4>, &fused_ws<R, D, 4>}}; \
        return (PO >= 1 && PO <= 4) ? &tab[PO - 1] : nullptr;                                        \
    }                                                                                                \
    const SampleEntry* sample_unit_##NAME() {                                                        \
        static const SampleEntry e = {&run_sample<R, D>, &sample_ws<R, D>, &run_sv_logpdf<R, D>, &sv_logpdf_ws<R, D>};                          \
        return &e;                                                                                   \
    }                                                                                                \
    }

}  // namespace ax

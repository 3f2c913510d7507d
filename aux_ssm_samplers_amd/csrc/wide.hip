// wide.hip -- the auxiliary-Kalman hot path for state / observation sizes beyond the register-resident per-lane kernels
// (dx > 4 or dy > 8, e.g. SURVEY config C5: dx = dy = 64).  Same math as kalman_math.h (reference
// aux_samplers/_primitives/kalman/{filtering,sampling,base}.py), different execution model:
//
//   ONE WORKGROUP (256 lanes) per time step / scan element.  Every d x d / p x d / p x p operand of the step lives in LDS
//   (odd leading dimension -> conflict-free column walks); products are 4 x 4 register-tiled GEMMs over LDS, the p x p
//   Cholesky, the triangular solves and the pivoted LU of the combine are cooperative right-looking sweeps.
//   Scans are the same three launches as the per-lane path (chunk reduce -> per-sequence aggregate scan -> down-sweep
//   that carries only (b, C) resp. e), with a workgroup walking a chunk sequentially.
//
// Scan element records in the workspace (dense, row-major):
//   filter : [A d*d | b d | C d*d | eta d | J d*d | z]      sampler : [G d*d | e d]
//   (z = log-scale of the element, kalman_math.h::FiltElem: the scale of the total product is the marginal log-likelihood)
#include <algorithm>
#include <cmath>
#include <string>

#include "ctx.h"

namespace ax {
namespace wide {

#ifndef AUXSSM_WIDE_NT
#define AUXSSM_WIDE_NT 1024
#endif
constexpr int NT = AUXSSM_WIDE_NT;         // lanes per workgroup: 4 waves per SIMD hide the LDS latency of the dependent sweeps
constexpr int NWV = NT / 64;     // waves per workgroup
constexpr size_t LDS_BUDGET = 160 * 1024 - 512;

__host__ __device__ inline int ldp_(int n) { return n | 1; }
__host__ __device__ inline size_t al16(size_t b) { return (b + 15) & ~(size_t)15; }

struct Bump {
    char* p;
    template <typename T> __device__ T* take(int n) {
        T* r = (T*)p;
        p += al16((size_t)n * sizeof(T));
        return r;
    }
};

// ---- cooperative primitives (all lanes of the workgroup call them; every one that WRITES LDS ends with a barrier unless called with
// sync = false; store_mat only reads LDS: its callers put a barrier before the source is overwritten) ---------------------------------

// dst (rows x cols, ld) <- contiguous row-major record
template <typename R> __device__ __forceinline__ void load_mat(R* dst, int ld, const R* __restrict__ src, int rows, int cols, int tid) {
    for (int r = tid / 64; r < rows; r += NWV)
        for (int c = tid & 63; c < cols; c += 64) dst[r * ld + c] = src[(long long)r * cols + c];
    __syncthreads();
}
template <typename R> __device__ __forceinline__ void store_mat(R* __restrict__ dst, const R* src, int ld, int rows, int cols, int tid) {
    for (int r = tid / 64; r < rows; r += NWV)
        for (int c = tid & 63; c < cols; c += 64) dst[(long long)r * cols + c] = src[r * ld + c];
}
template <typename R> __device__ __forceinline__ void load_vec(R* dst, const R* __restrict__ src, int n, int tid) {
    for (int i = tid; i < n; i += NT) dst[i] = src[i];
    __syncthreads();
}

// C (M x N, ldc) = alpha op(A) op(B) + beta C;  op(A) is M x K (TA: stored K x M), op(B) is K x N (TB: stored N x K).
// C must not alias A or B.  Matrix cores: each wave owns whole output tiles and walks K with f32 / f64 MFMAs whose A / B
// fragments are single LDS reads per lane (out-of-range rows / columns / k read as zero).
typedef double f64x4 __attribute__((ext_vector_type(4)));

// fp32: v_mfma_f32_16x16x4_f32 -- A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; D[row 4 (l >> 4) + r][col l & 15], r < 4.
// (16 x 16 tiles: a 64 x 64 product is 16 tiles = one per wave of the 1024-lane workgroup; four k-steps of fragments are
// fetched before their four MFMAs so the LDS latency is paid once per 16 k.)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool TA, bool TB>
__device__ __forceinline__ void gemm(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, float alpha, float beta, int tid,
                                     const float* Cin = nullptr, int ldi = 0, bool sync = true) {
    const int lane = tid & 63, wv = tid >> 6, lo = lane & 15, hi = lane >> 4;
    const int tn = (N + 15) >> 4, ntile = ((M + 15) >> 4) * tn;
    if (((M | N | K) & 15) == 0) {  // whole tiles (d = 16, 32, 48, 64): no bounds tests, the next 16 k of fragments are in flight during the MFMAs
        for (int tile = wv; tile < ntile; tile += NWV) {
            const int ti = tile / tn, i0 = ti << 4, j0 = (tile - ti * tn) << 4;
            const float* pa = TA ? A + hi * lda + i0 + lo : A + (i0 + lo) * lda + hi;
            const float* pb = TB ? B + (j0 + lo) * ldb + hi : B + hi * ldb + j0 + lo;
            const int sa = TA ? 4 * lda : 4, sb = TB ? 4 : 4 * ldb;
            f32x4 acc = {0, 0, 0, 0};
            if (K == 64) {  // the flagship size: all 16 k-steps of fragments issued up front, two accumulators.  The MFMA's k index is a
                // summation index, so lane (lo, hi) takes k = 16 hi + u instead of 4 u + hi: with an odd leading dimension the 64 lanes of
                // a fragment read then hit 64 different LDS banks (lo + 16 hi + u mod 64) for A, A^T, B and B^T alike
                const float* qa = TA ? A + 16 * hi * lda + i0 + lo : A + (i0 + lo) * lda + 16 * hi;
                const float* qb = TB ? B + (j0 + lo) * ldb + 16 * hi : B + 16 * hi * ldb + j0 + lo;
                const int ua = TA ? lda : 1, ub = TB ? 1 : ldb;
                // Software pipeline in four groups of four k-steps, pinned with sched_barrier: the loads of group g + 2 are issued BEFORE the MFMAs of
                // group g, so they are in flight while the wave waits for the matrix pipe (all 16 waves run this in lockstep: with every load
                // ahead of every MFMA a streamed product costs 3.6 k cycles, pipelined 3.4 k, against 2.1 k for its MFMAs alone and 1.3 k for its loads alone --
                // tools/micro/gj_bench.hip; staggering the odd waves by s_sleep made it slower).
                float fa[16], fb[16];
                f32x4 acc1 = {0, 0, 0, 0};
#define AX_LD(G)                                  \
    _Pragma("unroll") for (int u = 4 * (G); u < 4 * (G) + 4; ++u) fa[u] = qa[u * ua], fb[u] = qb[u * ub];
#define AX_MM(G)                                                                                  \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[4 * (G)], fb[4 * (G)], acc, 0, 0, 0);             \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[4 * (G) + 1], fb[4 * (G) + 1], acc1, 0, 0, 0);   \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[4 * (G) + 2], fb[4 * (G) + 2], acc, 0, 0, 0);     \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[4 * (G) + 3], fb[4 * (G) + 3], acc1, 0, 0, 0);
                AX_LD(0) AX_LD(1)
                __builtin_amdgcn_sched_barrier(0);
                AX_LD(2)
                AX_MM(0)
                __builtin_amdgcn_sched_barrier(0);
                AX_LD(3)
                AX_MM(1)
                __builtin_amdgcn_sched_barrier(0);
                AX_MM(2)
                AX_MM(3)
#undef AX_LD
#undef AX_MM
                acc += acc1;
                float* q = &C[(i0 + 4 * hi) * ldc + j0 + lo];
                if (beta != 0.f) {
                    const float* qi = Cin ? &Cin[(i0 + 4 * hi) * ldi + j0 + lo] : q;
                    const int li = Cin ? ldi : ldc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[r * ldc] = alpha * acc[r] + beta * qi[r * li];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[r * ldc] = alpha * acc[r];
                }
                continue;
            }
            float a[4], b[4], an[4], bn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = pa[u * sa], b[u] = pb[u * sb];
            for (int k0 = 16; k0 < K; k0 += 16) {
                pa += 4 * sa;
                pb += 4 * sb;
#pragma unroll
                for (int u = 0; u < 4; ++u) an[u] = pa[u * sa], bn[u] = pb[u * sb];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = an[u], b[u] = bn[u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            float* q = &C[(i0 + 4 * hi) * ldc + j0 + lo];
            if (beta != 0.f) {
                const float* qi = Cin ? &Cin[(i0 + 4 * hi) * ldi + j0 + lo] : q;
                const int li = Cin ? ldi : ldc;
#pragma unroll
                for (int r = 0; r < 4; ++r) q[r * ldc] = alpha * acc[r] + beta * qi[r * li];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) q[r * ldc] = alpha * acc[r];
            }
        }
        if (sync) __syncthreads();
        return;
    }
    for (int tile = wv; tile < ntile; tile += NWV) {
        const int ti = tile / tn, i0 = ti << 4, j0 = (tile - ti * tn) << 4;
        const int i = i0 + lo, j = j0 + lo;
        const bool iv = i < M, jv = j < N;
        const float* pa = TA ? A + i : A + i * lda;
        const float* pb = TB ? B + j * ldb : B + j;
        f32x4 acc = {0, 0, 0, 0};
        for (int k0 = 0; k0 < K; k0 += 16) {
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + hi;
                const bool kv = k < K;
                a[u] = (iv && kv) ? (TA ? pa[k * lda] : pa[k]) : 0.f;
                b[u] = (jv && kv) ? (TB ? pb[k] : pb[k * ldb]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
        }
        if (jv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + 4 * hi + r;
                if (row < M) {
                    float* q = &C[row * ldc + j];
                    *q = beta != 0.f ? alpha * acc[r] + beta * (Cin ? Cin[row * ldi + j] : *q) : alpha * acc[r];
                }
            }
        }
    }
    if (sync) __syncthreads();
}
// fp64: v_mfma_f64_16x16x4_f64 -- A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; D[row (l >> 4) + 4 r][col l & 15]
template <bool TA, bool TB>
__device__ __forceinline__ void gemm(int M, int N, int K, const double* A, int lda, const double* B, int ldb, double* C, int ldc, double alpha, double beta, int tid,
                                     const double* Cin = nullptr, int ldi = 0, bool sync = true) {
    const int lane = tid & 63, wv = tid >> 6, lo = lane & 15, hi = lane >> 4;
    const int tn = (N + 15) >> 4, ntile = ((M + 15) >> 4) * tn;
    for (int tile = wv; tile < ntile; tile += NWV) {
        const int ti = tile / tn, i0 = ti << 4, j0 = (tile - ti * tn) << 4;
        const int i = i0 + lo, j = j0 + lo;
        const bool iv = i < M, jv = j < N;
        const double* pa = TA ? A + i : A + i * lda;
        const double* pb = TB ? B + j * ldb : B + j;
        f64x4 acc = {0, 0, 0, 0};
        for (int k0 = 0; k0 < K; k0 += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + hi;
                const bool kv = k < K;
                a[u] = (iv && kv) ? (TA ? pa[k * lda] : pa[k]) : 0.0;
                b[u] = (jv && kv) ? (TB ? pb[k] : pb[k * ldb]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
        if (jv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + hi + 4 * r;
                if (row < M) {
                    double* q = &C[row * ldc + j];
                    *q = beta != 0.0 ? alpha * acc[r] + beta * (Cin ? Cin[row * ldi + j] : *q) : alpha * acc[r];
                }
            }
        }
    }
    if (sync) __syncthreads();
}
// y (M) = alpha op(A) x + beta y;  LPR lanes per row (16 when the whole product fits one pass, else 4), each a strided share of the k
// range, combined by shuffles
template <typename R, bool TA, int LPR> __device__ __forceinline__ void gemv_t(int M, int K, const R* A, int lda, const R* x, R* y, R alpha, R beta, int tid) {
    const int q = tid & (LPR - 1);
    for (int i0 = 0; i0 < M; i0 += NT / LPR) {
        const int i = i0 + tid / LPR;
        R s = 0;
        if (i < M)
            for (int k = q; k < K; k += LPR) s += (TA ? A[k * lda + i] : A[i * lda + k]) * x[k];
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
        if (i < M && q == 0) y[i] = beta != (R)0 ? alpha * s + beta * y[i] : alpha * s;
    }
}
template <typename R, bool TA> __device__ __forceinline__ void gemv(int M, int K, const R* A, int lda, const R* x, R* y, R alpha, R beta, int tid, bool sync = true) {
    if (M <= NT / 16) gemv_t<R, TA, 16>(M, K, A, lda, x, y, alpha, beta, tid);
    else gemv_t<R, TA, 4>(M, K, A, lda, x, y, alpha, beta, tid);
    if (sync) __syncthreads();
}
// y[i] = sum_j A[i lda + j] x[j], one wave per row, lanes over j: coalesced when A is a row-major record in global memory
template <typename R> __device__ __forceinline__ void gemv_rows(int M, int K, const R* __restrict__ A, long long lda, const R* x, R* y, int tid) {
    const int lane = tid & 63;
    for (int i = tid >> 6; i < M; i += NWV) {
        R s = 0;
        for (int j = lane; j < K; j += 64) s += A[i * lda + j] * x[j];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) y[i] = s;
    }
    __syncthreads();
}
// sum of one value per lane over the workgroup (every lane gets it); red = NWV reals of LDS scratch
template <typename R> __device__ __forceinline__ R block_sum(R v, R* red, int tid) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    R s = 0;
    for (int w = 0; w < NWV; ++w) s += red[w];
    __syncthreads();
    return s;
}
// M <- 0.5 (M + M^T)
template <typename R> __device__ __forceinline__ void symmetrise(R* M, int ld, int n, int tid) {
    for (int i = tid / 64; i < n; i += NWV)
        for (int j = i + 1 + (tid & 63); j < n; j += 64) {
            const R v = (R)0.5 * (M[i * ld + j] + M[j * ld + i]);
            M[i * ld + j] = v;
            M[j * ld + i] = v;
        }
    __syncthreads();
}

// The triangular sweeps below give wave ti = tid >> 6 the rows ti + NWV a and lane tj = tid & 63 the columns tj + 64 b, so a
// wave always touches 64 consecutive columns of one row.  Scalings are DEFERRED (a finished row / column is read unscaled and multiplied by its reciprocal pivot on the
// fly, then scaled once at the end), which leaves one barrier per elimination step.

// y[c] -= f x[c] for c = c0, c0 + 64, ... < c1 (a wave covers 64 consecutive columns of one row: conflict-free for any ld)
template <typename R> __device__ __forceinline__ void axpy64(R* __restrict__ y, const R* __restrict__ x, R f, int c0, int c1) {
    for (int c = c0; c < c1; c += 64) y[c] -= f * x[c];
}
// X[r][c] -= Lc[r * ldl] * xi for r = r0, r0 + NWV, ... < r1 (column c of X, leading dimension ld)
template <typename R> __device__ __forceinline__ void colupd(R* Xc, int ld, const R* Lc, int ldl, R xi, int r0, int r1) {
    int r = r0;
    for (; r + NWV < r1; r += 2 * NWV) {
        const R l0 = Lc[r * ldl], l1 = Lc[(r + NWV) * ldl];
        const R b0 = Xc[r * ld], b1 = Xc[(r + NWV) * ld];
        Xc[r * ld] = b0 - l0 * xi;
        Xc[(r + NWV) * ld] = b1 - l1 * xi;
    }
    for (; r < r1; r += NWV) Xc[r * ld] -= Lc[r * ldl] * xi;
}

// In-place lower Cholesky of the LOWER triangle of S (n x n).  skip[k] (may be null): index k is deleted (L_kk = 1,
// off-diagonals 0) -- the NaN-observation masking of filtering.py:89-100.  invd = 1 / diag; dg = scratch (n).  Returns false
// (uniformly) on a non-positive / NaN pivot; the factor then holds NaNs, as JAX's does.  Same operation order as smallmat.h.
template <typename R> __device__ __forceinline__ bool chol(R* S, int ld, int n, const unsigned char* skip, R* invd, R* dg, int* flag, int tid) {
    const int ti = tid >> 6, tj = tid & 63;
    if (tid == 0) *flag = 1;
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const bool skj = skip && skip[j];
        const R s = S[j * ld + j];
        const R ljj = skj ? (R)1 : sqrt_(s);
        const R inv = (R)1 / ljj;
        if (tid == 0) {
            dg[j] = ljj;
            invd[j] = inv;
            if (!skj && !(s > (R)0)) *flag = 0;
        }
        if (!skj)
            for (int r = j + 1 + ti; r < n; r += NWV) {
                if (skip && skip[r]) continue;
                const R lrj = S[r * ld + j] * inv;
                for (int c = j + 1 + tj; c <= r; c += 64)
                    if (!(skip && skip[c])) S[r * ld + c] -= lrj * (S[c * ld + j] * inv);
            }
        __syncthreads();
    }
    for (int r = ti; r < n; r += NWV)
        for (int c = tj; c <= r; c += 64)
            S[r * ld + c] = r == c ? dg[r] : ((skip && (skip[r] || skip[c])) ? (R)0 : S[r * ld + c] * invd[c]);
    __syncthreads();
    return *flag != 0;
}

// X (n x nc, ld) <- L^-1 X
template <typename R> __device__ __forceinline__ void trsm_l(const R* L, int ldl, int n, const R* invd, R* X, int ld, int nc, int tid) {
    const int ti = tid >> 6, tj = tid & 63;
    for (int i = 0; i + 1 < n; ++i) {
        const R inv = invd[i];
        for (int c = tj; c < nc; c += 64) colupd<R>(X + c, ld, L + i, ldl, X[i * ld + c] * inv, i + 1 + ti, n);
        __syncthreads();
    }
    for (int r = ti; r < n; r += NWV)
        for (int c = tj; c < nc; c += 64) X[r * ld + c] *= invd[r];
    __syncthreads();
}
// X <- L^-T X
template <typename R> __device__ __forceinline__ void trsm_lt(const R* L, int ldl, int n, const R* invd, R* X, int ld, int nc, int tid) {
    const int ti = tid >> 6, tj = tid & 63;
    for (int i = n - 1; i > 0; --i) {
        const R inv = invd[i];
        for (int c = tj; c < nc; c += 64) colupd<R>(X + c, ld, L + i * ldl, 1, X[i * ld + c] * inv, ti, i);
        __syncthreads();
    }
    for (int r = ti; r < n; r += NWV)
        for (int c = tj; c < nc; c += 64) X[r * ld + c] *= invd[r];
    __syncthreads();
}

// 32-bit order-preserving pivot key: the leading bits of |v| with the low 7 bits replaced by (127 - row): the largest
// magnitude wins up to a relative 2^-16 (fp32) / 2^-13 (fp64), ties and near-ties go to the smaller row, NaN never wins.
__device__ __forceinline__ unsigned int piv_key(float v, int r) {
    const float a = fabsf(v);
    const unsigned int bits = (a == a) ? __float_as_uint(a) : 0u;
    return (bits & ~0x7fu) | (unsigned int)(127 - r);
}
__device__ __forceinline__ unsigned int piv_key(double v, int r) {
    const double a = fabs(v);
    const unsigned int bits = (a == a) ? (unsigned int)((unsigned long long)__double_as_longlong(a) >> 32) : 0u;
    return (bits & ~0x7fu) | (unsigned int)(127 - r);
}

// value of lane `src` (wave-uniform index) broadcast to the wave through v_readlane (no LDS traffic)
__device__ __forceinline__ float bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ double bcast(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src), hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Z = [W (n x n) | RHS (n x (nct - n))] in LDS: RHS <- W^-1 RHS by Gauss-Jordan elimination with (implicit) partial
// pivoting, Z held in REGISTERS for the whole elimination: wave ti owns rows ti + NWV a (a < NRR), lane tj owns columns
// tj + 64 b (b < 4).  Column k of a wave's rows sits in that wave's lane k & 63, so multipliers are v_readlane broadcasts;
// only the pivot bid (one 32-bit LDS atomic max per wave) and the pivot row (nct values) cross waves, with two barriers
// per pivot:
//   [bids for pivot k were placed at the end of step k - 1] | barrier | the wave owning the pivot row publishes it and
//   1 / pivot | barrier | every lane updates its NRR x 4 registers and its wave bids for pivot k + 1.
// Rows are never swapped: pivot k stays in row perm[k]; the solution row k is written back from row perm[k] at the end.
// Needs n <= NWV * NRR and nct <= 256.  LDS scratch: rowbuf[nct + 1], pinv[n] reals; iperm[n] ints; key[2].
// (The reference's jnp.linalg.solve is LU + two triangular solves; same solution, rounding-level differences.)
template <typename R, int NRR>
__device__ __forceinline__ void gj_solve(R* Z, int ld, int n, int nct, R* rowbuf, R* pinv, int* iperm, unsigned int* key, int tid) {
    const int ti = tid >> 6, tj = tid & 63;
    R z[NRR][4];
    bool used[NRR];
#pragma unroll
    for (int a = 0; a < NRR; ++a) {
        used[a] = false;
        const int r = ti + NWV * a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = tj + 64 * b;
            z[a][b] = (r < n && c < nct) ? Z[r * ld + c] : (R)0;
        }
    }
    if (tid < 2) key[tid] = 0;
    __syncthreads();
    R colk[NRR];  // column k of this wave's rows (wave-uniform)
    auto column = [&](int k) {
        const int kb = k >> 6, src = k & 63;
        unsigned int best = 0;
#pragma unroll
        for (int a = 0; a < NRR; ++a) {
            const R v = kb == 0 ? z[a][0] : (kb == 1 ? z[a][1] : (kb == 2 ? z[a][2] : z[a][3]));
            colk[a] = bcast(v, src);
            const int r = ti + NWV * a;
            if (r < n && !used[a]) {
                const unsigned int ky = piv_key(colk[a], r);
                best = ky > best ? ky : best;
            }
        }
        if (tj == 0 && best) (void)__hip_atomic_fetch_max(key + (k & 1), best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    column(0);
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        const int pr = 127 - (int)(key[k & 1] & 0x7fu);
        const int pa = pr / NWV;
        if (ti == pr - pa * NWV) {  // this wave owns the pivot row
#pragma unroll
            for (int a = 0; a < NRR; ++a)
                if (a == pa) {
                    used[a] = true;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (tj + 64 * b < nct) rowbuf[tj + 64 * b] = z[a][b];
                    if (tj == 0) {
                        const R inv = (R)1 / colk[a];
                        rowbuf[nct] = inv;
                        pinv[pr] = inv;
                        iperm[pr] = k;
                        key[(k + 1) & 1] = 0;  // last read before this step's first barrier; bids for pivot k + 1 come after the second
                    }
                }
        }
        __syncthreads();
        const R inv = rowbuf[nct];
        R zk[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) zk[b] = (tj + 64 * b < nct) ? rowbuf[tj + 64 * b] : (R)0;
#pragma unroll
        for (int a = 0; a < NRR; ++a) {
            const R f = (ti + NWV * a != pr) ? colk[a] * inv : (R)0;
#pragma unroll
            for (int b = 0; b < 4; ++b) z[a][b] -= f * zk[b];
        }
        // rowbuf is rewritten only after the next step's first barrier, which every lane passes after the reads above
        if (k + 1 < n) column(k + 1);
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NRR; ++a) {
        const int r = ti + NWV * a;
        if (r < n) {
            const R inv = pinv[r];
            const int kr = iperm[r];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c = tj + 64 * b;
                if (c >= n && c < nct) Z[kr * ld + c] = z[a][b] * inv;
            }
        }
    }
    __syncthreads();
}
// 1 / v for the pivots of the blocked eliminations: hardware reciprocal + one Newton step (fp32: <= 1 ulp; the IEEE division sequence
// is 11 dependent instructions on the panel wave's critical path), IEEE division in fp64
__device__ __forceinline__ float rcp_nr(float v) {
    const float r = __builtin_amdgcn_rcpf(v);
    return fmaf(r, fmaf(-v, r, 1.0f), r);
}
__device__ __forceinline__ double rcp_nr(double v) { return 1.0 / v; }

// wave-wide maximum of one unsigned per lane (0 = identity) with DPP row shifts / row broadcasts (no LDS, no ds_bpermute): ~14 VALU
__device__ __forceinline__ unsigned int wave_umax_dpp(unsigned int v) {
    auto mx = [](unsigned int a, unsigned int b) { return a > b ? a : b; };
    v = mx(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = mx(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = mx(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = mx(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = mx(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
    v = mx(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

// The same elimination as gj_solve (same pivots: partial pivoting over the rows not yet used, deferred scaling, implicit permutation),
// BLOCKED by NB = 16 pivots so that the per-pivot synchronisation disappears, with the trailing update on the matrix cores:
//   Z lives in the MFMA accumulators of the 16 waves (ZTiles: wave w holds row tile w & 3 of column tiles (w >> 2) + 4 c);
//   A  the four waves holding the block's column tile drop the 64 x 16 panel into LDS;
//   B  ONE wave (lane = row) eliminates the panel in registers -- pivot search is a DPP wave reduction, pivot-row values are
//      v_readlane broadcasts, no barrier, no LDS -- and accumulates D = T E_p - E_p (64 x 16), T = T_15 .. T_0 the block's elementary
//      transformations, E_p the selector of its pivot rows;  T - I has non-zero columns only at the pivot rows, so for every column z of
//      the augmented matrix  T z = z + D z[p]  with z[p] the pivot rows BEFORE the block;
//   C  the owners publish those 16 rows (Zp, 16 x nct), then every wave adds D Zp to its tiles: four 16x16x4 MFMAs per tile.
// Three barriers per block of 16 pivots.  The LDS image of Z is free while Z lives in registers: the panel, D and the published rows are
// carved from it.  Needs 32 <= n <= 64 (one panel row per lane), nct <= 256 and NWV == 16.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f64x4 mfma16(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
template <typename R> struct AccT;
template <> struct AccT<float> {
    typedef f32x4 V;
    static __device__ __forceinline__ int row(int hi, int r) { return 4 * hi + r; }
};
template <> struct AccT<double> {
    typedef f64x4 V;
    static __device__ __forceinline__ int row(int hi, int r) { return hi + 4 * r; }
};
constexpr int BLK_NB = 16, BLK_PS = BLK_NB + 1;
// LDS reals the blocked elimination carves from the image of Z (panel, D, published rows, positions)
__host__ __device__ inline size_t blk_scratch(int n, int nct) { return (size_t)2 * n * BLK_PS + (size_t)BLK_NB * nct + n + 16; }
template <typename R> struct ZTiles {
    typename AccT<R>::V t[4];
    int rt, cg, lo, hi;  // row tile, first column tile, lane coordinates inside a tile
    __device__ __forceinline__ void init(int tid) {
        const int wv = tid >> 6, lane = tid & 63;
        rt = wv & 3, cg = wv >> 2, lo = lane & 15, hi = lane >> 4;
    }
    __device__ __forceinline__ int row(int r) const { return 16 * rt + AccT<R>::row(hi, r); }
    __device__ __forceinline__ int col(int c) const { return 16 * (cg + 4 * c) + lo; }
    // skip[k]: index k deleted -> unit row (the caller zeroed row / column k of the leading block)
    __device__ __forceinline__ void load(const R* Z, int ld, int n, int nct, const unsigned char* skip) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = row(r), j = col(c);
                R v = (i < n && j < nct) ? Z[i * ld + j] : (R)0;
                if (skip && i < n && skip[i]) v = i == j ? (R)1 : (R)0;
                t[c][r] = v;
            }
    }
    __device__ __forceinline__ void drop_panel(R* panel, int n, int k0) const {
        const int ctp = k0 >> 4;
        if (cg != (ctp & 3)) return;
        const int c = ctp >> 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = row(r);
            const R v = c == 0 ? t[0][r] : (c == 1 ? t[1][r] : (c == 2 ? t[2][r] : t[3][r]));
            if (i < n) panel[i * BLK_PS + lo] = v;
        }
    }
    // Zp[pos(i)][:] = row i for the block's pivot rows; pos(i) < 0: row i is not one of them
    template <typename POS> __device__ __forceinline__ void publish(R* Zp, int n, int nct, POS pos) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = row(r), pp = i < n ? pos(i) : -1;
            if (pp >= 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (col(c) < nct) Zp[pp * nct + col(c)] = t[c][r];
            }
        }
    }
    // tiles += D (n x nb, leading dimension BLK_PS) Zp (nb x nct) for the column tiles >= ctp (the earlier ones are finished pivot columns)
    __device__ __forceinline__ void update(const R* Dm, const R* Zp, int n, int nct, int nb, int ctp) {
        const int i = 16 * rt + lo;
        R a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = (i < n && 4 * u + hi < nb) ? Dm[i * BLK_PS + 4 * u + hi] : (R)0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ct = cg + 4 * c;
            if (ct < ctp || 16 * ct >= nct) continue;
            const int j = col(c);
            R b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = (j < nct && 4 * u + hi < nb) ? Zp[(4 * u + hi) * nct + j] : (R)0;
#pragma unroll
            for (int u = 0; u < 4; ++u) t[c] = mfma16(a[u], b[u], t[c]);
        }
    }
};
template <typename R>
__device__ __forceinline__ void gj_solve_blk(R* Z, int ld, int n, int nct, R* pinv, int* iperm, int tid) {
    constexpr int NB = BLK_NB, PS = BLK_PS;
    const int ti = tid >> 6, tj = tid & 63;
    ZTiles<R> z;
    z.init(tid);
    z.load(Z, ld, n, nct, nullptr);
    __syncthreads();  // Z is in registers: its LDS image is scratch until the write-back
    R* panel = Z;                 // [n][PS]
    R* Dm = panel + n * PS;       // [n][PS]
    R* Zp = Dm + n * PS;          // [NB][nct]
    int* pos = (int*)(Zp + NB * nct);  // [n] position of row r among the block's pivots, -1 if none
    bool used_lane = false;       // (wave 0) row tj already served as a pivot row
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = n - k0 < NB ? n - k0 : NB;
        z.drop_panel(panel, n, k0);
        __syncthreads();
        if (ti == 0) {
            const int r = tj;
            const bool valid = r < n;
            R pz[NB], g[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                pz[j] = (valid && j < nb) ? panel[r * PS + j] : (R)0;
                g[j] = 0;
            }
            int mypos = -1;
            R myinv = 0;
            auto step = [&](int j) {
                const unsigned int ky = (valid && !used_lane) ? piv_key(pz[j], r) : 0u;
                const unsigned int best = wave_umax_dpp(ky);
                const int pr = 127 - (int)(best & 0x7fu);
                const R inv = rcp_nr(bcast(pz[j], pr));
                const bool me = r == pr;
                const R f = me ? (R)0 : pz[j] * inv;
#pragma unroll
                for (int jj = j + 1; jj < NB; ++jj) pz[jj] -= f * bcast(pz[jj], pr);
#pragma unroll
                for (int i = 0; i < j; ++i) g[i] -= f * bcast(g[i], pr);
                g[j] = -f;
                used_lane = used_lane || me;
                mypos = me ? j : mypos;
                myinv = me ? inv : myinv;
            };
            if (nb == NB) {  // full block: no per-step branches
#pragma unroll
                for (int j = 0; j < NB; ++j) step(j);
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j < nb) step(j);
            }
            if (valid && mypos >= 0) {  // this lane's row was a pivot of the block: its reciprocal pivot and its position
                pinv[r] = myinv;
                iperm[r] = k0 + mypos;
            }
            if (valid) {
#pragma unroll
                for (int j = 0; j < NB; ++j) Dm[r * PS + j] = g[j];
                pos[r] = mypos;
            }
        }
        __syncthreads();
        z.publish(Zp, n, nct, [&](int i) { return pos[i]; });
        __syncthreads();
        z.update(Dm, Zp, n, nct, nb, k0 >> 4);
        // the next block's panel writes touch `panel` only; D / Zp / pos are rewritten after its first barrier, which no wave passes
        // before every wave has finished the update above
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = z.row(r);
        if (i < n) {
            const R inv = pinv[i];
            const int kr = iperm[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = z.col(c);
                if (j >= n && j < nct) Z[kr * ld + j] = z.t[c][r] * inv;
            }
        }
    }
    __syncthreads();
}
// Z = [S (n x n, symmetric positive definite, FULL storage) | RHS (n x (nct - n))] in LDS: RHS <- S^-1 RHS by Gauss-Jordan
// elimination WITHOUT pivoting (the pivots are the squared Cholesky diagonal, so the failure test "pivot <= 0 or NaN" and
// log|S| = sum log pivot are exactly what the Cholesky route gives), Z in registers as in gj_solve: one barrier per pivot.
// skip[k] (may be null): index k is deleted (row / column k of S must be zero on entry; treated as a unit row).
// Replaces chol + trsm_l (+ trsm_lt) wherever the factor itself is not needed.  Needs n <= NWV * NRR, nct <= 256.
// LDS scratch: rowbuf[2 (nct + 1)], piv[n] reals.  Returns ok (uniform); *half_logdet = 0.5 log|S| over the kept indices.
template <typename R, int NRR>
__device__ __forceinline__ bool spd_solve_t(R* Z, int ld, int n, int nct, const unsigned char* skip, R* rowbuf, R* piv, R* half_logdet, int tid, bool z_free,
                                            R* Lout = nullptr, int ldl = 0) {
    const int ti = tid >> 6, tj = tid & 63;
    // blocked elimination (the scheme of gj_solve_blk with the pivot of column k fixed to row k: no search at all): 16 pivots per three
    // barriers, trailing update on the matrix cores; scratch carved from the LDS image of Z, which is free while Z lives in registers
    // (z_free: the caller does not read S = Z[:, :n] again -- the unblocked path leaves it intact, this one does not)
    const bool blocked = z_free && NRR == 4 && NWV == 16 && n >= 32 && n <= 64 && blk_scratch(n, nct) <= (size_t)n * ld;
    if (blocked) {
        constexpr int NB = BLK_NB, PS = BLK_PS;
        ZTiles<R> zt;
        zt.init(tid);
        zt.load(Z, ld, n, nct, skip);
        __syncthreads();
        R* panel = Z;
        R* Dm = panel + n * PS;
        R* Zp = Dm + n * PS;
        for (int k0 = 0; k0 < n; k0 += NB) {
            const int nb = n - k0 < NB ? n - k0 : NB;
            zt.drop_panel(panel, n, k0);
            __syncthreads();
            if (ti == 0) {
                const int r = tj;
                const bool valid = r < n;
                R pz[NB], g[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    pz[j] = (valid && j < nb) ? panel[r * PS + j] : (R)0;
                    g[j] = 0;
                }
                R mypiv = 0;
                auto step = [&](int j) {
                    const int pr = k0 + j;
                    const R pv = bcast(pz[j], pr);
                    const R inv = rcp_nr(pv);
                    const bool me = r == pr;
                    const R f = me ? (R)0 : pz[j] * inv;
                    // the multipliers below the pivot are the columns of the unit-lower factor of S = L D L^T (blocked path only, see chol_blk_n2n)
                    if (Lout && valid && r > pr) Lout[r * ldl + pr] = f;
#pragma unroll
                    for (int jj = j + 1; jj < NB; ++jj) pz[jj] -= f * bcast(pz[jj], pr);
#pragma unroll
                    for (int i = 0; i < j; ++i) g[i] -= f * bcast(g[i], pr);
                    g[j] = -f;
                    mypiv = me ? pv : mypiv;
                };
                if (nb == NB) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) step(j);
                } else {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        if (j < nb) step(j);
                }
                if (valid && r >= k0 && r < k0 + nb) piv[r] = mypiv;
                if (valid) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) Dm[r * PS + j] = g[j];
                }
            }
            __syncthreads();
            zt.publish(Zp, n, nct, [&](int i) { return (i >= k0 && i < k0 + nb) ? i - k0 : -1; });
            __syncthreads();
            zt.update(Dm, Zp, n, nct, nb, k0 >> 4);
        }
        __syncthreads();
        int bad = 0;
        R hl = 0;
        for (int k = tid; k < n; k += NT) {
            const R d = piv[k];
            if (!(skip && skip[k])) {
                bad |= !(d > (R)0);
                hl += (R)0.5 * log_(d);
            }
        }
        const bool ok = !__syncthreads_or(bad);
        if (half_logdet) *half_logdet = block_sum<R>(hl, rowbuf, tid);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = zt.row(r);
            if (i < n) {
                const R inv = (R)1 / piv[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = zt.col(c);
                    if (j >= n && j < nct) Z[i * ld + j] = zt.t[c][r] * inv;
                }
            }
        }
        __syncthreads();
        return ok;
    }
    R z[NRR][4];
#pragma unroll
    for (int a = 0; a < NRR; ++a) {
        const int r = ti + NWV * a;
        const bool sk = skip && r < n && skip[r];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int c = tj + 64 * b;
            R v = (r < n && c < nct) ? Z[r * ld + c] : (R)0;
            if (sk) v = c == r ? (R)1 : (R)0;
            z[a][b] = v;
        }
    }
    __syncthreads();
    const int rb = nct + 1;
    for (int k = 0; k < n; ++k) {
        const int ka = k / NWV, kb = k >> 6, src = k & 63;
        R* rbuf = rowbuf + (k & 1) * rb;
        if (ti == k - ka * NWV) {  // the wave owning row k publishes it (current values) and the pivot
#pragma unroll
            for (int a = 0; a < NRR; ++a)
                if (a == ka) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (tj + 64 * b < nct) rbuf[tj + 64 * b] = z[a][b];
                        if (b == kb && tj == src) {
                            rbuf[nct] = (R)1 / z[a][b];
                            piv[k] = z[a][b];
                        }
                    }
                }
        }
        __syncthreads();
        const R inv = rbuf[nct];
        R zk[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) zk[b] = (tj + 64 * b < nct) ? rbuf[tj + 64 * b] : (R)0;
#pragma unroll
        for (int a = 0; a < NRR; ++a) {
            const R v = kb == 0 ? z[a][0] : (kb == 1 ? z[a][1] : (kb == 2 ? z[a][2] : z[a][3]));
            const R f = (ti + NWV * a != k) ? bcast(v, src) * inv : (R)0;
#pragma unroll
            for (int b = 0; b < 4; ++b) z[a][b] -= f * zk[b];
        }
        // rowbuf is double-buffered: the half read here is rewritten at step k + 2, after every lane has passed barrier k + 1
    }
    __syncthreads();
    int bad = 0;
    R hl = 0;
    for (int k = tid; k < n; k += NT) {
        const R d = piv[k];
        if (!(skip && skip[k])) {
            bad |= !(d > (R)0);
            hl += (R)0.5 * log_(d);
        }
    }
    const bool ok = !__syncthreads_or(bad);
    if (half_logdet) *half_logdet = block_sum<R>(hl, rowbuf, tid);
#pragma unroll
    for (int a = 0; a < NRR; ++a) {
        const int r = ti + NWV * a;
        if (r < n) {
            const R inv = (R)1 / piv[r];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c = tj + 64 * b;
                if (c >= n && c < nct) Z[r * ld + c] = z[a][b] * inv;
            }
        }
    }
    __syncthreads();
    return ok;
}
__host__ __device__ inline bool spd_fits(int n, int nct) { return n <= 8 * NWV && nct <= 256; }
template <typename R>
__device__ __forceinline__ bool spd_solve(R* Z, int ld, int n, int nct, const unsigned char* skip, R* rowbuf, R* piv, R* half_logdet, int tid, bool z_free = false) {
    if (n <= NWV * 4) return spd_solve_t<R, 4>(Z, ld, n, nct, skip, rowbuf, piv, half_logdet, tid, z_free);
    return spd_solve_t<R, 8>(Z, ld, n, nct, skip, rowbuf, piv, half_logdet, tid, z_free);
}
// 64 < n <= 128 (round 4: the concatenated observation of a d = 64 sweep is 64 + po wide): the blocked elimination holds one panel row per lane, i.e. n <= 64 --
// beyond it the unblocked path pays one barrier per pivot with eight rows per lane (the d = 64, po = 4 gain table: 12.4 ms against 1.5 at n = 64).  One level of
// block elimination brings both halves back under 64:  S = [A B; B^T C]:  [B | R1] <- A^-1 [B | R1] (blocked, n1 = 64: the right-hand sides B and R1 are
// contiguous columns of the top rows);  [C | R2] -= B^T (A^-1 [B | R1]) (one product: C becomes the Schur complement);  R2 <- Sc^-1 R2;  R1 -= (A^-1 B) R2.
// log|S| = log|A| + log|Sc|; a deleted index is a unit row in its own half, as before.  In place: the solution ends in columns n .. nct, as spd_solve's does.
// A SEPARATE function called by the one kernel that needs it (wk_gain_tab): inlined into every caller of spd_solve it cost each of them 40 registers and 250 bytes
// of scratch per lane (tests/test_kernel_resources.py), and the fp64 kernels at the register cap then produced wrong results.
// Gauss-Jordan without pivoting on a SMALL SPD block held in LDS (n2 < 32 rows of `ncols` columns, leading dimension ld): right-hand sides <- S^-1 right-hand sides in
// place, two barriers per pivot, every lane a share of the (row, column) updates.  A deleted index (zero row / column on entry) is a unit row.  (Not spd_solve_t: a
// second call site of it with these arguments changed the code generated for EVERY kernel of the unit that inlines it -- fp64 kernels 288 -> 384 bytes of scratch,
// the SV protocol's Kalman sweep 30.0k -> 16.9k sweeps/s; tests/test_kernel_resources.py.)
template <typename R>
__device__ __forceinline__ bool spd_small_inplace(R* Zs, int ld, int n2, int ncols, const unsigned char* skip, R* piv, R* colbuf, R* half_logdet, int tid) {
    for (int k = 0; k < n2; ++k) {
        __syncthreads();
        if (tid < n2) colbuf[tid] = Zs[tid * ld + k];
        __syncthreads();
        const bool sk = skip && skip[k];
        const R p = sk ? (R)1 : colbuf[k];
        if (tid == 0) piv[k] = p;
        const R ip = (R)1 / p;
        const int w = ncols - (k + 1);
        for (int e = tid; e < n2 * w; e += NT) {
            const int i = e / w, c = k + 1 + (e - i * w);
            if (i != k) Zs[i * ld + c] -= (colbuf[i] * ip) * Zs[k * ld + c];
        }
    }
    __syncthreads();
    int bad = 0;
    R hl = 0;
    for (int k = tid; k < n2; k += NT)
        if (!(skip && skip[k])) {
            bad |= !(piv[k] > (R)0);
            hl += (R)0.5 * log_(piv[k]);
        }
    const bool ok = !__syncthreads_or(bad);
    if (half_logdet) *half_logdet = block_sum<R>(hl, colbuf + 32, tid);
    for (int e = tid; e < n2 * (ncols - n2); e += NT) {
        const int i = e / (ncols - n2), c = n2 + (e - i * (ncols - n2));
        Zs[i * ld + c] *= (R)1 / piv[i];
    }
    __syncthreads();
    return ok;
}
template <typename R>
__device__ __forceinline__ bool spd_solve_split(R* Z, int ld, int n, int nct, const unsigned char* skip, R* rowbuf, R* piv, R* half_logdet, int tid) {
    constexpr int N1 = 64;
    const int n2 = n - N1, nr = nct - n;  // nr right-hand sides
    R hl1 = 0, hl2 = 0;
    const bool ok1 = spd_solve_t<R, 4>(Z, ld, N1, nct, skip, rowbuf, piv, &hl1, tid, true);  // the top rows as a system of 64 unknowns with nct columns: [A | B | R1]
    R* Zb = Z + N1 * ld;                                                                      // bottom rows: [B^T | C | R2]
    gemm<false, false>(n2, nct - N1, N1, Zb, ld, Z + N1, ld, Zb + N1, ld, (R)-1, (R)1, tid);
    const bool ok2 = spd_small_inplace<R>(Zb + N1, ld, n2, nct - N1, skip ? skip + N1 : nullptr, piv + N1, rowbuf, &hl2, tid);  // (n2 < 32: spd_split_fits)
    gemm<false, false>(N1, nr, n2, Z + N1, ld, Zb + N1 + n2, ld, Z + n, ld, (R)-1, (R)1, tid);
    if (half_logdet) *half_logdet = hl1 + hl2;
    return ok1 && ok2;
}
// (the second half stays below the blocked path's n >= 32: its scratch is carved from the image's first reals, which for a sub-image starting at column 64 would
// run past the image's end; every call passes z_free = true, as all callers of spd_solve_t do -- one caller with `false` changed the code generated for ALL of them:
// fp64 kernels 288 -> 384 bytes of scratch, the SV protocol's Kalman sweep 30.0k -> 16.9k sweeps/s)
__host__ __device__ inline bool spd_split_fits(int n, int nct, int ld) { return NWV == 16 && n > 64 && n < 96 && nct <= 256 && blk_scratch(64, nct) <= (size_t)64 * ld; }

template <typename R> __device__ __forceinline__ void lu_solve(R* Z, int ld, int n, int nct, R* rowbuf, R* pinv, int* iperm, unsigned int* key, int tid) {
    // blocked variant: one panel row per lane, and its scratch (panel, D, published rows, positions) must fit the LDS image of Z
    if (NWV == 16 && n >= 32 && n <= 64 && nct <= 256 && blk_scratch(n, nct) <= (size_t)n * ld) {
        gj_solve_blk<R>(Z, ld, n, nct, pinv, iperm, tid);
        return;
    }
    if (n <= NWV * 4) gj_solve<R, 4>(Z, ld, n, nct, rowbuf, pinv, iperm, key, tid);
    else if (n <= NWV * 8) gj_solve<R, 8>(Z, ld, n, nct, rowbuf, pinv, iperm, key, tid);
    else gj_solve<R, 128 / NWV>(Z, ld, n, nct, rowbuf, pinv, iperm, key, tid);
}

// ---- observation model of one time step, masked (filtering.py:89-100, :204-213) ------------------------------------------
template <typename R> struct Obs {
    R* H_;   // p x d (leading dimension chosen by the kernel); missing rows zeroed
    R* c_;   // p
    R* y;    // p (raw)
    unsigned char* nan;
    int* cnt;  // #observed components
};
template <typename R> __device__ __forceinline__ bool load_obs(const Obs<R>& o, const R* Hg, const R* cg, const R* yg, int p, int d, int ldh, int tid) {
    const int ldd = ldh;
    if (tid == 0) *o.cnt = 0;
    __syncthreads();
    for (int k = tid; k < p; k += NT) {
        const R y = yg[k];
        const bool nn = !finite_(y);
        o.nan[k] = nn;
        o.y[k] = y;
        o.c_[k] = nn ? (R)0 : cg[k];
        if (!nn) atomicAdd(o.cnt, 1);
    }
    __syncthreads();
    for (int k = tid / 64; k < p; k += NWV)
        for (int j = tid & 63; j < d; j += 64) o.H_[k * ldd + j] = o.nan[k] ? (R)0 : Hg[(long long)k * d + j];
    __syncthreads();
    return *o.cnt > 0;
}
// S (p x p, FULL symmetric storage, leading dimension lds) = H_ P_ H_^T + R_;  PHt (d x p, ld ldp_(p)) = P_ H_^T.  Rg: the p x p
// covariance record in global memory (its upper entries are read, as the per-lane path does); deleted indices get a zero row /
// column (spd_solve treats them as unit rows).
template <typename R>
__device__ __forceinline__ void innovation(const Obs<R>& o, int ldh, const R* P_, const R* Rg, int p, int d, R* PHt, R* S, int lds, int tid) {
    const int ldd = ldp_(d), ldp = ldp_(p);
    gemm<false, true>(d, p, d, P_, ldd, o.H_, ldh, PHt, ldp, (R)1, (R)0, tid);
    gemm<false, false>(p, p, d, o.H_, ldh, PHt, ldp, S, lds, (R)1, (R)0, tid);
    for (int i = tid / 64; i < p; i += NWV)
        for (int j = tid & 63; j <= i; j += 64) {
            const R r = (o.nan[i] || o.nan[j]) ? (R)0 : Rg[(long long)j * p + i];
            const R v = S[j * lds + i] + r;  // the value the reference computes for the upper (j, i) entry
            S[i * lds + j] = v;
            S[j * lds + i] = v;
        }
    __syncthreads();
}
// log N from the solved system: -0.5 r^T S^-1 r - 0.5 log|S| - dim/2 log 2 pi; NaN / failed factor -> 0 (the reference's nansum)
template <typename R> __device__ __forceinline__ R ell_value(R q, R half_logdet, int dim, bool ok) {
    R ell = (R)-0.5 * q - half_logdet - (R)(0.5 * LOG_2PI) * (R)dim;
    if (!ok) ell = r_nan<R>();
    return isnan_(ell) ? (R)0 : ell;
}

// ---- t = 0 measurement update (sequential_update, filtering.py:83-130); one workgroup per sequence ------------------------
static size_t lds_filter_t0(size_t s, int d, int p) {
    const size_t ldd = ldp_(d), ldz = ldp_(p + d + 1);
    return al16(d * ldd * s) + 2 * al16(p * ldd * s) + al16(p * ldz * s) + al16(d * s) * 2 + 4 * al16(p * s) + al16((2 * (p + d + 2) + NWV) * s) + al16(p) + 128;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_filter_t0(FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x, c = s / a.d.B, b = s % a.d.B, d = a.dx, p = a.dy;
    const int ldd = ldp_(d), nct = p + d + 1, ldz = ldp_(nct);
    Bump L{smem};
    R* P = L.take<R>(d * ldd);
    Obs<R> o;
    o.H_ = L.take<R>(p * ldd);
    R* HP = L.take<R>(p * ldd);
    R* Z = L.take<R>(p * ldz);  // [S | H_ P | yd] -> [S | S^-1 H_ P | S^-1 yd]
    R* m = L.take<R>(d);
    R* dm = L.take<R>(d);
    o.c_ = L.take<R>(p);
    o.y = L.take<R>(p);
    R* yd = L.take<R>(p);
    R* piv = L.take<R>(p);
    R* rowbuf = L.take<R>(2 * (nct + 1) + NWV);
    o.nan = L.take<unsigned char>(p);
    o.cnt = L.take<int>(1);
    load_mat<R>(P, ldd, at<R>(a.P0, c, 0, b), d, d, tid);
    load_vec<R>(m, at<R>(a.m0, c, 0, b), d, tid);
    const bool any = load_obs<R>(o, at<R>(a.Hs, c, 0, b), at<R>(a.cs, c, 0, b), at<R>(a.ys, c, 0, b), p, d, ldd, tid);
    R* mo = const_cast<R*>(at<R>(a.ms, c, 0, b));
    R* Po = const_cast<R*>(at<R>(a.Ps, c, 0, b));
    if (!any) {  // _passthrough :127-130
        for (int i = tid; i < d; i += NT) mo[i] = m[i];
        store_mat<R>(Po, P, ldd, d, d, tid);
        if (tid == 0) ((R*)a.ell0)[s] = 0;
        return;
    }
    gemv<R, false>(p, d, o.H_, ldd, m, yd, (R)1, (R)0, tid);
    for (int k = tid; k < p; k += NT) {
        yd[k] = o.nan[k] ? (R)0 : o.y[k] - (yd[k] + o.c_[k]);
        Z[k * ldz + p + d] = yd[k];
    }
    // HP = H_ P (p x d);  S = HP H_^T + R_
    gemm<false, false>(p, d, d, o.H_, ldd, P, ldd, HP, ldd, (R)1, (R)0, tid);
    gemm<false, true>(p, p, d, HP, ldd, o.H_, ldd, Z, ldz, (R)1, (R)0, tid);
    const R* Rg = at<R>(a.Rs, c, 0, b);
    for (int i = tid / 64; i < p; i += NWV)
        for (int j = tid & 63; j <= i; j += 64) {
            const R v = Z[i * ldz + j] + ((o.nan[i] || o.nan[j]) ? (R)0 : Rg[(long long)j * p + i]);
            Z[i * ldz + j] = v;
            Z[j * ldz + i] = v;
        }
    for (int i = tid / 64; i < p; i += NWV)
        for (int j = tid & 63; j < d; j += 64) Z[i * ldz + p + j] = HP[i * ldd + j];
    __syncthreads();
    R hl;
    const bool ok = spd_solve<R>(Z, ldz, p, nct, o.nan, rowbuf, piv, &hl, tid, true);  // X = S^-1 HP (gain^T, :117), S^-1 yd
    R q = 0;
    for (int k = tid; k < p; k += NT) q += yd[k] * Z[k * ldz + p + d];
    q = block_sum<R>(q, rowbuf, tid);
    const R ell = ell_value<R>(q, hl, *o.cnt, ok);
    // m += X^T yd;  P <- sym(P - X^T HP)
    gemv<R, true>(d, p, Z + p, ldz, yd, dm, (R)1, (R)0, tid);
    gemm<true, false>(d, d, p, Z + p, ldz, HP, ldd, P, ldd, (R)-1, (R)1, tid);
    symmetrise<R>(P, ldd, d, tid);
    const R bad = r_nan<R>();
    for (int i = tid; i < d; i += NT) mo[i] = ok ? m[i] + dm[i] : bad;
    for (int r = tid / 64; r < d; r += NWV)
        for (int q2 = tid & 63; q2 < d; q2 += 64) Po[(long long)r * d + q2] = ok ? P[r * ldd + q2] : bad;
    if (tid == 0) ((R*)a.ell0)[s] = ell;
}

// ---- scan element of transition i -> i + 1 (_filtering_init_one, filtering.py:196-250), information form of kalman_math.h ---
static size_t lds_filter_init(size_t s, int d, int p) {
    const size_t ldd = ldp_(d), ldp = ldp_(p), ldz = ldp_(p + d + 2);
    return 5 * al16(d * ldd * s) + al16(p * ldd * s) + al16(d * std::max(ldp, ldd) * s) + al16(p * ldz * s) + 6 * al16(d * s) +
           5 * al16(p * s) + al16((2 * (p + d + 3) + NWV) * s) + al16(p) + 128;
}
__host__ __device__ inline long long fe_size(int d) { return 3ll * d * d + 2 * d + 1; }  // [A | b | C | eta | J | z]
__host__ __device__ inline long long pre_size(int d) { return (long long)d * d + d + 1; }    // [b | C | z]

template <typename R> __global__ void __launch_bounds__(NT) wk_filter_init(FilterArgs a, R* __restrict__ elem) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, p = a.dy, n = a.d.T - 1;
    const int s = blockIdx.x / n, i = blockIdx.x - s * n, c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    const int ldd = ldp_(d), ldp = ldp_(p), ldt = ldp > ldd ? ldp : ldd, nct = p + d + 2, ldz = ldp_(nct);
    Bump L{smem};
    R* F = L.take<R>(d * ldd);
    R* P_ = L.take<R>(d * ldd);
    R* M = L.take<R>(d * ldd);
    R* MF = L.take<R>(d * ldd);
    R* O = L.take<R>(d * ldd);
    Obs<R> o;
    o.H_ = L.take<R>(p * ldd);
    R* Tm = L.take<R>(d * ldt);
    R* Z = L.take<R>(p * ldz);  // [S | H_ | rm | rb] -> [S | S^-1 H_ | S^-1 rm | S^-1 rb]
    R* bd = L.take<R>(d);
    R* m_ = L.take<R>(d);
    R* vm = L.take<R>(d);
    R* vb = L.take<R>(d);
    R* m0p = L.take<R>(d);
    R* tv = L.take<R>(d);
    o.c_ = L.take<R>(p);
    o.y = L.take<R>(p);
    R* rm = L.take<R>(p);
    R* rb = L.take<R>(p);
    R* piv = L.take<R>(p);
    R* rowbuf = L.take<R>(2 * (nct + 1) + NWV);
    o.nan = L.take<unsigned char>(p);
    o.cnt = L.take<int>(1);
    R* e = elem + ((long long)s * n + i) * fe_size(d);
    R* eA = e;
    R* eb = e + d * d;
    R* eC = eb + d;
    R* eeta = eC + d * d;
    R* eJ = eeta + d;

    load_mat<R>(F, ldd, at<R>(a.Fs, c, i, b), d, d, tid);
    load_mat<R>(P_, ldd, at<R>(a.Qs, c, i, b), d, d, tid);
    load_vec<R>(bd, at<R>(a.bs, c, i, b), d, tid);
    const bool any = load_obs<R>(o, at<R>(a.Hs, c, t, b), at<R>(a.cs, c, t, b), at<R>(a.ys, c, t, b), p, d, ldd, tid);
    if (i == 0) {  // built around predict(m0+, P0+), not symmetrised (filtering.py:200-201)
        load_mat<R>(M, ldd, at<R>(a.Ps, c, 0, b), d, d, tid);
        load_vec<R>(m0p, at<R>(a.ms, c, 0, b), d, tid);
        gemm<false, false>(d, d, d, F, ldd, M, ldd, Tm, ldt, (R)1, (R)0, tid);
        gemm<false, true>(d, d, d, Tm, ldt, F, ldd, P_, ldd, (R)1, (R)1, tid);
        gemv<R, false>(d, d, F, ldd, m0p, m_, (R)1, (R)0, tid);
        for (int k = tid; k < d; k += NT) m_[k] += bd[k];
    } else {
        for (int k = tid; k < d; k += NT) m_[k] = bd[k];
    }
    __syncthreads();
    if (!any) {  // _passthrough :239-248
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) {
                eA[r * d + q] = F[r * ldd + q];
                eC[r * d + q] = r == q ? P_[r * ldd + r] : (R)0.5 * (P_[r * ldd + q] + P_[q * ldd + r]);
                eJ[r * d + q] = 0;
            }
        for (int k = tid; k < d; k += NT) eb[k] = m_[k], eeta[k] = 0;
        if (tid == 0) eJ[d * d] = 0;  // z
        return;
    }
    innovation<R>(o, ldd, P_, at<R>(a.Rs, c, t, b), p, d, Tm, Z, ldz, tid);
    gemv<R, false>(p, d, o.H_, ldd, m_, rm, (R)1, (R)0, tid);
    gemv<R, false>(p, d, o.H_, ldd, bd, rb, (R)1, (R)0, tid);
    for (int k = tid; k < p; k += NT) {
        Z[k * ldz + p + d] = o.nan[k] ? (R)0 : o.y[k] - (rm[k] + o.c_[k]);
        Z[k * ldz + p + d + 1] = o.nan[k] ? (R)0 : o.y[k] - (rb[k] + o.c_[k]);
    }
    for (int k = tid / 64; k < p; k += NWV)
        for (int j = tid & 63; j < d; j += 64) Z[k * ldz + p + j] = o.H_[k * ldd + j];
    __syncthreads();
    for (int k = tid; k < p; k += NT) rm[k] = Z[k * ldz + p + d];  // the residual y - H_ m_ - c_ itself (rm held H_ m_ so far)
    __syncthreads();
    R hl;
    const bool ok = spd_solve<R>(Z, ldz, p, nct, o.nan, rowbuf, piv, &hl, tid, true);
    {  // the element's log-scale: log N(y; H_ m_ + c_, S)
        R q = 0;
        for (int k = tid; k < p; k += NT) q += rm[k] * Z[k * ldz + p + d];
        q = block_sum<R>(q, rowbuf, tid);
        const R zs = ok ? (R)-0.5 * q - hl - (R)(0.5 * LOG_2PI) * (R)*o.cnt : r_nan<R>();
        if (tid == 0) eJ[d * d] = zs;
    }
    for (int k = tid; k < p; k += NT) rm[k] = Z[k * ldz + p + d], rb[k] = Z[k * ldz + p + d + 1];
    __syncthreads();
    // M = H_^T S^-1 H_, vm = H_^T S^-1 rm, vb = H_^T S^-1 rb  (the information quantities of kalman_math.h::filter_elem)
    gemm<true, false>(d, d, p, o.H_, ldd, Z + p, ldz, M, ldd, (R)1, (R)0, tid);
    symmetrise<R>(M, ldd, d, tid);
    gemv<R, true>(d, p, o.H_, ldd, rm, vm, (R)1, (R)0, tid);
    gemv<R, true>(d, p, o.H_, ldd, rb, vb, (R)1, (R)0, tid);
    if (!ok) {
        const R bad = r_nan<R>();
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) M[r * ldd + q] = bad;
        for (int k = tid; k < d; k += NT) vm[k] = bad, vb[k] = bad;
        __syncthreads();
    }
    // A = F - P_ M F;  b = m_ + P_ vm;  C = sym(P_ - P_ M P_);  eta = F^T vb;  J = sym(F^T M F)
    gemm<false, false>(d, d, d, P_, ldd, M, ldd, Tm, ldt, (R)1, (R)0, tid);   // PM
    gemm<false, false>(d, d, d, M, ldd, F, ldd, MF, ldd, (R)1, (R)0, tid);
    gemm<false, false>(d, d, d, Tm, ldt, F, ldd, O, ldd, (R)1, (R)0, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) eA[r * d + q] = F[r * ldd + q] - O[r * ldd + q];
    __syncthreads();
    gemm<false, false>(d, d, d, Tm, ldt, P_, ldd, O, ldd, (R)1, (R)0, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) {
            const R v1 = P_[r * ldd + q] - O[r * ldd + q], v2 = P_[q * ldd + r] - O[q * ldd + r];
            eC[r * d + q] = r == q ? v1 : (R)0.5 * (v1 + v2);
        }
    __syncthreads();
    gemm<true, false>(d, d, d, F, ldd, MF, ldd, O, ldd, (R)1, (R)0, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) eJ[r * d + q] = r == q ? O[r * ldd + r] : (R)0.5 * (O[r * ldd + q] + O[q * ldd + r]);
    gemv<R, false>(d, d, P_, ldd, vm, tv, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) eb[k] = m_[k] + tv[k];
    __syncthreads();
    gemv<R, true>(d, d, F, ldd, vb, tv, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) eeta[k] = tv[k];
}

// ---- the associative operator of the parallel filter (_filtering_op_impl, filtering.py:163-183; one LU as in kalman_math.h) -
// The running prefix lives in LDS inside ONE augmented matrix Z = [W | A | C | v] (d x (3d + 1)) so that the pivoted LU of
// W = I + C1 J2 sweeps its right-hand sides [A1 | C1 | b1 + C1 eta2] in the same pass; the down-sweep, which carries only
// (b, C), uses Z = [W | C | v].
template <typename R> struct Agg {
    R *Z, *A, *C, *J, *b, *eta;  // A, C: views into Z (leading dimension ldz); J: d x d, ld ldp_(d)
    R z;                         // log-scale of the prefix (every lane holds the same value)
    int ldz, nct;                // columns of Z
    R *T1, *T2, *Eb;             // d x d scratch
    R *v, *w, *e2;               // d
    R *rowbuf, *pinv;            // elimination scratch: 3d + 2, d
    int* iperm;
    unsigned int* key;
};
static size_t lds_combine(size_t s, int d) { return al16(d * (size_t)ldp_(3 * d + 1) * s) + 4 * al16(d * (size_t)ldp_(d) * s) + 6 * al16(d * s) + al16((3 * d + 2 + NWV) * s) + al16(d * 4) + 64; }

template <typename R> __device__ __forceinline__ void carve_combine(Bump& L, Agg<R>& g, int d, bool full) {
    const int ldd = ldp_(d);
    g.nct = full ? 3 * d + 1 : 2 * d + 1;
    g.ldz = ldp_(g.nct);
    g.Z = L.take<R>(d * g.ldz);
    g.A = full ? g.Z + d : nullptr;
    g.C = g.Z + (full ? 2 * d : d);
    g.J = L.take<R>(d * ldd);
    g.T1 = L.take<R>(d * ldd);
    g.T2 = L.take<R>(d * ldd);
    g.Eb = L.take<R>(d * ldd);
    g.b = L.take<R>(d);
    g.eta = L.take<R>(d);
    g.v = L.take<R>(d);
    g.w = L.take<R>(d);
    g.e2 = L.take<R>(d);
    g.rowbuf = L.take<R>(3 * d + 2 > NWV ? 3 * d + 2 : NWV);
    g.pinv = L.take<R>(d);
    g.iperm = L.take<int>(d);
    g.key = L.take<unsigned int>(2);
}
template <typename R> __device__ __forceinline__ void agg_load(Agg<R>& g, const R* __restrict__ e, int d, int tid) {
    const int ldd = ldp_(d);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) {
            if (g.A) g.A[r * g.ldz + q] = e[r * d + q];
            g.C[r * g.ldz + q] = e[d * d + d + r * d + q];
            g.J[r * ldd + q] = e[2 * d * d + 2 * d + r * d + q];
        }
    for (int k = tid; k < d; k += NT) g.b[k] = e[d * d + k], g.eta[k] = e[2 * d * d + d + k];
    g.z = e[fe_size(d) - 1];
    __syncthreads();
}
template <typename R> __device__ __forceinline__ void agg_store(R* __restrict__ e, const Agg<R>& g, int d, int tid) {
    const int ldd = ldp_(d);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) {
            e[r * d + q] = g.A[r * g.ldz + q];
            e[d * d + d + r * d + q] = g.C[r * g.ldz + q];
            e[2 * d * d + 2 * d + r * d + q] = g.J[r * ldd + q];
        }
    for (int k = tid; k < d; k += NT) e[d * d + k] = g.b[k], e[2 * d * d + d + k] = g.eta[k];
    if (tid == 0) e[fe_size(d) - 1] = g.z;
}
// g <- g (+) e2   (g = earlier prefix a1, e2 = later element a2 in global memory)
//   W = I + C1 J2;  [X | Y | z] = W^-1 [A1 | C1 | b1 + C1 eta2]
//   A = A2 X;  b = A2 z + b2;  C = sym(A2 Y A2^T + C2);  eta = X^T (eta2 - J2 b1) + eta1;  J = sym(X^T (J2 A1) + J1)
// full = false: only (b, C) are updated (the down-sweep; they depend on a1 only through (b1, C1)).
template <typename R> __device__ __forceinline__ void combine(Agg<R>& g, const R* __restrict__ e2, int d, bool full, int tid) {
    asm volatile("" : "+v"(tid));  // opaque per call: keeps the lane addresses of the products from being hoisted out of the caller's loop
    asm volatile("" : "+s"(d));    // (see fold_step)
    const int ldd = ldp_(d), ldz = g.ldz;
    const R* A2 = e2;
    const R* b2 = e2 + d * d;
    const R* C2 = b2 + d;
    const R* eta2 = C2 + d * d;
    const R* J2 = eta2 + d;
    load_mat<R>(g.Eb, ldd, J2, d, d, tid);
    load_vec<R>(g.e2, eta2, d, tid);
    gemm<false, false>(d, d, d, g.C, ldz, g.Eb, ldd, g.Z, ldz, (R)1, (R)0, tid);  // W = C1 J2 (+ I below)
    gemv<R, false>(d, d, g.C, ldz, g.e2, g.v, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) {
        g.Z[k * ldz + k] += (R)1;
        g.Z[k * ldz + g.nct - 1] = g.v[k] + g.b[k];
    }
    if (full) gemm<false, false>(d, d, d, g.Eb, ldd, g.A, ldz, g.T1, ldd, (R)1, (R)0, tid);  // J2 A1
    gemv<R, false>(d, d, g.Eb, ldd, g.b, g.w, (R)1, (R)0, tid);                                // J2 b1
    __syncthreads();
    lu_solve<R>(g.Z, ldz, d, g.nct, g.rowbuf, g.pinv, g.iperm, g.key, tid);
    for (int k = tid; k < d; k += NT) g.v[k] = g.Z[k * ldz + g.nct - 1];  // W^-1 (b1 + C1 eta2)
    __syncthreads();
    {  // log-scale of the product: z1 + z2 - log|W|/2 + eta2.u + eta2.v/2 - (J2 b1).u/2, u = W^-1 b1, v = W^-1 C1 eta2 = Y eta2
        gemv<R, false>(d, d, g.C, ldz, g.e2, g.T2, (R)1, (R)0, tid);  // v (T2 is free here)
        R t = 0;
        for (int k = tid; k < d; k += NT) {
            const R vi = g.T2[k], ui = g.v[k] - vi;
            t += g.e2[k] * ui + (R)0.5 * g.e2[k] * vi - (R)0.5 * g.w[k] * ui - (R)0.5 * log_(abs_((R)1 / g.pinv[k]));
        }
        t = block_sum<R>(t, g.rowbuf, tid);
        g.z = g.z + e2[fe_size(d) - 1] + t;
    }
    for (int k = tid; k < d; k += NT) g.w[k] = g.e2[k] - g.w[k];  // eta2 - J2 b1
    __syncthreads();
    if (full) {
        gemm<true, false>(d, d, d, g.A, ldz, g.T1, ldd, g.J, ldd, (R)1, (R)1, tid);  // J1 + X^T (J2 A1)
        symmetrise<R>(g.J, ldd, d, tid);
        gemv<R, true>(d, d, g.A, ldz, g.w, g.eta, (R)1, (R)1, tid);
    }
    load_mat<R>(g.Eb, ldd, A2, d, d, tid);
    gemm<false, false>(d, d, d, g.Eb, ldd, g.C, ldz, g.T2, ldd, (R)1, (R)0, tid);  // A2 Y
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) g.C[r * ldz + q] = C2[r * d + q];
    __syncthreads();
    gemm<false, true>(d, d, d, g.T2, ldd, g.Eb, ldd, g.C, ldz, (R)1, (R)1, tid);
    symmetrise<R>(g.C, ldz, d, tid);
    gemv<R, false>(d, d, g.Eb, ldd, g.v, g.b, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) g.b[k] += b2[k];
    if (full) {
        gemm<false, false>(d, d, d, g.Eb, ldd, g.A, ldz, g.T1, ldd, (R)1, (R)0, tid);  // A2 X
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) g.A[r * ldz + q] = g.T1[r * ldd + q];
    }
    __syncthreads();
}

// chunk aggregate: elements [ch E, min(n, (ch+1) E))
template <typename R> __global__ void __launch_bounds__(NT) wk_scan_reduce(const R* __restrict__ elem, R* __restrict__ aggs, int n, int E, int nchunk, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x / nchunk, ch = blockIdx.x - s * nchunk;
    Bump L{smem};
    Agg<R> g;
    carve_combine<R>(L, g, d, true);
    const long long ne = fe_size(d);
    const int i0 = ch * E, i1 = min(n, i0 + E);
    agg_load<R>(g, elem + ((long long)s * n + i0) * ne, d, tid);
    for (int i = i0 + 1; i < i1; ++i) combine<R>(g, elem + ((long long)s * n + i) * ne, d, true, tid);
    agg_store<R>(aggs + ((long long)s * nchunk + ch) * ne, g, d, tid);
}
// exclusive scan of the chunk aggregates of one sequence; pre[ch] = (b, C) of the prefix before chunk ch (ch >= 1)
template <typename R> __global__ void __launch_bounds__(NT) wk_scan_aggs(const R* __restrict__ aggs, R* __restrict__ pre, int nchunk, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x;
    Bump L{smem};
    Agg<R> g;
    carve_combine<R>(L, g, d, true);
    const long long ne = fe_size(d), np = pre_size(d);
    agg_load<R>(g, aggs + (long long)s * nchunk * ne, d, tid);
    for (int ch = 1; ch < nchunk; ++ch) {
        R* q = pre + ((long long)s * nchunk + ch) * np;
        for (int k = tid; k < d; k += NT) q[k] = g.b[k];
        store_mat<R>(q + d, g.C, g.ldz, d, d, tid);
        if (tid == 0) q[np - 1] = g.z;
        if (ch + 1 < nchunk) combine<R>(g, aggs + ((long long)s * nchunk + ch) * ne, d, true, tid);
    }
}
// second level: exclusive (b, C) prefixes of a run of aggregates.  Workgroup (s, c1) walks aggregates [c1 E1, min(n0, (c1+1) E1)) of
// sequence s; its own incoming prefix is pre1[s][c1] (exclusive prefix over the level-1 chunks, c1 >= 1).  pre0[s][i] = (b, C) of
// aggs0[0] (+) ... (+) aggs0[i-1] for i >= 1.
template <typename R>
__global__ void __launch_bounds__(NT) wk_scan_down_pre(const R* __restrict__ aggs0, const R* __restrict__ pre1, R* __restrict__ pre0, int n0, int E1,
                                                       int nchunk1, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x / nchunk1, c1 = blockIdx.x - s * nchunk1;
    Bump L{smem};
    Agg<R> g;
    carve_combine<R>(L, g, d, false);
    const long long ne = fe_size(d), np = pre_size(d);
    const int i0 = c1 * E1, i1 = min(n0, i0 + E1);
    bool have = false;
    if (c1 > 0) {
        const R* q = pre1 + ((long long)s * nchunk1 + c1) * np;
        load_vec<R>(g.b, q, d, tid);
        load_mat<R>(g.C, g.ldz, q + d, d, d, tid);
        g.z = q[np - 1];
        have = true;
    }
    for (int i = i0; i < i1; ++i) {
        if (i > 0) {
            R* q = pre0 + ((long long)s * n0 + i) * np;
            for (int k = tid; k < d; k += NT) q[k] = g.b[k];
            store_mat<R>(q + d, g.C, g.ldz, d, d, tid);
            if (tid == 0) q[np - 1] = g.z;
            __syncthreads();
        }
        if (i + 1 < i1) {  // the prefix after the chunk's last aggregate belongs to the next workgroup
            const R* e = aggs0 + ((long long)s * n0 + i) * ne;
            if (have) combine<R>(g, e, d, false, tid);
            else {
                agg_load<R>(g, e, d, tid);
                have = true;
            }
        }
    }
}

// down-sweep: filtered moments ms[i + 1], Ps[i + 1] = (b, C) of the inclusive prefix i
template <typename R> __global__ void __launch_bounds__(NT) wk_scan_down(FilterArgs a, const R* __restrict__ elem, const R* __restrict__ pre, int E, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, n = a.d.T - 1;
    const int s = blockIdx.x / nchunk, ch = blockIdx.x - s * nchunk, c = s / a.d.B, b = s % a.d.B;
    Bump L{smem};
    Agg<R> g;
    carve_combine<R>(L, g, d, false);
    const long long ne = fe_size(d), np = pre_size(d);
    const int i0 = ch * E, i1 = min(n, i0 + E);
    if (ch == 0) {
        agg_load<R>(g, elem + (long long)s * n * ne, d, tid);  // prefix 0 = element 0 itself
    } else {
        const R* q = pre + ((long long)s * nchunk + ch) * np;
        load_vec<R>(g.b, q, d, tid);
        load_mat<R>(g.C, g.ldz, q + d, d, d, tid);
        g.z = q[np - 1];
    }
    for (int i = i0; i < i1; ++i) {
        if (!(ch == 0 && i == 0)) combine<R>(g, elem + ((long long)s * n + i) * ne, d, false, tid);
        R* mo = const_cast<R*>(at<R>(a.ms, c, (long long)i + 1, b));
        R* Po = const_cast<R*>(at<R>(a.Ps, c, (long long)i + 1, b));
        for (int k = tid; k < d; k += NT) mo[k] = g.b[k];
        store_mat<R>(Po, g.C, g.ldz, d, d, tid);
        if (i == n - 1 && tid == 0) ((R*)a.ellz)[s] = g.z;  // log-scale of the full product = log p(y_1..T-1 | y_0)
        __syncthreads();
    }
}

// ---- level 0 of the filter scan WITHOUT scan elements: fold one step onto the running prefix (kalman_math.h::filter_fold_step) ----
// The chunk-serial part of the scan never needs a step's own element (A2, b2, C2, eta2, J2): with the step's observation information
// (Lam = H_^T R_^-1 H_, g0 = H_^T R_^-1 (y - c_), q0 = (y - c_)^T R_^-1 (y - c_), missing components deleted -- one InfoRow per step,
// wk_obs_info) the prefix advances by
//   FA = F A,  mb = F b + b_dyn,  Pp = sym(F C F^T + Q)                               (predict, filtering.py:134-139)
//   W = I + Lam Pp,  [M | v] = W^-1 [Lam | g0 - Lam mb]                               (M = H^T S^-1 H, v = H^T S^-1 (y - H mb - c))
//   A' = FA - Pp M FA,  b' = mb + Pp v,  C' = sym(Pp - Pp M Pp)                       (update, filtering.py:83-130, information form)
//   eta' = eta + FA^T v,  J' = sym(J + FA^T M FA),  z' = z + log N(y; H mb + c, S)
// which IS prefix (+) element(step) of filtering.py:163-183: nine d^3 products and one pivoted elimination with d + 1 right-hand sides,
// no element build (wk_filter_init: a p x p elimination + eight products), no element record in HBM (3 d^2 + 2 d + 1 reals written and
// read twice per step; an InfoRow is d^2 + d + 3).  The aggregate levels keep the general combine.  Info row: [Lam | g0 | q0, ldR, dim].
__host__ __device__ inline long long info_size(int d) { return (long long)d * d + d + 3; }
static size_t lds_obs_info(size_t s, int d, int p) {
    const size_t ldd = ldp_(d), ldz = ldp_(p + d + 1);
    return al16(p * ldd * s) + al16(p * ldz * s) + al16(d * ldd * s) + al16(d * s) + 5 * al16(p * s) + al16((2 * (p + d + 2) + NWV) * s) + al16(p) + 128;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_obs_info(FilterArgs a, R* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, p = a.dy, n = a.d.T - 1;
    const int s = blockIdx.x / n, i = blockIdx.x - s * n, c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    const int ldd = ldp_(d), nct = p + d + 1, ldz = ldp_(nct);
    Bump L{smem};
    Obs<R> o;
    o.H_ = L.take<R>(p * ldd);
    R* Z = L.take<R>(p * ldz);  // [R_ | H_ | r] -> [. | R_^-1 H_ | R_^-1 r]
    R* Lam = L.take<R>(d * ldd);
    R* g0 = L.take<R>(d);
    o.c_ = L.take<R>(p);
    o.y = L.take<R>(p);
    R* rr = L.take<R>(p);
    R* w = L.take<R>(p);
    R* piv = L.take<R>(p);
    R* rowbuf = L.take<R>(2 * (nct + 1) + NWV);
    o.nan = L.take<unsigned char>(p);
    o.cnt = L.take<int>(1);
    R* e = info + ((long long)s * n + i) * info_size(d);
    const R* Rg = at<R>(a.Rs, c, t, b);
    const bool small = p <= 64 && d <= 64;  // then every record of the step is fetched at once (four entries of H and of R per lane): ONE exposed
                                            // HBM latency instead of four dependent load phases
    R ph[4], pr[4];
    if (small) {
        const R* Hg = at<R>(a.Hs, c, t, b);
        const int q = tid & 63;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = min((tid >> 6) + NWV * k, p - 1);
            ph[k] = Hg[(long long)r * d + min(q, d - 1)];
            pr[k] = Rg[(long long)min(q, r) * p + r];  // the upper entry (q, r), q <= r, as the per-lane path reads it
        }
    }
    bool any;
    if (small) {
        const R* cg = at<R>(a.cs, c, t, b);
        const R* yg = at<R>(a.ys, c, t, b);
        const R yv = yg[min(tid, p - 1)], cv = cg[min(tid, p - 1)];
        const bool nn = !finite_(yv);
        if (tid < p) {
            o.nan[tid] = nn;
            o.y[tid] = yv;
            o.c_[tid] = nn ? (R)0 : cv;
        }
        any = __syncthreads_or(tid < p && !nn);  // (publishes nan / y / c_)
        if (tid == 0) *o.cnt = 0;
        __syncthreads();
        if (tid < p && !nn) atomicAdd(o.cnt, 1);
    } else {
        any = load_obs<R>(o, at<R>(a.Hs, c, t, b), at<R>(a.cs, c, t, b), at<R>(a.ys, c, t, b), p, d, ldd, tid);
    }
    if (!any) {  // nothing observed: Lam = 0 makes the fold a pure prediction (_passthrough, filtering.py:239-248)
        for (long long k = tid; k < info_size(d); k += NT) e[k] = 0;
        return;
    }
    int offd = 0;
    if (small) {
        const int q = tid & 63;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (tid >> 6) + NWV * k;
            if (r < p) {
                if (q < d) {
                    const R hv = o.nan[r] ? (R)0 : ph[k];
                    o.H_[r * ldd + q] = hv;
                    Z[r * ldz + p + q] = hv;
                }
                if (q <= r) {
                    const R v = (o.nan[r] || o.nan[q]) ? (R)0 : pr[k];
                    Z[r * ldz + q] = v;
                    Z[q * ldz + r] = v;
                    offd |= (q < r && v != (R)0) ? 1 : 0;
                }
            }
        }
    } else {
        for (int r = tid / 64; r < p; r += NWV)
            for (int q = tid & 63; q <= r; q += 64) {  // the upper entry (q, r) of the record, as the per-lane path reads it
                const R v = (o.nan[r] || o.nan[q]) ? (R)0 : Rg[(long long)q * p + r];
                Z[r * ldz + q] = v;
                Z[q * ldz + r] = v;
                offd |= (q < r && v != (R)0) ? 1 : 0;
            }
        for (int k = tid / 64; k < p; k += NWV)
            for (int j = tid & 63; j < d; j += 64) Z[k * ldz + p + j] = o.H_[k * ldd + j];
    }
    for (int k = tid; k < p; k += NT) {
        rr[k] = o.nan[k] ? (R)0 : o.y[k] - o.c_[k];
        Z[k * ldz + p + d] = rr[k];
    }
    // diagonal R_ (the usual observation noise; the masked entries are zero already): R_^-1 is a row scaling, no elimination
    const bool diag = !__syncthreads_or(offd);  // (also the barrier that publishes Z)
    R hl;
    bool ok;
    if (diag) {
        int bad = 0;
        R hs = 0;
        for (int k = tid; k < p; k += NT)
            if (!o.nan[k]) {
                const R rk = Z[k * ldz + k];
                bad |= !(rk > (R)0);
                hs += (R)0.5 * log_(rk);
            }
        for (int k = tid / 64; k < p; k += NWV) {
            const R inv = o.nan[k] ? (R)0 : (R)1 / Z[k * ldz + k];
            for (int j = tid & 63; j <= d; j += 64) Z[k * ldz + p + j] *= inv;
        }
        ok = !__syncthreads_or(bad);
        hl = block_sum<R>(hs, rowbuf, tid);
    } else {
        ok = spd_solve<R>(Z, ldz, p, nct, o.nan, rowbuf, piv, &hl, tid, true);
    }
    R q0 = 0;
    for (int k = tid; k < p; k += NT) {
        w[k] = Z[k * ldz + p + d];
        q0 += rr[k] * w[k];
    }
    q0 = block_sum<R>(q0, rowbuf, tid);
    gemm<true, false>(d, d, p, o.H_, ldd, Z + p, ldz, Lam, ldd, (R)1, (R)0, tid);
    symmetrise<R>(Lam, ldd, d, tid);
    gemv<R, true>(d, p, o.H_, ldd, w, g0, (R)1, (R)0, tid);
    const R bad = r_nan<R>();
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) e[r * d + q] = ok ? Lam[r * ldd + q] : bad;
    for (int k = tid; k < d; k += NT) e[d * d + k] = ok ? g0[k] : bad;
    if (tid == 0) {
        e[d * d + d] = ok ? q0 : bad;
        e[d * d + d + 1] = hl;
        e[d * d + d + 2] = (R)*o.cnt;
    }
}

template <typename R> struct Fold {
    R *A, *C, *J, *F, *Fn, *Pp, *T1, *Z;  // Z = [W | Lam -> M | g -> v] (d x (2d + 1)); after the elimination its W part holds Pp M; Fn: the next step's F
    R *b, *eta, *mb, *bd, *g0, *lm, *g, *v, *pg, *tv;
    R *rowbuf, *pinv;
    int* iperm;
    unsigned int* key;
    R z;
    int ldz;
#ifdef AUXSSM_FOLD_PROF  // tools/micro/fold_phase.hip: s_memtime at the phase boundaries of fold_step
    long long ph[16], t0;
#endif
};
#ifdef AUXSSM_FOLD_PROF
#define FOLD_TICK(k)                   \
    do {                               \
        const long long t_ = clock64(); \
        g.ph[k] += t_ - g.t0;          \
        g.t0 = t_;                     \
    } while (0)
#else
#define FOLD_TICK(k)
#endif
__host__ __device__ inline size_t lds_fold(size_t s, int d, bool full) {
    const size_t ldd = ldp_(d), ldz = ldp_(2 * d + 1);
    return (full ? 7 : 6) * al16(d * ldd * s) + al16(d * ldz * s) + 9 * al16(d * s) + al16((d + 3) * s) + al16((2 * d + 2 + NWV) * s) + al16(d * s) + al16(d * 4) + 128;
}
__host__ __device__ inline bool fold_fits(size_t s, int d) { return d <= 64 && lds_fold(s, d, true) <= LDS_BUDGET; }
template <typename R> __device__ __forceinline__ void carve_fold(Bump& L, Fold<R>& g, int d, bool full) {
    const int ldd = ldp_(d);
    g.ldz = ldp_(2 * d + 1);
    g.A = L.take<R>(d * ldd);  // the (b, C, z) half keeps Lam here
    g.C = L.take<R>(d * ldd);
    g.J = full ? L.take<R>(d * ldd) : nullptr;
    g.F = L.take<R>(d * ldd);
    g.Fn = L.take<R>(d * ldd);
    g.Pp = L.take<R>(d * ldd);
    g.T1 = L.take<R>(d * ldd);
    g.Z = L.take<R>(d * g.ldz);
    g.b = L.take<R>(d);
    g.eta = L.take<R>(d);
    g.mb = L.take<R>(d);
    g.bd = L.take<R>(d);
    g.g0 = L.take<R>(d + 3);
    g.lm = L.take<R>(d);
    g.g = L.take<R>(d);
    g.v = L.take<R>(d);
    g.pg = L.take<R>(d);
    g.tv = L.take<R>(d);
    g.rowbuf = L.take<R>(2 * d + 2 > NWV ? 2 * d + 2 : NWV);
    g.pinv = L.take<R>(d);
    g.iperm = L.take<int>(d);
    g.key = L.take<unsigned int>(2);
}
// The records of step i + 1 travel HBM -> registers -> LDS inside step i: the loads are issued right after the elimination, the registers
// are dropped into LDS buffers that step i has finished with (Q over Pp, Lam over M, F into the spare matrix Fn, the vectors into their
// own doubles) after the next barrier-closed group of products, so they are live across four products only and the HBM latency is hidden.
template <typename R> struct StepSrc {
    const R *Fg, *Qg, *bdg, *info;  // records of one step in global memory (info: its InfoRow)
};
template <typename R> struct StepRegs {
    R f[4], q[4], l[4];
    R bd, g0;  // lane k < d: entry k of b_dyn and of g0; lanes d .. d + 2 of g0: q0, ldR, dim (the tail of the InfoRow)
};
// every load is unconditional (clamped address; the mask is applied when the registers are dropped into LDS)
template <typename R> __device__ __forceinline__ void step_fetch(StepRegs<R>& sr, const StepSrc<R>& src, int d, int tid) {
    // explicit global address space: through the StepSrc aggregate the compiler loses it and emits FLAT loads, which also count on
    // lgkmcnt -- the next LDS wait would then sit out the whole HBM latency
    typedef const R __attribute__((address_space(1))) * GP;
    const GP Fg = (GP)src.Fg, Qg = (GP)src.Qg, bdg = (GP)src.bdg, info = (GP)src.info;
    const int cq = min(tid & 63, d - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long o = (long long)min((tid >> 6) + NWV * k, d - 1) * d + cq;
        sr.f[k] = Fg[o];
        sr.q[k] = Qg[o];
        sr.l[k] = info[o];
    }
    sr.bd = bdg[min(tid, d - 1)];
    sr.g0 = info[(long long)d * d + min(tid, d + 2)];
}
template <typename R> __device__ __forceinline__ void step_drop(Fold<R>& g, const StepRegs<R>& sr, R* Fdst, R* Ldst, int ldl, int d, int tid) {
    const int ldd = ldp_(d), cq = tid & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = (tid >> 6) + NWV * k;
        if (r < d && cq < d) {
            Fdst[r * ldd + cq] = sr.f[k];
            g.Pp[r * ldd + cq] = sr.q[k];
            Ldst[r * ldl + cq] = sr.l[k];
        }
    }
    if (tid < d) g.bd[tid] = sr.bd;
    if (tid < d + 3) g.g0[tid] = sr.g0;
}
// prefix <- prefix (+) step.  FULL: (A, b, C, eta, J, z); else (b, C, z) only.  On entry the step's F, Q (in Pp), Lam (in Z[:, d:2d]), b_dyn and
// the InfoRow tail are in LDS; on exit those of step `next` are (next == nullptr: none) and g.F / g.Fn have changed roles.
template <typename R, bool FULL> __device__ __forceinline__ void fold_step(Fold<R>& g, const StepSrc<R>* next, int d, int tid) {
    // opaque per step: without this the compiler hoists every lane address of the nine products out of the step loop, runs out of
    // registers and reloads them from scratch inside the loop (each reload a vmcnt(0) that also sits out the prefetch below)
    asm volatile("" : "+v"(tid));
    asm volatile("" : "+s"(d));
    const int ldd = ldp_(d), ldz = g.ldz;
    R* Lam = g.Z + d;  // becomes M
    FOLD_TICK(15);
    const R q0 = g.g0[d], ldR = g.g0[d + 1], dim = g.g0[d + 2];
    gemm<false, false>(d, d, d, g.F, ldd, g.C, ldd, g.T1, ldd, (R)1, (R)0, tid, (const R*)nullptr, 0, false);  // F C
    gemv<R, false>(d, d, g.F, ldd, g.b, g.mb, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) g.mb[k] += g.bd[k];
    gemm<false, true>(d, d, d, g.T1, ldd, g.F, ldd, g.Pp, ldd, (R)1, (R)1, tid);  // Pp = F C F^T + Q (symmetric up to rounding; C' and J' are symmetrised)
    FOLD_TICK(1);
    if (FULL) gemm<false, false>(d, d, d, g.F, ldd, g.A, ldd, g.T1, ldd, (R)1, (R)0, tid, (const R*)nullptr, 0, false);  // FA
    gemm<false, false>(d, d, d, Lam, ldz, g.Pp, ldd, g.Z, ldz, (R)1, (R)0, tid, (const R*)nullptr, 0, false);             // W - I
    gemv<R, false>(d, d, Lam, ldz, g.mb, g.lm, (R)1, (R)0, tid);
    R t = 0;  // per-lane share of q - q0 - corr + log|W| (summed once below)
    for (int k = tid; k < d; k += NT) {
        g.Z[k * ldz + k] += (R)1;
        const R gk = g.g0[k] - g.lm[k];
        g.g[k] = gk;
        g.Z[k * ldz + 2 * d] = gk;
        t += g.mb[k] * (g.lm[k] - (R)2 * g.g0[k]);
    }
    __syncthreads();
    FOLD_TICK(2);
    lu_solve<R>(g.Z, ldz, d, 2 * d + 1, g.rowbuf, g.pinv, g.iperm, g.key, tid);
    FOLD_TICK(3);
    StepRegs<R> sr;
    if (next) step_fetch<R>(sr, *next, d, tid);
    for (int k = tid; k < d; k += NT) g.v[k] = g.Z[k * ldz + 2 * d];
    gemm<false, false>(d, d, d, g.Pp, ldd, Lam, ldz, g.Z, ldz, (R)1, (R)0, tid);  // PM (over W, which is spent; its barrier publishes v)
    gemv<R, false>(d, d, g.Pp, ldd, g.g, g.pg, (R)1, (R)0, tid, false);
    gemv<R, false>(d, d, g.Pp, ldd, g.v, g.tv, (R)1, (R)0, tid, false);
    if (FULL) {
        gemv<R, true>(d, d, g.T1, ldd, g.v, g.eta, (R)1, (R)1, tid, false);
        gemm<false, false>(d, d, d, Lam, ldz, g.T1, ldd, g.F, ldd, (R)1, (R)0, tid, (const R*)nullptr, 0, false);  // M FA (over F, which is spent)
        gemm<false, false>(d, d, d, g.Z, ldz, g.T1, ldd, g.A, ldd, (R)-1, (R)1, tid, g.T1, ldd, false);            // A' = FA - PM FA
    }
    gemm<false, false>(d, d, d, g.Z, ldz, g.Pp, ldd, g.C, ldd, (R)-1, (R)1, tid, g.Pp, ldd);                       // C' = Pp - PM Pp
    FOLD_TICK(4);
    if (next) step_drop<R>(g, sr, g.Fn, g.Z + d, ldz, d, tid);  // Pp, M, b_dyn, g0 are spent; F still holds M FA for the J product below
    for (int k = tid; k < d; k += NT) {
        t += log_(abs_((R)1 / g.pinv[k])) - g.pg[k] * g.v[k];
        g.b[k] = g.mb[k] + g.tv[k];
    }
    t = block_sum<R>(t, g.rowbuf, tid);
    g.z += (R)-0.5 * (q0 + t) - ldR - (R)(0.5 * LOG_2PI) * dim;
    if (FULL) gemm<true, false>(d, d, d, g.T1, ldd, g.F, ldd, g.J, ldd, (R)1, (R)1, tid, (const R*)nullptr, 0, false);  // J + FA^T (M FA)
    symmetrise<R>(g.C, ldd, d, tid);  // (its barrier also closes the J product; J itself is symmetrised once, when the chunk's aggregate is stored)
    FOLD_TICK(5);
    R* sw = g.F;
    g.F = g.Fn;
    g.Fn = sw;
    FOLD_TICK(6);
}
// The (b, C, z) half alone (the down-sweep; the sequential filter): with G = I - Pp M = (I + Pp Lam)^-1 (push-through identity)
//   C' = G Pp,  b' = G (mb + Pp g0)   ->   one elimination  (I + Pp Lam) [C' | b'] = [Pp | mb + Pp g0]
// and Pp v = b' - mb, log|I + Pp Lam| = log|W| for the log-scale: three products instead of five, no M.  LDS roles here: g.Pp holds Q,
// g.A holds Lam (leading dimension ldd), Pp itself is formed inside Z[:, d:2d].
template <typename R> __device__ __forceinline__ void fold_step_down(Fold<R>& g, const StepSrc<R>* next, int d, int tid) {
    asm volatile("" : "+v"(tid));  // (see fold_step)
    asm volatile("" : "+s"(d));
    const int ldd = ldp_(d), ldz = g.ldz;
    R* Pp = g.Z + d;
    const R* Lam = g.A;
    FOLD_TICK(15);
    const R q0 = g.g0[d], ldR = g.g0[d + 1], dim = g.g0[d + 2];
    gemm<false, false>(d, d, d, g.F, ldd, g.C, ldd, g.T1, ldd, (R)1, (R)0, tid, (const R*)nullptr, 0, false);  // F C
    gemv<R, false>(d, d, g.F, ldd, g.b, g.mb, (R)1, (R)0, tid, false);
    __syncthreads();
    for (int k = tid; k < d; k += NT) g.mb[k] += g.bd[k];
    gemm<false, true>(d, d, d, g.T1, ldd, g.F, ldd, Pp, ldz, (R)1, (R)1, tid, g.Pp, ldd);  // Pp = F C F^T + Q
    FOLD_TICK(1);
    gemm<false, false>(d, d, d, Pp, ldz, Lam, ldd, g.Z, ldz, (R)1, (R)0, tid, (const R*)nullptr, 0, false);  // (I + Pp Lam) - I
    gemv<R, false>(d, d, Lam, ldd, g.mb, g.lm, (R)1, (R)0, tid, false);
    gemv<R, false>(d, d, Pp, ldz, g.g0, g.tv, (R)1, (R)0, tid);
    R t = 0;
    for (int k = tid; k < d; k += NT) {
        g.Z[k * ldz + k] += (R)1;
        g.g[k] = g.g0[k] - g.lm[k];
        g.Z[k * ldz + 2 * d] = g.mb[k] + g.tv[k];
        t += g.mb[k] * (g.lm[k] - (R)2 * g.g0[k]);
    }
    StepRegs<R> sr;
    if (next) step_fetch<R>(sr, *next, d, tid);
    __syncthreads();
    FOLD_TICK(2);
    lu_solve<R>(g.Z, ldz, d, 2 * d + 1, g.rowbuf, g.pinv, g.iperm, g.key, tid);
    FOLD_TICK(3);
    if (next) step_drop<R>(g, sr, g.Fn, g.A, ldd, d, tid);  // F, Q, Lam, b_dyn, g0 are spent (g, mb and the scalars were copied out)
    for (int k = tid; k < d; k += NT) {
        const R bk = g.Z[k * ldz + 2 * d];
        t += log_(abs_((R)1 / g.pinv[k])) - g.g[k] * (bk - g.mb[k]);
        g.b[k] = bk;
    }
    for (int r = tid / 64; r < d; r += NWV)  // C' = sym(G Pp)
        for (int q = tid & 63; q < d; q += 64) g.C[r * ldd + q] = r == q ? Pp[r * ldz + r] : (R)0.5 * (Pp[r * ldz + q] + Pp[q * ldz + r]);
    t = block_sum<R>(t, g.rowbuf, tid);
    g.z += (R)-0.5 * (q0 + t) - ldR - (R)(0.5 * LOG_2PI) * dim;
    R* sw = g.F;
    g.F = g.Fn;
    g.Fn = sw;
    FOLD_TICK(5);
    FOLD_TICK(6);
}
template <typename R> __device__ __forceinline__ StepSrc<R> step_src(const FilterArgs& a, const R* __restrict__ info, int s, int c, int b, int n, long long i) {
    return StepSrc<R>{at<R>(a.Fs, c, i, b), at<R>(a.Qs, c, i, b), at<R>(a.bs, c, i, b), info + ((long long)s * n + i) * info_size(a.dx)};
}
// chunk aggregate of steps [ch E, min(n, (ch+1) E)) of sequence s, record layout of the general combine ([A | b | C | eta | J | z]).
// Chunk 0 starts from the t = 0 posterior (A = 0, b = m0+, C = P0+: its A / eta / J never reach an output), the others from the identity.
template <typename R> __global__ void __launch_bounds__(NT) wk_fold_reduce(FilterArgs a, const R* __restrict__ info, R* __restrict__ aggs, int E, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, n = a.d.T - 1;
    const int s = blockIdx.x / nchunk, ch = blockIdx.x - s * nchunk, c = s / a.d.B, b = s % a.d.B;
    const int ldd = ldp_(d);
    Bump L{smem};
    Fold<R> g;
    carve_fold<R>(L, g, d, true);
    const long long ne = fe_size(d);
    const int i0 = ch * E, i1 = min(n, i0 + E);
    {
        StepRegs<R> sr;
        step_fetch<R>(sr, step_src<R>(a, info, s, c, b, n, i0), d, tid);
        step_drop<R>(g, sr, g.F, g.Z + d, g.ldz, d, tid);
    }
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) {
            g.A[r * ldd + q] = (ch > 0 && r == q) ? (R)1 : (R)0;
            g.C[r * ldd + q] = ch > 0 ? (R)0 : at<R>(a.Ps, c, 0, b)[(long long)r * d + q];
            g.J[r * ldd + q] = 0;
        }
    for (int k = tid; k < d; k += NT) g.b[k] = ch > 0 ? (R)0 : at<R>(a.ms, c, 0, b)[k], g.eta[k] = 0;
    g.z = 0;
    __syncthreads();
    for (int i = i0; i < i1; ++i) {
        const StepSrc<R> nx = step_src<R>(a, info, s, c, b, n, i + 1 < i1 ? i + 1 : i);
        fold_step<R, true>(g, i + 1 < i1 ? &nx : nullptr, d, tid);
    }
    R* e = aggs + ((long long)s * nchunk + ch) * ne;
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) {
            e[r * d + q] = g.A[r * ldd + q];
            e[d * d + d + r * d + q] = g.C[r * ldd + q];
            e[2 * d * d + 2 * d + r * d + q] = r == q ? g.J[r * ldd + r] : (R)0.5 * (g.J[r * ldd + q] + g.J[q * ldd + r]);  // J = sum of FA^T M FA: symmetric up to rounding
        }
    for (int k = tid; k < d; k += NT) e[d * d + k] = g.b[k], e[2 * d * d + d + k] = g.eta[k];
    if (tid == 0) e[ne - 1] = g.z;
}
// down-sweep: filtered moments ms[i + 1], Ps[i + 1] = (b, C) of the inclusive prefix i, one Kalman step (information form) per time step
template <typename R> __global__ void __launch_bounds__(NT) wk_fold_down(FilterArgs a, const R* __restrict__ info, const R* __restrict__ pre, int E, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, n = a.d.T - 1;
    const int s = blockIdx.x / nchunk, ch = blockIdx.x - s * nchunk, c = s / a.d.B, b = s % a.d.B;
    const int ldd = ldp_(d);
    Bump L{smem};
    Fold<R> g;
    carve_fold<R>(L, g, d, false);
    const long long np = pre_size(d);
    const int i0 = ch * E, i1 = min(n, i0 + E);
    {
        StepRegs<R> sr;
        step_fetch<R>(sr, step_src<R>(a, info, s, c, b, n, i0), d, tid);
        step_drop<R>(g, sr, g.F, g.A, ldd, d, tid);
    }
    if (ch == 0) {
        load_vec<R>(g.b, at<R>(a.ms, c, 0, b), d, tid);
        load_mat<R>(g.C, ldd, at<R>(a.Ps, c, 0, b), d, d, tid);
        g.z = 0;
    } else {
        const R* q = pre + ((long long)s * nchunk + ch) * np;
        load_vec<R>(g.b, q, d, tid);
        load_mat<R>(g.C, ldd, q + d, d, d, tid);
        g.z = q[np - 1];
    }
    for (int i = i0; i < i1; ++i) {
        const StepSrc<R> nx = step_src<R>(a, info, s, c, b, n, i + 1 < i1 ? i + 1 : i);
        fold_step_down<R>(g, i + 1 < i1 ? &nx : nullptr, d, tid);
        R* mo = const_cast<R*>(at<R>(a.ms, c, (long long)i + 1, b));
        R* Po = const_cast<R*>(at<R>(a.Ps, c, (long long)i + 1, b));
        for (int k = tid; k < d; k += NT) mo[k] = g.b[k];
        store_mat<R>(Po, g.C, ldd, d, d, tid);
        if (i == n - 1 && tid == 0) ((R*)a.ellz)[s] = g.z;
        __syncthreads();
    }
}

// out[r] = sum_{b < B} ( add0[r B + b] + sum_{i < n} part[(r B + b) n + i] ), fixed order; one workgroup per output
// (O: the type of the sum -- Acc for the sweep's log-density totals, smallmat.h)
template <typename R, typename O = R>
__global__ void __launch_bounds__(NT) wk_reduce(const R* __restrict__ part, const R* __restrict__ add0, int B, long long n, O* __restrict__ out) {
    __shared__ O sh[NT];
    const int tid = threadIdx.x, r = blockIdx.x;
    O acc = 0;
    for (int b = 0; b < B; ++b) {
        const R* q = part + ((long long)r * B + b) * n;
        for (long long i = tid; i < n; i += NT) acc += (O)q[i];
    }
    sh[tid] = acc;
    __syncthreads();
    for (int off = NT / 2; off > 0; off >>= 1) {
        if (tid < off) sh[tid] += sh[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        O v = sh[0];
        if (add0)
            for (int b = 0; b < B; ++b) v += (O)add0[(long long)r * B + b];
        out[r] = v;
    }
}

// ---- pathwise sampler (sampling.py:60-124): scan position j <-> time T - 1 - j; element [G d*d | e d] ------------------------
static size_t lds_sample_init(size_t s, int d) {
    return 5 * al16(d * (size_t)ldp_(d) * s) + al16(d * (size_t)ldp_(2 * d) * s) + 8 * al16(d * s) + al16((2 * (2 * d + 1) + NWV) * s) + 128;
}

// Lc <- lower Cholesky factor of the symmetric matrix in Lc (full storage), nan_to_num'ed; a failed factorisation is all zero
template <typename R> __device__ __forceinline__ void chol_n2n(R* Lc, int ld, int n, R* invd, R* dg, int* flag, int tid) {
    const bool ok = chol<R>(Lc, ld, n, nullptr, invd, dg, flag, tid);
    for (int r = tid / 64; r < n; r += NWV)
        for (int q = tid & 63; q < n; q += 64) Lc[r * ld + q] = (q <= r && ok) ? nan_to_num<R>(Lc[r * ld + q]) : (R)0;
    __syncthreads();
}
// The same factor from the BLOCKED SPD elimination (41 k cycles at n = 64 against 87 k for the column-by-column Cholesky above): eliminating
// S without pivoting is S = L' D L'^T with the multipliers below each pivot as the columns of the unit-lower L' and the pivots as D, so
// chol(S) = L' sqrt(D).  X: S in full storage on entry, the nan_to_num'ed lower factor on exit (all zero if a pivot is <= 0 or NaN, as
// chol_n2n).  Zs: scratch of n x ldp_(n + 1) reals at least (the elimination runs on a copy with one dummy right-hand side).
template <typename R> __device__ __forceinline__ void chol_blk_n2n(R* X, int ld, int n, R* Zs, int ldzs, R* rowbuf, R* piv, R* invd, R* dg, int* flag, int tid) {
    const int nct = n + 1;
    if (!(NWV == 16 && n >= 32 && n <= 64 && blk_scratch(n, nct) <= (size_t)n * ldzs)) {
        chol_n2n<R>(X, ld, n, invd, dg, flag, tid);
        return;
    }
    for (int r = tid / 64; r < n; r += NWV)
        for (int q = tid & 63; q <= n; q += 64) Zs[r * ldzs + q] = q < n ? X[r * ld + q] : (R)0;
    __syncthreads();
    const bool ok = spd_solve_t<R, 4>(Zs, ldzs, n, nct, nullptr, rowbuf, piv, (R*)nullptr, tid, true, X, ld);
    for (int r = tid / 64; r < n; r += NWV)
        for (int q = tid & 63; q < n; q += 64) {
            R v = 0;
            if (ok && q <= r) v = nan_to_num<R>(q == r ? sqrt_(piv[r]) : X[r * ld + q] * sqrt_(piv[q]));
            X[r * ld + q] = v;
        }
    __syncthreads();
}
// ltab (optional): the factor Lc of every position goes to ltab[j] (d x d) as well -- the shared sampler's second table (wide_shared.h)
template <typename R> __global__ void __launch_bounds__(NT) wk_sample_init(SampleArgs a, R* __restrict__ elem, R* __restrict__ ltab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, T = a.d.T;
    const int s = blockIdx.x / T, j = blockIdx.x - s * T, c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)T - 1 - j;
    const int ldd = ldp_(d), ldz = ldp_(2 * d);
    Bump L{smem};
    R* F = L.take<R>(d * ldd);
    R* P = L.take<R>(d * ldd);
    R* T1 = L.take<R>(d * ldd);
    R* X = L.take<R>(d * ldd);
    R* G = L.take<R>(d * ldd);
    R* Z = L.take<R>(d * ldz);  // [S | F] -> [S | S^-1 F]
    R* m = L.take<R>(d);
    R* eps = L.take<R>(d);
    R* bd = L.take<R>(d);
    R* pm = L.take<R>(d);
    R* tv = L.take<R>(d);
    R* invd = L.take<R>(d);
    R* dg = L.take<R>(d);
    R* piv = L.take<R>(d);
    R* rowbuf = L.take<R>(2 * (2 * d + 1) + NWV);
    int* flag = L.take<int>(1);
    R* e = elem + ((long long)s * T + j) * ((long long)d * d + d);
    load_mat<R>(P, ldd, at<R>(a.Ps, c, t, b), d, d, tid);
    load_vec<R>(m, at<R>(a.ms, c, t, b), d, tid);
    load_vec<R>(eps, at<R>(a.eps, c, t, b), d, tid);
    if (j == 0) {  // _sample_last_step :115-124
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) X[r * ldd + q] = r == q ? P[r * ldd + r] : (R)0.5 * (P[r * ldd + q] + P[q * ldd + r]);
        __syncthreads();
        chol_blk_n2n<R>(X, ldd, d, Z, ldz, rowbuf, piv, invd, dg, flag, tid);
        if (ltab) store_mat<R>(ltab + (long long)j * d * d, X, ldd, d, d, tid);
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) e[r * d + q] = 0;
        gemv<R, false>(d, d, X, ldd, eps, tv, (R)1, (R)0, tid);  // Lc eps (the factor's upper part is zero)
        for (int k = tid; k < d; k += NT) e[d * d + k] = m[k] + tv[k];
        return;
    }
    load_mat<R>(F, ldd, at<R>(a.Fs, c, t, b), d, d, tid);
    load_mat<R>(Z, ldz, at<R>(a.Qs, c, t, b), d, d, tid);
    load_vec<R>(bd, at<R>(a.bs, c, t, b), d, tid);
    // S = sym(F P F^T + Q);  gain = P (S^-1 F)^T  (mean_and_chol :84-97)
    gemm<false, false>(d, d, d, F, ldd, P, ldd, T1, ldd, (R)1, (R)0, tid);
    gemm<false, true>(d, d, d, T1, ldd, F, ldd, Z, ldz, (R)1, (R)1, tid);
    symmetrise<R>(Z, ldz, d, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) Z[r * ldz + d + q] = F[r * ldd + q];
    __syncthreads();
    // (S is needed again below: kept in X, because the blocked elimination uses the LDS image of Z as scratch)
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) X[r * ldd + q] = Z[r * ldz + q];
    __syncthreads();
    const bool ok = spd_solve<R>(Z, ldz, d, 2 * d, nullptr, rowbuf, piv, (R*)nullptr, tid, true);
    gemm<false, true>(d, d, d, P, ldd, Z + d, ldz, G, ldd, (R)1, (R)0, tid);
    if (!ok) {
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) G[r * ldd + q] = r_nan<R>();
        __syncthreads();
    }
    // Sig = sym(P - G S G^T);  Lc = nan_to_num(chol(Sig))  (:98-104)
    gemm<false, false>(d, d, d, G, ldd, X, ldd, T1, ldd, (R)1, (R)0, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) X[r * ldd + q] = P[r * ldd + q];
    __syncthreads();
    gemm<false, true>(d, d, d, T1, ldd, G, ldd, X, ldd, (R)-1, (R)1, tid);
    symmetrise<R>(X, ldd, d, tid);
    chol_blk_n2n<R>(X, ldd, d, Z, ldz, rowbuf, piv, invd, dg, flag, tid);
    if (ltab) store_mat<R>(ltab + (long long)j * d * d, X, ldd, d, d, tid);
    // inc = m - G (F m + b) + Lc eps  (:108-112)
    gemv<R, false>(d, d, F, ldd, m, pm, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) pm[k] += bd[k];
    __syncthreads();
    gemv<R, false>(d, d, G, ldd, pm, tv, (R)1, (R)0, tid, false);
    gemv<R, false>(d, d, X, ldd, eps, invd, (R)1, (R)0, tid, false);  // Lc eps (upper part of the factor is zero; invd is free after the factorisation)
    __syncthreads();
    store_mat<R>(e, G, ldd, d, d, tid);
    for (int k = tid; k < d; k += NT) e[d * d + k] = m[k] - tv[k] + invd[k];
}

// _sampling_op_impl (sampling.py:51-55): acc = later times already composed, cur = this step: G = Gc Ga, e = Gc ea + ec
static size_t lds_sample_scan(size_t s, int d) { return 3 * al16(d * (size_t)ldp_(d) * s) + 3 * al16(d * s) + 64; }
template <typename R> struct SAgg {
    R *G, *Gc, *Go, *e, *tv, *ec;
};
template <typename R> __device__ __forceinline__ void carve_sample(Bump& L, SAgg<R>& g, int d) {
    const int ldd = ldp_(d);
    g.G = L.take<R>(d * ldd);
    g.Gc = L.take<R>(d * ldd);
    g.Go = L.take<R>(d * ldd);
    g.e = L.take<R>(d);
    g.tv = L.take<R>(d);
    g.ec = L.take<R>(d);
}
template <typename R> __device__ __forceinline__ void sample_combine_w(SAgg<R>& g, const R* __restrict__ cur, int d, bool full, int tid) {
    const int ldd = ldp_(d);
    load_mat<R>(g.Gc, ldd, cur, d, d, tid);
    load_vec<R>(g.ec, cur + d * d, d, tid);
    gemv<R, false>(d, d, g.Gc, ldd, g.e, g.tv, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) g.e[k] = g.tv[k] + g.ec[k];
    if (full) {
        gemm<false, false>(d, d, d, g.Gc, ldd, g.G, ldd, g.Go, ldd, (R)1, (R)0, tid);
        R* sw = g.G;
        g.G = g.Go;
        g.Go = sw;
    }
    __syncthreads();
}
template <typename R> __global__ void __launch_bounds__(NT) wk_sscan_reduce(const R* __restrict__ elem, R* __restrict__ aggs, int n, int E, int nchunk, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x / nchunk, ch = blockIdx.x - s * nchunk;
    Bump L{smem};
    SAgg<R> g;
    carve_sample<R>(L, g, d);
    const long long ne = (long long)d * d + d;
    const int ldd = ldp_(d);
    const int i0 = ch * E, i1 = min(n, i0 + E);
    load_mat<R>(g.G, ldd, elem + ((long long)s * n + i0) * ne, d, d, tid);
    load_vec<R>(g.e, elem + ((long long)s * n + i0) * ne + d * d, d, tid);
    for (int i = i0 + 1; i < i1; ++i) sample_combine_w<R>(g, elem + ((long long)s * n + i) * ne, d, true, tid);
    R* q = aggs + ((long long)s * nchunk + ch) * ne;
    store_mat<R>(q, g.G, ldd, d, d, tid);
    for (int k = tid; k < d; k += NT) q[d * d + k] = g.e[k];
}
// One workgroup per sequence walks the chunk aggregates in order.  Only e of the exclusive prefix is handed down (the final pass composes
// e' = G e + e_c), so the walk is a mat-vec per aggregate, no product; the next aggregate's record is fetched into registers while the
// current one is applied (d <= 64: four entries of G per lane; wider states load it directly).
template <typename R> __global__ void __launch_bounds__(NT) wk_sscan_aggs(const R* __restrict__ aggs, R* __restrict__ pre, int nchunk, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x;
    Bump L{smem};
    SAgg<R> g;
    carve_sample<R>(L, g, d);
    const long long ne = (long long)d * d + d;
    const int ldd = ldp_(d);
    const R* base = aggs + (long long)s * nchunk * ne;
    load_vec<R>(g.e, base + d * d, d, tid);
    const bool small = d <= 64;
    const int cq = min(tid & 63, d - 1);
    R pg[4], pe = 0;
    auto fetch = [&](int ch) {
        const R* q = base + (long long)ch * ne;
#pragma unroll
        for (int k = 0; k < 4; ++k) pg[k] = q[(long long)min((tid >> 6) + NWV * k, d - 1) * d + cq];
        pe = q[d * d + min(tid, d - 1)];
    };
    if (small && nchunk > 2) fetch(1);
    for (int ch = 1; ch < nchunk; ++ch) {
        R* q = pre + ((long long)s * nchunk + ch) * d;
        for (int k = tid; k < d; k += NT) q[k] = g.e[k];
        if (ch + 1 < nchunk) {
            if (small) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = (tid >> 6) + NWV * k;
                    if (r < d && (tid & 63) < d) g.Gc[r * ldd + (tid & 63)] = pg[k];
                }
                if (tid < d) g.ec[tid] = pe;
                __syncthreads();
                if (ch + 2 < nchunk) fetch(ch + 1);
                gemv<R, false>(d, d, g.Gc, ldd, g.e, g.tv, (R)1, (R)0, tid);
                for (int k = tid; k < d; k += NT) g.e[k] = g.tv[k] + g.ec[k];
                __syncthreads();
            } else {
                __syncthreads();
                sample_combine_w<R>(g, base + (long long)ch * ne, d, false, tid);
            }
        }
    }
}
template <typename R> __global__ void __launch_bounds__(NT) wk_sscan_down(SampleArgs a, const R* __restrict__ elem, const R* __restrict__ pre, int E, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, n = a.d.T;
    const int s = blockIdx.x / nchunk, ch = blockIdx.x - s * nchunk, c = s / a.d.B, b = s % a.d.B;
    Bump L{smem};
    SAgg<R> g;
    carve_sample<R>(L, g, d);
    const long long ne = (long long)d * d + d;
    const int i0 = ch * E, i1 = min(n, i0 + E);
    if (ch == 0) load_vec<R>(g.e, elem + (long long)s * n * ne + d * d, d, tid);
    else load_vec<R>(g.e, pre + ((long long)s * nchunk + ch) * d, d, tid);
    for (int i = i0; i < i1; ++i) {
        if (!(ch == 0 && i == 0)) sample_combine_w<R>(g, elem + ((long long)s * n + i) * ne, d, false, tid);
        R* xo = const_cast<R*>(at<R>(a.xs, c, (long long)n - 1 - i, b));
        for (int k = tid; k < d; k += NT) xo[k] = g.e[k];
        __syncthreads();
    }
}

// ---- Gaussian log-densities (math/mvn/base.py:15-58) ----------------------------------------------------------------------
// Cholesky of the covariance record `cov` (n x n in global memory, upper entries read) with deleted components `skip`, then
// up to two residuals solved in place.  Returns through o1 / o2 (lane-0 values; 0 where the reference's nansum drops the term).
template <typename R>
__device__ __forceinline__ void gauss2(const R* __restrict__ cov, int n, const unsigned char* skip, R* r1, R* r2, R* Z, R* piv, R* rowbuf, int tid, R& o1, R& o2) {
    const int nct = n + 2, ldz = ldp_(nct);
    // Z = [cov | r1 | r2]; the covariance record is symmetric: its lower triangle is read (row-contiguous, coalesced)
    for (int i = tid / 64; i < n; i += NWV)
        for (int j = tid & 63; j <= i; j += 64) {
            const bool sk = skip && (skip[i] || skip[j]);
            const R v = sk ? (R)0 : cov[(long long)i * n + j];
            Z[i * ldz + j] = v;
            Z[j * ldz + i] = v;
        }
    int b1 = 0, b2 = 0;
    R dm = 0;
    for (int k = tid; k < n; k += NT) {
        const bool sk = skip && skip[k];
        dm += sk ? (R)0 : (R)1;
        b1 |= (!sk && !finite_(r1[k])) ? 1 : 0;
        b2 |= (!sk && r2 && !finite_(r2[k])) ? 1 : 0;
        if (sk) {
            r1[k] = 0;
            if (r2) r2[k] = 0;
        }
        Z[k * ldz + n] = sk ? (R)0 : r1[k];
        Z[k * ldz + n + 1] = (sk || !r2) ? (R)0 : r2[k];
    }
    const bool bad1 = __syncthreads_or(b1), bad2 = __syncthreads_or(b2);
    // a diagonal covariance (the usual noise model; the kept entries only) needs no elimination: the solve is a division per component
    int offd = 0;
    for (int i = tid / 64; i < n; i += NWV)
        for (int j = tid & 63; j < i; j += 64) offd |= Z[i * ldz + j] != (R)0 ? 1 : 0;
    const bool diag = !__syncthreads_or(offd);
    R hl;
    bool ok;
    if (diag) {
        int bad = 0;
        R hs = 0;
        for (int k = tid; k < n; k += NT) {
            const bool sk = skip && skip[k];
            const R ck = Z[k * ldz + k];
            if (!sk) {
                bad |= !(ck > (R)0);
                hs += (R)0.5 * log_(ck);
            }
            const R inv = sk ? (R)0 : (R)1 / ck;
            Z[k * ldz + n] *= inv;
            Z[k * ldz + n + 1] *= inv;
        }
        ok = !__syncthreads_or(bad);
        hl = block_sum<R>(hs, rowbuf, tid);
    } else {
        ok = spd_solve<R>(Z, ldz, n, nct, skip, rowbuf, piv, &hl, tid, true);
    }
    R q1 = 0, q2 = 0;
    for (int k = tid; k < n; k += NT) {
        q1 += r1[k] * Z[k * ldz + n];
        if (r2) q2 += r2[k] * Z[k * ldz + n + 1];
    }
    q1 = block_sum<R>(q1, rowbuf, tid);
    q2 = block_sum<R>(q2, rowbuf, tid);
    const R dimr = block_sum<R>(dm, rowbuf, tid);
    const R cst = -hl - (R)(0.5 * LOG_2PI) * dimr;
    o1 = ok ? (R)-0.5 * q1 + cst : r_nan<R>();
    o2 = ok ? (R)-0.5 * q2 + cst : r_nan<R>();
    if (bad1 || isnan_(o1)) o1 = 0;
    if (bad2 || isnan_(o2)) o2 = 0;
    __syncthreads();
}

// joint log-density (base.py:99-166): item (s, t): observation term at t + transition into t (t >= 1) or initial term (t = 0)
static size_t lds_logpdf(size_t s, int d, int p) {
    const int n = std::max(d, p);
    return al16(n * (size_t)ldp_(n + 2) * s) + 4 * al16(d * s) + 3 * al16(n * s) + al16((2 * (n + 3) + NWV) * s) + al16(n) + 512;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_logpdf(LogpdfArgs a, R* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, p = a.dy, T = a.d.T, nmax = d > p ? d : p;
    const int s = blockIdx.x / T, t = blockIdx.x - s * T, c = s / a.d.B, b = s % a.d.B;
    Bump L{smem};
    R* Lb = L.take<R>(nmax * ldp_(nmax + 2));
    R* x = L.take<R>(d);
    R* xq = L.take<R>(d);
    R* rd_ = L.take<R>(d);
    (void)L.take<R>(d);
    R* ro = L.take<R>(nmax);
    R* piv = L.take<R>(nmax);
    R* rowbuf = L.take<R>(2 * (nmax + 3) + NWV);
    unsigned char* skip = L.take<unsigned char>(nmax);
    load_vec<R>(x, at<R>(a.xs, c, t, b), d, tid);
    const R* Hg = at<R>(a.Hs, c, t, b);
    const R* cg = at<R>(a.cs, c, t, b);
    const R* yg = at<R>(a.ys, c, t, b);
    gemv_rows<R>(p, d, Hg, d, x, ro, tid);
    for (int k = tid; k < p; k += NT) {
        ro[k] = yg[k] - (cg[k] + ro[k]);
        skip[k] = (a.nan_policy == 1) && !finite_(yg[k]);
    }
    __syncthreads();
    R o_obs, o_dyn, dummy;
    gauss2<R>(at<R>(a.Rs, c, t, b), p, a.nan_policy == 1 ? skip : nullptr, ro, nullptr, Lb, piv, rowbuf, tid, o_obs, dummy);
    if (t == 0) {
        const R* m0 = at<R>(a.m0, c, 0, b);
        for (int k = tid; k < d; k += NT) rd_[k] = x[k] - m0[k];
        __syncthreads();
        gauss2<R>(at<R>(a.P0, c, 0, b), d, nullptr, rd_, nullptr, Lb, piv, rowbuf, tid, o_dyn, dummy);
    } else {
        load_vec<R>(xq, at<R>(a.xs, c, t - 1, b), d, tid);
        const R* Fg = at<R>(a.Fs, c, t - 1, b);
        const R* bg = at<R>(a.bs, c, t - 1, b);
        gemv_rows<R>(d, d, Fg, d, xq, rd_, tid);
        for (int k = tid; k < d; k += NT) rd_[k] = x[k] - (rd_[k] + bg[k]);
        __syncthreads();
        gauss2<R>(at<R>(a.Qs, c, t - 1, b), d, nullptr, rd_, nullptr, Lb, piv, rowbuf, tid, o_dyn, dummy);
    }
    if (tid == 0) part[(long long)s * T + t] = o_obs + o_dyn;
}

// the five sums of one sweep of the LG_CONCAT device model (body_sweep_logpdf of kalman_bodies.h); part [5][C][T]
static size_t lds_sweep_logpdf(size_t s, int d, int po) {
    const int n = std::max(d, po);
    return al16(n * (size_t)ldp_(n + 2) * s) + 7 * al16(d * s) + 4 * al16(n * s) + al16((2 * (n + 3) + NWV) * s) + al16(n) + 512;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_sweep_logpdf(SweepLogpdfArgs a, R* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, po = a.po, T = a.d.T, C = a.d.C, nmax = d > po ? d : po;
    const int c = blockIdx.x / T, t = blockIdx.x - c * T;
    Bump L{smem};
    R* Lb = L.take<R>(nmax * ldp_(nmax + 2));
    R* x = L.take<R>(d);
    R* xp = L.take<R>(d);
    R* u = L.take<R>(d);
    R* xq = L.take<R>(d);
    R* xpq = L.take<R>(d);
    R* d1 = L.take<R>(d);
    R* d2 = L.take<R>(d);
    R* r1 = L.take<R>(nmax);
    R* r2 = L.take<R>(nmax);
    R* piv = L.take<R>(nmax);
    R* rowbuf = L.take<R>(2 * (nmax + 3) + NWV);
    unsigned char* skip = L.take<unsigned char>(nmax);
    load_vec<R>(x, at<R>(a.x, c, t, 0), d, tid);
    load_vec<R>(xp, at<R>(a.xp, c, t, 0), d, tid);
    load_vec<R>(u, at<R>(a.u, c, t, 0), d, tid);
    const R* Hg = at<R>(a.Hs, c, t, 0);
    const R* cg = at<R>(a.cs, c, t, 0);
    const R* yg = at<R>(a.ys, c, t, 0);
    gemv_rows<R>(po, d, Hg, d, xp, r1, tid);
    gemv_rows<R>(po, d, Hg, d, x, r2, tid);
    int bp = 0, bx = 0;
    for (int k = tid; k < po; k += NT) {
        r1[k] = yg[k] - (cg[k] + r1[k]);
        r2[k] = yg[k] - (cg[k] + r2[k]);
        skip[k] = (a.nan_policy == 1) && !finite_(yg[k]);
        bp |= (!skip[k] && !finite_(r1[k])) ? 1 : 0;
        bx |= (!skip[k] && !finite_(r2[k])) ? 1 : 0;
    }
    const bool badobs_p = __syncthreads_or(bp), badobs_x = __syncthreads_or(bx);
    R ob_p, ob_x, pr_p, pr_x;
    gauss2<R>(at<R>(a.Rs, c, t, 0), po, a.nan_policy == 1 ? skip : nullptr, r1, r2, Lb, piv, rowbuf, tid, ob_p, ob_x);
    // auxiliary block N(u; x, delta/2 I) and the MH correction (generic.py:103-105)
    const R hd = (R)(0.5 * arg_delta(a)), sd = sqrt_(hd);
    R q1 = 0, q2 = 0, corr = 0;
    int ib1 = 0, ib2 = 0;
    for (int k = tid; k < d; k += NT) {
        const R e1 = u[k] - xp[k], e2 = u[k] - x[k];
        ib1 |= finite_(e1) ? 0 : 1;
        ib2 |= finite_(e2) ? 0 : 1;
        const R z1 = e1 / sd, z2 = e2 / sd;
        q1 += z1 * z1;
        q2 += z2 * z2;
        const R f1 = xp[k] - u[k], f2 = x[k] - u[k];
        corr += (f1 * f1 - f2 * f2) / (R)arg_delta(a);
    }
    const bool b1 = __syncthreads_or(ib1), b2 = __syncthreads_or(ib2);
    q1 = block_sum<R>(q1, rowbuf, tid);
    q2 = block_sum<R>(q2, rowbuf, tid);
    corr = block_sum<R>(corr, rowbuf, tid);
    const R cst = -(R)d * log_(sd) - (R)(0.5 * LOG_2PI) * (R)d;
    const R ax_p = b1 ? (R)0 : (R)-0.5 * q1 + cst, ax_x = b2 ? (R)0 : (R)-0.5 * q2 + cst;
    const bool ref = a.nan_policy == 0;
    const R cc_p = (ref && (b1 || badobs_p)) ? (R)0 : ax_p + ob_p;
    const R cc_x = (ref && (b2 || badobs_x)) ? (R)0 : ax_x + ob_x;
    if (t == 0) {
        const R* m0 = at<R>(a.m0, c, 0, 0);
        for (int k = tid; k < d; k += NT) d1[k] = xp[k] - m0[k], d2[k] = x[k] - m0[k];
        __syncthreads();
        gauss2<R>(at<R>(a.P0, c, 0, 0), d, nullptr, d1, d2, Lb, piv, rowbuf, tid, pr_p, pr_x);
    } else {
        load_vec<R>(xq, at<R>(a.x, c, t - 1, 0), d, tid);
        load_vec<R>(xpq, at<R>(a.xp, c, t - 1, 0), d, tid);
        const R* Fg = at<R>(a.Fs, c, t - 1, 0);
        const R* bg = at<R>(a.bs, c, t - 1, 0);
        gemv_rows<R>(d, d, Fg, d, xpq, d1, tid);
        gemv_rows<R>(d, d, Fg, d, xq, d2, tid);
        for (int k = tid; k < d; k += NT) {
            d1[k] = xp[k] - (d1[k] + bg[k]);
            d2[k] = x[k] - (d2[k] + bg[k]);
        }
        __syncthreads();
        gauss2<R>(at<R>(a.Qs, c, t - 1, 0), d, nullptr, d1, d2, Lb, piv, rowbuf, tid, pr_p, pr_x);
    }
    if (tid == 0) {
        const long long CT = (long long)C * T, o = (long long)c * T + t;
        part[o] = cc_p + pr_p;
        part[CT + o] = cc_x + pr_x;
        part[2 * CT + o] = ob_p + pr_p;
        part[3 * CT + o] = ob_x + pr_x;
        part[4 * CT + o] = corr;
    }
}

#include "wide_shared.h"

// ---- host side -----------------------------------------------------------------------------------------------------------------
constexpr int WMAXLEV = 4;
struct WPlan {
    int E, nchunk;  // level 0: chunks of E consecutive elements
    // filter scan only: the nchunk level-0 aggregates are scanned by a tree.  Level l (l < nlev) groups El[l] consecutive aggregates of the cnt[l]
    // below it into cnt[l + 1] = ceil(cnt[l] / El[l]); the top cnt[nlev] are scanned by one workgroup per sequence.  cnt[0] = nchunk.
    // Critical path ~ sum_l 1.7 (El[l] - 1) + cnt[nlev] - 1 general combines (reduce + the cheaper (b, C) down pass per level): groups of
    // four keep it at ~18 for 256 chunks where one level of 21 cost 51.
    int nlev, El[WMAXLEV], cnt[WMAXLEV + 1];
};
static WPlan plan(const auxssm_ctx* h, int S, int n, int parallel) {
    WPlan p{};
    p.E = n > 0 ? n : 1;
    p.nchunk = 1;
    p.cnt[0] = 1;
    if (!parallel || n <= 3) return p;
    // one sequence: a chunk per CU; many sequences: two rounds of workgroups; never chunks shorter than ~sqrt(n / 2) steps
    long long nchunk = std::max(1ll, std::min(((long long)2 * h->num_cu + S - 1) / S, std::max((long long)std::sqrt(2.0 * n), (long long)h->num_cu / S)));
    if (const char* ev = getenv("AUXSSM_WIDE_NCHUNK")) {  // tuning/debug override
        const long long v = atoll(ev);
        if (v >= 1 && v <= n) nchunk = v;
    }
    nchunk = std::min<long long>(nchunk, n);
    p.E = (int)((n + nchunk - 1) / nchunk);
    p.nchunk = (n + p.E - 1) / p.E;
    p.cnt[0] = p.nchunk;
    int group = 4;
    if (const char* ev = getenv("AUXSSM_WIDE_GROUP")) group = std::max(2, atoi(ev));
    while (p.nlev < WMAXLEV && p.cnt[p.nlev] > 2 * group && !getenv("AUXSSM_WIDE_ONE_LEVEL")) {
        p.El[p.nlev] = group;
        p.cnt[p.nlev + 1] = (p.cnt[p.nlev] + group - 1) / group;
        ++p.nlev;
    }
    return p;
}

template <typename K> static int set_lds(K kern, size_t bytes) {
    if (bytes > LDS_BUDGET) {
        set_error("internal: wide-path kernel needs %zu bytes of LDS (> %zu)", bytes, LDS_BUDGET);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    if (bytes > 48 * 1024) AX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return AUXSSM_OK;
}
#define WK_LAUNCH(kern, grid, lds, ...)                                                           \
    do {                                                                                          \
        int _rc = set_lds(kern, lds);                                                             \
        if (_rc) return _rc;                                                                      \
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3(NT), lds, h->stream, __VA_ARGS__);  \
    } while (0)

static bool fold_enabled() {
    static const bool on = !getenv("AUXSSM_WIDE_NO_FOLD");
    return on;
}
template <typename R> static bool use_fold(int d, int p) { return fold_enabled() && fold_fits(sizeof(R), d) && lds_obs_info(sizeof(R), d, p) <= LDS_BUDGET; }
template <typename R> static size_t filter_ws_one(const auxssm_ctx* h, int S, int n, int parallel, int d) {
    const WPlan p = plan(h, S, n, parallel);
    const size_t ne = (size_t)fe_size(d);
    size_t nagg = 0;
    for (int l = 0; l <= p.nlev; ++l) nagg += p.cnt[l];
    return ((size_t)S * std::max(n, 1) * ne + (size_t)S * nagg * (ne + (size_t)pre_size(d)) + (size_t)S * (std::max(n, 1) + 2)) * sizeof(R) + 8192 + 256 * (2 * (WMAXLEV + 1) + 4);
}
// the chain-shared filter (wide_shared.h): blocks of CB sequences ride as matrix columns; chunks of E transitions
struct SPlan {
    int CB, ncb, E, nchunk;
};
static SPlan shared_plan(const auxssm_ctx* h, int S, int n, int d, int p, size_t sR) {
    SPlan sp{};
    sp.CB = 64;
    while (sp.CB > 16 && (sp.CB / 2 >= S || (long long)p * sp.CB > (long long)SH_NY * NT || std::max(lds_mean_down(sR, d, p, sp.CB), lds_mean_reduce(sR, d, p, sp.CB)) > LDS_BUDGET))
        sp.CB /= 2;
    sp.ncb = (S + sp.CB - 1) / sp.CB;
    long long nchunk = std::max(1, std::min(n, std::max(h->num_cu / sp.ncb, 32)));
    if (const char* ev = getenv("AUXSSM_WIDE_SHARED_NCHUNK")) {
        const long long v = atoll(ev);
        if (v >= 1 && v <= n) nchunk = v;
    }
    sp.E = (int)((n + nchunk - 1) / nchunk);
    sp.nchunk = (n + sp.E - 1) / sp.E;
    return sp;
}
template <typename R> static size_t shared_ws(const auxssm_ctx* h, int S, int n, int d, int p) {
    const SPlan sp = shared_plan(h, S, n, d, p, sizeof(R));
    const GRow g(d, p);
    const size_t Spad = (size_t)sp.ncb * sp.CB;
    return ((size_t)n * g.size + 2 * (size_t)sp.nchunk * d * ldp_(d) + 4 * (size_t)sp.nchunk * d * Spad + (size_t)S * sp.nchunk + 2 * (size_t)S + 64) * sizeof(R) + 20 * 256;
}
template <typename R> static size_t filter_ws_d(const auxssm_ctx* h, const KDims& kd, int parallel, int d, int p) {
    const int S = kd.S(), n = kd.n();
    size_t need = filter_ws_one<R>(h, S, n, parallel, d);
    if (S >= 2 && parallel && n >= 4 && p > 0) need = std::max(need, filter_ws_one<R>(h, 1, n, parallel, d) + shared_ws<R>(h, S, n, d, p));
    return need;
}
template <typename R> static int run_filter_shared(auxssm_ctx* h, const FilterArgs& a, void* ell_out);

template <typename R> int run_filter(auxssm_ctx* h, const FilterArgs& a, int parallel, void* ell_out) {
    const int S = a.d.S(), n = a.d.n(), d = a.dx, p = a.dy;
    if (S >= 2 && parallel && n >= 4) {  // sequences that share every model parameter: one matrix recursion, the means as columns (wide_shared.h)
        const int rc = run_filter_shared<R>(h, a, ell_out);
        if (rc != 1) return rc;  // 1: not applicable (strides, option, observation patterns, LDS), nothing written -- the per-sequence path below
    }
    const WPlan pl = plan(h, S, n, parallel);
    const bool fold = use_fold<R>(d, p);
    const size_t ne = (size_t)fe_size(d), np = (size_t)pre_size(d);
    R* elem = (R*)ws_take(h, (size_t)S * std::max(n, 1) * (fold ? (size_t)info_size(d) : ne) * sizeof(R));  // fold: one InfoRow per step
    R *aggs[WMAXLEV + 1], *pre[WMAXLEV + 1];
    for (int l = 0; l <= pl.nlev; ++l) {
        aggs[l] = (R*)ws_take(h, (size_t)S * pl.cnt[l] * ne * sizeof(R));
        pre[l] = (R*)ws_take(h, (size_t)S * pl.cnt[l] * np * sizeof(R));
        if (!aggs[l] || !pre[l]) return AUXSSM_ERR_NOMEM;
    }
    R* ell0 = (R*)ws_take(h, (size_t)S * sizeof(R));
    R* ellz = (R*)ws_take(h, (size_t)S * sizeof(R));
    if (!elem || !ell0 || !ellz) return AUXSSM_ERR_NOMEM;
    FilterArgs fa = a;
    fa.ell0 = ell0;
    fa.ellz = ellz;
    {
        ProfScope ps(h, AUXSSM_K_FILTER_INIT);
        WK_LAUNCH((wk_filter_t0<R>), S, lds_filter_t0(sizeof(R), d, p), fa);
        if (n > 0) {
            if (fold) WK_LAUNCH((wk_obs_info<R>), (long long)S * n, lds_obs_info(sizeof(R), d, p), fa, elem);
            else WK_LAUNCH((wk_filter_init<R>), (long long)S * n, lds_filter_init(sizeof(R), d, p), fa, elem);
        }
    }
    if (n > 0) {
        const size_t lc = lds_combine(sizeof(R), d);
        {
            ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
            if (pl.nchunk > 1) {
                if (fold) WK_LAUNCH((wk_fold_reduce<R>), (long long)S * pl.nchunk, lds_fold(sizeof(R), d, true), fa, (const R*)elem, aggs[0], pl.E, pl.nchunk);
                else WK_LAUNCH((wk_scan_reduce<R>), (long long)S * pl.nchunk, lc, (const R*)elem, aggs[0], n, pl.E, pl.nchunk, d);
                // the tree over the chunk aggregates: up (group aggregates), top (one workgroup per sequence), down ((b, C) prefixes per group)
                for (int l = 0; l < pl.nlev; ++l)
                    WK_LAUNCH((wk_scan_reduce<R>), (long long)S * pl.cnt[l + 1], lc, (const R*)aggs[l], aggs[l + 1], pl.cnt[l], pl.El[l], pl.cnt[l + 1], d);
                WK_LAUNCH((wk_scan_aggs<R>), S, lc, (const R*)aggs[pl.nlev], pre[pl.nlev], pl.cnt[pl.nlev], d);
                for (int l = pl.nlev - 1; l >= 0; --l)
                    WK_LAUNCH((wk_scan_down_pre<R>), (long long)S * pl.cnt[l + 1], lc, (const R*)aggs[l], (const R*)pre[l + 1], pre[l], pl.cnt[l], pl.El[l], pl.cnt[l + 1], d);
            }
            if (fold) WK_LAUNCH((wk_fold_down<R>), (long long)S * pl.nchunk, lds_fold(sizeof(R), d, false), fa, (const R*)elem, (const R*)pre[0], pl.E, pl.nchunk);
            else WK_LAUNCH((wk_scan_down<R>), (long long)S * pl.nchunk, lc, fa, (const R*)elem, (const R*)pre[0], pl.E, pl.nchunk);
        }
    }
    // ell = t = 0 term + the scan's log-scale (the reference's second pass, filtering.py:60-62, is not needed)
    hipLaunchKernelGGL((wk_reduce<R>), dim3(a.d.C), dim3(NT), 0, h->stream, (const R*)ellz, (const R*)ell0, a.d.B, (long long)(n > 0 ? 1 : 0), (R*)ell_out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// returns 1 when the shared form does not apply (the caller runs the per-sequence path), else a status
template <typename R> static int run_filter_shared(auxssm_ctx* h, const FilterArgs& a, void* ell_out) {
    const int S = a.d.S(), n = a.d.n(), d = a.dx, p = a.dy;
    static const bool off = getenv("AUXSSM_WIDE_SHARED") && atoi(getenv("AUXSSM_WIDE_SHARED")) == 0;
    if (off || !h->share_model || a.aux_on || a.tab || p < 1) return 1;
    if (a.pc && a.tab_ready && !a.mask_ys.ptr) return 1;  // (a reused table was built on the carrier's pattern)
    for (const Arr* q : {&a.P0, &a.Fs, &a.Qs, &a.bs, &a.Hs, &a.Rs, &a.cs})
        if (q->sc != 0 || q->sb != 0) return 1;
    const SPlan sp = shared_plan(h, S, n, d, p, sizeof(R));
    if ((long long)p * sp.CB > (long long)SH_NY * NT || (long long)d * sp.CB > 6ll * NT || (long long)std::max(d, p) * ldp_(std::max(d, p)) > (long long)SH_NR * NT ||
        d + 2 * p + 2 > 3 * NT || !spd_fits(p, 2 * p + d))
        return 1;
    const size_t l_tab = lds_gain_tab(sizeof(R), d, p), l_red = lds_mean_reduce(sizeof(R), d, p, sp.CB), l_agg = lds_mean_aggs(sizeof(R), d, sp.CB),
                 l_down = lds_mean_down(sizeof(R), d, p, sp.CB);
    if (std::max({l_tab, l_red, l_agg, l_down}) > LDS_BUDGET) return 1;
    const GRow g(d, p);
    const size_t Spad = (size_t)sp.ncb * sp.CB;
    const size_t mark = h->ws_off;
    int* flag = (int*)ws_take(h, 256);
    R* tab = a.pc ? (R*)a.pc : (R*)ws_take(h, (size_t)n * g.size * sizeof(R));  // (the caller's buffer: a sweep's second filter reuses the rows)
    const bool reuse = a.pc && a.tab_ready;
    R* aggA = (R*)ws_take(h, (size_t)sp.nchunk * d * ldp_(d) * sizeof(R));
    R* aggG = (R*)ws_take(h, (size_t)sp.nchunk * d * Spad * sizeof(R));
    R* pre = (R*)ws_take(h, (size_t)sp.nchunk * d * Spad * sizeof(R));
    // second level of the chunk-composite scan: groups of GRP composites (none below 2 GRP chunks)
    static const int GRP = getenv("AUXSSM_WIDE_SHARED_GROUP") ? std::max(2, atoi(getenv("AUXSSM_WIDE_SHARED_GROUP"))) : 16;
    const int nsup = sp.nchunk >= 2 * GRP ? (sp.nchunk + GRP - 1) / GRP : 1;
    const size_t l_grp = lds_mean_group(sizeof(R), d, sp.CB);
    R *supA = nullptr, *supG = nullptr, *presup = nullptr;
    if (nsup > 1) {
        if (l_grp > LDS_BUDGET) {
            h->ws_off = mark;
            return 1;
        }
        supA = (R*)ws_take(h, (size_t)nsup * d * ldp_(d) * sizeof(R));
        supG = (R*)ws_take(h, (size_t)nsup * d * Spad * sizeof(R));
        presup = (R*)ws_take(h, (size_t)nsup * d * Spad * sizeof(R));
        if (!supA || !supG || !presup) return AUXSSM_ERR_NOMEM;
    }
    R* ellpart = (R*)ws_take(h, (size_t)S * sp.nchunk * sizeof(R));
    R* ell0 = (R*)ws_take(h, (size_t)S * sizeof(R));
    R* ell_seq0 = (R*)ws_take(h, 256);
    if (!flag || !tab || !aggA || !aggG || !pre || !ellpart || !ell0 || !ell_seq0) return AUXSSM_ERR_NOMEM;
    // the one decision that needs the data: do all sequences miss the same observations?  4 bytes back to the host and one stream synchronisation -- for
    // auxssm_kalman_filter only (documented there).  A SWEEP never takes it: its caller passes the pattern as a chain-independent carrier (mask_ys; the first-order
    // SV factory's pseudo-observations are finite by construction), so the sweeps of a sampling loop stay asynchronous.
    if (!a.mask_ys.ptr) {
        AX_HIP(hipMemsetAsync(flag, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL((wk_mask_check<R>), dim3(std::min<long long>(1024, ((long long)a.d.T * p + 255) / 256)), dim3(256), 0, h->stream, a, flag);
        int differ = 0;
        AX_HIP(hipMemcpyAsync(&differ, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        AX_HIP(hipStreamSynchronize(h->stream));
        if (differ) {
            h->ws_off = mark;
            return 1;
        }
    }
    FilterArgs fa = a;
    fa.ell0 = ell0;
    if (!reuse) {   // the matrix filter: sequence 0's slots through the per-sequence path, on sequence 0's observations or on the caller's pattern carrier (its Ps rows are
        // final; its means and its ell are recomputed below with the others')
        FilterArgs a1 = a;
        a1.pc = nullptr;
        a1.tab_ready = 0;
        a1.d = KDims{1, a.d.T, 1};
        if (a.mask_ys.ptr) a1.ys = a.mask_ys;
        a1.mask_ys = Arr{nullptr, 0, 0, 0, 1};
        const int rc = run_filter<R>(h, a1, 1, ell_seq0);
        if (rc) return rc;
    }
    {
        ProfScope ps(h, AUXSSM_K_FILTER_INIT);
        WK_LAUNCH((wk_filter_t0<R>), S, lds_filter_t0(sizeof(R), d, p), fa);  // every sequence's own t = 0 update: ms[., 0], Ps[., 0], its ell term (after the matrix
                                                                                // filter, whose t = 0 mean on a carrier is not sequence 0's)
    }
    // the other sequences' covariance slots are copies of sequence 0's: pure memory traffic that only needs the matrix filter -- on the fork stream,
    // beside the gain table (matrix cores / latency) and the one-workgroup pass over the chunk composites
    bool forked = false;
    const bool prof_all = h->prof.kernel_id == AUXSSM_K_ALL && h->prof.max_launches > 0;
    const bool ps_once = a.Ps.sc == 0 && a.Ps.sb == 0;  // the caller keeps ONE copy of the covariances (the matrix filter wrote it): nothing to broadcast
    if (ps_once) {
    } else if (!prof_all && !getenv("AUXSSM_WIDE_NO_FORK")) {
        if (!h->fork_stream) {
            AX_HIP(hipStreamCreateWithFlags(&h->fork_stream, hipStreamNonBlocking));
            AX_HIP(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
            AX_HIP(hipEventCreateWithFlags(&h->join_ev, hipEventDisableTiming));
        }
        AX_HIP(hipEventRecord(h->fork_ev, h->stream));
        AX_HIP(hipStreamWaitEvent(h->fork_stream, h->fork_ev, 0));
        hipLaunchKernelGGL((wk_ps_bcast<R>), dim3(n), dim3(NT), 0, h->fork_stream, fa);
        AX_HIP(hipEventRecord(h->join_ev, h->fork_stream));
        forked = true;
    } else {
        ProfScope ps(h, AUXSSM_K_SELECT);  // (profiled runs keep everything on the one stream: the group times then add up)
        hipLaunchKernelGGL((wk_ps_bcast<R>), dim3(n), dim3(NT), 0, h->stream, fa);
    }
    if (!reuse) {
        ProfScope ps(h, AUXSSM_K_FILTER_TAB);
        WK_LAUNCH((wk_gain_tab<R>), n, l_tab, fa, tab);
    }
    {
        ProfScope ps(h, AUXSSM_K_FILTER_ELL);
#define AX_MEAN(NRI)                                                                                                                                                         \
    do {                                                                                                                                                                  \
        if (sp.nchunk > 1) WK_LAUNCH((wk_mean_reduce<R, NRI>), (long long)sp.nchunk * sp.ncb, l_red, fa, (const R*)tab, aggA, aggG, sp.E, sp.ncb, sp.CB);                  \
        if (nsup > 1) {  /* two levels: group composites, their sequential pass, then every group's chunk starts in parallel */                                         \
            WK_LAUNCH((wk_mean_group<R, NRI>), (long long)nsup * sp.ncb, l_grp, fa, (const R*)aggA, (const R*)aggG, supA, supG, sp.nchunk, sp.ncb, sp.CB, GRP);            \
            WK_LAUNCH((wk_mean_aggs<R, NRI>), sp.ncb, l_agg, fa, (const R*)supA, (const R*)supG, presup, nsup, sp.ncb, sp.CB, nsup, (const R*)nullptr);                    \
            WK_LAUNCH((wk_mean_aggs<R, NRI>), (long long)nsup * sp.ncb, l_agg, fa, (const R*)aggA, (const R*)aggG, pre, sp.nchunk, sp.ncb, sp.CB, GRP, (const R*)presup);  \
        } else {                                                                                                                                                          \
            WK_LAUNCH((wk_mean_aggs<R, NRI>), sp.ncb, l_agg, fa, (const R*)aggA, (const R*)aggG, pre, sp.nchunk, sp.ncb, sp.CB, sp.nchunk, (const R*)nullptr);            \
        }                                                                                                                                                                 \
        WK_LAUNCH((wk_mean_down<R, NRI>), (long long)sp.nchunk * sp.ncb, l_down, fa, (const R*)tab, (const R*)pre, ellpart, sp.E, sp.nchunk, sp.ncb, sp.CB);              \
    } while (0)
        if ((long long)std::max(d, p) * ldp_(std::max(d, p)) <= 5ll * NT) AX_MEAN(5);
        else AX_MEAN(SH_NR);
#undef AX_MEAN
    }
    hipLaunchKernelGGL((wk_reduce<R>), dim3(a.d.C), dim3(NT), 0, h->stream, (const R*)ellpart, (const R*)ell0, a.d.B, (long long)sp.nchunk, (R*)ell_out);
    if (forked) AX_HIP(hipStreamWaitEvent(h->stream, h->join_ev, 0));
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// chunks of the shared sampler's vector scan: the chunk walks (twice E = T / nchunk dependent products) against the one-workgroup pass over the nchunk
// composites -- nchunk ~ sqrt(2 T) balances them (T = 250: 22 chunks of 12; the filter's plan would make 250 chunks of one step and a 250-step serial pass)
static WPlan samp_plan(int T, int parallel) {
    WPlan p{};
    p.E = T > 0 ? T : 1;
    p.nchunk = 1;
    p.cnt[0] = 1;
    if (!parallel || T <= 3) return p;
    long long nchunk = std::max(1ll, (long long)std::sqrt(2.0 * T));
    if (const char* ev = getenv("AUXSSM_WIDE_SAMP_NCHUNK")) {
        const long long v = atoll(ev);
        if (v >= 1 && v <= T) nchunk = v;
    }
    p.E = (int)((T + nchunk - 1) / nchunk);
    p.nchunk = (T + p.E - 1) / p.E;
    p.cnt[0] = p.nchunk;
    return p;
}
// column blocks of the shared sampler: as many sequences per workgroup as keep the grid at two rounds of workgroups
static int samp_cb(const auxssm_ctx* h, int S, int nchunk) {
    int CB = 64;
    while (CB > 16 && (long long)((S + CB - 1) / CB) * nchunk < 2ll * h->num_cu && CB / 2 >= 8) CB /= 2;
    return std::min(CB, std::max(S, 1));
}
// sequences sharing the model AND the filtered covariances (Ps with chain / batch stride 0: what the shared filter leaves when its caller asks for one copy):
// the gain / factor tables once, the sequences as columns (wide_shared.h).  Returns 1 when not applicable.
template <typename R> static int run_sample_shared(auxssm_ctx* h, const SampleArgs& a, int parallel) {
    const int S = a.d.S(), T = a.d.T, d = a.dx;
    static const bool off = getenv("AUXSSM_WIDE_SHARED") && atoi(getenv("AUXSSM_WIDE_SHARED")) == 0;
    if (off || !h->share_model || S < 2 || T < 2) return 1;
    for (const Arr* q : {&a.Ps, &a.Fs, &a.Qs, &a.bs})
        if (q->sc != 0 || q->sb != 0) return 1;
    const WPlan pl = samp_plan(T, parallel);
    const int CB = samp_cb(h, S, pl.nchunk), ncb = (S + CB - 1) / CB;
    const size_t l_e = lds_samp_evec(sizeof(R), d, CB), l_s = lds_samp_scan(sizeof(R), d, CB);
    if (std::max(l_e, l_s) > LDS_BUDGET) return 1;
    const size_t ne = (size_t)d * d + d;
    R* gtab = (R*)ws_take(h, (size_t)T * ne * sizeof(R));
    R* ltab = (R*)ws_take(h, (size_t)T * d * d * sizeof(R));
    R* ec = (R*)ws_take(h, (size_t)T * S * d * sizeof(R));
    R* gagg = (R*)ws_take(h, (size_t)pl.nchunk * ne * sizeof(R));
    R* eagg = (R*)ws_take(h, (size_t)pl.nchunk * S * d * sizeof(R));
    R* pre = (R*)ws_take(h, (size_t)pl.nchunk * S * d * sizeof(R));
    if (!gtab || !ltab || !ec || !gagg || !eagg || !pre) return AUXSSM_ERR_NOMEM;
    {
        ProfScope ps(h, AUXSSM_K_SAMPLE_INIT);
        SampleArgs a1 = a;  // the tables: sequence 0's records (its own increments come out of the column pass like everybody's)
        a1.d = KDims{1, T, 1};
        WK_LAUNCH((wk_sample_init<R>), (long long)T, lds_sample_init(sizeof(R), d), a1, gtab, ltab);
        WK_LAUNCH((wk_samp_evec<R>), (long long)T * ncb, l_e, a, (const R*)gtab, (const R*)ltab, ec, ncb, CB);
    }
    ProfScope ps(h, AUXSSM_K_SAMPLE_SCAN);
    if (pl.nchunk > 1) {
        WK_LAUNCH((wk_sscan_reduce<R>), (long long)pl.nchunk, lds_sample_scan(sizeof(R), d), (const R*)gtab, gagg, T, pl.E, pl.nchunk, d);
        WK_LAUNCH((wk_samp_reduce<R>), (long long)pl.nchunk * ncb, l_s, (const R*)gtab, (const R*)ec, eagg, T, pl.E, pl.nchunk, d, S, ncb, CB);
        WK_LAUNCH((wk_samp_aggs<R>), (long long)ncb, l_s, (const R*)gagg, (const R*)eagg, pre, pl.nchunk, d, S, CB);
    }
    WK_LAUNCH((wk_samp_down<R>), (long long)pl.nchunk * ncb, l_s, a, (const R*)gtab, (const R*)ec, (const R*)pre, pl.E, pl.nchunk, ncb, CB);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

template <typename R> int run_sample(auxssm_ctx* h, const SampleArgs& a, int parallel) {
    const int S = a.d.S(), T = a.d.T, d = a.dx;
    if (S >= 2) {
        const int rc = run_sample_shared<R>(h, a, parallel);
        if (rc != 1) return rc;
    }
    const WPlan pl = plan(h, S, T, parallel);
    const size_t ne = (size_t)d * d + d;
    R* elem = (R*)ws_take(h, (size_t)S * T * ne * sizeof(R));
    R* aggs = (R*)ws_take(h, (size_t)S * pl.nchunk * ne * sizeof(R));
    R* pre = (R*)ws_take(h, (size_t)S * pl.nchunk * d * sizeof(R));
    if (!elem || !aggs || !pre) return AUXSSM_ERR_NOMEM;
    {
        ProfScope ps(h, AUXSSM_K_SAMPLE_INIT);
        WK_LAUNCH((wk_sample_init<R>), (long long)S * T, lds_sample_init(sizeof(R), d), a, elem, (R*)nullptr);
    }
    ProfScope ps(h, AUXSSM_K_SAMPLE_SCAN);
    const size_t ls = lds_sample_scan(sizeof(R), d);
    if (pl.nchunk > 1) {
        WK_LAUNCH((wk_sscan_reduce<R>), (long long)S * pl.nchunk, ls, (const R*)elem, aggs, T, pl.E, pl.nchunk, d);
        WK_LAUNCH((wk_sscan_aggs<R>), S, ls, (const R*)aggs, pre, pl.nchunk, d);
    }
    WK_LAUNCH((wk_sscan_down<R>), (long long)S * pl.nchunk, ls, a, (const R*)elem, (const R*)pre, pl.E, pl.nchunk);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

template <typename R> int run_logpdf(auxssm_ctx* h, const LogpdfArgs& a, void* out) {
    const int S = a.d.S(), T = a.d.T;
    R* part = (R*)ws_take(h, (size_t)S * T * sizeof(R));
    if (!part) return AUXSSM_ERR_NOMEM;
    ProfScope ps(h, AUXSSM_K_LOGPDF);
    WK_LAUNCH((wk_logpdf<R>), (long long)S * T, lds_logpdf(sizeof(R), a.dx, a.dy), a, part);
    hipLaunchKernelGGL((wk_reduce<R>), dim3(a.d.C), dim3(NT), 0, h->stream, (const R*)part, (const R*)nullptr, a.d.B, (long long)T, (R*)out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}
// chains on one model: the covariances' inverses and log-determinants once per time step, the chains as columns (wide_shared.h::wk_lp_tab / wk_lp_cols).
// The observations may be per chain under the reference NaN policy (SweepLogpdfArgs::ys_x).  Returns 1 when the form does not apply (nothing enqueued).
template <typename R> int run_sweep_logpdf_shared(auxssm_ctx* h, const SweepLogpdfArgs& a, void* out) {
    const int C = a.d.C, T = a.d.T;
    static const bool sh_off = getenv("AUXSSM_WIDE_SHARED") && atoi(getenv("AUXSSM_WIDE_SHARED")) == 0;
    if (sh_off || !h->share_model || C < 2 || a.u_fly || NT != 1024 || !spd_fits(std::max(a.dx, a.po), 2 * std::max(a.dx, a.po))) return 1;
    for (const Arr* q : {&a.m0, &a.P0, &a.Fs, &a.Qs, &a.bs, &a.Hs, &a.Rs, &a.cs})
        if (q->sc != 0 || q->sb != 0) return 1;
    if ((a.ys.sc != 0 || a.ys_x.ptr) && a.nan_policy != 0) return 1;
    int CB = std::min(64, C);
    while (CB > 8 && lds_lp_cols(sizeof(R), a.dx, a.po, CB) > LDS_BUDGET) CB /= 2;
    const size_t l_tab = lds_lp_tab(sizeof(R), a.dx, a.po), l_cols = lds_lp_cols(sizeof(R), a.dx, a.po, CB);
    if (l_tab > LDS_BUDGET || l_cols > LDS_BUDGET) return 1;
    const size_t mark = h->ws_off;
    R* part = (R*)ws_take(h, (size_t)5 * C * T * sizeof(R));
    R* tab = (R*)ws_take(h, (size_t)T * lp_row(a.dx, a.po) * sizeof(R));
    if (!part || !tab) {
        h->ws_off = mark;
        return 1;
    }
    ProfScope ps(h, AUXSSM_K_LOGPDF);
    const int ncb = (C + CB - 1) / CB;
    WK_LAUNCH((wk_lp_tab<R>), (long long)T, l_tab, a, tab);
    WK_LAUNCH((wk_lp_cols<R>), (long long)T * ncb, l_cols, a, (const R*)tab, part, ncb, CB);
    hipLaunchKernelGGL((wk_reduce<R, Acc>), dim3(5 * C), dim3(NT), 0, h->stream, (const R*)part, (const R*)nullptr, 1, (long long)T, (Acc*)out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}
template <typename R> int run_sweep_logpdf(auxssm_ctx* h, const SweepLogpdfArgs& a, void* out) {
    const int C = a.d.C, T = a.d.T;
    {
        const int rc = run_sweep_logpdf_shared<R>(h, a, out);
        if (rc != 1) return rc;
    }
    R* part = (R*)ws_take(h, (size_t)5 * C * T * sizeof(R));
    if (!part) return AUXSSM_ERR_NOMEM;
    ProfScope ps(h, AUXSSM_K_LOGPDF);
    WK_LAUNCH((wk_sweep_logpdf<R>), (long long)C * T, lds_sweep_logpdf(sizeof(R), a.dx, a.po), a, part);
    hipLaunchKernelGGL((wk_reduce<R, Acc>), dim3(5 * C), dim3(NT), 0, h->stream, (const R*)part, (const R*)nullptr, 1, (long long)T, (Acc*)out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

}  // namespace wide

// ---- workspace sizes: the entry-table signatures carry no (dx, dy), so api.hip asks through these ---------------------------
size_t wide_filter_ws(const auxssm_ctx* h, int dtype, const KDims& kd, int parallel, int d, int p) {
    return dtype == AUXSSM_F32 ? wide::filter_ws_d<float>(h, kd, parallel, d, p) : wide::filter_ws_d<double>(h, kd, parallel, d, p);
}
size_t wide_gain_tab_bytes(int dtype, int T, int d, int p) { return (size_t)std::max(T - 1, 1) * (size_t)wide::GRow(d, p).size * (dtype == AUXSSM_F32 ? 4 : 8) + 256; }
size_t wide_sample_ws(const auxssm_ctx* h, int dtype, const KDims& kd, int parallel, int d) {
    const size_t s = dtype == AUXSSM_F32 ? 4 : 8;
    const wide::WPlan p = wide::plan(h, kd.S(), kd.T, parallel);
    const size_t ne = (size_t)d * d + d;
    const size_t per_seq = ((size_t)kd.S() * kd.T * ne + (size_t)kd.S() * p.nchunk * (ne + d)) * s + 4096;
    const wide::WPlan p1 = wide::samp_plan(kd.T, parallel);  // shared form: two tables, the increments, the chunk composites
    const size_t shared = ((size_t)kd.T * (ne + (size_t)d * d + (size_t)kd.S() * d) + (size_t)p1.nchunk * (ne + 2 * (size_t)kd.S() * d)) * s + 8 * 256;
    return std::max(per_seq, shared);
}
size_t wide_logpdf_ws(int dtype, const KDims& kd) {  // (+ the shared form's table: Q^-1 of at most 85 x 85 and R^-1 of at most 128 x 128 per time step -- wide_fits)
    return ((size_t)5 * kd.S() * kd.T + (size_t)kd.T * (85 * 85 + 128 * 128 + 8)) * (dtype == AUXSSM_F32 ? 4 : 8) + 4096;
}

static size_t ws_unused_f(const auxssm_ctx*, const KDims&, int) { return 0; }
static size_t ws_unused_l(const auxssm_ctx*, const KDims&) { return 0; }

const KalmanEntry* wide_kalman_entry(int dtype) {
    static const KalmanEntry f32{&wide::run_filter<float>, &ws_unused_f, &wide::run_logpdf<float>, &ws_unused_l};
    static const KalmanEntry f64{&wide::run_filter<double>, &ws_unused_f, &wide::run_logpdf<double>, &ws_unused_l};
    return dtype == AUXSSM_F32 ? &f32 : &f64;
}
const SampleEntry* wide_sample_entry(int dtype) {
    static const SampleEntry f32{&wide::run_sample<float>, &ws_unused_f};
    static const SampleEntry f64{&wide::run_sample<double>, &ws_unused_f};
    return dtype == AUXSSM_F32 ? &f32 : &f64;
}
const SweepLogpdfEntry* wide_sweep_logpdf_entry(int dtype) {
    static const SweepLogpdfEntry f32 = [] {
        SweepLogpdfEntry e{&wide::run_sweep_logpdf<float>, &ws_unused_l};
        e.wide_shared = &wide::run_sweep_logpdf_shared<float>;
        return e;
    }();
    static const SweepLogpdfEntry f64 = [] {
        SweepLogpdfEntry e{&wide::run_sweep_logpdf<double>, &ws_unused_l};
        e.wide_shared = &wide::run_sweep_logpdf_shared<double>;
        return e;
    }();
    return dtype == AUXSSM_F32 ? &f32 : &f64;
}
bool wide_fits(int dtype, int dx, int dy, std::string* why) {
    const size_t s = dtype == AUXSSM_F32 ? 4 : 8;
    size_t need = std::max(wide::lds_combine(s, dx), wide::lds_sample_init(s, dx));
    if (dy > 0) need = std::max({need, wide::lds_filter_init(s, dx, dy), wide::lds_filter_t0(s, dx, dy), wide::lds_logpdf(s, dx, dy)});
    const bool regs_ok = dx <= 8 * wide::NWV && 3 * dx + 1 <= 256 && (dy == 0 || (dy <= 8 * wide::NWV && dy + dx + 2 <= 256));
    if (!regs_ok) {
        if (why) {
            char buf[256];
            snprintf(buf, sizeof buf, "(dx=%d, dy=%d) exceeds the register-resident solves of the wide-state path (LDS / register plan: dx <= 85, dy + dx <= 254)", dx, dy);
            *why = buf;
        }
        return false;
    }
    if (need <= wide::LDS_BUDGET) return true;
    if (why) {
        char buf[256];
        snprintf(buf, sizeof buf, "(dx=%d, dy=%d, %s) needs %zu bytes of LDS per workgroup, the device has %zu", dx, dy,
                 dtype == AUXSSM_F32 ? "fp32" : "fp64", need, wide::LDS_BUDGET);
        *why = buf;
    }
    return false;
}

}  // namespace ax

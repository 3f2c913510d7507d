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
//   filter : [A d*d | b d | C d*d | eta d | J d*d]          sampler : [G d*d | e d]
#include <algorithm>
#include <cmath>
#include <string>

#include "ctx.h"

namespace ax {
namespace wide {

constexpr int NT = 256;          // lanes per workgroup
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

// ---- cooperative primitives (all lanes of the workgroup call them; every one ENDS with a barrier) -------------------------

// dst (rows x cols, ld) <- contiguous row-major record
template <typename R> __device__ void load_mat(R* dst, int ld, const R* __restrict__ src, int rows, int cols, int tid) {
    for (int r = tid / 64; r < rows; r += NWV)
        for (int c = tid & 63; c < cols; c += 64) dst[r * ld + c] = src[(long long)r * cols + c];
    __syncthreads();
}
template <typename R> __device__ void store_mat(R* __restrict__ dst, const R* src, int ld, int rows, int cols, int tid) {
    for (int r = tid / 64; r < rows; r += NWV)
        for (int c = tid & 63; c < cols; c += 64) dst[(long long)r * cols + c] = src[r * ld + c];
}
template <typename R> __device__ void load_vec(R* dst, const R* __restrict__ src, int n, int tid) {
    for (int i = tid; i < n; i += NT) dst[i] = src[i];
    __syncthreads();
}

// C (M x N, ldc) = alpha op(A) op(B) + beta C;  op(A) is M x K (TA: stored K x M), op(B) is K x N (TB: stored N x K).
// C must not alias A or B.  4 x 4 register tile per lane.
template <typename R, bool TA, bool TB>
__device__ void gemm(int M, int N, int K, const R* A, int lda, const R* B, int ldb, R* C, int ldc, R alpha, R beta, int tid) {
    const int tn = (N + 3) >> 2, ntile = ((M + 3) >> 2) * tn;
    for (int tile = tid; tile < ntile; tile += NT) {
        const int ti = tile / tn;
        const int i0 = ti << 2, j0 = (tile - ti * tn) << 2;
        int ia[4], jb[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int i = i0 + x < M ? i0 + x : M - 1, j = j0 + x < N ? j0 + x : N - 1;
            ia[x] = TA ? i : i * lda;
            jb[x] = TB ? j * ldb : j;
        }
        R acc[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[x] = 0;
        for (int k = 0; k < K; ++k) {
            const int ka = TA ? k * lda : k, kb = TB ? k : k * ldb;
            R a[4], b[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) a[x] = A[ia[x] + ka], b[x] = B[jb[x] + kb];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x * 4 + y] += a[x] * b[y];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (i0 + x < M && j0 + y < N) {
                    R* q = &C[(i0 + x) * ldc + j0 + y];
                    *q = beta != (R)0 ? alpha * acc[x * 4 + y] + beta * *q : alpha * acc[x * 4 + y];
                }
    }
    __syncthreads();
}
// y (M) = alpha op(A) x + beta y
template <typename R, bool TA> __device__ void gemv(int M, int K, const R* A, int lda, const R* x, R* y, R alpha, R beta, int tid) {
    for (int i = tid; i < M; i += NT) {
        R s = 0;
        for (int k = 0; k < K; ++k) s += (TA ? A[k * lda + i] : A[i * lda + k]) * x[k];
        y[i] = beta != (R)0 ? alpha * s + beta * y[i] : alpha * s;
    }
    __syncthreads();
}
// M <- 0.5 (M + M^T)
template <typename R> __device__ void symmetrise(R* M, int ld, int n, int tid) {
    for (int i = tid / 64; i < n; i += NWV)
        for (int j = i + 1 + (tid & 63); j < n; j += 64) {
            const R v = (R)0.5 * (M[i * ld + j] + M[j * ld + i]);
            M[i * ld + j] = v;
            M[j * ld + i] = v;
        }
    __syncthreads();
}

// In-place lower Cholesky of the LOWER triangle of S (n x n).  skip[k] (may be null): index k is deleted (L_kk = 1,
// off-diagonals 0) -- the NaN-observation masking of filtering.py:89-100.  invd = 1 / diag.  Returns false (uniformly)
// on a non-positive / NaN pivot; the factor then holds NaNs, as JAX's does.  Same subtraction order as smallmat.h.
template <typename R> __device__ bool chol(R* S, int ld, int n, const unsigned char* skip, R* invd, int* flag, int tid) {
    if (tid == 0) *flag = 1;
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const bool skj = skip && skip[j];
        const R s = S[j * ld + j];
        const R ljj = skj ? (R)1 : sqrt_(s);
        const R inv = (R)1 / ljj;
        __syncthreads();
        if (tid == 0) {
            S[j * ld + j] = ljj;
            invd[j] = inv;
            if (!skj && !(s > (R)0)) *flag = 0;
        }
        for (int i = j + 1 + tid; i < n; i += NT) S[i * ld + j] = (skj || (skip && skip[i])) ? (R)0 : S[i * ld + j] * inv;
        __syncthreads();
        for (int i = j + 1 + tid / 64; i < n; i += NWV) {
            const R lij = S[i * ld + j];
            for (int k = j + 1 + (tid & 63); k <= i; k += 64) S[i * ld + k] -= lij * S[k * ld + j];
        }
        __syncthreads();
    }
    return *flag != 0;
}

// right-hand sides of a cooperative solve: a matrix block (nc columns) plus up to two vectors riding along as extra columns
template <typename R> struct Rhs {
    R* B;
    int ldb, nc;
    R* v1;
    R* v2;
    __device__ int ncol() const { return nc + (v1 ? 1 : 0) + (v2 ? 1 : 0); }
    __device__ R& at(int i, int c) const { return c < nc ? B[i * ldb + c] : (c == nc ? v1[i] : v2[i]); }
};

// X <- L^-1 X
template <typename R> __device__ void trsm_l(const R* L, int ld, int n, const R* invd, const Rhs<R>& X, int tid) {
    const int ncol = X.ncol();
    for (int i = 0; i < n; ++i) {
        for (int c = tid; c < ncol; c += NT) X.at(i, c) *= invd[i];
        __syncthreads();
        for (int r = i + 1 + tid / 64; r < n; r += NWV) {
            const R l = L[r * ld + i];
            for (int c = tid & 63; c < ncol; c += 64) X.at(r, c) -= l * X.at(i, c);
        }
        __syncthreads();
    }
}
// X <- L^-T X
template <typename R> __device__ void trsm_lt(const R* L, int ld, int n, const R* invd, const Rhs<R>& X, int tid) {
    const int ncol = X.ncol();
    for (int i = n - 1; i >= 0; --i) {
        for (int c = tid; c < ncol; c += NT) X.at(i, c) *= invd[i];
        __syncthreads();
        for (int r = tid / 64; r < i; r += NWV) {
            const R l = L[i * ld + r];
            for (int c = tid & 63; c < ncol; c += 64) X.at(r, c) -= l * X.at(i, c);
        }
        __syncthreads();
    }
}

// LU with partial pivoting of W (n x n, destroyed); two RHS groups overwritten by W^-1 RHS (X1 may have nc = 0 and no
// vectors).  scratch: fcol[n], ipiv[n] reals, *piv int.  Returns log|det W| (thread-uniform).
template <typename R>
__device__ R lu_solve(R* W, int ld, int n, const Rhs<R>& X0, const Rhs<R>& X1, R* fcol, R* ipiv, int* piv, int tid) {
    const int n0 = X0.ncol(), n1 = X1.ncol();
    R logdet = 0;
    for (int k = 0; k < n; ++k) {
        if (tid < 64) {  // first row r >= k with the largest |W[r][k]| (NaNs never win), wave 0
            R best = abs_(W[k * ld + k]);
            int idx = k;
            for (int r = k + 1 + tid; r < n; r += 64) {
                const R v = abs_(W[r * ld + k]);
                if (v > best) best = v, idx = r;
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const R ob = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(idx, off, 64);
                if (ob > best || (ob == best && oi < idx)) best = ob, idx = oi;
            }
            if (tid == 0) *piv = idx;
        }
        __syncthreads();
        const int pr = *piv;
        if (pr != k) {
            const int wc = n - k;
            for (int c = tid; c < wc + n0 + n1; c += NT) {
                R* a;
                R* b;
                if (c < wc) a = &W[k * ld + k + c], b = &W[pr * ld + k + c];
                else if (c < wc + n0) a = &X0.at(k, c - wc), b = &X0.at(pr, c - wc);
                else a = &X1.at(k, c - wc - n0), b = &X1.at(pr, c - wc - n0);
                const R t = *a;
                *a = *b;
                *b = t;
            }
        }
        __syncthreads();
        const R inv = (R)1 / W[k * ld + k];
        logdet -= log_(abs_(inv));
        for (int r = k + 1 + tid; r < n; r += NT) fcol[r] = W[r * ld + k] * inv;
        if (tid == 0) ipiv[k] = inv;
        __syncthreads();
        const int wc = n - k - 1;
        for (int r = k + 1 + tid / 64; r < n; r += NWV) {
            const R f = fcol[r];
            for (int c = tid & 63; c < wc + n0 + n1; c += 64) {
                if (c < wc) W[r * ld + k + 1 + c] -= f * W[k * ld + k + 1 + c];
                else if (c < wc + n0) X0.at(r, c - wc) -= f * X0.at(k, c - wc);
                else X1.at(r, c - wc - n0) -= f * X1.at(k, c - wc - n0);
            }
        }
        __syncthreads();
    }
    for (int k = n - 1; k >= 0; --k) {  // back substitution, right-looking
        const R inv = ipiv[k];
        for (int c = tid; c < n0 + n1; c += NT) {
            if (c < n0) X0.at(k, c) *= inv;
            else X1.at(k, c - n0) *= inv;
        }
        __syncthreads();
        for (int r = tid / 64; r < k; r += NWV) {
            const R u = W[r * ld + k];
            for (int c = tid & 63; c < n0 + n1; c += 64) {
                if (c < n0) X0.at(r, c) -= u * X0.at(k, c);
                else X1.at(r, c - n0) -= u * X1.at(k, c - n0);
            }
        }
        __syncthreads();
    }
    return logdet;
}

// ---- observation model of one time step, masked (filtering.py:89-100, :204-213) ------------------------------------------
template <typename R> struct Obs {
    R* H_;   // p x d, ld = ldp_(d); missing rows zeroed
    R* c_;   // p
    R* y;    // p (raw)
    unsigned char* nan;
    int* cnt;  // #observed components
};
template <typename R> __device__ bool load_obs(const Obs<R>& o, const R* Hg, const R* cg, const R* yg, int p, int d, int tid) {
    const int ldd = ldp_(d);
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
// S (lower triangle valid, ld ldp_(p)) = H_ P_ H_^T + R_;  PHt (d x p, ld ldp_(p)) = P_ H_^T.  Rg: the p x p record in
// global memory, upper entries read (as the per-lane path does).
template <typename R>
__device__ void innovation(const Obs<R>& o, const R* P_, const R* Rg, int p, int d, R* PHt, R* S, int tid) {
    const int ldd = ldp_(d), ldp = ldp_(p);
    gemm<R, false, true>(d, p, d, P_, ldd, o.H_, ldd, PHt, ldp, (R)1, (R)0, tid);
    gemm<R, false, false>(p, p, d, o.H_, ldd, PHt, ldp, S, ldp, (R)1, (R)0, tid);
    for (int i = tid / 64; i < p; i += NWV)
        for (int j = tid & 63; j <= i; j += 64) {
            // lower (i, j) <- the value the reference computes for the upper (j, i) entry
            const R r = (o.nan[i] || o.nan[j]) ? (R)0 : Rg[(long long)j * p + i];
            S[i * ldp + j] = S[j * ldp + i] + r;
        }
    __syncthreads();
}
// -0.5 |z|^2 - sum log L_kk - dim/2 log 2 pi over the observed components; NaN / failed factor -> 0 (nansum).  Lane 0's value.
template <typename R> __device__ R ell_from(const R* L, int ldp, const R* z, const unsigned char* nan, int p, int dim, bool ok) {
    R q = 0, logdet = 0;
    for (int k = 0; k < p; ++k) {
        q += z[k] * z[k];
        logdet += (nan && nan[k]) ? (R)0 : log_(L[k * ldp + k]);
    }
    R ell = (R)-0.5 * q - logdet - (R)(0.5 * LOG_2PI) * (R)dim;
    if (!ok) ell = r_nan<R>();
    return isnan_(ell) ? (R)0 : ell;
}

// ---- t = 0 measurement update (sequential_update, filtering.py:83-130); one workgroup per sequence ------------------------
static size_t lds_filter_t0(size_t s, int d, int p) {
    const size_t ldd = ldp_(d), ldp = ldp_(p);
    return al16(d * ldd * s) + 3 * al16(p * ldd * s) + al16(p * ldp * s) + al16(d * s) * 2 + 5 * al16(p * s) + al16(p) + 64;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_filter_t0(FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x, c = s / a.d.B, b = s % a.d.B, d = a.dx, p = a.dy;
    const int ldd = ldp_(d), ldp = ldp_(p);
    Bump L{smem};
    R* P = L.take<R>(d * ldd);
    Obs<R> o;
    o.H_ = L.take<R>(p * ldd);
    R* HP = L.take<R>(p * ldd);
    R* X = L.take<R>(p * ldd);
    R* S = L.take<R>(p * ldp);
    R* m = L.take<R>(d);
    R* dm = L.take<R>(d);
    o.c_ = L.take<R>(p);
    o.y = L.take<R>(p);
    R* yd = L.take<R>(p);
    R* z = L.take<R>(p);
    R* invd = L.take<R>(p);
    o.nan = L.take<unsigned char>(p);
    o.cnt = L.take<int>(1);
    int* flag = L.take<int>(1);
    load_mat<R>(P, ldd, at<R>(a.P0, c, 0, b), d, d, tid);
    load_vec<R>(m, at<R>(a.m0, c, 0, b), d, tid);
    const bool any = load_obs<R>(o, at<R>(a.Hs, c, 0, b), at<R>(a.cs, c, 0, b), at<R>(a.ys, c, 0, b), p, d, tid);
    R* mo = const_cast<R*>(at<R>(a.ms, c, 0, b));
    R* Po = const_cast<R*>(at<R>(a.Ps, c, 0, b));
    if (!any) {  // _passthrough :127-130
        for (int i = tid; i < d; i += NT) mo[i] = m[i];
        store_mat<R>(Po, P, ldd, d, d, tid);
        if (tid == 0) ((R*)a.ell0)[s] = 0;
        return;
    }
    for (int k = tid; k < p; k += NT) {
        R yh = o.c_[k];
        for (int j = 0; j < d; ++j) yh += o.H_[k * ldd + j] * m[j];
        yd[k] = o.nan[k] ? (R)0 : o.y[k] - yh;
        z[k] = yd[k];
    }
    // HP = H_ P (p x d);  S = HP H_^T + R_
    gemm<R, false, false>(p, d, d, o.H_, ldd, P, ldd, HP, ldd, (R)1, (R)0, tid);
    gemm<R, false, true>(p, p, d, HP, ldd, o.H_, ldd, S, ldp, (R)1, (R)0, tid);
    const R* Rg = at<R>(a.Rs, c, 0, b);
    for (int i = tid / 64; i < p; i += NWV)
        for (int j = tid & 63; j <= i; j += 64) S[i * ldp + j] += (o.nan[i] || o.nan[j]) ? (R)0 : Rg[(long long)j * p + i];
    for (int i = tid / 64; i < p; i += NWV)
        for (int j = tid & 63; j < d; j += 64) X[i * ldd + j] = HP[i * ldd + j];
    __syncthreads();
    const bool ok = chol<R>(S, ldp, p, o.nan, invd, flag, tid);
    // z = L^-1 yd;  X = S^-1 HP  (gain^T, :117)
    trsm_l<R>(S, ldp, p, invd, Rhs<R>{X, ldd, d, z, nullptr}, tid);
    const R ell = ell_from<R>(S, ldp, z, o.nan, p, *o.cnt, ok);
    trsm_lt<R>(S, ldp, p, invd, Rhs<R>{X, ldd, d, nullptr, nullptr}, tid);
    // m += X^T yd;  P <- sym(P - X^T HP)
    gemv<R, true>(d, p, X, ldd, yd, dm, (R)1, (R)0, tid);
    gemm<R, true, false>(d, d, p, X, ldd, HP, ldd, P, ldd, (R)-1, (R)1, tid);
    symmetrise<R>(P, ldd, d, tid);
    const R bad = r_nan<R>();
    for (int i = tid; i < d; i += NT) mo[i] = ok ? m[i] + dm[i] : bad;
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) Po[(long long)r * d + q] = ok ? P[r * ldd + q] : bad;
    if (tid == 0) ((R*)a.ell0)[s] = ell;
}

// ---- scan element of transition i -> i + 1 (_filtering_init_one, filtering.py:196-250), information form of kalman_math.h ---
static size_t lds_filter_init(size_t s, int d, int p) {
    const size_t ldd = ldp_(d), ldp = ldp_(p);
    return 5 * al16(d * ldd * s) + al16(p * ldd * s) + al16(d * std::max(ldp, ldd) * s) + al16(p * ldp * s) + 6 * al16(d * s) +
           6 * al16(p * s) + al16(p) + 64;
}
__host__ __device__ inline long long fe_size(int d) { return 3ll * d * d + 2 * d; }

template <typename R> __global__ void __launch_bounds__(NT) wk_filter_init(FilterArgs a, R* __restrict__ elem) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, p = a.dy, n = a.d.T - 1;
    const int s = blockIdx.x / n, i = blockIdx.x - s * n, c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    const int ldd = ldp_(d), ldp = ldp_(p), ldt = ldp > ldd ? ldp : ldd;
    Bump L{smem};
    R* F = L.take<R>(d * ldd);
    R* P_ = L.take<R>(d * ldd);
    R* M = L.take<R>(d * ldd);
    R* MF = L.take<R>(d * ldd);
    R* O = L.take<R>(d * ldd);
    Obs<R> o;
    o.H_ = L.take<R>(p * ldd);
    R* Tm = L.take<R>(d * ldt);
    R* S = L.take<R>(p * ldp);
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
    R* invd = L.take<R>(p);
    (void)L.take<R>(p);
    o.nan = L.take<unsigned char>(p);
    o.cnt = L.take<int>(1);
    int* flag = L.take<int>(1);
    R* e = elem + ((long long)s * n + i) * fe_size(d);
    R* eA = e;
    R* eb = e + d * d;
    R* eC = eb + d;
    R* eeta = eC + d * d;
    R* eJ = eeta + d;

    load_mat<R>(F, ldd, at<R>(a.Fs, c, i, b), d, d, tid);
    load_mat<R>(P_, ldd, at<R>(a.Qs, c, i, b), d, d, tid);
    load_vec<R>(bd, at<R>(a.bs, c, i, b), d, tid);
    const bool any = load_obs<R>(o, at<R>(a.Hs, c, t, b), at<R>(a.cs, c, t, b), at<R>(a.ys, c, t, b), p, d, tid);
    if (i == 0) {  // built around predict(m0+, P0+), not symmetrised (filtering.py:200-201)
        load_mat<R>(M, ldd, at<R>(a.Ps, c, 0, b), d, d, tid);
        load_vec<R>(m0p, at<R>(a.ms, c, 0, b), d, tid);
        gemm<R, false, false>(d, d, d, F, ldd, M, ldd, Tm, ldt, (R)1, (R)0, tid);
        gemm<R, false, true>(d, d, d, Tm, ldt, F, ldd, P_, ldd, (R)1, (R)1, tid);
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
        return;
    }
    innovation<R>(o, P_, at<R>(a.Rs, c, t, b), p, d, Tm, S, tid);
    for (int k = tid; k < p; k += NT) {
        R hm = o.c_[k], hb = o.c_[k];
        for (int j = 0; j < d; ++j) hm += o.H_[k * ldd + j] * m_[j], hb += o.H_[k * ldd + j] * bd[j];
        rm[k] = o.nan[k] ? (R)0 : o.y[k] - hm;
        rb[k] = o.nan[k] ? (R)0 : o.y[k] - hb;
    }
    __syncthreads();
    const bool ok = chol<R>(S, ldp, p, o.nan, invd, flag, tid);
    trsm_l<R>(S, ldp, p, invd, Rhs<R>{o.H_, ldd, d, rm, rb}, tid);  // H_ <- W = L^-1 H_
    // M = W^T W, vm = W^T rm, vb = W^T rb
    gemm<R, true, false>(d, d, p, o.H_, ldd, o.H_, ldd, M, ldd, (R)1, (R)0, tid);
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
    gemm<R, false, false>(d, d, d, P_, ldd, M, ldd, Tm, ldt, (R)1, (R)0, tid);   // PM
    gemm<R, false, false>(d, d, d, M, ldd, F, ldd, MF, ldd, (R)1, (R)0, tid);
    gemm<R, false, false>(d, d, d, Tm, ldt, F, ldd, O, ldd, (R)1, (R)0, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) eA[r * d + q] = F[r * ldd + q] - O[r * ldd + q];
    __syncthreads();
    gemm<R, false, false>(d, d, d, Tm, ldt, P_, ldd, O, ldd, (R)1, (R)0, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) {
            const R v1 = P_[r * ldd + q] - O[r * ldd + q], v2 = P_[q * ldd + r] - O[q * ldd + r];
            eC[r * d + q] = r == q ? v1 : (R)0.5 * (v1 + v2);
        }
    __syncthreads();
    gemm<R, true, false>(d, d, d, F, ldd, MF, ldd, O, ldd, (R)1, (R)0, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) eJ[r * d + q] = r == q ? O[r * ldd + r] : (R)0.5 * (O[r * ldd + q] + O[q * ldd + r]);
    gemv<R, false>(d, d, P_, ldd, vm, tv, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) eb[k] = m_[k] + tv[k];
    __syncthreads();
    gemv<R, true>(d, d, F, ldd, vb, tv, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) eeta[k] = tv[k];
}

// ---- the associative operator of the parallel filter (_filtering_op_impl, filtering.py:163-183; one LU as in kalman_math.h) -
template <typename R> struct Agg {  // running prefix in LDS
    R *A, *C, *J, *b, *eta;
};
template <typename R> struct CombTmp {
    R *W, *T1, *T2, *Eb;           // d x d
    R *v, *w, *e2, *fcol, *ipiv;   // d
    int* piv;
};
static size_t lds_combine(size_t s, int d) { return 7 * al16(d * (size_t)ldp_(d) * s) + 7 * al16(d * s) + 64; }

template <typename R> __device__ void agg_load(const Agg<R>& g, const R* __restrict__ e, int d, int tid) {
    const int ldd = ldp_(d);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) {
            g.A[r * ldd + q] = e[r * d + q];
            g.C[r * ldd + q] = e[d * d + d + r * d + q];
            g.J[r * ldd + q] = e[2 * d * d + 2 * d + r * d + q];
        }
    for (int k = tid; k < d; k += NT) g.b[k] = e[d * d + k], g.eta[k] = e[2 * d * d + d + k];
    __syncthreads();
}
template <typename R> __device__ void agg_store(R* __restrict__ e, const Agg<R>& g, int d, int tid) {
    const int ldd = ldp_(d);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) {
            e[r * d + q] = g.A[r * ldd + q];
            e[d * d + d + r * d + q] = g.C[r * ldd + q];
            e[2 * d * d + 2 * d + r * d + q] = g.J[r * ldd + q];
        }
    for (int k = tid; k < d; k += NT) e[d * d + k] = g.b[k], e[2 * d * d + d + k] = g.eta[k];
}
// g <- g (+) e2   (g = earlier prefix a1, e2 = later element a2 in global memory)
//   W = I + C1 J2;  [X | Y | z] = W^-1 [A1 | C1 | b1 + C1 eta2]
//   A = A2 X;  b = A2 z + b2;  C = sym(A2 Y A2^T + C2);  eta = X^T (eta2 - J2 b1) + eta1;  J = sym(X^T (J2 A1) + J1)
// full = false: only (b, C) are updated (the down-sweep; they depend on a1 only through (b1, C1)).
template <typename R> __device__ void combine(const Agg<R>& g, const CombTmp<R>& t, const R* __restrict__ e2, int d, bool full, int tid) {
    const int ldd = ldp_(d);
    const R* A2 = e2;
    const R* b2 = e2 + d * d;
    const R* C2 = b2 + d;
    const R* eta2 = C2 + d * d;
    const R* J2 = eta2 + d;
    load_mat<R>(t.Eb, ldd, J2, d, d, tid);
    load_vec<R>(t.e2, eta2, d, tid);
    gemm<R, false, false>(d, d, d, g.C, ldd, t.Eb, ldd, t.W, ldd, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) t.W[k * ldd + k] += (R)1;
    gemv<R, false>(d, d, g.C, ldd, t.e2, t.v, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) t.v[k] += g.b[k];
    if (full) {
        gemm<R, false, false>(d, d, d, t.Eb, ldd, g.A, ldd, t.T1, ldd, (R)1, (R)0, tid);  // J2 A1
        gemv<R, false>(d, d, t.Eb, ldd, g.b, t.w, (R)1, (R)0, tid);
        for (int k = tid; k < d; k += NT) t.w[k] = t.e2[k] - t.w[k];
    }
    __syncthreads();
    if (full) lu_solve<R>(t.W, ldd, d, Rhs<R>{g.A, ldd, d, nullptr, nullptr}, Rhs<R>{g.C, ldd, d, t.v, nullptr}, t.fcol, t.ipiv, t.piv, tid);
    else lu_solve<R>(t.W, ldd, d, Rhs<R>{g.C, ldd, d, t.v, nullptr}, Rhs<R>{nullptr, 0, 0, nullptr, nullptr}, t.fcol, t.ipiv, t.piv, tid);
    if (full) {
        gemm<R, true, false>(d, d, d, g.A, ldd, t.T1, ldd, g.J, ldd, (R)1, (R)1, tid);  // J1 + X^T (J2 A1)
        symmetrise<R>(g.J, ldd, d, tid);
        gemv<R, true>(d, d, g.A, ldd, t.w, g.eta, (R)1, (R)1, tid);
    }
    load_mat<R>(t.Eb, ldd, A2, d, d, tid);
    gemm<R, false, false>(d, d, d, t.Eb, ldd, g.C, ldd, t.T2, ldd, (R)1, (R)0, tid);  // A2 Y
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) g.C[r * ldd + q] = C2[r * d + q];
    __syncthreads();
    gemm<R, false, true>(d, d, d, t.T2, ldd, t.Eb, ldd, g.C, ldd, (R)1, (R)1, tid);
    symmetrise<R>(g.C, ldd, d, tid);
    gemv<R, false>(d, d, t.Eb, ldd, t.v, g.b, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) g.b[k] += b2[k];
    if (full) {
        gemm<R, false, false>(d, d, d, t.Eb, ldd, g.A, ldd, t.T1, ldd, (R)1, (R)0, tid);  // A2 X
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) g.A[r * ldd + q] = t.T1[r * ldd + q];
    }
    __syncthreads();
}
template <typename R> __device__ void carve_combine(Bump& L, Agg<R>& g, CombTmp<R>& t, int d) {
    const int ldd = ldp_(d);
    g.A = L.take<R>(d * ldd);
    g.C = L.take<R>(d * ldd);
    g.J = L.take<R>(d * ldd);
    t.W = L.take<R>(d * ldd);
    t.T1 = L.take<R>(d * ldd);
    t.T2 = L.take<R>(d * ldd);
    t.Eb = L.take<R>(d * ldd);
    g.b = L.take<R>(d);
    g.eta = L.take<R>(d);
    t.v = L.take<R>(d);
    t.w = L.take<R>(d);
    t.e2 = L.take<R>(d);
    t.fcol = L.take<R>(d);
    t.ipiv = L.take<R>(d);
    t.piv = L.take<int>(1);
}

// chunk aggregate: elements [ch E, min(n, (ch+1) E))
template <typename R> __global__ void __launch_bounds__(NT) wk_scan_reduce(const R* __restrict__ elem, R* __restrict__ aggs, int n, int E, int nchunk, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x / nchunk, ch = blockIdx.x - s * nchunk;
    Bump L{smem};
    Agg<R> g;
    CombTmp<R> t;
    carve_combine<R>(L, g, t, d);
    const long long ne = fe_size(d);
    const int i0 = ch * E, i1 = min(n, i0 + E);
    agg_load<R>(g, elem + ((long long)s * n + i0) * ne, d, tid);
    for (int i = i0 + 1; i < i1; ++i) combine<R>(g, t, elem + ((long long)s * n + i) * ne, d, true, tid);
    agg_store<R>(aggs + ((long long)s * nchunk + ch) * ne, g, d, tid);
}
// exclusive scan of the chunk aggregates of one sequence; pre[ch] = (b, C) of the prefix before chunk ch (ch >= 1)
template <typename R> __global__ void __launch_bounds__(NT) wk_scan_aggs(const R* __restrict__ aggs, R* __restrict__ pre, int nchunk, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x;
    Bump L{smem};
    Agg<R> g;
    CombTmp<R> t;
    carve_combine<R>(L, g, t, d);
    const long long ne = fe_size(d), np = (long long)d * d + d;
    const int ldd = ldp_(d);
    agg_load<R>(g, aggs + (long long)s * nchunk * ne, d, tid);
    for (int ch = 1; ch < nchunk; ++ch) {
        R* q = pre + ((long long)s * nchunk + ch) * np;
        for (int k = tid; k < d; k += NT) q[k] = g.b[k];
        store_mat<R>(q + d, g.C, ldd, d, d, tid);
        if (ch + 1 < nchunk) combine<R>(g, t, aggs + ((long long)s * nchunk + ch) * ne, d, true, tid);
    }
}
// down-sweep: filtered moments ms[i + 1], Ps[i + 1] = (b, C) of the inclusive prefix i
template <typename R> __global__ void __launch_bounds__(NT) wk_scan_down(FilterArgs a, const R* __restrict__ elem, const R* __restrict__ pre, int E, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, n = a.d.T - 1;
    const int s = blockIdx.x / nchunk, ch = blockIdx.x - s * nchunk, c = s / a.d.B, b = s % a.d.B;
    Bump L{smem};
    Agg<R> g;
    CombTmp<R> t;
    carve_combine<R>(L, g, t, d);
    const long long ne = fe_size(d), np = (long long)d * d + d;
    const int ldd = ldp_(d);
    const int i0 = ch * E, i1 = min(n, i0 + E);
    int i = i0;
    if (ch == 0) {
        agg_load<R>(g, elem + (long long)s * n * ne, d, tid);  // prefix 0 = element 0 itself
    } else {
        const R* q = pre + ((long long)s * nchunk + ch) * np;
        load_vec<R>(g.b, q, d, tid);
        load_mat<R>(g.C, ldd, q + d, d, d, tid);
    }
    for (; i < i1; ++i) {
        if (!(ch == 0 && i == 0)) combine<R>(g, t, elem + ((long long)s * n + i) * ne, d, false, tid);
        R* mo = const_cast<R*>(at<R>(a.ms, c, (long long)i + 1, b));
        R* Po = const_cast<R*>(at<R>(a.Ps, c, (long long)i + 1, b));
        for (int k = tid; k < d; k += NT) mo[k] = g.b[k];
        store_mat<R>(Po, g.C, ldd, d, d, tid);
        __syncthreads();
    }
}

// ---- log-likelihood increments (filtering.py:60-62): predict from the filtered moments at i, ell_inc of step i + 1 ---------
static size_t lds_filter_ell(size_t s, int d, int p) {
    const size_t ldd = ldp_(d), ldp = ldp_(p);
    return 4 * al16(d * ldd * s) + al16(p * ldd * s) + al16(d * ldp * s) + al16(p * ldp * s) + 3 * al16(d * s) + 5 * al16(p * s) + al16(p) + 64;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_filter_ell(FilterArgs a, R* __restrict__ ellinc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, p = a.dy, n = a.d.T - 1;
    const int s = blockIdx.x / n, i = blockIdx.x - s * n, c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    const int ldd = ldp_(d), ldp = ldp_(p);
    Bump L{smem};
    R* F = L.take<R>(d * ldd);
    R* P = L.take<R>(d * ldd);
    R* Tm = L.take<R>(d * ldd);
    R* P_ = L.take<R>(d * ldd);
    Obs<R> o;
    o.H_ = L.take<R>(p * ldd);
    R* PHt = L.take<R>(d * ldp);
    R* S = L.take<R>(p * ldp);
    R* m = L.take<R>(d);
    R* m_ = L.take<R>(d);
    R* bd = L.take<R>(d);
    o.c_ = L.take<R>(p);
    o.y = L.take<R>(p);
    R* yd = L.take<R>(p);
    R* invd = L.take<R>(p);
    (void)L.take<R>(p);
    o.nan = L.take<unsigned char>(p);
    o.cnt = L.take<int>(1);
    int* flag = L.take<int>(1);
    const bool any = load_obs<R>(o, at<R>(a.Hs, c, t, b), at<R>(a.cs, c, t, b), at<R>(a.ys, c, t, b), p, d, tid);
    if (!any) {
        if (tid == 0) ellinc[(long long)s * n + i] = 0;
        return;
    }
    load_mat<R>(F, ldd, at<R>(a.Fs, c, i, b), d, d, tid);
    load_mat<R>(P, ldd, at<R>(a.Ps, c, i, b), d, d, tid);
    load_mat<R>(P_, ldd, at<R>(a.Qs, c, i, b), d, d, tid);
    load_vec<R>(m, at<R>(a.ms, c, i, b), d, tid);
    load_vec<R>(bd, at<R>(a.bs, c, i, b), d, tid);
    // sequential_predict :134-139
    gemv<R, false>(d, d, F, ldd, m, m_, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) m_[k] += bd[k];
    gemm<R, false, false>(d, d, d, F, ldd, P, ldd, Tm, ldd, (R)1, (R)0, tid);
    gemm<R, false, true>(d, d, d, Tm, ldd, F, ldd, P_, ldd, (R)1, (R)1, tid);
    symmetrise<R>(P_, ldd, d, tid);
    innovation<R>(o, P_, at<R>(a.Rs, c, t, b), p, d, PHt, S, tid);
    for (int k = tid; k < p; k += NT) {
        R yh = o.c_[k];
        for (int j = 0; j < d; ++j) yh += o.H_[k * ldd + j] * m_[j];
        yd[k] = o.nan[k] ? (R)0 : o.y[k] - yh;
    }
    __syncthreads();
    const bool ok = chol<R>(S, ldp, p, o.nan, invd, flag, tid);
    trsm_l<R>(S, ldp, p, invd, Rhs<R>{nullptr, 0, 0, yd, nullptr}, tid);
    if (tid == 0) ellinc[(long long)s * n + i] = ell_from<R>(S, ldp, yd, o.nan, p, *o.cnt, ok);
}

// out[r] = sum_{b < B} ( add0[r B + b] + sum_{i < n} part[(r B + b) n + i] ), fixed order; one workgroup per output
template <typename R> __global__ void __launch_bounds__(NT) wk_reduce(const R* __restrict__ part, const R* __restrict__ add0, int B, long long n, R* __restrict__ out) {
    __shared__ R sh[NT];
    const int tid = threadIdx.x, r = blockIdx.x;
    R acc = 0;
    for (int b = 0; b < B; ++b) {
        const R* q = part + ((long long)r * B + b) * n;
        for (long long i = tid; i < n; i += NT) acc += q[i];
    }
    sh[tid] = acc;
    __syncthreads();
    for (int off = NT / 2; off > 0; off >>= 1) {
        if (tid < off) sh[tid] += sh[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        R v = sh[0];
        if (add0)
            for (int b = 0; b < B; ++b) v += add0[(long long)r * B + b];
        out[r] = v;
    }
}

// ---- pathwise sampler (sampling.py:60-124): scan position j <-> time T - 1 - j; element [G d*d | e d] ------------------------
static size_t lds_sample_init(size_t s, int d) { return 7 * al16(d * (size_t)ldp_(d) * s) + 8 * al16(d * s) + 64; }
template <typename R> __device__ R nan_to_num_(R x) { return nan_to_num<R>(x); }

// Lc <- lower Cholesky factor of the symmetric matrix in Lc (full storage), nan_to_num'ed; a failed factorisation is all zero
template <typename R> __device__ void chol_n2n(R* Lc, int ld, int n, R* invd, int* flag, int tid) {
    const bool ok = chol<R>(Lc, ld, n, nullptr, invd, flag, tid);
    for (int r = tid / 64; r < n; r += NWV)
        for (int q = tid & 63; q < n; q += 64) Lc[r * ld + q] = (q <= r && ok) ? nan_to_num<R>(Lc[r * ld + q]) : (R)0;
    __syncthreads();
}
template <typename R> __global__ void __launch_bounds__(NT) wk_sample_init(SampleArgs a, R* __restrict__ elem) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, T = a.d.T;
    const int s = blockIdx.x / T, j = blockIdx.x - s * T, c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)T - 1 - j;
    const int ldd = ldp_(d);
    Bump L{smem};
    R* F = L.take<R>(d * ldd);
    R* P = L.take<R>(d * ldd);
    R* T1 = L.take<R>(d * ldd);
    R* S = L.take<R>(d * ldd);
    R* S0 = L.take<R>(d * ldd);
    R* X = L.take<R>(d * ldd);
    R* G = L.take<R>(d * ldd);
    R* m = L.take<R>(d);
    R* eps = L.take<R>(d);
    R* bd = L.take<R>(d);
    R* pm = L.take<R>(d);
    R* tv = L.take<R>(d);
    R* invd = L.take<R>(d);
    (void)L.take<R>(2 * d);
    int* flag = L.take<int>(1);
    R* e = elem + ((long long)s * T + j) * ((long long)d * d + d);
    load_mat<R>(P, ldd, at<R>(a.Ps, c, t, b), d, d, tid);
    load_vec<R>(m, at<R>(a.ms, c, t, b), d, tid);
    load_vec<R>(eps, at<R>(a.eps, c, t, b), d, tid);
    if (j == 0) {  // _sample_last_step :115-124
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) X[r * ldd + q] = r == q ? P[r * ldd + r] : (R)0.5 * (P[r * ldd + q] + P[q * ldd + r]);
        __syncthreads();
        chol_n2n<R>(X, ldd, d, invd, flag, tid);
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) e[r * d + q] = 0;
        for (int k = tid; k < d; k += NT) {
            R v = m[k];
            for (int q = 0; q <= k; ++q) v += X[k * ldd + q] * eps[q];
            e[d * d + k] = v;
        }
        return;
    }
    load_mat<R>(F, ldd, at<R>(a.Fs, c, t, b), d, d, tid);
    load_mat<R>(S, ldd, at<R>(a.Qs, c, t, b), d, d, tid);
    load_vec<R>(bd, at<R>(a.bs, c, t, b), d, tid);
    // S = sym(F P F^T + Q);  gain = P (S^-1 F)^T  (mean_and_chol :84-97)
    gemm<R, false, false>(d, d, d, F, ldd, P, ldd, T1, ldd, (R)1, (R)0, tid);
    gemm<R, false, true>(d, d, d, T1, ldd, F, ldd, S, ldd, (R)1, (R)1, tid);
    symmetrise<R>(S, ldd, d, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) S0[r * ldd + q] = S[r * ldd + q], X[r * ldd + q] = F[r * ldd + q];
    __syncthreads();
    const bool ok = chol<R>(S, ldd, d, nullptr, invd, flag, tid);
    trsm_l<R>(S, ldd, d, invd, Rhs<R>{X, ldd, d, nullptr, nullptr}, tid);
    trsm_lt<R>(S, ldd, d, invd, Rhs<R>{X, ldd, d, nullptr, nullptr}, tid);
    gemm<R, false, true>(d, d, d, P, ldd, X, ldd, G, ldd, (R)1, (R)0, tid);
    if (!ok) {
        for (int r = tid / 64; r < d; r += NWV)
            for (int q = tid & 63; q < d; q += 64) G[r * ldd + q] = r_nan<R>();
        __syncthreads();
    }
    // Sig = sym(P - G S G^T);  Lc = nan_to_num(chol(Sig))  (:98-104)
    gemm<R, false, false>(d, d, d, G, ldd, S0, ldd, T1, ldd, (R)1, (R)0, tid);
    for (int r = tid / 64; r < d; r += NWV)
        for (int q = tid & 63; q < d; q += 64) X[r * ldd + q] = P[r * ldd + q];
    __syncthreads();
    gemm<R, false, true>(d, d, d, T1, ldd, G, ldd, X, ldd, (R)-1, (R)1, tid);
    symmetrise<R>(X, ldd, d, tid);
    chol_n2n<R>(X, ldd, d, invd, flag, tid);
    // inc = m - G (F m + b) + Lc eps  (:108-112)
    gemv<R, false>(d, d, F, ldd, m, pm, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) pm[k] += bd[k];
    __syncthreads();
    gemv<R, false>(d, d, G, ldd, pm, tv, (R)1, (R)0, tid);
    store_mat<R>(e, G, ldd, d, d, tid);
    for (int k = tid; k < d; k += NT) {
        R v = m[k] - tv[k];
        for (int q = 0; q <= k; ++q) v += X[k * ldd + q] * eps[q];
        e[d * d + k] = v;
    }
}

// _sampling_op_impl (sampling.py:51-55): acc = later times already composed, cur = this step: G = Gc Ga, e = Gc ea + ec
static size_t lds_sample_scan(size_t s, int d) { return 3 * al16(d * (size_t)ldp_(d) * s) + 3 * al16(d * s) + 64; }
template <typename R> struct SAgg {
    R *G, *Gc, *Go, *e, *tv, *ec;
};
template <typename R> __device__ void carve_sample(Bump& L, SAgg<R>& g, int d) {
    const int ldd = ldp_(d);
    g.G = L.take<R>(d * ldd);
    g.Gc = L.take<R>(d * ldd);
    g.Go = L.take<R>(d * ldd);
    g.e = L.take<R>(d);
    g.tv = L.take<R>(d);
    g.ec = L.take<R>(d);
}
template <typename R> __device__ void sample_combine_w(SAgg<R>& g, const R* __restrict__ cur, int d, bool full, int tid) {
    const int ldd = ldp_(d);
    load_mat<R>(g.Gc, ldd, cur, d, d, tid);
    load_vec<R>(g.ec, cur + d * d, d, tid);
    gemv<R, false>(d, d, g.Gc, ldd, g.e, g.tv, (R)1, (R)0, tid);
    for (int k = tid; k < d; k += NT) g.e[k] = g.tv[k] + g.ec[k];
    if (full) {
        gemm<R, false, false>(d, d, d, g.Gc, ldd, g.G, ldd, g.Go, ldd, (R)1, (R)0, tid);
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
template <typename R> __global__ void __launch_bounds__(NT) wk_sscan_aggs(const R* __restrict__ aggs, R* __restrict__ pre, int nchunk, int d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s = blockIdx.x;
    Bump L{smem};
    SAgg<R> g;
    carve_sample<R>(L, g, d);
    const long long ne = (long long)d * d + d;
    const int ldd = ldp_(d);
    load_mat<R>(g.G, ldd, aggs + (long long)s * nchunk * ne, d, d, tid);
    load_vec<R>(g.e, aggs + (long long)s * nchunk * ne + d * d, d, tid);
    for (int ch = 1; ch < nchunk; ++ch) {
        R* q = pre + ((long long)s * nchunk + ch) * d;
        for (int k = tid; k < d; k += NT) q[k] = g.e[k];
        __syncthreads();
        if (ch + 1 < nchunk) sample_combine_w<R>(g, aggs + ((long long)s * nchunk + ch) * ne, d, true, tid);
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
__device__ void gauss2(const R* __restrict__ cov, int n, const unsigned char* skip, R* r1, R* r2, R* Lb, R* invd, int* flag, int tid, R& o1, R& o2) {
    const int ld = ldp_(n);
    for (int i = tid / 64; i < n; i += NWV)
        for (int j = tid & 63; j <= i; j += 64) Lb[i * ld + j] = cov[(long long)j * n + i];
    __syncthreads();
    int dim = 0;
    bool bad1 = false, bad2 = false;
    for (int k = 0; k < n; ++k) {  // every lane computes these (cheap, uniform)
        const bool sk = skip && skip[k];
        dim += sk ? 0 : 1;
        bad1 = bad1 || (!sk && !finite_(r1[k]));
        if (r2) bad2 = bad2 || (!sk && !finite_(r2[k]));
    }
    __syncthreads();
    if (skip)
        for (int k = tid; k < n; k += NT)
            if (skip[k]) {
                r1[k] = 0;
                if (r2) r2[k] = 0;
            }
    __syncthreads();
    const bool ok = chol<R>(Lb, ld, n, skip, invd, flag, tid);
    trsm_l<R>(Lb, ld, n, invd, Rhs<R>{nullptr, 0, 0, r1, r2}, tid);
    R q1 = 0, q2 = 0, logdet = 0;
    for (int k = 0; k < n; ++k) {
        q1 += r1[k] * r1[k];
        if (r2) q2 += r2[k] * r2[k];
        logdet += (skip && skip[k]) ? (R)0 : log_(Lb[k * ld + k]);
    }
    const R cst = -logdet - (R)(0.5 * LOG_2PI) * (R)dim;
    o1 = ok ? (R)-0.5 * q1 + cst : r_nan<R>();
    o2 = ok ? (R)-0.5 * q2 + cst : r_nan<R>();
    if (bad1 || isnan_(o1)) o1 = 0;
    if (bad2 || isnan_(o2)) o2 = 0;
    __syncthreads();
}

// joint log-density (base.py:99-166): item (s, t): observation term at t + transition into t (t >= 1) or initial term (t = 0)
static size_t lds_logpdf(size_t s, int d, int p) {
    const int n = std::max(d, p);
    return al16(n * (size_t)ldp_(n) * s) + 4 * al16(d * s) + 4 * al16(n * s) + al16(n) + 64;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_logpdf(LogpdfArgs a, R* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, p = a.dy, T = a.d.T, nmax = d > p ? d : p;
    const int s = blockIdx.x / T, t = blockIdx.x - s * T, c = s / a.d.B, b = s % a.d.B;
    Bump L{smem};
    R* Lb = L.take<R>(nmax * ldp_(nmax));
    R* x = L.take<R>(d);
    R* xq = L.take<R>(d);
    R* rd_ = L.take<R>(d);
    (void)L.take<R>(d);
    R* ro = L.take<R>(nmax);
    R* invd = L.take<R>(nmax);
    (void)L.take<R>(2 * nmax);
    unsigned char* skip = L.take<unsigned char>(nmax);
    int* flag = L.take<int>(1);
    load_vec<R>(x, at<R>(a.xs, c, t, b), d, tid);
    const R* Hg = at<R>(a.Hs, c, t, b);
    const R* cg = at<R>(a.cs, c, t, b);
    const R* yg = at<R>(a.ys, c, t, b);
    for (int k = tid; k < p; k += NT) {
        R pr = cg[k];
        for (int j = 0; j < d; ++j) pr += Hg[(long long)k * d + j] * x[j];
        ro[k] = yg[k] - pr;
        skip[k] = (a.nan_policy == 1) && !finite_(yg[k]);
    }
    __syncthreads();
    R o_obs, o_dyn, dummy;
    gauss2<R>(at<R>(a.Rs, c, t, b), p, a.nan_policy == 1 ? skip : nullptr, ro, nullptr, Lb, invd, flag, tid, o_obs, dummy);
    if (t == 0) {
        const R* m0 = at<R>(a.m0, c, 0, b);
        for (int k = tid; k < d; k += NT) rd_[k] = x[k] - m0[k];
        __syncthreads();
        gauss2<R>(at<R>(a.P0, c, 0, b), d, nullptr, rd_, nullptr, Lb, invd, flag, tid, o_dyn, dummy);
    } else {
        load_vec<R>(xq, at<R>(a.xs, c, t - 1, b), d, tid);
        const R* Fg = at<R>(a.Fs, c, t - 1, b);
        const R* bg = at<R>(a.bs, c, t - 1, b);
        for (int k = tid; k < d; k += NT) {
            R pr = 0;
            for (int j = 0; j < d; ++j) pr += Fg[(long long)k * d + j] * xq[j];
            rd_[k] = x[k] - (pr + bg[k]);
        }
        __syncthreads();
        gauss2<R>(at<R>(a.Qs, c, t - 1, b), d, nullptr, rd_, nullptr, Lb, invd, flag, tid, o_dyn, dummy);
    }
    if (tid == 0) part[(long long)s * T + t] = o_obs + o_dyn;
}

// the five sums of one sweep of the LG_CONCAT device model (body_sweep_logpdf of kalman_bodies.h); part [5][C][T]
static size_t lds_sweep_logpdf(size_t s, int d, int po) {
    const int n = std::max(d, po);
    return al16(n * (size_t)ldp_(n) * s) + 7 * al16(d * s) + 4 * al16(n * s) + al16(n) + 64;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_sweep_logpdf(SweepLogpdfArgs a, R* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, po = a.po, T = a.d.T, C = a.d.C, nmax = d > po ? d : po;
    const int c = blockIdx.x / T, t = blockIdx.x - c * T;
    Bump L{smem};
    R* Lb = L.take<R>(nmax * ldp_(nmax));
    R* x = L.take<R>(d);
    R* xp = L.take<R>(d);
    R* u = L.take<R>(d);
    R* xq = L.take<R>(d);
    R* xpq = L.take<R>(d);
    R* d1 = L.take<R>(d);
    R* d2 = L.take<R>(d);
    R* r1 = L.take<R>(nmax);
    R* r2 = L.take<R>(nmax);
    R* invd = L.take<R>(nmax);
    (void)L.take<R>(nmax);
    unsigned char* skip = L.take<unsigned char>(nmax);
    int* flag = L.take<int>(1);
    load_vec<R>(x, at<R>(a.x, c, t, 0), d, tid);
    load_vec<R>(xp, at<R>(a.xp, c, t, 0), d, tid);
    load_vec<R>(u, at<R>(a.u, c, t, 0), d, tid);
    const R* Hg = at<R>(a.Hs, c, t, 0);
    const R* cg = at<R>(a.cs, c, t, 0);
    const R* yg = at<R>(a.ys, c, t, 0);
    for (int k = tid; k < po; k += NT) {
        R p1 = cg[k], p2 = cg[k];
        for (int j = 0; j < d; ++j) p1 += Hg[(long long)k * d + j] * xp[j], p2 += Hg[(long long)k * d + j] * x[j];
        r1[k] = yg[k] - p1;
        r2[k] = yg[k] - p2;
        skip[k] = (a.nan_policy == 1) && !finite_(yg[k]);
    }
    __syncthreads();
    bool badobs_p = false, badobs_x = false;
    for (int k = 0; k < po; ++k) {
        badobs_p = badobs_p || (!skip[k] && !finite_(r1[k]));
        badobs_x = badobs_x || (!skip[k] && !finite_(r2[k]));
    }
    R ob_p, ob_x, pr_p, pr_x;
    gauss2<R>(at<R>(a.Rs, c, t, 0), po, a.nan_policy == 1 ? skip : nullptr, r1, r2, Lb, invd, flag, tid, ob_p, ob_x);
    // auxiliary block N(u; x, delta/2 I) and the MH correction (generic.py:103-105)
    const R hd = (R)(0.5 * a.delta), sd = sqrt_(hd);
    R q1 = 0, q2 = 0, corr = 0;
    bool b1 = false, b2 = false;
    for (int k = 0; k < d; ++k) {
        const R e1 = u[k] - xp[k], e2 = u[k] - x[k];
        b1 = b1 || !finite_(e1);
        b2 = b2 || !finite_(e2);
        const R z1 = e1 / sd, z2 = e2 / sd;
        q1 += z1 * z1;
        q2 += z2 * z2;
        const R f1 = xp[k] - u[k], f2 = x[k] - u[k];
        corr += (f1 * f1 - f2 * f2) / (R)a.delta;
    }
    const R cst = -(R)d * log_(sd) - (R)(0.5 * LOG_2PI) * (R)d;
    const R ax_p = b1 ? (R)0 : (R)-0.5 * q1 + cst, ax_x = b2 ? (R)0 : (R)-0.5 * q2 + cst;
    const bool ref = a.nan_policy == 0;
    const R cc_p = (ref && (b1 || badobs_p)) ? (R)0 : ax_p + ob_p;
    const R cc_x = (ref && (b2 || badobs_x)) ? (R)0 : ax_x + ob_x;
    if (t == 0) {
        const R* m0 = at<R>(a.m0, c, 0, 0);
        for (int k = tid; k < d; k += NT) d1[k] = xp[k] - m0[k], d2[k] = x[k] - m0[k];
        __syncthreads();
        gauss2<R>(at<R>(a.P0, c, 0, 0), d, nullptr, d1, d2, Lb, invd, flag, tid, pr_p, pr_x);
    } else {
        load_vec<R>(xq, at<R>(a.x, c, t - 1, 0), d, tid);
        load_vec<R>(xpq, at<R>(a.xp, c, t - 1, 0), d, tid);
        const R* Fg = at<R>(a.Fs, c, t - 1, 0);
        const R* bg = at<R>(a.bs, c, t - 1, 0);
        for (int k = tid; k < d; k += NT) {
            R m1 = 0, m2 = 0;
            for (int j = 0; j < d; ++j) m1 += Fg[(long long)k * d + j] * xpq[j], m2 += Fg[(long long)k * d + j] * xq[j];
            d1[k] = xp[k] - (m1 + bg[k]);
            d2[k] = x[k] - (m2 + bg[k]);
        }
        __syncthreads();
        gauss2<R>(at<R>(a.Qs, c, t - 1, 0), d, nullptr, d1, d2, Lb, invd, flag, tid, pr_p, pr_x);
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

// ---- host side -----------------------------------------------------------------------------------------------------------------
struct WPlan {
    int E, nchunk;
};
static WPlan plan(const auxssm_ctx* h, int S, int n, int parallel) {
    WPlan p{n > 0 ? n : 1, 1};
    if (!parallel || n <= 3) return p;
    long long nchunk = ((long long)2 * h->num_cu + S - 1) / S;
    const long long cap = (long long)std::sqrt(2.0 * n);
    nchunk = std::max(1ll, std::min(nchunk, cap));
    if (const char* ev = getenv("AUXSSM_WIDE_NCHUNK")) {  // tuning/debug override
        const long long v = atoll(ev);
        if (v >= 1 && v <= n) nchunk = v;
    }
    p.E = (int)((n + nchunk - 1) / nchunk);
    p.nchunk = (n + p.E - 1) / p.E;
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

template <typename R> static size_t filter_ws_d(const auxssm_ctx* h, const KDims& kd, int parallel, int d) {
    const int S = kd.S(), n = kd.n();
    const WPlan p = plan(h, S, n, parallel);
    const size_t ne = (size_t)fe_size(d);
    return ((size_t)S * std::max(n, 1) * ne + (size_t)S * p.nchunk * (ne + (size_t)d * d + d) + (size_t)S * (std::max(n, 1) + 1)) * sizeof(R) + 4096;
}

template <typename R> int run_filter(auxssm_ctx* h, const FilterArgs& a, int parallel, void* ell_out) {
    const int S = a.d.S(), n = a.d.n(), d = a.dx, p = a.dy;
    const WPlan pl = plan(h, S, n, parallel);
    const size_t ne = (size_t)fe_size(d), np = (size_t)d * d + d;
    R* elem = (R*)ws_take(h, (size_t)S * std::max(n, 1) * ne * sizeof(R));
    R* aggs = (R*)ws_take(h, (size_t)S * pl.nchunk * ne * sizeof(R));
    R* pre = (R*)ws_take(h, (size_t)S * pl.nchunk * np * sizeof(R));
    R* ell0 = (R*)ws_take(h, (size_t)S * sizeof(R));
    R* ellinc = (R*)ws_take(h, (size_t)S * std::max(n, 1) * sizeof(R));
    if (!elem || !aggs || !pre || !ell0 || !ellinc) return AUXSSM_ERR_NOMEM;
    FilterArgs fa = a;
    fa.ell0 = ell0;
    {
        ProfScope ps(h, AUXSSM_K_FILTER_INIT);
        WK_LAUNCH((wk_filter_t0<R>), S, lds_filter_t0(sizeof(R), d, p), fa);
        if (n > 0) WK_LAUNCH((wk_filter_init<R>), (long long)S * n, lds_filter_init(sizeof(R), d, p), fa, elem);
    }
    if (n > 0) {
        const size_t lc = lds_combine(sizeof(R), d);
        {
            ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
            if (pl.nchunk > 1) {
                WK_LAUNCH((wk_scan_reduce<R>), (long long)S * pl.nchunk, lc, (const R*)elem, aggs, n, pl.E, pl.nchunk, d);
                WK_LAUNCH((wk_scan_aggs<R>), S, lc, (const R*)aggs, pre, pl.nchunk, d);
            }
            WK_LAUNCH((wk_scan_down<R>), (long long)S * pl.nchunk, lc, fa, (const R*)elem, (const R*)pre, pl.E, pl.nchunk);
        }
        ProfScope ps(h, AUXSSM_K_FILTER_ELL);
        WK_LAUNCH((wk_filter_ell<R>), (long long)S * n, lds_filter_ell(sizeof(R), d, p), fa, ellinc);
    }
    hipLaunchKernelGGL((wk_reduce<R>), dim3(a.d.C), dim3(NT), 0, h->stream, (const R*)ellinc, (const R*)ell0, a.d.B, (long long)std::max(n, 0), (R*)ell_out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

template <typename R> int run_sample(auxssm_ctx* h, const SampleArgs& a, int parallel) {
    const int S = a.d.S(), T = a.d.T, d = a.dx;
    const WPlan pl = plan(h, S, T, parallel);
    const size_t ne = (size_t)d * d + d;
    R* elem = (R*)ws_take(h, (size_t)S * T * ne * sizeof(R));
    R* aggs = (R*)ws_take(h, (size_t)S * pl.nchunk * ne * sizeof(R));
    R* pre = (R*)ws_take(h, (size_t)S * pl.nchunk * d * sizeof(R));
    if (!elem || !aggs || !pre) return AUXSSM_ERR_NOMEM;
    {
        ProfScope ps(h, AUXSSM_K_SAMPLE_INIT);
        WK_LAUNCH((wk_sample_init<R>), (long long)S * T, lds_sample_init(sizeof(R), d), a, elem);
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
template <typename R> int run_sweep_logpdf(auxssm_ctx* h, const SweepLogpdfArgs& a, void* out) {
    const int C = a.d.C, T = a.d.T;
    R* part = (R*)ws_take(h, (size_t)5 * C * T * sizeof(R));
    if (!part) return AUXSSM_ERR_NOMEM;
    ProfScope ps(h, AUXSSM_K_LOGPDF);
    WK_LAUNCH((wk_sweep_logpdf<R>), (long long)C * T, lds_sweep_logpdf(sizeof(R), a.dx, a.po), a, part);
    hipLaunchKernelGGL((wk_reduce<R>), dim3(5 * C), dim3(NT), 0, h->stream, (const R*)part, (const R*)nullptr, 1, (long long)T, (R*)out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

}  // namespace wide

// ---- workspace sizes: the entry-table signatures carry no (dx, dy), so api.hip asks through these ---------------------------
size_t wide_filter_ws(const auxssm_ctx* h, int dtype, const KDims& kd, int parallel, int d) {
    return dtype == AUXSSM_F32 ? wide::filter_ws_d<float>(h, kd, parallel, d) : wide::filter_ws_d<double>(h, kd, parallel, d);
}
size_t wide_sample_ws(const auxssm_ctx* h, int dtype, const KDims& kd, int parallel, int d) {
    const size_t s = dtype == AUXSSM_F32 ? 4 : 8;
    const wide::WPlan p = wide::plan(h, kd.S(), kd.T, parallel);
    const size_t ne = (size_t)d * d + d;
    return ((size_t)kd.S() * kd.T * ne + (size_t)kd.S() * p.nchunk * (ne + d)) * s + 4096;
}
size_t wide_logpdf_ws(int dtype, const KDims& kd) { return (size_t)5 * kd.S() * kd.T * (dtype == AUXSSM_F32 ? 4 : 8) + 4096; }

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
    static const SweepLogpdfEntry f32{&wide::run_sweep_logpdf<float>, &ws_unused_l};
    static const SweepLogpdfEntry f64{&wide::run_sweep_logpdf<double>, &ws_unused_l};
    return dtype == AUXSSM_F32 ? &f32 : &f64;
}
bool wide_fits(int dtype, int dx, int dy, std::string* why) {
    const size_t s = dtype == AUXSSM_F32 ? 4 : 8;
    size_t need = std::max(wide::lds_combine(s, dx), wide::lds_sample_init(s, dx));
    if (dy > 0) need = std::max({need, wide::lds_filter_init(s, dx, dy), wide::lds_filter_t0(s, dx, dy), wide::lds_filter_ell(s, dx, dy), wide::lds_logpdf(s, dx, dy)});
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

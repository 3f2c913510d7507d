// smallmat.h -- register-resident small dense linear algebra for one lane (= one time step / scan element).
// Everything is a fully unrolled template on the compile-time sizes so that arrays live in VGPRs.
// AX_HD functions also compile for the host so tests can single-step a kernel body without a GPU
// (tests/hostsim only; the product path is the HIP build).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AX_HD __host__ __device__ __forceinline__
#else
#define AX_HD inline
#endif

namespace ax {

template <typename R> AX_HD R r_inf() { return (R)INFINITY; }
template <typename R> AX_HD R r_nan() { return (R)NAN; }
AX_HD bool finite_(float x) { return __builtin_isfinite(x); }
AX_HD bool finite_(double x) { return __builtin_isfinite(x); }
AX_HD bool isnan_(float x) { return __builtin_isnan(x); }
AX_HD bool isnan_(double x) { return __builtin_isnan(x); }
AX_HD float sqrt_(float x) { return sqrtf(x); }
AX_HD double sqrt_(double x) { return sqrt(x); }
AX_HD float log_(float x) { return logf(x); }
AX_HD double log_(double x) { return log(x); }
AX_HD float exp_(float x) { return expf(x); }
AX_HD double exp_(double x) { return exp(x); }
AX_HD float abs_(float x) { return fabsf(x); }
AX_HD double abs_(double x) { return fabs(x); }
template <typename R> AX_HD R max_(R a, R b) { return a > b ? a : b; }
template <typename R> AX_HD R min_(R a, R b) { return a < b ? a : b; }

constexpr double LOG_2PI = 1.8378770664093454835606594728112;

// The per-chain log-density totals of a sweep (sums over T terms of the Metropolis-Hastings ratio) are accumulated, reduced and compared in
// fp64 whatever the working precision: at T = 65536 an fp32 total is ~3e5 with a 0.03 ulp, and log alpha is a difference of six of them.
typedef double Acc;

// packed symmetric storage: upper triangle, row-major.  (i <= j)
AX_HD constexpr int symsize(int D) { return D * (D + 1) / 2; }
AX_HD constexpr int sidx_u(int D, int i, int j) { return i * D - (i * (i - 1)) / 2 + (j - i); }
AX_HD constexpr int sidx(int D, int i, int j) { return i <= j ? sidx_u(D, i, j) : sidx_u(D, j, i); }
// packed lower-triangular storage (Cholesky factors): row-major, j <= i
AX_HD constexpr int lidx(int i, int j) { return i * (i + 1) / 2 + j; }

// ---- plain loads/stores of a record of N reals --------------------------------------------------
template <typename R, int N> AX_HD void ld(const R* __restrict__ p, R* out) {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = p[i];
}
template <typename R, int N> AX_HD void st(R* __restrict__ p, const R* v) {
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = v[i];
}

// ---- 16-byte vector loads/stores of an internal record (pointer 16-byte aligned, N reals, any tail scalar) -------------------
#if defined(__HIPCC__)
template <typename R> struct Vec16;
template <> struct Vec16<float> { typedef float __attribute__((ext_vector_type(4))) type; static constexpr int W = 4; };
template <> struct Vec16<double> { typedef double __attribute__((ext_vector_type(2))) type; static constexpr int W = 2; };
template <typename R, int N> AX_HD void ldv(const R* __restrict__ p, R* out) {
    using V = typename Vec16<R>::type;
    constexpr int W = Vec16<R>::W;
    const V* q = reinterpret_cast<const V*>(__builtin_assume_aligned(p, 16));
#pragma unroll
    for (int i = 0; i < N / W; ++i) {
        const V v = q[i];
#pragma unroll
        for (int k = 0; k < W; ++k) out[i * W + k] = v[k];
    }
#pragma unroll
    for (int i = N / W * W; i < N; ++i) out[i] = p[i];
}
template <typename R, int N> AX_HD void stv(R* __restrict__ p, const R* v) {
    using V = typename Vec16<R>::type;
    constexpr int W = Vec16<R>::W;
    V* q = reinterpret_cast<V*>(__builtin_assume_aligned(p, 16));
#pragma unroll
    for (int i = 0; i < N / W; ++i) {
        V t;
#pragma unroll
        for (int k = 0; k < W; ++k) t[k] = v[i * W + k];
        q[i] = t;
    }
#pragma unroll
    for (int i = N / W * W; i < N; ++i) p[i] = v[i];
}
#else  // host build (tests/hostsim): plain scalar copies
template <typename R, int N> AX_HD void ldv(const R* p, R* out) { ld<R, N>(p, out); }
template <typename R, int N> AX_HD void stv(R* p, const R* v) { st<R, N>(p, v); }
#endif

// ---- dense products (row-major) -----------------------------------------------------------------
// C[M][N] = A[M][K] * B[K][N]
template <typename R, int M, int K, int N> AX_HD void mm(const R* A, const R* B, R* C) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            R s = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * N + j];
            C[i * N + j] = s;
        }
}
// C[M][N] = A[M][K] * B[N][K]^T
template <typename R, int M, int K, int N> AX_HD void mmt(const R* A, const R* B, R* C) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            R s = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) s += A[i * K + k] * B[j * K + k];
            C[i * N + j] = s;
        }
}
// C[M][N] = A[K][M]^T * B[K][N]
template <typename R, int M, int K, int N> AX_HD void tmm(const R* A, const R* B, R* C) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            R s = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) s += A[k * M + i] * B[k * N + j];
            C[i * N + j] = s;
        }
}
// y[M] = A[M][K] x[K]
template <typename R, int M, int K> AX_HD void mv(const R* A, const R* x, R* y) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
        R s = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += A[i * K + k] * x[k];
        y[i] = s;
    }
}
// y[M] = A[K][M]^T x[K]
template <typename R, int M, int K> AX_HD void tmv(const R* A, const R* x, R* y) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
        R s = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += A[k * M + i] * x[k];
        y[i] = s;
    }
}
// symmetric-packed (D) times dense: C[D][N] = S * B[D][N]
template <typename R, int D, int N> AX_HD void symm(const R* S, const R* B, R* C) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            R s = 0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += S[sidx(D, i, k)] * B[k * N + j];
            C[i * N + j] = s;
        }
}
template <typename R, int D> AX_HD void symv(const R* S, const R* x, R* y) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R s = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) s += S[sidx(D, i, k)] * x[k];
        y[i] = s;
    }
}
// pack 0.5*(M + M^T) of a dense D x D into symmetric-packed
template <typename R, int D> AX_HD void sympack(const R* M, R* S) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) S[sidx_u(D, i, j)] = (i == j) ? M[i * D + i] : (R)0.5 * (M[i * D + j] + M[j * D + i]);
}
template <typename R, int D> AX_HD void symunpack(const R* S, R* M) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) M[i * D + j] = S[sidx(D, i, j)];
}

// ---- Cholesky, packed lower, LAPACK/JAX semantics ------------------------------------------------
// In: packed-symmetric A (upper storage, size symsize(N)).  Out: packed-lower L and the reciprocal diagonal invd
// (one division per column; the triangular solves below multiply by it -- fp64 division is ~10x an fma on gfx950).
// `skip[k]` marks an index that is treated as deleted (L_kk = 1, off-diagonals 0).  Returns false on failure
// (pivot <= 0 or NaN); the caller decides what a failed factor means (the reference gets an all-NaN factor from JAX).
template <typename R, int N> AX_HD bool chol_packed(const R* A, R* L, R* invd, const bool* skip) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        R s = A[sidx_u(N, j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[lidx(j, k)] * L[lidx(j, k)];
        const bool sk = skip ? skip[j] : false;
        ok = ok && (sk || (s > (R)0));
        const R ljj = sk ? (R)1 : sqrt_(s);
        L[lidx(j, j)] = ljj;
        const R inv = (R)1 / ljj;
        invd[j] = inv;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            R t = A[sidx_u(N, j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= L[lidx(i, k)] * L[lidx(j, k)];
            const bool ski = skip ? (skip[i] || sk) : false;
            L[lidx(i, j)] = ski ? (R)0 : t * inv;
        }
    }
    return ok;
}
// In-place variant: A holds the symmetric matrix in LOWER-packed storage (lidx) on entry and L on exit.
template <typename R, int N> AX_HD bool chol_inplace(R* A, R* invd, const bool* skip) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        R s = A[lidx(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= A[lidx(j, k)] * A[lidx(j, k)];
        const bool sk = skip ? skip[j] : false;
        ok = ok && (sk || (s > (R)0));
        const R ljj = sk ? (R)1 : sqrt_(s);
        A[lidx(j, j)] = ljj;
        const R inv = (R)1 / ljj;
        invd[j] = inv;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            R t = A[lidx(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= A[lidx(i, k)] * A[lidx(j, k)];
            const bool ski = skip ? (skip[i] || sk) : false;
            A[lidx(i, j)] = ski ? (R)0 : t * inv;
        }
    }
    return ok;
}
// solve L z = b in place
template <typename R, int N> AX_HD void lsolve(const R* L, const R* invd, R* b) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        R s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[lidx(i, k)] * b[k];
        b[i] = s * invd[i];
    }
}
// solve L^T z = b in place
template <typename R, int N> AX_HD void ltsolve(const R* L, const R* invd, R* b) {
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        R s = b[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= L[lidx(k, i)] * b[k];
        b[i] = s * invd[i];
    }
}
template <typename R, int N> AX_HD void cho_solve(const R* L, const R* invd, R* b) {
    lsolve<R, N>(L, invd, b);
    ltsolve<R, N>(L, invd, b);
}
// column `col` of a row-major [N][NC] matrix, in place
template <typename R, int N, int NC> AX_HD void lsolve_col(const R* L, const R* invd, R* Bm, int col) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        R s = Bm[i * NC + col];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[lidx(i, k)] * Bm[k * NC + col];
        Bm[i * NC + col] = s * invd[i];
    }
}
template <typename R, int N, int NC> AX_HD void cho_solve_col(const R* L, const R* invd, R* Bm, int col) {
    lsolve_col<R, N, NC>(L, invd, Bm, col);
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        R s = Bm[i * NC + col];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= L[lidx(k, i)] * Bm[k * NC + col];
        Bm[i * NC + col] = s * invd[i];
    }
}

// ---- LU with partial pivoting, W [D][D] destroyed, RHS [D][NR] overwritten by W^{-1} RHS -----------------
template <typename R, int D, int NR, bool RDET = false> AX_HD R lu_solve_logdet(R* W, R* B, R* pr = nullptr);
template <typename R, int D, int NR> AX_HD void lu_solve(R* W, R* B) { (void)lu_solve_logdet<R, D, NR>(W, B); }
// returns log |det W|.  RDET: no logarithm -- pr[(D + 1) / 2] receives the products of PAIRS of reciprocal pivots (1 / |det W| = their product), returns 0: the
// chunk-serial filter passes multiply them into a running (mantissa, exponent) pair and take ONE logarithm per chunk (kalman_math.h::LogProd)
template <typename R, int D, int NR, bool RDET> AX_HD R lu_solve_logdet(R* W, R* B, R* pr) {
    R ipiv[D];
    R ld_ = 0, pp = 1;  // log|det| from products of PAIRS of reciprocal pivots: half the logarithms, no under/overflow in practice
#pragma unroll
    for (int k = 0; k < D; ++k) {
        // bring the largest |W[r][k]|, r >= k, to row k by compare-and-swap (branch-free selects)
#pragma unroll
        for (int r = k + 1; r < D; ++r) {
            const bool sw = abs_(W[r * D + k]) > abs_(W[k * D + k]);
#pragma unroll
            for (int j = k; j < D; ++j) {
                const R a = W[k * D + j], b = W[r * D + j];
                W[k * D + j] = sw ? b : a;
                W[r * D + j] = sw ? a : b;
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const R a = B[k * NR + j], b = B[r * NR + j];
                B[k * NR + j] = sw ? b : a;
                B[r * NR + j] = sw ? a : b;
            }
        }
        const R inv = (R)1 / W[k * D + k];
        ipiv[k] = inv;
        pp *= abs_(inv);
        if ((k & 1) == 1 || k == D - 1) {
            if constexpr (RDET) pr[k >> 1] = pp;
            else ld_ -= log_(pp);
            pp = 1;
        }
#pragma unroll
        for (int r = k + 1; r < D; ++r) {
            const R f = W[r * D + k] * inv;
#pragma unroll
            for (int j = k + 1; j < D; ++j) W[r * D + j] -= f * W[k * D + j];
#pragma unroll
            for (int j = 0; j < NR; ++j) B[r * NR + j] -= f * B[k * NR + j];
        }
    }
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        const R inv = ipiv[k];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            R s = B[k * NR + j];
#pragma unroll
            for (int c = k + 1; c < D; ++c) s -= W[k * D + c] * B[c * NR + j];
            B[k * NR + j] = s * inv;
        }
    }
    return ld_;
}

}  // namespace ax

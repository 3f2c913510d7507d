// loop.hip -- the MCMC loop around the sweeps: running statistics, acceptance averages, step-size adaptation and the
// Lorenz-63 theta step, as device kernels on the handle's stream so that a run of sweeps never returns to the host.
// Reference: the body of `loop` in examples/stochastic_volatility/experiment.py:88-128 / examples/lorenz/experiment.py:120-169,
// aux_samplers/common.py:4-32 (delta_adaptation), examples/lorenz/model.py:59-79 (theta_posterior_mean_and_chol).
// Built with -ffp-contract=off: the running means are then the NumPy expressions bit for bit.
#include "ctx.h"
#include "rng.h"

namespace ax {

template <typename R> __device__ inline R fold(R i, R u, R v) { return (i * u + v) / (i + (R)1); }

template <typename R>
__global__ void k_stats_update(long long n, long long iter, const R* __restrict__ x_prev, const R* __restrict__ x_next, R* __restrict__ sq_jump,
                               R* __restrict__ mean, R* __restrict__ sq_mean) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const R i = (R)iter, xn = x_next[g], dj = xn - x_prev[g];
    sq_jump[g] = fold<R>(i, sq_jump[g], dj * dj);
    mean[g] = fold<R>(i, mean[g], xn);
    sq_mean[g] = fold<R>(i, sq_mean[g], xn * xn);
}

template <typename R>
__global__ void k_accept_update(long long n, long long iter, R beta, R one_minus_beta, const int32_t* __restrict__ flags, R* __restrict__ avg,
                                R* __restrict__ window) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const R f = flags[g] != 0 ? (R)1 : (R)0;
    avg[g] = fold<R>((R)iter, avg[g], f);
    window[g] = beta * f + one_minus_beta * window[g];
}

// one lane per time step j: chain-pooled windowed acceptance, then common.py:29-32
template <typename R>
__global__ void k_delta_adapt(int C, int m, const R* __restrict__ window, R target, R rate, R lo, R hi, R* __restrict__ delta,
                              R* __restrict__ shd) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    R s = 0;
    for (int c = 0; c < C; ++c) s += window[(long long)c * m + j];
    const R acc = s / (R)C;
    R d = delta[j] * exp(rate * (acc - target));
    d = d < lo ? lo : (d > hi ? hi : d);
    delta[j] = d;
    if (shd) shd[j] = sqrt((R)0.5 * d);
}

// one workgroup per chain: X = dt phi(x_t), Y = x_{t+1} - x_t - dt phi_0(x_t); sums in double whatever R is
template <typename R>
__global__ void __launch_bounds__(256) k_lorenz_theta(int C, int T, int cfast, const R* __restrict__ x, double sigma_theta, double sigma_x,
                                                      const R* __restrict__ eps, R* __restrict__ par, R* __restrict__ mean_chol) {
    __shared__ double sh[6][256];
    const int c = blockIdx.x, tid = threadIdx.x;
    // element (t, k) of chain c: dense (C, T, 3) or chain-minor (T, 3, C)
    const R* xc = cfast ? x + c : x + (long long)c * T * 3;
    const long long ks = cfast ? C : 1, ts = 3 * ks;
    R* pc = par + (long long)c * 4;
    const R dt = pc[3];
    double a[6] = {0, 0, 0, 0, 0, 0};
    for (long long t = tid; t + 1 < T; t += 256) {
        const R x1 = xc[t * ts], x2 = xc[t * ts + ks], x3 = xc[t * ts + 2 * ks];
        const R X[3] = {dt * (x2 - x1), dt * x1, dt * (-x3)};
        const R p0[3] = {(R)0, -x2 - x1 * x3, x1 * x2};
        for (int k = 0; k < 3; ++k) {
            const R Y = (xc[(t + 1) * ts + k * ks] - xc[t * ts + k * ks]) - dt * p0[k];
            a[k] += (double)(X[k] * X[k]);
            a[3 + k] += (double)(X[k] * Y);
        }
    }
    for (int k = 0; k < 6; ++k) sh[k][tid] = a[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off)
            for (int k = 0; k < 6; ++k) sh[k][tid] += sh[k][tid + off];
        __syncthreads();
    }
    if (tid < 3) {
        const double Sigma = 1.0 / (sh[tid][0] + 1.0 / (sigma_theta * sigma_theta));
        const double mean = Sigma * sh[3 + tid][0];
        const double chol = sigma_x * sqrt((double)dt) * sqrt(Sigma);
        pc[tid] = (R)(mean + chol * (double)eps[c * 3 + tid]);
        if (mean_chol) {
            mean_chol[c * 6 + tid] = (R)mean;
            mean_chol[c * 6 + 3 + tid] = (R)chol;
        }
    }
}

#define AX_NEED_H(h)                         \
    do {                                     \
        if (!(h)) {                          \
            set_error("handle is NULL");     \
            return AUXSSM_ERR_ARG;           \
        }                                    \
        AX_HIP(hipSetDevice((h)->device));   \
        ++(h)->api_calls;                    \
    } while (0)

static int need_dtype(int dtype) {
    if (dtype == AUXSSM_F32 || dtype == AUXSSM_F64) return AUXSSM_OK;
    set_error("dtype must be AUXSSM_F32 or AUXSSM_F64");
    return AUXSSM_ERR_ARG;
}

template <typename R>
static int stats_update(auxssm_ctx* h, int64_t n, int64_t iter, const void* xp, const void* xn, void* sj, void* mn, void* sq) {
    hipLaunchKernelGGL((k_stats_update<R>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (long long)n, (long long)iter,
                       (const R*)xp, (const R*)xn, (R*)sj, (R*)mn, (R*)sq);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}
template <typename R> static int accept_update(auxssm_ctx* h, int64_t n, int64_t iter, double beta, const int32_t* flags, void* avg, void* win) {
    hipLaunchKernelGGL((k_accept_update<R>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (long long)n, (long long)iter, (R)beta,
                       (R)(1.0 - beta), flags, (R*)avg, (R*)win);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}
template <typename R>
static int delta_adapt(auxssm_ctx* h, int C, int m, const void* win, double target, double rate, double lo, double hi, void* delta, void* shd) {
    hipLaunchKernelGGL((k_delta_adapt<R>), dim3((unsigned)((m + 127) / 128)), dim3(128), 0, h->stream, C, m, (const R*)win, (R)target, (R)rate,
                       (R)lo, (R)hi, (R*)delta, (R*)shd);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}
template <typename R>
static int lorenz_theta(auxssm_ctx* h, int C, int T, int cfast, const void* x, double sth, double sx, const void* eps, void* par, void* mc) {
    hipLaunchKernelGGL((k_lorenz_theta<R>), dim3(C), dim3(256), 0, h->stream, C, T, cfast, (const R*)x, sth, sx, (const R*)eps, (R*)par, (R*)mc);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// mvn.logpdf (math/mvn/base.py:15-58) + tril_log_det (:108-128) for n independent (x, m, chol) triplets of runtime dimension dim <= 64, one
// lane per triplet, IEEE-literal: non-finite entries of chol become +inf before the forward substitution (nan_to_num(chol, nan = inf, ...),
// :52), the dimension counts the finite diagonal entries (:50) and non-finite diagonal entries drop out of the log-determinant (:123-128).
constexpr int MVN_MAX_DIM = 64;
template <typename R>
__global__ void k_mvn_logpdf(long long n, int dim, const R* __restrict__ x, long long sx, const R* __restrict__ m, long long sm,
                             const R* __restrict__ chol, long long sl, R* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const R* xg = x + g * sx;
    const R* mg = m + g * sm;
    const R* Lg = chol + g * sl;
    const R inf = (R)INFINITY;
    R y[MVN_MAX_DIM];
    R nrm = 0, logdet = 0;
    int nfin = 0;
    for (int i = 0; i < dim; ++i) {
        R s = xg[i] - mg[i];
        for (int k = 0; k < i; ++k) {
            R l = Lg[(long long)i * dim + k];
            if (!isfinite(l)) l = inf;
            s -= l * y[k];
        }
        const R d = Lg[(long long)i * dim + i];
        const bool fin = isfinite(d);
        y[i] = s / (fin ? d : inf);
        nrm += y[i] * y[i];
        if (fin) {
            ++nfin;
            const R l = log(fabs(d));
            if (!isnan(l)) logdet += l;  // nansum (:128)
        }
    }
    out[g] = (R)-0.5 * nrm - (logdet + (R)0.5 * (R)nfin * (R)1.8378770664093453);
}

// ---- mvn.get_optimal_covariance (_primitives/math/mvn/base.py:78-105): the dominating covariance of Section 3 of the paper ---------------------------------
// Y = chol_P^-1 chol_Sig (forward substitution, :98), (w, V) = eigh(Y^T Y) (:99), w <- min(w, 1), L = chol_Sig V diag(w^-1/2) (:100-103), out = chol(L L^T) (:104).
// One workgroup, dim <= 64, four dim x dim images in LDS.  The symmetric eigen-decomposition is a cyclic Jacobi iteration (rotations (p, q) in row order, every lane a
// row / column index; the result -- chol of L L^T -- depends on neither the order nor the signs of the eigenvectors); convergence: off-diagonal mass below
// eps^2 x the diagonal's, at most 40 sweeps.  vector != 0: the scalar / diagonal branch (:94-95), out = max(chol_P, chol_Sig) elementwise over `dim` entries.
template <typename R> __global__ void __launch_bounds__(256) k_opt_cov(int n, int vector, const R* __restrict__ LPg, const R* __restrict__ LSg, R* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_oc[];
    const int tid = threadIdx.x, NTH = 256;
    if (vector) {
        for (int k = tid; k < n; k += NTH) out[k] = LPg[k] > LSg[k] ? LPg[k] : LSg[k];
        return;
    }
    R* A = (R*)smem_oc;      // chol_P, then Y^T Y (rotated to diagonal), then L L^T and its factor
    R* LS = A + n * n;       // chol_Sig
    R* Y = LS + n * n;       // Y, then L
    R* V = Y + n * n;        // eigenvectors
    __shared__ R red[256];
    for (int e = tid; e < n * n; e += NTH) {
        const int i = e / n, j = e - i * n;
        A[e] = j <= i ? LPg[e] : (R)0;
        LS[e] = j <= i ? LSg[e] : (R)0;
        V[e] = i == j ? (R)1 : (R)0;
    }
    __syncthreads();
    // Y = chol_P^-1 chol_Sig: lane j solves column j
    for (int j = tid; j < n; j += NTH)
        for (int i = 0; i < n; ++i) {
            R acc = LS[i * n + j];
            for (int k = 0; k < i; ++k) acc -= A[i * n + k] * Y[k * n + j];
            Y[i * n + j] = acc / A[i * n + i];
        }
    __syncthreads();
    for (int e = tid; e < n * n; e += NTH) {  // A = Y^T Y
        const int i = e / n, j = e - i * n;
        R acc = 0;
        for (int k = 0; k < n; ++k) acc += Y[k * n + i] * Y[k * n + j];
        A[e] = acc;
    }
    __syncthreads();
    const R eps = sizeof(R) == 4 ? (R)1.2e-7 : (R)2.3e-16;
    for (int sweep = 0; sweep < 40; ++sweep) {
        R off = 0, dg = 0;
        for (int e = tid; e < n * n; e += NTH) {
            const int i = e / n, j = e - i * n;
            if (i == j) dg += A[e] * A[e];
            else off += A[e] * A[e];
        }
        red[tid] = off;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) red[tid] += red[tid + w];
            __syncthreads();
        }
        off = red[0];
        __syncthreads();
        red[tid] = dg;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) red[tid] += red[tid + w];
            __syncthreads();
        }
        dg = red[0];
        __syncthreads();
        if (!(off > eps * eps * dg)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const R apq = A[p * n + q], app = A[p * n + p], aqq = A[q * n + q];
                __syncthreads();  // (every lane holds the three entries before anybody rotates)
                if (apq != (R)0) {
                    const R theta = (aqq - app) / ((R)2 * apq);
                    const R t = (theta >= (R)0 ? (R)1 : (R)-1) / (fabs(theta) + sqrt(theta * theta + (R)1));
                    const R c = (R)1 / sqrt(t * t + (R)1), sn = t * c;
                    for (int k = tid; k < n; k += NTH) {
                        const R vkp = V[k * n + p], vkq = V[k * n + q];
                        V[k * n + p] = c * vkp - sn * vkq;
                        V[k * n + q] = sn * vkp + c * vkq;
                        if (k != p && k != q) {
                            const R akp = A[k * n + p], akq = A[k * n + q];
                            const R np_ = c * akp - sn * akq, nq_ = sn * akp + c * akq;
                            A[k * n + p] = np_;
                            A[p * n + k] = np_;
                            A[k * n + q] = nq_;
                            A[q * n + k] = nq_;
                        }
                    }
                    if (tid == 0) {
                        A[p * n + p] = app - t * apq;
                        A[q * n + q] = aqq + t * apq;
                        A[p * n + q] = 0;
                        A[q * n + p] = 0;
                    }
                }
                __syncthreads();
            }
    }
    // L = chol_Sig (V diag(min(w, 1)^-1/2))  -> Y
    for (int e = tid; e < n * n; e += NTH) {
        const int i = e / n, j = e - i * n;
        R wj = A[j * n + j];
        wj = wj < (R)1 ? wj : (R)1;
        R acc = 0;
        for (int k = 0; k <= i; ++k) acc += LS[i * n + k] * V[k * n + j];
        Y[e] = acc / sqrt(wj);
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NTH) {  // A = L L^T
        const int i = e / n, j = e - i * n;
        R acc = 0;
        for (int k = 0; k < n; ++k) acc += Y[i * n + k] * Y[j * n + k];
        A[e] = acc;
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {  // Cholesky, column by column (jnp.linalg.cholesky: a failed factorisation is NaN)
        if (tid == 0) A[j * n + j] = sqrt(A[j * n + j]);
        __syncthreads();
        const R d = A[j * n + j];
        for (int i = j + 1 + tid; i < n; i += NTH) A[i * n + j] /= d;
        __syncthreads();
        for (int e = tid; e < (n - j - 1) * (n - j - 1); e += NTH) {
            const int i = j + 1 + e / (n - j - 1), k = j + 1 + e % (n - j - 1);
            if (k <= i) A[i * n + k] -= A[i * n + j] * A[k * n + j];
        }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += NTH) {
        const int i = e / n, j = e - i * n;
        out[e] = j <= i ? A[e] : (R)0;
    }
}

// ---- effective sample size (examples/rare_event/ess.py:28-160: BlackJAX's estimator with the option of dividing by the TRUE variance) ----------------------------
// a: (M chains, N draws, K series) dense.  Means per chain, the biased autocovariances of every lag averaged over the chains (direct sums in double: the reference's
// FFT gives the same numbers to rounding), then Geyer's initial positive / monotone sequence per series (one lane per series: a scan over N / 2 pairs).
template <typename R> __global__ void __launch_bounds__(256) k_ess_mean(long long M, long long N, long long K, const R* __restrict__ a, double* __restrict__ cm) {
    __shared__ double red[256];
    const long long mk = blockIdx.x, m = mk / K, k = mk - m * K;
    double acc = 0;
    for (long long i = threadIdx.x; i < N; i += 256) acc += (double)a[(m * N + i) * K + k];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) cm[mk] = red[0] / (double)N;
}
template <typename R>
__global__ void __launch_bounds__(256) k_ess_acov(long long M, long long N, long long K, long long nlag, const R* __restrict__ a, const double* __restrict__ cm, double* __restrict__ acov) {
    __shared__ double red[256];
    const long long l = blockIdx.x, k = blockIdx.y;
    double acc = 0;
    for (long long m = 0; m < M; ++m) {
        const double mu = cm[m * K + k];
        const R* am = a + m * N * K + k;
        for (long long i = threadIdx.x; i + l < N; i += 256) acc += ((double)am[i * K] - mu) * ((double)am[(i + l) * K] - mu);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) acov[l * K + k] = red[0] / (double)N / (double)M;
}
template <typename R>
__global__ void k_ess_geyer(long long M, long long N, long long K, const double* __restrict__ cm, const double* __restrict__ acov, const R* __restrict__ var, R* __restrict__ out) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const long long n_even = N - N % 2, J = n_even / 2;
    double var0 = acov[k] * (double)N / ((double)N - 1.0);
    double wvar = var0 * ((double)N - 1.0) / (double)N;
    if (M > 1) {  // + the variance of the chain means (ddof = 1)
        double mu = 0, ss = 0;
        for (long long m = 0; m < M; ++m) mu += cm[m * K + k];
        mu /= (double)M;
        for (long long m = 0; m < M; ++m) ss += (cm[m * K + k] - mu) * (cm[m * K + k] - mu);
        wvar += ss / ((double)M - 1.0);
    }
    if (var) wvar = (double)var[k], var0 = wvar;
    auto rho = [&](long long l) { return l == 0 ? 1.0 : 1.0 - (var0 - acov[l * K + k]) / wvar; };
    // pass 1: the length of the initial positive run of P_j = rho_2j + rho_2j+1
    long long L = J;
    for (long long j = 0; j < J; ++j)
        if (!(rho(2 * j) + rho(2 * j + 1) > 0.0)) {
            L = j;
            break;
        }
    const long long last = L > 0 ? L - 1 : 0;
    // pass 2: truncated terms, the initial monotone sequence, the sum
    double run = 0, sum = 0, extra = 0;
    for (long long j = 0; j < J; ++j) {
        double e = rho(2 * j), o = rho(2 * j + 1);
        if (j >= L) o = 0.0;
        bool keep = j < L;
        if (j == last + 1) keep = e > 0.0;  // "improve estimation": one more even term if it is positive (ess.py:140-146)
        if (!keep) e = 0.0;
        const double s = e + o;
        const double prev = j == 0 ? s : run;   // running minimum of the terms before j (the term itself at j = 0)
        const bool upd = s > prev;
        run = j == 0 ? s : (s < run ? s : run);
        const double ef = upd ? run / 2.0 : e, of = upd ? run / 2.0 : o;
        sum += ef + of;
        if (j == (last + 1 < J - 1 ? last + 1 : J - 1)) extra = ef;  // (ess.py:156: the gather clamps an out-of-range index to the last even term)
    }
    double tau = -1.0 + 2.0 * sum - extra;
    const double floor_ = 1.0 / log10((double)M * (double)N);
    tau = tau > floor_ ? tau : floor_;
    out[k] = (R)((double)M * (double)N / tau);
}


// ---- statistical / Taylor linearisation on the device (_primitives/linearisation.py: extended :11-44, gauss_hermite :47-75, cubature :78-104,
// _generic_sigma_points :107-127) for the closed family of conditional means the device knows: AFFINE mean(x) = A x + a (the reference's own
// test, test_linearisation.py) and LORENZ63 mean(x) = x + dt (phi_0(x) + theta * phi(x)) (examples/lorenz/model.py:10-25); cov(x) = Qc.
// One lane per linearisation point, D <= 4.  Sigma-point methods: L = chol(P*), points x* + L xi_j, two passes over the points (the mean of the
// images, then the two covariances around it -- the images are recomputed, not stored), F = (P*^-1 Psi)^T by the two triangular solves of
// cho_solve, Q = Phi - (F L)(F L)^T + sum_j w_j Qc, b = m_f - F x*.  The 1-D rule (nodes, weights) comes from the host: cubature = the 2 D points
// +- sqrt(D) e_i with weights 1 / 2D, Gauss-Hermite = the order^D tensor grid of the probabilists' rule.
struct LinRule { int method, order; double node[8], weight[8]; };
template <typename R, int DX, int DY> struct LinFn {
    int kind;              // 0 affine, 1 lorenz63
    const R *A, *a, *Qc;   // affine: A (DY, DX), a (DY); lorenz63: A = (theta_0..2, dt); Qc (DY, DY)
    __device__ void mean(const R* x, R* mu) const {
        if constexpr (DX == 3 && DY == 3) {
            if (kind == 1) {
                const R dt = A[3];
                mu[0] = x[0] + dt * (A[0] * (x[1] - x[0]));
                mu[1] = x[1] + dt * (A[1] * x[0] - x[1] - x[0] * x[2]);
                mu[2] = x[2] + dt * (x[0] * x[1] - A[2] * x[2]);
                return;
            }
        }
#pragma unroll
        for (int r = 0; r < DY; ++r) {
            R acc = a[r];
#pragma unroll
            for (int k = 0; k < DX; ++k) acc += A[r * DX + k] * x[k];
            mu[r] = acc;
        }
    }
    __device__ void jac(const R* x, R* F) const {
        if constexpr (DX == 3 && DY == 3) {
            if (kind == 1) {
                const R dt = A[3];
                const R J[9] = {-A[0], A[0], 0, A[1] - x[2], (R)-1, -x[0], x[1], x[0], -A[2]};
#pragma unroll
                for (int k = 0; k < 9; ++k) F[k] = ((k / 3 == k % 3) ? (R)1 : (R)0) + dt * J[k];
                return;
            }
        }
#pragma unroll
        for (int k = 0; k < DY * DX; ++k) F[k] = A[k];
    }
};
template <typename R, int D> __device__ inline void lin_point(const LinRule& rule, int j, const R* L, const R* xs, R* dx, R* pt, R* w) {
    R xi[D];
    if (rule.method == 1) {  // cubature: j < D: +sqrt(D) e_j, else -sqrt(D) e_{j - D}
#pragma unroll
        for (int k = 0; k < D; ++k) xi[k] = (j % D == k) ? (j < D ? (R)rule.node[0] : -(R)rule.node[0]) : (R)0;
        *w = (R)rule.weight[0];
    } else {  // Gauss-Hermite: digits of j in base `order`
        R ww = 1;
        int q = j;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int i = q % rule.order;
            q /= rule.order;
            xi[k] = (R)rule.node[i];
            ww *= (R)rule.weight[i];
        }
        *w = ww;
    }
#pragma unroll
    for (int r = 0; r < D; ++r) {
        R acc = 0;
#pragma unroll
        for (int k = 0; k <= r; ++k) acc += L[r * D + k] * xi[k];
        dx[r] = acc;
        pt[r] = xs[r] + acc;
    }
}
template <typename R, int DX, int DY>
__global__ void k_linearise(long long n, LinRule rule, int npts, LinFn<R, DX, DY> fn, const R* __restrict__ xstar, const R* __restrict__ Pstar, long long sP,
                            R* __restrict__ Fo, R* __restrict__ Qo, R* __restrict__ bo) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    R xs[DX], F[DY * DX], Q[DY * DY], b[DY];
#pragma unroll
    for (int k = 0; k < DX; ++k) xs[k] = xstar[g * DX + k];
    if (rule.method == 0) {  // extended: analytic Jacobian in place of jacfwd / jacrev
        R mu[DY];
        fn.mean(xs, mu);
        fn.jac(xs, F);
#pragma unroll
        for (int k = 0; k < DY * DY; ++k) Q[k] = fn.Qc[k];
#pragma unroll
        for (int r = 0; r < DY; ++r) {
            R acc = mu[r];
#pragma unroll
            for (int k = 0; k < DX; ++k) acc -= F[r * DX + k] * xs[k];
            b[r] = acc;
        }
    } else {
        R L[DX * DX];
        const R* P = Pstar + g * sP;
#pragma unroll
        for (int k = 0; k < DX * DX; ++k) L[k] = 0;
#pragma unroll
        for (int j = 0; j < DX; ++j) {  // Cholesky (a failed factorisation is NaN, as jnp.linalg.cholesky)
            R dsum = P[j * DX + j];
#pragma unroll
            for (int k = 0; k < j; ++k) dsum -= L[j * DX + k] * L[j * DX + k];
            const R dj = sqrt(dsum);
            L[j * DX + j] = dj;
#pragma unroll
            for (int i = j + 1; i < DX; ++i) {
                R acc = P[i * DX + j];
#pragma unroll
                for (int k = 0; k < j; ++k) acc -= L[i * DX + k] * L[j * DX + k];
                L[i * DX + j] = acc / dj;
            }
        }
        R mf[DY], wsum = 0;
#pragma unroll
        for (int k = 0; k < DY; ++k) mf[k] = 0;
        for (int j = 0; j < npts; ++j) {
            R dx[DX], pt[DX], f[DY], w;
            lin_point<R, DX>(rule, j, L, xs, dx, pt, &w);
            fn.mean(pt, f);
            wsum += w;
#pragma unroll
            for (int k = 0; k < DY; ++k) mf[k] += w * f[k];
        }
        R Psi[DX * DY], Phi[DY * DY];
#pragma unroll
        for (int k = 0; k < DX * DY; ++k) Psi[k] = 0;
#pragma unroll
        for (int k = 0; k < DY * DY; ++k) Phi[k] = 0;
        for (int j = 0; j < npts; ++j) {
            R dx[DX], pt[DX], f[DY], w;
            lin_point<R, DX>(rule, j, L, xs, dx, pt, &w);
            fn.mean(pt, f);
#pragma unroll
            for (int c = 0; c < DY; ++c) {
                const R dfc = f[c] - mf[c];
#pragma unroll
                for (int r = 0; r < DX; ++r) Psi[r * DY + c] += (dx[r] * w) * dfc;
#pragma unroll
                for (int r = 0; r < DY; ++r) Phi[r * DY + c] += ((f[r] - mf[r]) * w) * dfc;
            }
        }
        // F^T = P^-1 Psi (DX x DY): L z = Psi, L^T Ft = z, column by column
        R Ft[DX * DY];
#pragma unroll
        for (int c = 0; c < DY; ++c) {
            R z[DX];
#pragma unroll
            for (int r = 0; r < DX; ++r) {
                R acc = Psi[r * DY + c];
#pragma unroll
                for (int k = 0; k < r; ++k) acc -= L[r * DX + k] * z[k];
                z[r] = acc / L[r * DX + r];
            }
#pragma unroll
            for (int r = DX - 1; r >= 0; --r) {
                R acc = z[r];
#pragma unroll
                for (int k = r + 1; k < DX; ++k) acc -= L[k * DX + r] * Ft[k * DY + c];
                Ft[r * DY + c] = acc / L[r * DX + r];
            }
        }
#pragma unroll
        for (int r = 0; r < DY; ++r)
#pragma unroll
            for (int c = 0; c < DX; ++c) F[r * DX + c] = Ft[c * DY + r];
        R FL[DY * DX];
#pragma unroll
        for (int r = 0; r < DY; ++r)
#pragma unroll
            for (int c = 0; c < DX; ++c) {
                R acc = 0;
#pragma unroll
                for (int k = c; k < DX; ++k) acc += F[r * DX + k] * L[k * DX + c];
                FL[r * DX + c] = acc;
            }
#pragma unroll
        for (int r = 0; r < DY; ++r)
#pragma unroll
            for (int c = 0; c < DY; ++c) {
                R acc = 0;
#pragma unroll
                for (int k = 0; k < DX; ++k) acc += FL[r * DX + k] * FL[c * DX + k];
                Q[r * DY + c] = Phi[r * DY + c] - acc + wsum * fn.Qc[r * DY + c];
            }
#pragma unroll
        for (int r = 0; r < DY; ++r) {
            R acc = mf[r];
#pragma unroll
            for (int k = 0; k < DX; ++k) acc -= F[r * DX + k] * xs[k];
            b[r] = acc;
        }
    }
#pragma unroll
    for (int k = 0; k < DY * DX; ++k) Fo[g * DY * DX + k] = F[k];
#pragma unroll
    for (int k = 0; k < DY * DY; ++k) Qo[g * DY * DY + k] = Q[k];
#pragma unroll
    for (int k = 0; k < DY; ++k) bo[g * DY + k] = b[k];
}


// ---- jax.random's own bit stream (threefry2x32, the non-partitionable layout: JAX's default up to 0.4.x, what the reference ran on) ----------------------------
// jax/_src/prng.py: threefry_2x32(key, iota(m)) splits the counters in two halves (one 0 appended when m is odd), runs one block per pair and concatenates the two
// output halves -- value i < h = ceil(m / 2) is word 0 of block (i, h + i) [second counter 0 for the appended one], value i >= h is word 1 of block (i - h, i);
// 64-bit values take m = 2 n counters and are (word 0 << 32) | word 1 of block (i, n + i).  jax/_src/random.py: uniform = (mantissa bits | exponent of 1) - 1, scaled to
// [minval, maxval) and clamped below; normal = sqrt(2) erfinv(uniform(nextafter(-1, 0), 1)).  One launch fills n values for EACH of nkeys keys (a vmap over keys),
// written with a key stride and an element stride (dense or chain-minor targets).  oracle/rng_np.py::jax_* restates this on the host, pinned by the values JAX's
// documentation prints (tests/test_rng.py); erfinv: float32 = XLA's ErfInvF32 polynomials (M. Giles), float64 = the same start + two Newton steps on erf (XLA uses its
// own rational form there: agreement to rounding).
template <typename R> struct JaxBits;
template <> struct JaxBits<float> {
    static __device__ float unit(uint32_t w0, uint32_t) { return __uint_as_float((w0 >> 9) | 0x3F800000u) - 1.0f; }
};
__device__ inline float jax_erfinv(float x) {
    float w = -log1pf(-x * x);
    const bool lt = w < 5.0f;
    w = lt ? w - 2.5f : sqrtf(w) - 3.0f;
    float p = lt ? 2.81022636e-08f : -0.000200214257f;
    p = (lt ? 3.43273939e-07f : 0.000100950558f) + p * w;
    p = (lt ? -3.5233877e-06f : 0.00134934322f) + p * w;
    p = (lt ? -4.39150654e-06f : -0.00367342844f) + p * w;
    p = (lt ? 0.00021858087f : 0.00573950773f) + p * w;
    p = (lt ? -0.00125372503f : -0.0076224613f) + p * w;
    p = (lt ? -0.00417768164f : 0.00943887047f) + p * w;
    p = (lt ? 0.246640727f : 1.00167406f) + p * w;
    p = (lt ? 1.50140941f : 2.83297682f) + p * w;
    return p * x;
}
__device__ inline double jax_erfinv(double x) {
    if (!(fabs(x) < 1.0)) return x == 1.0 ? (double)INFINITY : (x == -1.0 ? -(double)INFINITY : (double)NAN);
    // start: the float32 polynomials evaluated in double on w = -log((1 - |x|)(1 + |x|)) (1 - |x| is exact for |x| >= 1/2, so the tails keep their digits: a float
    // argument would round |x| > 1 - 6e-8 to 1 and start from infinity); beyond float's range (w > 16) they extrapolate to a few per cent, which two more steps absorb
    const double ax = fabs(x), c = 1.0 - ax;
    double w = -log(c * (1.0 + ax));
    const bool lt = w < 5.0, far = w > 16.0;
    w = lt ? w - 2.5 : sqrt(w) - 3.0;
    double p = lt ? 2.81022636e-08 : -0.000200214257;
    p = (lt ? 3.43273939e-07 : 0.000100950558) + p * w;
    p = (lt ? -3.5233877e-06 : 0.00134934322) + p * w;
    p = (lt ? -4.39150654e-06 : -0.00367342844) + p * w;
    p = (lt ? 0.00021858087 : 0.00573950773) + p * w;
    p = (lt ? -0.00125372503 : -0.0076224613) + p * w;
    p = (lt ? -0.00417768164 : 0.00943887047) + p * w;
    p = (lt ? 0.246640727 : 1.00167406) + p * w;
    p = (lt ? 1.50140941 : 2.83297682) + p * w;
    double ya = p * ax;
    // Newton with the second-order (Halley) correction, d/dy erf = 2 / sqrt(pi) exp(-y^2); beyond |x| = 1/2 on erfc(|y|) = 1 - |x| (erfc keeps its RELATIVE accuracy in
    // the tail, where erf(y) - x would lose it to cancellation)
    const bool tail = ax > 0.5;
    const int nit = far ? 4 : 2;
    for (int it = 0; it < nit; ++it) {
        const double d = 1.1283791670955126 * exp(-ya * ya);
        const double e = tail ? c - erfc(ya) : erf(ya) - ax;   // = erf(ya) - |x| either way
        const double st = e / d;
        ya -= st / (1.0 + ya * st);
    }
    return x < 0 ? -ya : ya;
}
template <typename R>
__global__ void __launch_bounds__(256) k_rng_jax(int kind, long long nkeys, long long n, const uint32_t* __restrict__ keys, R lo, R hi, R* __restrict__ out,
                                                 long long skey, long long selem) {
    const long long per = sizeof(R) == 4 ? (n + 1) / 2 : n;   // blocks per key
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= nkeys * per) return;
    // consecutive lanes write consecutive addresses: along the values of a key (dense targets) or along the keys (chain-minor targets: key stride 1)
    const bool keys_fast = skey < selem;
    const long long c = keys_fast ? g % nkeys : g / per, i = keys_fast ? g / nkeys : g - (g / per) * per;
    const uint32_t k0 = keys[2 * c], k1 = keys[2 * c + 1];
    auto finish = [&](R f) {   // f in [0, 1)
        R v = f * (hi - lo) + lo;
        v = v > lo ? v : lo;
        if (kind == 1) v = (R)1.4142135623730951 * jax_erfinv(v);
        return v;
    };
    if constexpr (sizeof(R) == 4) {
        const long long second = per + i;                 // the value that takes word 1 of this block
        uint32_t x0 = (uint32_t)i, x1 = second < n ? (uint32_t)second : 0u;
        threefry2x32(k0, k1, x0, x1);
        out[c * skey + i * selem] = finish(__uint_as_float((x0 >> 9) | 0x3F800000u) - 1.0f);
        if (second < n) out[c * skey + second * selem] = finish(__uint_as_float((x1 >> 9) | 0x3F800000u) - 1.0f);
    } else {
        uint32_t x0 = (uint32_t)i, x1 = (uint32_t)(n + i);
        threefry2x32(k0, k1, x0, x1);
        const unsigned long long b = ((unsigned long long)x0 << 32) | x1;
        out[c * skey + i * selem] = finish(__longlong_as_double((long long)((b >> 12) | 0x3FF0000000000000ull)) - 1.0);
    }
}

}  // namespace ax

using namespace ax;

template <typename R, int DX, int DY>
static void launch_linearise(auxssm_ctx* h, int64_t n, const LinRule& rule, int npts, int fn_kind, const void* A, const void* a, const void* Qc, const void* xstar,
                             const void* Pstar, int64_t sP, void* F, void* Q, void* b) {
    LinFn<R, DX, DY> fn{fn_kind, (const R*)A, (const R*)a, (const R*)Qc};
    hipLaunchKernelGGL((k_linearise<R, DX, DY>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, h->stream, (long long)n, rule, npts, fn, (const R*)xstar,
                       (const R*)Pstar, (long long)sP, (R*)F, (R*)Q, (R*)b);
}
template <typename R, int DX>
static void launch_linearise_dy(auxssm_ctx* h, int dy, int64_t n, const LinRule& rule, int npts, int fn_kind, const void* A, const void* a, const void* Qc,
                                const void* xstar, const void* Pstar, int64_t sP, void* F, void* Q, void* b) {
    switch (dy) {
        case 1: launch_linearise<R, DX, 1>(h, n, rule, npts, fn_kind, A, a, Qc, xstar, Pstar, sP, F, Q, b); break;
        case 2: launch_linearise<R, DX, 2>(h, n, rule, npts, fn_kind, A, a, Qc, xstar, Pstar, sP, F, Q, b); break;
        case 3: launch_linearise<R, DX, 3>(h, n, rule, npts, fn_kind, A, a, Qc, xstar, Pstar, sP, F, Q, b); break;
        default: launch_linearise<R, DX, 4>(h, n, rule, npts, fn_kind, A, a, Qc, xstar, Pstar, sP, F, Q, b);
    }
}
template <typename R>
static void launch_linearise_dx(auxssm_ctx* h, int dx, int dy, int64_t n, const LinRule& rule, int npts, int fn_kind, const void* A, const void* a, const void* Qc,
                                const void* xstar, const void* Pstar, int64_t sP, void* F, void* Q, void* b) {
    switch (dx) {
        case 1: launch_linearise_dy<R, 1>(h, dy, n, rule, npts, fn_kind, A, a, Qc, xstar, Pstar, sP, F, Q, b); break;
        case 2: launch_linearise_dy<R, 2>(h, dy, n, rule, npts, fn_kind, A, a, Qc, xstar, Pstar, sP, F, Q, b); break;
        case 3: launch_linearise_dy<R, 3>(h, dy, n, rule, npts, fn_kind, A, a, Qc, xstar, Pstar, sP, F, Q, b); break;
        default: launch_linearise_dy<R, 4>(h, dy, n, rule, npts, fn_kind, A, a, Qc, xstar, Pstar, sP, F, Q, b);
    }
}

extern "C" {

int auxssm_stats_attach(auxssm_handle h, int dtype, int64_t n, const void* x, void* sq_jump, void* mean, void* sq_mean, int64_t iter) {
    AX_NEED_H(h);
    const int nn = (sq_jump != nullptr) + (mean != nullptr) + (sq_mean != nullptr);
    if (nn != 0 && nn != 3) {
        set_error("sq_jump, mean and sq_mean must be all non-NULL (attach) or all NULL (detach)");
        return AUXSSM_ERR_ARG;
    }
    if (nn == 3) {
        if (int rc = need_dtype(dtype)) return rc;
        if (!x || n < 1) {
            set_error("attach needs the resident state x (non-NULL) and its element count n >= 1");
            return AUXSSM_ERR_ARG;
        }
    }
    if (iter < 0) {
        set_error("iter must be >= 0");
        return AUXSSM_ERR_ARG;
    }
    h->st_sq_jump = sq_jump;
    h->st_mean = mean;
    h->st_sq_mean = sq_mean;
    h->st_iter = iter;
    h->st_x = nn == 3 ? x : nullptr;
    h->st_n = nn == 3 ? n : 0;
    h->st_dtype = nn == 3 ? dtype : -1;
    return AUXSSM_OK;
}

int auxssm_stats_update(auxssm_handle h, int dtype, int64_t n, int64_t iter, const void* x_prev, const void* x_next, void* sq_jump,
                        void* mean, void* sq_mean) {
    AX_NEED_H(h);
    if (int rc = need_dtype(dtype)) return rc;
    if (n < 0 || iter < 0) {
        set_error("n and iter must be >= 0");
        return AUXSSM_ERR_ARG;
    }
    if (n == 0) return AUXSSM_OK;
    if (!x_prev || !x_next || !sq_jump || !mean || !sq_mean) {
        set_error("x_prev/x_next/sq_jump/mean/sq_mean must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    return dtype == AUXSSM_F32 ? stats_update<float>(h, n, iter, x_prev, x_next, sq_jump, mean, sq_mean)
                               : stats_update<double>(h, n, iter, x_prev, x_next, sq_jump, mean, sq_mean);
}

int auxssm_accept_update(auxssm_handle h, int dtype, int32_t C, int32_t m, int64_t iter, double beta, const int32_t* flags, void* avg,
                         void* window) {
    AX_NEED_H(h);
    --h->api_calls;  // (reads the sweep's flags, writes its own averages: a chain-shared sweep's model stage may still run ahead of it, ctx.h::SideStage)
    if (int rc = need_dtype(dtype)) return rc;
    if (C < 0 || m < 0 || iter < 0) {
        set_error("C, m and iter must be >= 0");
        return AUXSSM_ERR_ARG;
    }
    const int64_t n = (int64_t)C * m;
    if (n == 0) return AUXSSM_OK;
    if (!flags || !avg || !window) {
        set_error("flags/avg/window must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    return dtype == AUXSSM_F32 ? accept_update<float>(h, n, iter, beta, flags, avg, window)
                               : accept_update<double>(h, n, iter, beta, flags, avg, window);
}

int auxssm_delta_adapt(auxssm_handle h, int dtype, int32_t C, int32_t m, const void* window, double target, double rate, double min_delta,
                       double max_delta, void* delta, void* sqrt_half_delta) {
    AX_NEED_H(h);
    if (int rc = need_dtype(dtype)) return rc;
    if (C < 1 || m < 0) {
        set_error("C must be >= 1 and m >= 0");
        return AUXSSM_ERR_ARG;
    }
    if (m == 0) return AUXSSM_OK;
    if (!window || !delta) {
        set_error("window/delta must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    return dtype == AUXSSM_F32 ? delta_adapt<float>(h, C, m, window, target, rate, min_delta, max_delta, delta, sqrt_half_delta)
                               : delta_adapt<double>(h, C, m, window, target, rate, min_delta, max_delta, delta, sqrt_half_delta);
}

int auxssm_lorenz_theta_update(auxssm_handle h, int dtype, int32_t C, int32_t T, int layout, const void* x, double sigma_theta, double sigma_x,
                               const void* eps, void* par, void* mean_chol) {
    AX_NEED_H(h);
    if (int rc = need_dtype(dtype)) return rc;
    if (C < 0 || T < 1) {
        set_error("C must be >= 0 and T >= 1");
        return AUXSSM_ERR_ARG;
    }
    if (!(sigma_theta > 0) || !(sigma_x > 0)) {
        set_error("sigma_theta and sigma_x must be positive");
        return AUXSSM_ERR_ARG;
    }
    if (layout != AUXSSM_LAYOUT_DENSE && layout != AUXSSM_LAYOUT_CHAIN_MINOR) {
        set_error("layout must be AUXSSM_LAYOUT_DENSE (0) or AUXSSM_LAYOUT_CHAIN_MINOR (1)");
        return AUXSSM_ERR_ARG;
    }
    if (C == 0) return AUXSSM_OK;
    if (!x || !eps || !par) {
        set_error("x/eps/par must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    const int cf = layout == AUXSSM_LAYOUT_CHAIN_MINOR ? 1 : 0;
    return dtype == AUXSSM_F32 ? lorenz_theta<float>(h, C, T, cf, x, sigma_theta, sigma_x, eps, par, mean_chol)
                               : lorenz_theta<double>(h, C, T, cf, x, sigma_theta, sigma_x, eps, par, mean_chol);
}

int auxssm_mvn_logpdf(auxssm_handle h, int dtype, int64_t n, int32_t dim, const void* x, int64_t sx, const void* m, int64_t sm, const void* chol,
                      int64_t sl, void* out) {
    AX_NEED_H(h);
    if (int rc = need_dtype(dtype)) return rc;
    if (n < 0 || dim < 1 || dim > MVN_MAX_DIM) {
        set_error("n must be >= 0 and 1 <= dim <= %d (got n=%lld, dim=%d)", MVN_MAX_DIM, (long long)n, dim);
        return AUXSSM_ERR_ARG;
    }
    if (n == 0) return AUXSSM_OK;
    if (!x || !m || !chol || !out || sx < 0 || sm < 0 || sl < 0) {
        set_error("x/m/chol/out must be non-NULL and the strides >= 0");
        return AUXSSM_ERR_ARG;
    }
    const unsigned grid = (unsigned)((n + 63) / 64);
    if (dtype == AUXSSM_F32)
        hipLaunchKernelGGL((k_mvn_logpdf<float>), dim3(grid), dim3(64), 0, h->stream, (long long)n, dim, (const float*)x, (long long)sx, (const float*)m,
                           (long long)sm, (const float*)chol, (long long)sl, (float*)out);
    else
        hipLaunchKernelGGL((k_mvn_logpdf<double>), dim3(grid), dim3(64), 0, h->stream, (long long)n, dim, (const double*)x, (long long)sx, (const double*)m,
                           (long long)sm, (const double*)chol, (long long)sl, (double*)out);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

int auxssm_mvn_optimal_covariance(auxssm_handle h, int dtype, int32_t dim, int vector, const void* chol_P, const void* chol_Sig, void* out) {
    AX_NEED_H(h);
    if (int rc = need_dtype(dtype)) return rc;
    if (dim < 1 || (!vector && dim > MVN_MAX_DIM)) {
        set_error("1 <= dim <= %d (got %d)", MVN_MAX_DIM, dim);
        return AUXSSM_ERR_ARG;
    }
    if (!chol_P || !chol_Sig || !out) {
        set_error("chol_P/chol_Sig/out must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    const size_t s = dtype == AUXSSM_F32 ? 4 : 8, lds = vector ? 0 : (size_t)4 * dim * dim * s;
    if (dtype == AUXSSM_F32) {
        if (lds > 48 * 1024) AX_HIP(hipFuncSetAttribute((const void*)k_opt_cov<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_opt_cov<float>), dim3(1), dim3(256), lds, h->stream, dim, vector, (const float*)chol_P, (const float*)chol_Sig, (float*)out);
    } else {
        if (lds > 48 * 1024) AX_HIP(hipFuncSetAttribute((const void*)k_opt_cov<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_opt_cov<double>), dim3(1), dim3(256), lds, h->stream, dim, vector, (const double*)chol_P, (const double*)chol_Sig, (double*)out);
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

int auxssm_ess(auxssm_handle h, int dtype, int64_t M, int64_t N, int64_t K, const void* a, const void* var, void* out) {
    AX_NEED_H(h);
    if (int rc = need_dtype(dtype)) return rc;
    if (M < 1 || N < 4 || K < 1 || N > 0x7fffffffLL || K > 65535 || M * K > 0x7fffffffLL) {
        set_error("need M >= 1 chains, 4 <= N draws, 1 <= K <= 65535 series (got M=%lld N=%lld K=%lld)", (long long)M, (long long)N, (long long)K);
        return AUXSSM_ERR_ARG;
    }
    if (!a || !out) {
        set_error("a/out must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    const long long nlag = N - N % 2;
    if (int rc = ws_reserve(h, (size_t)(M * K + nlag * K) * sizeof(double) + 1024)) return rc;
    double* cm = (double*)ws_take(h, (size_t)M * K * sizeof(double));
    double* acov = (double*)ws_take(h, (size_t)nlag * K * sizeof(double));
    if (!cm || !acov) return AUXSSM_ERR_NOMEM;
    if (dtype == AUXSSM_F32) {
        hipLaunchKernelGGL((k_ess_mean<float>), dim3((unsigned)(M * K)), dim3(256), 0, h->stream, (long long)M, (long long)N, (long long)K, (const float*)a, cm);
        hipLaunchKernelGGL((k_ess_acov<float>), dim3((unsigned)nlag, (unsigned)K), dim3(256), 0, h->stream, (long long)M, (long long)N, (long long)K, nlag, (const float*)a, (const double*)cm, acov);
        hipLaunchKernelGGL((k_ess_geyer<float>), dim3((unsigned)((K + 63) / 64)), dim3(64), 0, h->stream, (long long)M, (long long)N, (long long)K, (const double*)cm, (const double*)acov,
                           (const float*)var, (float*)out);
    } else {
        hipLaunchKernelGGL((k_ess_mean<double>), dim3((unsigned)(M * K)), dim3(256), 0, h->stream, (long long)M, (long long)N, (long long)K, (const double*)a, cm);
        hipLaunchKernelGGL((k_ess_acov<double>), dim3((unsigned)nlag, (unsigned)K), dim3(256), 0, h->stream, (long long)M, (long long)N, (long long)K, nlag, (const double*)a, (const double*)cm, acov);
        hipLaunchKernelGGL((k_ess_geyer<double>), dim3((unsigned)((K + 63) / 64)), dim3(64), 0, h->stream, (long long)M, (long long)N, (long long)K, (const double*)cm, (const double*)acov,
                           (const double*)var, (double*)out);
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

int auxssm_linearise(auxssm_handle h, int dtype, int method, int order, int fn_kind, int64_t n, int32_t dim, int32_t dim_out, const void* A, const void* a,
                     const void* Qc, const void* nodes, const void* weights, const void* x_star, const void* P_star, int64_t sP, void* F, void* Q, void* b) {
    AX_NEED_H(h);
    if (int rc = need_dtype(dtype)) return rc;
    if (method < AUXSSM_LIN_EXTENDED || method > AUXSSM_LIN_GAUSS_HERMITE || fn_kind < AUXSSM_FN_AFFINE || fn_kind > AUXSSM_FN_LORENZ63) {
        set_error("unknown linearisation method %d / function kind %d", method, fn_kind);
        return AUXSSM_ERR_ARG;
    }
    if (n < 0 || dim < 1 || dim > 4 || dim_out < 1 || dim_out > 4 || (fn_kind == AUXSSM_FN_LORENZ63 && (dim != 3 || dim_out != 3))) {
        set_error("need n >= 0 and 1 <= dim, dim_out <= 4 (3, 3 for LORENZ63); got n=%lld dim=%d dim_out=%d", (long long)n, dim, dim_out);
        return AUXSSM_ERR_ARG;
    }
    if (!A || !Qc || (fn_kind == AUXSSM_FN_AFFINE && !a) || !x_star || !F || !Q || !b || (method != AUXSSM_LIN_EXTENDED && !P_star)) {
        set_error("NULL argument (A, a [affine], Qc, x_star, P_star [sigma-point methods], F, Q, b)");
        return AUXSSM_ERR_ARG;
    }
    LinRule rule{};
    rule.method = method;
    int npts = 0;
    if (method == AUXSSM_LIN_CUBATURE) {
        rule.order = 1;
        rule.node[0] = sqrt((double)dim);
        rule.weight[0] = 0.5 / dim;
        npts = 2 * dim;
    } else if (method == AUXSSM_LIN_GAUSS_HERMITE) {
        if (order < 1 || order > 8 || !nodes || !weights) {
            set_error("Gauss-Hermite: 1 <= order <= 8 with its nodes / weights for N(0, 1) as host doubles (got order %d)", order);
            return AUXSSM_ERR_ARG;
        }
        rule.order = order;
        npts = 1;
        for (int k = 0; k < dim; ++k) npts *= order;
        for (int k = 0; k < order; ++k) rule.node[k] = ((const double*)nodes)[k], rule.weight[k] = ((const double*)weights)[k];
    }
    if (n == 0) return AUXSSM_OK;
    if (dtype == AUXSSM_F32) launch_linearise_dx<float>(h, dim, dim_out, n, rule, npts, fn_kind, A, a, Qc, x_star, P_star, sP, F, Q, b);
    else launch_linearise_dx<double>(h, dim, dim_out, n, rule, npts, fn_kind, A, a, Qc, x_star, P_star, sP, F, Q, b);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

int auxssm_rng_jax(auxssm_handle h, int dtype, int kind, int64_t nkeys, int64_t n, const uint32_t* keys, double minval, double maxval, void* out, int64_t key_stride,
                   int64_t elem_stride) {
    AX_NEED_H(h);
    if (int rc = need_dtype(dtype)) return rc;
    if (kind < 0 || kind > 1 || nkeys < 0 || n < 0 || (2 * n > 0xffffffffLL) || !keys || !out) {
        set_error("kind 0 (uniform) / 1 (normal), nkeys, n >= 0, 2 n < 2^32, keys / out non-NULL (got kind=%d nkeys=%lld n=%lld)", kind, (long long)nkeys, (long long)n);
        return AUXSSM_ERR_ARG;
    }
    if (nkeys == 0 || n == 0) return AUXSSM_OK;
    const long long per = dtype == AUXSSM_F32 ? (n + 1) / 2 : n, tot = nkeys * per;
    if ((tot + 255) / 256 > 0x7fffffffLL) {
        set_error("too many values for one launch");
        return AUXSSM_ERR_ARG;
    }
    const unsigned grid = (unsigned)((tot + 255) / 256);
    ProfScope ps(h, AUXSSM_K_RNG);
    if (dtype == AUXSSM_F32) {
        float lo = (float)minval, hi = (float)maxval;
        if (kind == 1) lo = nextafterf(-1.0f, 0.0f), hi = 1.0f;
        hipLaunchKernelGGL((k_rng_jax<float>), dim3(grid), dim3(256), 0, h->stream, kind, (long long)nkeys, (long long)n, keys, lo, hi, (float*)out, (long long)key_stride,
                           (long long)elem_stride);
    } else {
        double lo = minval, hi = maxval;
        if (kind == 1) lo = nextafter(-1.0, 0.0), hi = 1.0;
        hipLaunchKernelGGL((k_rng_jax<double>), dim3(grid), dim3(256), 0, h->stream, kind, (long long)nkeys, (long long)n, keys, lo, hi, (double*)out, (long long)key_stride,
                           (long long)elem_stride);
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

}  // extern "C"

// loop.hip -- the MCMC loop around the sweeps: running statistics, acceptance averages, step-size adaptation and the
// Lorenz-63 theta step, as device kernels on the handle's stream so that a run of sweeps never returns to the host.
// Reference: the body of `loop` in examples/stochastic_volatility/experiment.py:88-128 / examples/lorenz/experiment.py:120-169,
// aux_samplers/common.py:4-32 (delta_adaptation), examples/lorenz/model.py:59-79 (theta_posterior_mean_and_chol).
// Built with -ffp-contract=off: the running means are then the NumPy expressions bit for bit.
#include "ctx.h"

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

}  // namespace ax

using namespace ax;

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

}  // extern "C"

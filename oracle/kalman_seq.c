/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (never linked or loaded by the product).
 *
 * Plain-C restatement of the reference's SEQUENTIAL auxiliary-Kalman sweep -- the `parallel=False` algorithms, which is the code
 * path the reference itself runs on CPU (examples/stochastic_volatility/experiment.sh:1-6):
 *   aux_samplers/_primitives/kalman/filtering.py   _sequential_filtering :66-79, sequential_update :83-130,
 *                                                  sequential_predict :134-139
 *   aux_samplers/_primitives/kalman/sampling.py    sampling :11-40 (the lax.scan branch :34-39), mean_and_chol :60-105,
 *                                                  _sample_last_step :115-124
 *   aux_samplers/_primitives/kalman/base.py        posterior_logpdf :72-96, prior_logpdf :99-134, log_likelihood :137-166
 *   aux_samplers/_primitives/math/mvn/base.py      logpdf :15-58
 *   aux_samplers/kalman/generic.py                 kernel :53-76, do_one :78-90, _get_alpha :98-106
 * for the linear-Gaussian model with the auxiliary observations concatenated to the real ones (the factory pattern of
 * examples/lorenz/auxiliary_kalman.py:26-35; aux_ssm_samplers_amd/kalman/models.py::LGConcatModel):
 *   ys = [u ; y], Hs = [I ; Hobs], Rs = blkdiag(delta/2 I, Robs), cs = [0 ; cobs],
 *   log_likelihood_fn(x) = prior_logpdf(x) + nansum_t log N(y_t; Hobs_t x_t + cobs_t, Robs_t).
 * The factories of this model ignore the linearisation point, so the reverse LGSSM (generic.py:67) equals the forward one and its
 * filter pass is run once (as the device sweep does; under jit XLA's CSE would do the same).
 *
 * Two uses: (1) bench.py's `cpu_baseline` leg ("C++ restatement: sequential filter / sampler ... OMP_NUM_THREADS = all
 * cores, chains as the parallel dimension; plus a single-thread figure") -- chains are distributed over OpenMP threads;
 * (2) a second, independent restatement the NumPy oracle is cross-checked against (tests/test_oracle_kalman_seq.py).
 * Missing data: the update runs on the observed sub-vector (what the reference's inf-padding is algebraically, and what its own test
 * oracle does by deleting rows, test_kalman/common.py:65-77); log_likelihood drops a time step with any NaN residual component
 * (jnp.nansum over per-step logpdfs, base.py:166).
 * Pinned by: tests/test_oracle_kalman_seq.py (== oracle/kalman_np.py::kalman_sweep(parallel=False), itself pinned to the reference's
 * known answers, tests/test_golden.py).
 *
 * Build: gcc -O2 -fopenmp -fPIC -shared kalman_seq.c -o _build/libkalman_seq.so -lm   (oracle/Makefile)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 8
#define MAXP 16
#define LOG_2PI 1.8378770664093453

typedef struct {
    int T, d, po;
    const double *m0, *P0;        /* (d), (d,d) */
    const double *Fs, *Qs, *bs;   /* (T-1,d,d), (T-1,d,d), (T-1,d); time strides sF, sQ, sb (0 = time-invariant) */
    const double *Hobs, *Robs, *cobs, *yobs; /* (T,po,d), (T,po,po), (T,po), (T,po); time strides sH, sR, sc (yobs dense) */
    long sF, sQ, sb, sH, sR, sc;
} kseq_model;

/* lower Cholesky of the n x n matrix a (row-major, leading dimension n) in place; returns 0 and fills NaN on failure (LAPACK potrf as
 * jnp.linalg.cholesky wraps it) */
static int chol(int n, double* a) {
    for (int j = 0; j < n; ++j) {
        double s = a[j * n + j];
        for (int k = 0; k < j; ++k) s -= a[j * n + k] * a[j * n + k];
        if (!(s > 0.0)) {
            for (int i = 0; i < n * n; ++i) a[i] = NAN;
            return 0;
        }
        const double l = sqrt(s);
        a[j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double t = a[i * n + j];
            for (int k = 0; k < j; ++k) t -= a[i * n + k] * a[j * n + k];
            a[i * n + j] = t / l;
        }
        for (int i = 0; i < j; ++i) a[i * n + j] = 0.0;
    }
    return 1;
}
/* x <- L^-1 x */
static void lsolve(int n, const double* L, double* x) {
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
}
/* x <- L^-T x */
static void ltsolve(int n, const double* L, double* x) {
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}
/* mvn.logpdf of a residual r with Cholesky factor L (n x n): mvn/base.py:49-58 on finite inputs; NaN in -> NaN out */
static double mvn_logpdf_chol(int n, const double* L, const double* r) {
    double y[MAXP], q = 0, ld = 0;
    memcpy(y, r, n * sizeof(double));
    lsolve(n, L, y);
    for (int i = 0; i < n; ++i) q += y[i] * y[i], ld += log(fabs(L[i * n + i]));
    return -0.5 * q - ld - 0.5 * n * LOG_2PI;
}

/* sequential_update (filtering.py:83-130) on the observed sub-vector; y (p), H (p,d), c (p), R (p,p).  Returns the ell increment. */
static double seq_update(int d, int p, const double* y, const double* H, const double* c, const double* R, double* m, double* P) {
    int idx[MAXP], q = 0;
    for (int k = 0; k < p; ++k)
        if (isfinite(y[k])) idx[q++] = k;
    if (q == 0) return 0.0; /* _passthrough :127-130 */
    double Ho[MAXP * MAXD], r[MAXP], S[MAXP * MAXP], HP[MAXP * MAXD], G[MAXD * MAXP];
    for (int a = 0; a < q; ++a) {
        const int k = idx[a];
        double yh = c[k];
        for (int j = 0; j < d; ++j) Ho[a * d + j] = H[k * d + j], yh += H[k * d + j] * m[j];
        r[a] = y[k] - yh;
    }
    for (int a = 0; a < q; ++a)
        for (int j = 0; j < d; ++j) {
            double s = 0;
            for (int i = 0; i < d; ++i) s += Ho[a * d + i] * P[i * d + j];
            HP[a * d + j] = s;
        }
    for (int a = 0; a < q; ++a)
        for (int b = 0; b < q; ++b) {
            double s = R[idx[a] * p + idx[b]];
            for (int j = 0; j < d; ++j) s += HP[a * d + j] * Ho[b * d + j];
            S[a * q + b] = s;
        }
    double ell;
    if (p == 1) { /* scalar branch :108-111 */
        const double sd = sqrt(S[0]), z = r[0] / sd;
        ell = -0.5 * z * z - log(sd) - 0.5 * LOG_2PI;
        for (int i = 0; i < d; ++i) G[i] = HP[i] / S[0];
    } else {
        double L[MAXP * MAXP];
        memcpy(L, S, q * q * sizeof(double));
        const int ok = chol(q, L);
        ell = ok ? mvn_logpdf_chol(q, L, r) : NAN;
        /* G = (S^-1 H P)^T :117 */
        for (int j = 0; j < d; ++j) {
            double col[MAXP];
            for (int a = 0; a < q; ++a) col[a] = HP[a * d + j];
            lsolve(q, L, col);
            ltsolve(q, L, col);
            for (int a = 0; a < q; ++a) G[j * q + a] = col[a];
        }
    }
    for (int i = 0; i < d; ++i) {
        double s = 0;
        for (int a = 0; a < q; ++a) s += G[i * q + a] * r[a];
        m[i] += s;
    }
    /* P - G S G^T, symmetrised :122-124 */
    double GS[MAXD * MAXP], Pn[MAXD * MAXD];
    for (int i = 0; i < d; ++i)
        for (int b = 0; b < q; ++b) {
            double s = 0;
            for (int a = 0; a < q; ++a) s += G[i * q + a] * S[a * q + b];
            GS[i * q + b] = s;
        }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = 0;
            for (int a = 0; a < q; ++a) s += GS[i * q + a] * G[j * q + a];
            Pn[i * d + j] = P[i * d + j] - s;
        }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) P[i * d + j] = 0.5 * (Pn[i * d + j] + Pn[j * d + i]);
    return isnan(ell) ? 0.0 : ell;
}

/* sequential_predict (filtering.py:134-139) */
static void seq_predict(int d, const double* F, const double* b, const double* Q, double* m, double* P) {
    double t[MAXD], FP[MAXD * MAXD], Pn[MAXD * MAXD];
    for (int i = 0; i < d; ++i) {
        double s = b[i];
        for (int j = 0; j < d; ++j) s += F[i * d + j] * m[j];
        t[i] = s;
    }
    memcpy(m, t, d * sizeof(double));
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = 0;
            for (int k = 0; k < d; ++k) s += F[i * d + k] * P[k * d + j];
            FP[i * d + j] = s;
        }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = Q[i * d + j];
            for (int k = 0; k < d; ++k) s += FP[i * d + k] * F[j * d + k];
            Pn[i * d + j] = s;
        }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) P[i * d + j] = 0.5 * (Pn[i * d + j] + Pn[j * d + i]);
}

/* concatenated observation at time t: y = [u_t ; yobs_t], H = [I ; Hobs_t], c = [0 ; cobs_t], R = blkdiag(delta/2 I, Robs_t) */
static void concat_obs(const kseq_model* M, long t, const double* u_t, double delta, double* y, double* H, double* c, double* R) {
    const int d = M->d, po = M->po, p = d + po;
    memset(H, 0, sizeof(double) * p * d);
    memset(R, 0, sizeof(double) * p * p);
    for (int k = 0; k < d; ++k) y[k] = u_t[k], c[k] = 0.0, H[k * d + k] = 1.0, R[k * p + k] = 0.5 * delta;
    const double* Ho = M->Hobs + t * M->sH;
    const double* Ro = M->Robs + t * M->sR;
    const double* co = M->cobs + t * M->sc;
    for (int k = 0; k < po; ++k) {
        y[d + k] = M->yobs[t * po + k];
        c[d + k] = co[k];
        for (int j = 0; j < d; ++j) H[(d + k) * d + j] = Ho[k * d + j];
        for (int l = 0; l < po; ++l) R[(d + k) * p + d + l] = Ro[k * po + l];
    }
}

/* log N(y; H x + c, R) of one time step as mvn.logpdf(y, pred, chol(R)) (base.py:159-165); NaN if any residual component is */
static double obs_logpdf(int d, int p, const double* y, const double* H, const double* c, const double* R, const double* x) {
    double r[MAXP], L[MAXP * MAXP];
    for (int k = 0; k < p; ++k) {
        double s = c[k];
        for (int j = 0; j < d; ++j) s += H[k * d + j] * x[j];
        r[k] = y[k] - s;
    }
    if (p == 1) {
        const double sd = sqrt(R[0]), z = r[0] / sd;
        return -0.5 * z * z - log(sd) - 0.5 * LOG_2PI;
    }
    memcpy(L, R, p * p * sizeof(double));
    if (!chol(p, L)) return NAN;
    return mvn_logpdf_chol(p, L, r);
}

/* prior_logpdf(xs) (base.py:99-134) */
static double prior_logpdf(const kseq_model* M, const double* xs) {
    const int d = M->d, T = M->T;
    double r[MAXD], L[MAXD * MAXD], tot;
    for (int k = 0; k < d; ++k) r[k] = xs[k] - M->m0[k];
    if (d == 1) {
        const double sd = sqrt(M->P0[0]), z = r[0] / sd;
        tot = -0.5 * z * z - log(sd) - 0.5 * LOG_2PI;
    } else {
        memcpy(L, M->P0, d * d * sizeof(double));
        tot = chol(d, L) ? mvn_logpdf_chol(d, L, r) : NAN;
    }
    if (isnan(tot)) tot = 0.0;
    for (long t = 1; t < T; ++t) {
        const double* F = M->Fs + (t - 1) * M->sF;
        const double* Q = M->Qs + (t - 1) * M->sQ;
        const double* b = M->bs + (t - 1) * M->sb;
        const double* xp = xs + (t - 1) * d;
        for (int i = 0; i < d; ++i) {
            double s = b[i];
            for (int j = 0; j < d; ++j) s += F[i * d + j] * xp[j];
            r[i] = xs[t * d + i] - s;
        }
        double v;
        if (d == 1) {
            const double sd = sqrt(Q[0]), z = r[0] / sd;
            v = -0.5 * z * z - log(sd) - 0.5 * LOG_2PI;
        } else {
            memcpy(L, Q, d * d * sizeof(double));
            v = chol(d, L) ? mvn_logpdf_chol(d, L, r) : NAN;
        }
        if (!isnan(v)) tot += v; /* nansum */
    }
    return tot;
}

/* One sweep of chain `x` (T, d) in place.  ws: (T*(3d + d*d)) doubles.  Returns accepted; logs[5] = log_alpha, lp_prop, lp_rev, lt_prop,
 * lt_rev (may be NULL); xprop_out (T, d) may be NULL. */
static int sweep_one(const kseq_model* M, double delta, double* x, const double* eps_aux, const double* eps_samp, double u_acc, double* ws,
                     double* logs, double* xprop_out) {
    const int d = M->d, po = M->po, p = d + po, T = M->T;
    double* u = ws;
    double* ms = u + (size_t)T * d;
    double* xp = ms + (size_t)T * d;
    double* Ps = xp + (size_t)T * d;
    const double shd = sqrt(0.5 * delta);
    for (long i = 0; i < (long)T * d; ++i) u[i] = x[i] + shd * eps_aux[i]; /* generic.py:61 */
    double y[MAXP], H[MAXP * MAXD], c[MAXP], R[MAXP * MAXP];
    /* _sequential_filtering (filtering.py:66-79) */
    double m[MAXD], P[MAXD * MAXD], ell;
    memcpy(m, M->m0, d * sizeof(double));
    memcpy(P, M->P0, d * d * sizeof(double));
    concat_obs(M, 0, u, delta, y, H, c, R);
    ell = seq_update(d, p, y, H, c, R, m, P);
    memcpy(ms, m, d * sizeof(double));
    memcpy(Ps, P, d * d * sizeof(double));
    for (long t = 1; t < T; ++t) {
        seq_predict(d, M->Fs + (t - 1) * M->sF, M->bs + (t - 1) * M->sb, M->Qs + (t - 1) * M->sQ, m, P);
        concat_obs(M, t, u + t * d, delta, y, H, c, R);
        ell += seq_update(d, p, y, H, c, R, m, P);
        memcpy(ms + t * d, m, d * sizeof(double));
        memcpy(Ps + t * d * d, P, d * d * sizeof(double));
    }
    /* sampling, sequential branch (sampling.py:34-39) on mean_and_chol (:60-105) and _sample_last_step (:115-124) */
    {
        double L[MAXD * MAXD], xn[MAXD];
        const double* Pl = Ps + (size_t)(T - 1) * d * d;
        if (d == 1) {
            L[0] = sqrt(Pl[0]);
            if (isnan(L[0])) L[0] = 0.0;
        } else {
            memcpy(L, Pl, d * d * sizeof(double));
            if (!chol(d, L)) memset(L, 0, sizeof(double) * d * d); /* nan_to_num */
        }
        for (int i = 0; i < d; ++i) {
            double s = ms[(size_t)(T - 1) * d + i];
            for (int k = 0; k <= i; ++k) s += L[i * d + k] * eps_samp[(size_t)(T - 1) * d + k];
            xp[(size_t)(T - 1) * d + i] = s;
        }
        for (long t = T - 2; t >= 0; --t) {
            const double* F = M->Fs + t * M->sF;
            const double* Q = M->Qs + t * M->sQ;
            const double* b = M->bs + t * M->sb;
            const double* mt = ms + t * d;
            const double* Pt = Ps + t * d * d;
            double FP[MAXD * MAXD], S[MAXD * MAXD], G[MAXD * MAXD], Sig[MAXD * MAXD], GS[MAXD * MAXD], pm[MAXD];
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    double s = 0;
                    for (int k = 0; k < d; ++k) s += F[i * d + k] * Pt[k * d + j];
                    FP[i * d + j] = s;
                }
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    double s = Q[i * d + j];
                    for (int k = 0; k < d; ++k) s += FP[i * d + k] * F[j * d + k];
                    Sig[i * d + j] = s;
                }
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) S[i * d + j] = 0.5 * (Sig[i * d + j] + Sig[j * d + i]);
            if (d == 1) {
                G[0] = Pt[0] * F[0] / S[0];
            } else {
                double Ls[MAXD * MAXD];
                memcpy(Ls, S, d * d * sizeof(double));
                const int ok = chol(d, Ls);
                /* gain = P (S^-1 F)^T  == (S^-1 F P)^T */
                for (int j = 0; j < d; ++j) {
                    double col[MAXD];
                    for (int i = 0; i < d; ++i) col[i] = FP[i * d + j];
                    if (ok) {
                        lsolve(d, Ls, col);
                        ltsolve(d, Ls, col);
                    }
                    for (int i = 0; i < d; ++i) G[j * d + i] = ok ? col[i] : NAN;
                }
            }
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    double s = 0;
                    for (int k = 0; k < d; ++k) s += G[i * d + k] * S[k * d + j];
                    GS[i * d + j] = s;
                }
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    double s = 0;
                    for (int k = 0; k < d; ++k) s += GS[i * d + k] * G[j * d + k];
                    Sig[i * d + j] = Pt[i * d + j] - s;
                }
            for (int i = 0; i < d; ++i)
                for (int j = i; j < d; ++j) L[i * d + j] = L[j * d + i] = 0.5 * (Sig[i * d + j] + Sig[j * d + i]);
            if (d == 1) {
                L[0] = sqrt(L[0]);
                if (isnan(L[0])) L[0] = 0.0;
            } else if (!chol(d, L)) {
                memset(L, 0, sizeof(double) * d * d);
            }
            for (int i = 0; i < d; ++i) {
                double s = b[i];
                for (int j = 0; j < d; ++j) s += F[i * d + j] * mt[j];
                pm[i] = s;
            }
            for (int i = 0; i < d; ++i) {
                double inc = mt[i];
                for (int j = 0; j < d; ++j) inc -= G[i * d + j] * pm[j];
                for (int k = 0; k <= i; ++k) inc += L[i * d + k] * eps_samp[t * d + k];
                double s = inc;
                for (int j = 0; j < d; ++j) s += G[i * d + j] * xp[(t + 1) * d + j];
                xn[i] = s;
            }
            memcpy(xp + t * d, xn, d * sizeof(double));
        }
    }
    /* posterior_logpdf of both moves (generic.py:88, base.py:72-96) and the target (generic.py:89) */
    double ll_c_prop = 0, ll_c_rev = 0, ll_o_prop = 0, ll_o_rev = 0, corr = 0;
    for (long t = 0; t < T; ++t) {
        concat_obs(M, t, u + t * d, delta, y, H, c, R);
        double v = obs_logpdf(d, p, y, H, c, R, xp + t * d);
        if (!isnan(v)) ll_c_prop += v;
        v = obs_logpdf(d, p, y, H, c, R, x + t * d);
        if (!isnan(v)) ll_c_rev += v;
        const double* Ho = M->Hobs + t * M->sH;
        const double* Ro = M->Robs + t * M->sR;
        const double* co = M->cobs + t * M->sc;
        v = obs_logpdf(d, po, M->yobs + t * po, Ho, co, Ro, xp + t * d);
        if (!isnan(v)) ll_o_prop += v;
        v = obs_logpdf(d, po, M->yobs + t * po, Ho, co, Ro, x + t * d);
        if (!isnan(v)) ll_o_rev += v;
        for (int k = 0; k < d; ++k) {
            const double a = (xp[t * d + k] - u[t * d + k]) / sqrt(delta), b2 = (x[t * d + k] - u[t * d + k]) / sqrt(delta);
            corr += a * a - b2 * b2;
        }
    }
    const double pr_prop = prior_logpdf(M, xp), pr_rev = prior_logpdf(M, x);
    const double lp_prop = ll_c_prop - ell + pr_prop, lp_rev = ll_c_rev - ell + pr_rev;
    const double lt_prop = pr_prop + ll_o_prop, lt_rev = pr_rev + ll_o_rev;
    double la = lt_prop - lt_rev; /* _get_alpha :98-106 */
    la += lp_rev - lp_prop;
    la -= corr;
    const double alpha = exp(la != la ? la : (la < 0.0 ? la : 0.0)); /* jnp.minimum(0, nan) = nan: a NaN ratio rejects */
    const int acc = u_acc < alpha;
    if (logs) logs[0] = la, logs[1] = lp_prop, logs[2] = lp_rev, logs[3] = lt_prop, logs[4] = lt_rev;
    if (xprop_out) memcpy(xprop_out, xp, sizeof(double) * T * d);
    if (acc) memcpy(x, xp, sizeof(double) * T * d);
    return acc;
}

/* C chains: x (C,T,d) in/out, eps_aux / eps_samp (C,T,d), u_acc (C), accepted (C), logs (C,5) or NULL, xprop (C,T,d) or NULL.
 * nthreads <= 0: the OpenMP default.  Returns 0, or -1 on bad sizes / allocation failure. */
int kseq_sweep(const kseq_model* M, int C, double delta, double* x, const double* eps_aux, const double* eps_samp, const double* u_acc,
               int* accepted, double* logs, double* xprop, int nthreads) {
    if (!M || M->d < 1 || M->d > MAXD || M->po < 1 || M->d + M->po > MAXP || M->T < 1 || C < 1) return -1;
    const size_t per = (size_t)M->T * (3 * M->d + M->d * M->d);
    const size_t TD = (size_t)M->T * M->d;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        double* ws = (double*)malloc(per * sizeof(double));
        if (!ws) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            fail = 1;
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int c = 0; c < C; ++c) {
            if (!ws) continue;
            accepted[c] = sweep_one(M, delta, x + c * TD, eps_aux + c * TD, eps_samp + c * TD, u_acc[c], ws, logs ? logs + 5 * c : NULL,
                                    xprop ? xprop + c * TD : NULL);
        }
        free(ws);
    }
    return fail ? -1 : 0;
}

int kseq_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// affine_shared.h -- chain-shared model parameters: what is left PER CHAIN of the parallel filter / sampler.
//
// When (F, Q, b, H, R, c, P0) are the same for every chain -- the factories of a linear-Gaussian model ignore the linearisation
// point; under jax.vmap the reference leaves all of it unbatched -- every MATRIX of the filter's associative scan
// (filtering.py:163-183: A, C, J of every element and of every prefix) is chain-independent.  The d x d block-affine combine then
// runs ONCE per sweep, on one sequence (the matrix filter: the same k_filter_init / k_scan_* kernels with S = 1), and yields the
// filtered covariances P_t.  From them one table row per transition holds the operators of the chain's part of the prefixes:
//     K_t  = P_t^- H^T S_t^-1            (P_t^- = F P_{t-1} F^T + Q, S_t = H P_t^- H^T + R; masked rows deleted)
//     Mb_t = F - K_t H F,   kc_t = b - K_t (H b + c)          m_t = Mb_t m_{t-1} + kc_t + K_t y_t
//     Si_t = S_t^-1, c0_t = -log|S_t|/2 - dim/2 log 2 pi        ell += -r^T Si_t r / 2 + c0_t,  r = y_t - HF_t m_{t-1} - ym_t
//     HF_t = H F, ym_t = H b + c                                (r = y_t - H (F m_{t-1} + b) - c)
// i.e. the (b, z) components of prefix k of the scan written as a recursion in k -- the composition of the same chain-shared
// operators in the same order; the eta / J components are only needed to build prefixes of unknown incoming covariance, which the
// matrix filter has already done.  What a chain carries is an AFFINE map with chain-shared matrices:
//     reduce : one lane per (chain, chunk) folds its chunk:  h <- Mb_i h + kc_i + K_i y_i             (no matrix product per chain)
//     aggs   : the generic (G, e) scan of kernels.hip.h over the chunk aggregates (G = product of the chunk's Mb, from a table)
//     down   : re-walks the chunk from its incoming mean, writes the filtered means and accumulates the log-likelihood.
// The pathwise sampler (sampling.py:51-55) is the same recursion backwards in time: x_t = G_t x_{t+1} + inc_t, G_t chain-shared.
// Lanes run over chains (chain-minor buffers), table rows are wave-uniform scalar loads.
#pragma once
#include "rng.h"
#include "kalman_bodies.h"

namespace ax {

template <typename R, int D, int P> struct GainRow {
    static constexpr int PS = symsize(P);
    static constexpr int oM = 0, oKc = D * D, oK = oKc + D, oHF = oK + D * P, oYm = oHF + P * D, oSi = oYm + P, oC0 = oSi + PS, N = oC0 + 1;
    static constexpr int VEC = 16 / sizeof(R);
    static constexpr int NPAD = (N + VEC - 1) / VEC * VEC;
};

// table row of transition t-1 -> t from the filtered covariance of t-1 (Pprev, dense) and the model at t; y carries the mask only
template <typename R, int D, int P>
AX_HD void gain_row(const R* F, const R* bdyn, const R* Q, const R* Pprev, const R* H, const R* c, const R* __restrict__ Rm, const R* y, R* row) {
    using T = GainRow<R, D, P>;
    R P_[D * D];
    {
        R FP[D * D], Pn[D * D];
        mm<R, D, D, D>(F, Pprev, FP);
        mmt<R, D, D, D>(FP, F, Pn);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) P_[i * D + j] = (i == j) ? Pn[i * D + i] + Q[i * D + i] : (R)0.5 * (Pn[i * D + j] + Pn[j * D + i]) + (R)0.5 * (Q[i * D + j] + Q[j * D + i]);
    }
    bool nan[P];
    R H_[P * D], c_[P];
    const bool any = obs_mask<R, D, P>(y, H, c, nan, H_, c_);
#pragma unroll
    for (int i = 0; i < T::N; ++i) row[i] = 0;
    if (!any) {  // _passthrough (filtering.py:127-130): m_t = F m + b, no likelihood term
#pragma unroll
        for (int i = 0; i < D * D; ++i) row[T::oM + i] = F[i];
#pragma unroll
        for (int i = 0; i < D; ++i) row[T::oKc + i] = bdyn[i];
        return;
    }
    R L[symsize(P)], invd[P], PHt[D * P];
    innovation_cov<R, D, P>(P_, H_, Rm, nan, PHt, L);
    const bool ok = chol_inplace<R, P>(L, invd, nan);
    R K[D * P];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R g[P];
#pragma unroll
        for (int k = 0; k < P; ++k) g[k] = PHt[i * P + k];
        cho_solve<R, P>(L, invd, g);
#pragma unroll
        for (int k = 0; k < P; ++k) K[i * P + k] = nan[k] ? (R)0 : g[k];
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {  // column k of S^-1 (deleted components: zero)
        R e[P];
#pragma unroll
        for (int l = 0; l < P; ++l) e[l] = (l == k && !nan[k]) ? (R)1 : (R)0;
        cho_solve<R, P>(L, invd, e);
#pragma unroll
        for (int l = k; l < P; ++l) row[T::oSi + sidx_u(P, k, l)] = (nan[k] || nan[l]) ? (R)0 : e[l];
    }
    R HF[P * D], ym[P];
    mm<R, P, D, D>(H_, F, HF);
    R logdet = 0;
    int dim = 0;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        R s = c_[k];
#pragma unroll
        for (int j = 0; j < D; ++j) s += H_[k * D + j] * bdyn[j];
        ym[k] = s;
        logdet += nan[k] ? (R)0 : log_(L[lidx(k, k)]);
        dim += nan[k] ? 0 : 1;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R kc = bdyn[i];
#pragma unroll
        for (int k = 0; k < P; ++k) kc -= K[i * P + k] * ym[k];
        row[T::oKc + i] = kc;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R s = F[i * D + j];
#pragma unroll
            for (int k = 0; k < P; ++k) s -= K[i * P + k] * HF[k * D + j];
            row[T::oM + i * D + j] = s;
        }
    }
#pragma unroll
    for (int i = 0; i < D * P; ++i) row[T::oK + i] = K[i], row[T::oHF + i] = HF[i];
#pragma unroll
    for (int k = 0; k < P; ++k) row[T::oYm + k] = ym[k];
    row[T::oC0] = -logdet - (R)(0.5 * LOG_2PI) * (R)dim;
    if (!ok) {  // a failed Cholesky is an all-NaN factor in the reference (jnp.linalg.cholesky)
        const R bad = r_nan<R>();
#pragma unroll
        for (int i = 0; i < T::N; ++i) row[i] = bad;
    }
}

// lane i = transition i -> i + 1; Ps1 = the matrix filter's covariances, dense (T, D, D); the observation mask is chain 0's / the data's
template <typename R, int D, int P> AX_HD void body_gain_tab(const FilterArgs& a, const R* __restrict__ Ps1, int i) {
    using T = GainRow<R, D, P>;
    const long long t = (long long)i + 1;
    R F[D * D], bd[D], Q[D * D], Pp[D * D], H[P * D], cv[P], y[P], Rm[P * P], row[T::N];
    rd<R, D * D>(a.Fs, 0, i, 0, F);
    rd<R, D>(a.bs, 0, i, 0, bd);
    rd<R, D * D>(a.Qs, 0, i, 0, Q);
    rd<R, P * D>(a.Hs, 0, t, 0, H);
    rd<R, P>(a.cs, 0, t, 0, cv);
    if (a.aux_on) {
#pragma unroll
        for (int k = 0; k < P; ++k) y[k] = k < D ? (R)0 : at<R>(a.aux_yobs, 0, t, 0)[k - D];
    } else if (a.mask_ys.ptr) {
        rd<R, P>(a.mask_ys, 0, t, 0, y);
    } else {
        rd<R, P>(a.ys, 0, t, 0, y);
    }
    rd_upper<R, P>(a.Rs, 0, t, 0, Rm);
    ld<R, D * D>(Ps1 + (long long)i * D * D, Pp);
    gain_row<R, D, P>(F, bd, Q, Pp, H, cv, Rm, y, row);
    if constexpr (P > D) {
        // concatenated observations y_t = [u_t ; yobs_t]: the yobs part is the data's, the same for every chain, so its contribution goes into the
        // row once -- kc += K[:, D:] yobs_t and ym[D:] -= yobs_t (masked components: K columns, HF rows and ym are zero already) -- and a chain's
        // fold / walk only touch its own D auxiliary components (FilterMeanOp, `folded`)
        if (a.aux_on) {
#pragma unroll
            for (int k = D; k < P; ++k) {
                const R yk = finite_(y[k]) ? y[k] : (R)0;
#pragma unroll
                for (int r = 0; r < D; ++r) row[T::oKc + r] += row[T::oK + r * P + k] * yk;
                row[T::oYm + k] -= yk;
            }
        }
    }
    stv<R, T::N>((R*)a.tab + (long long)i * T::NPAD, row);
}

// the observation of chain s at time t (>= 1): given, or the concatenated auxiliary observation built on the fly (FilterArgs::aux_*)
// GEN: this is the FIRST reader of eps_aux[c, t] and the sweep is keyed (FilterArgs::aux_gen): draw it here and store it for the later readers
template <typename R, int D, int P, bool WRITE_U, bool GEN = false> AX_HD void aff_obs(const FilterArgs& a, int s, long long t, R* y) {
    const int c = s / a.d.B, b = s % a.d.B;
    if (a.aux_on) {
        if constexpr (P > D) {
            R xv[D], ev[D], uv[D];
            rd<R, D>(a.aux_x, c, t, b, xv);
            if constexpr (GEN) {
                normals_cm<R, D>(a.gen_k0, a.gen_k1, (long long)c * a.aux_eps.sc + t * a.aux_eps.st + (long long)b * a.aux_eps.sb, a.aux_eps.se, ev);
                wr<R, D>(a.aux_eps, c, t, b, ev);
            } else {
                rd<R, D>(a.aux_eps, c, t, b, ev);
            }
#pragma unroll
            for (int k = 0; k < D; ++k) uv[k] = xv[k] + (R)arg_aux_shd(a) * ev[k], y[k] = uv[k];
            if constexpr (WRITE_U) wr<R, D>(a.aux_u, c, t, b, uv);
            const UniformRow<R> yo = uniform_row<R>(at<R>(a.aux_yobs, 0, t, 0));
#pragma unroll
            for (int k = D; k < P; ++k) y[k] = yo[k - D];
        }
    } else {
        rd<R, P>(a.ys, c, t, b, y);
    }
}

// the chain's own part of a concatenated observation: u_t = x_t + shd eps_t (GEN: drawn here, as in aff_obs)
template <typename R, int D, bool GEN> AX_HD void aff_aux(const FilterArgs& a, int s, long long t, R* u) {
    const int c = s / a.d.B, b = s % a.d.B;
    R xv[D], ev[D];
    rd<R, D>(a.aux_x, c, t, b, xv);
    if constexpr (GEN) {
        normals_cm<R, D>(a.gen_k0, a.gen_k1, (long long)c * a.aux_eps.sc + t * a.aux_eps.st + (long long)b * a.aux_eps.sb, a.aux_eps.se, ev);
        wr<R, D>(a.aux_eps, c, t, b, ev);
    } else {
        rd<R, D>(a.aux_eps, c, t, b, ev);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) u[k] = xv[k] + (R)arg_aux_shd(a) * ev[k];
}

// ---- the two chain-shared affine recursions behind one interface ------------------------------------------------------------------
//   N                    number of scan positions
//   mat(a, j, G)         the chain-shared matrix of position j                              (table kernels, lane = chunk)
//   init(a, s, h)        the state entering position 0
//   fold(a, s, j, h)     h <- G_j h + v_j(s)                                                (reduce pass)
//   walk(a, s, j, h, acc) the same + writes the outputs of position j; acc += log-likelihood increment (down pass)
template <typename R_, int D, int P> struct FilterMeanOp {
    using R = R_;
    using Args = FilterArgs;
    static constexpr int kAffWaves = 11;
    using T = GainRow<R, D, P>;
    static AX_HD int length(const Args& a) { return a.d.n(); }
    static AX_HD void mat(const Args& a, int i, R* G) {
        const R* row = (const R*)a.tab + (long long)i * T::NPAD;
#pragma unroll
        for (int k = 0; k < D * D; ++k) G[k] = row[T::oM + k];
    }
    static AX_HD void init(const Args& a, int s, R* h) { rd<R, D>(a.ms, s / a.d.B, 0, s % a.d.B, h); }  // m0+ of the chain (k_filter_t0)
    static AX_HD void fold(const Args& a, int s, int i, R* h) {
        const UniformRow<R> row = uniform_row<R>((const R*)a.tab + (long long)i * T::NPAD);
        if constexpr (P > D) {
            if (a.aux_on) {  // folded rows (body_gain_tab): only the chain's D auxiliary components are left
                R u[D], o[D];
                if (a.aux_gen) aff_aux<R, D, true>(a, s, (long long)i + 1, u);  // (the reduce pass reads every position first)
                else aff_aux<R, D, false>(a, s, (long long)i + 1, u);
#pragma unroll
                for (int r = 0; r < D; ++r) {
                    R v = row[T::oKc + r];
#pragma unroll
                    for (int k = 0; k < D; ++k) v += row[T::oM + r * D + k] * h[k];
#pragma unroll
                    for (int k = 0; k < D; ++k) v += row[T::oK + r * P + k] * (finite_(u[k]) ? u[k] : (R)0);
                    o[r] = v;
                }
#pragma unroll
                for (int r = 0; r < D; ++r) h[r] = o[r];
                return;
            }
        }
        R y[P];
        if (a.aux_gen) aff_obs<R, D, P, false, true>(a, s, (long long)i + 1, y);  // (the reduce pass reads every position first)
        else aff_obs<R, D, P, false>(a, s, (long long)i + 1, y);
        R o[D];
#pragma unroll
        for (int r = 0; r < D; ++r) {
            R v = row[T::oKc + r];
#pragma unroll
            for (int k = 0; k < D; ++k) v += row[T::oM + r * D + k] * h[k];
#pragma unroll
            for (int k = 0; k < P; ++k) v += row[T::oK + r * P + k] * (finite_(y[k]) ? y[k] : (R)0);
            o[r] = v;
        }
#pragma unroll
        for (int r = 0; r < D; ++r) h[r] = o[r];
    }
    // m_t = Mb m + kc + K y (the same affine step as fold) and the innovation r = y - H (F m + b) - c = y - HF m - ym for the
    // log-likelihood increment -r^T S^-1 r / 2 + c0; both read the incoming mean only, so they issue independently
    static AX_HD void walk(const Args& a, int s, int i, R* m, R& acc) {
        if constexpr (P > D) {
            if (a.aux_on) return walk_impl<true>(a, s, i, m, acc);
        }
        walk_impl<false>(a, s, i, m, acc);
    }
    // FOLDED: rows of body_gain_tab with the data's part of a concatenated observation folded in -- y[D:] does not appear: its K columns are in kc,
    // and r[k >= D] = -(HF_k m + ym'_k) with ym' = ym - yobs (all zero for a masked component)
    template <bool FOLDED> static AX_HD void walk_impl(const Args& a, int s, int i, R* m, R& acc) {
        constexpr int PA = FOLDED ? D : P;  // the chain's own components
        const long long t = (long long)i + 1;
        const UniformRow<R> row = uniform_row<R>((const R*)a.tab + (long long)i * T::NPAD);
        R y[PA];
        if constexpr (FOLDED) aff_aux<R, D, false>(a, s, t, y);
        else aff_obs<R, D, P, false>(a, s, t, y);
        R r[P], o[D];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            R v = row[T::oYm + k];
#pragma unroll
            for (int j = 0; j < D; ++j) v += row[T::oHF + k * D + j] * m[j];
            if (k < PA) {
                const bool fin = finite_(y[k < PA ? k : 0]);
                y[k < PA ? k : 0] = fin ? y[k < PA ? k : 0] : (R)0;
                r[k] = fin ? y[k < PA ? k : 0] - v : (R)0;
            } else {
                r[k] = -v;
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            R v = row[T::oKc + k];
#pragma unroll
            for (int j = 0; j < D; ++j) v += row[T::oM + k * D + j] * m[j];
#pragma unroll
            for (int l = 0; l < PA; ++l) v += row[T::oK + k * P + l] * y[l];
            o[k] = v;
        }
        R q = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            R sk = 0;
#pragma unroll
            for (int l = 0; l < P; ++l) sk += row[T::oSi + sidx(P, k, l)] * r[l];
            q += r[k] * sk;
        }
        const R inc = (R)-0.5 * q + row[T::oC0];
        acc += isnan_(inc) ? (R)0 : inc;  // nansum (filtering.py:62)
#pragma unroll
        for (int k = 0; k < D; ++k) m[k] = o[k];
        wr<R, D>(a.ms, s / a.d.B, t, s % a.d.B, m);
    }
};

// scan position j <-> time t = T-1-j; table rows of kalman_math.h::SampShared [G | M1 | gb | Lc]
template <typename R_, int D> struct SampleAffOp {
    using R = R_;
    using Args = SampleArgs;
    using T = SampShared<R, D>;
    static AX_HD int length(const Args& a) { return a.d.T; }
    static AX_HD void mat(const Args& a, int j, R* G) {
        const R* row = (const R*)a.tab + ((long long)a.d.T - 1 - j) * T::NPAD;
#pragma unroll
        for (int k = 0; k < D * D; ++k) G[k] = row[T::oG + k];
    }
    static AX_HD void init(const Args&, int, R* h) {
#pragma unroll
        for (int k = 0; k < D; ++k) h[k] = 0;  // G_{T-1} = 0: the first position ignores the incoming state
    }
    static AX_HD void fold(const Args& a, int s, int j, R* h) {  // the reduce pass: the first reader of eps
        if (a.eps_gen) step<true>(a, s, j, h);
        else step<false>(a, s, j, h);
    }
    template <bool GEN> static AX_HD void step(const Args& a, int s, int j, R* h) {
        const int c = s / a.d.B, b = s % a.d.B;
        const long long t = (long long)a.d.T - 1 - j;
        const UniformRow<R> row = uniform_row<R>((const R*)a.tab + t * T::NPAD);
        R m[D], eps[D], o[D];
        rd<R, D>(a.ms, c, t, b, m);
        if constexpr (GEN) {
            normals_cm<R, D>(a.gen_k0, a.gen_k1, (long long)c * a.eps.sc + t * a.eps.st + (long long)b * a.eps.sb, a.eps.se, eps);
            wr<R, D>(a.eps, c, t, b, eps);
        } else {
            rd<R, D>(a.eps, c, t, b, eps);
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            R v = -row[T::oGb + i];
#pragma unroll
            for (int k = 0; k < D; ++k) v += row[T::oM + i * D + k] * m[k];
#pragma unroll
            for (int k = 0; k <= i; ++k) v += row[T::oL + i * D + k] * eps[k];
#pragma unroll
            for (int k = 0; k < D; ++k) v += row[T::oG + i * D + k] * h[k];
            o[i] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) h[i] = o[i];
    }
    static AX_HD void walk(const Args& a, int s, int j, R* h, R&) {
        step<false>(a, s, j, h);
        wr<R, D>(a.xs, s / a.d.B, (long long)a.d.T - 1 - j, s % a.d.B, h);
    }
};

}  // namespace ax

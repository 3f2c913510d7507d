// kalman_bodies.h -- per-thread bodies of the Kalman kernels and the two scan operators.
// A body takes (args, s, i): s = sequence (chain c = s / B, batch b = s % B), i = time / element index.
// They are AX_HD so tests/hostsim can run them on the CPU; the product launches them from kernels.hip.
#pragma once
#include "kalman_math.h"

namespace ax {

struct Arr {
    const void* ptr;
    long long sc, st, sb;
};
template <typename R> AX_HD const R* at(const Arr& a, int c, long long t, int b) {
    return (const R*)a.ptr + (long long)c * a.sc + t * a.st + (long long)b * a.sb;
}

struct KDims {
    int C, T, B;
    AX_HD int S() const { return C * B; }
    AX_HD int n() const { return T - 1; }
    // dense (C, T, B, ...) offset in records
    AX_HD long long rec(int s, long long t) const { return ((long long)(s / B) * T + t) * B + (s % B); }
};

struct FilterArgs {
    KDims d;
    Arr m0, P0, Fs, Qs, bs, Hs, Rs, cs, ys;
    void* ms;       // dense (C,T,B,D)
    void* Ps;       // dense (C,T,B,D,D)
    void* elem;     // [S][n][FiltElem::NPAD]
    void* ell0;     // [S]
};

// ---- t = 0 measurement update (filtering.py:52) -------------------------------------------------
template <typename R, int D, int P> AX_HD void body_filter_t0(const FilterArgs& a, int s) {
    const int c = s / a.d.B, b = s % a.d.B;
    R m[D], Pd[D * D], H[P * D], cv[P], y[P];
    ld<R, D>(at<R>(a.m0, c, 0, b), m);
    ld<R, D * D>(at<R>(a.P0, c, 0, b), Pd);
    ld<R, P * D>(at<R>(a.Hs, c, 0, b), H);
    ld<R, P>(at<R>(a.cs, c, 0, b), cv);
    ld<R, P>(at<R>(a.ys, c, 0, b), y);
    const R ell = kalman_update<R, D, P>(m, Pd, H, cv, at<R>(a.Rs, c, 0, b), y);
    const long long r = a.d.rec(s, 0);
    st<R, D>((R*)a.ms + r * D, m);
    st<R, D * D>((R*)a.Ps + r * D * D, Pd);
    ((R*)a.ell0)[s] = ell;
}

// ---- scan element for transition i -> i+1 (filtering.py:188-250) ---------------------------------
template <typename R, int D, int P> AX_HD void body_filter_init(const FilterArgs& a, int s, int i) {
    const int c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    R F[D * D], bd[D], m_[D], P_[D * D];
    ld<R, D * D>(at<R>(a.Fs, c, i, b), F);
    ld<R, D>(at<R>(a.bs, c, i, b), bd);
    if (i == 0) {
        const long long r = a.d.rec(s, 0);
        R Q[D * D];
        ld<R, D>((const R*)a.ms + r * D, m_);
        ld<R, D * D>((const R*)a.Ps + r * D * D, P_);
        ld<R, D * D>(at<R>(a.Qs, c, i, b), Q);
        // m_ = F m + b ; P_ = F P F^T + Q   (not symmetrised: filtering.py:200-201)
        R tm[D], FP[D * D], Pn[D * D];
        mv<R, D, D>(F, m_, tm);
        mm<R, D, D, D>(F, P_, FP);
        mmt<R, D, D, D>(FP, F, Pn);
#pragma unroll
        for (int k = 0; k < D; ++k) m_[k] = tm[k] + bd[k];
#pragma unroll
        for (int k = 0; k < D * D; ++k) P_[k] = Pn[k] + Q[k];
    } else {
        // (m, P) = (0, 0): m_ = b, P_ = Q exactly
#pragma unroll
        for (int k = 0; k < D; ++k) m_[k] = bd[k];
        ld<R, D * D>(at<R>(a.Qs, c, i, b), P_);
    }
    R H[P * D], cv[P], y[P];
    ld<R, P * D>(at<R>(a.Hs, c, t, b), H);
    ld<R, P>(at<R>(a.cs, c, t, b), cv);
    ld<R, P>(at<R>(a.ys, c, t, b), y);
    FiltElem<R, D> e;
    filter_elem<R, D, P>(F, bd, m_, P_, H, cv, at<R>(a.Rs, c, t, b), y, e);
    fe_store<R, D>((R*)a.elem + ((long long)s * a.d.n() + i) * FiltElem<R, D>::NPAD, e);
}

// ---- log-likelihood increment of step i+1 from the filtered moments at i (filtering.py:60) ---------
template <typename R, int D, int P> AX_HD R body_filter_ell(const FilterArgs& a, int s, int i) {
    const int c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    const long long r = a.d.rec(s, i);
    R m[D], Pd[D * D], F[D * D], bd[D], Q[D * D];
    ld<R, D>((const R*)a.ms + r * D, m);
    ld<R, D * D>((const R*)a.Ps + r * D * D, Pd);
    ld<R, D * D>(at<R>(a.Fs, c, i, b), F);
    ld<R, D>(at<R>(a.bs, c, i, b), bd);
    ld<R, D * D>(at<R>(a.Qs, c, i, b), Q);
    kalman_predict<R, D>(m, Pd, F, bd, Q);
    R H[P * D], cv[P], y[P];
    ld<R, P * D>(at<R>(a.Hs, c, t, b), H);
    ld<R, P>(at<R>(a.cs, c, t, b), cv);
    ld<R, P>(at<R>(a.ys, c, t, b), y);
    return kalman_update<R, D, P>(m, Pd, H, cv, at<R>(a.Rs, c, t, b), y);
}

// ---- scan operator: parallel filter ----------------------------------------------------------------
struct ScanBufs {
    void* agg;  // [S][nchunk][Full::NPAD]
    void* pre;  // [S][nchunk][Pre::NPAD]
};

template <typename R_, int D> struct FilterOp {
    using R = R_;
    using Full = FiltElem<R, D>;
    using Pre = FiltPre<R, D>;
    using Args = FilterArgs;
    static constexpr int DS = symsize(D);
    static AX_HD int length(const Args& a) { return a.d.n(); }
    static AX_HD void load(const Args& a, int s, int i, Full& e) {
        fe_load<R, D>((const R*)a.elem + ((long long)s * a.d.n() + i) * Full::NPAD, e);
    }
    static AX_HD void load_rec(const R* p, Full& e) { fe_load<R, D>(p, e); }
    static AX_HD void store_rec(R* p, const Full& e) { fe_store<R, D>(p, e); }
    static AX_HD void identity(Full& e) { fe_identity<R, D>(e); }
    static AX_HD void combine(const Full& a1, const Full& a2, Full& o) { filter_combine<R, D>(a1, a2, o); }
    static AX_HD void to_pre(const Full& f, Pre& p) {
#pragma unroll
        for (int i = 0; i < D; ++i) p.b[i] = f.b[i];
#pragma unroll
        for (int i = 0; i < DS; ++i) p.C[i] = f.C[i];
    }
    static AX_HD void store_pre(R* q, const Pre& p) {
        st<R, D>(q, p.b);
        st<R, DS>(q + D, p.C);
    }
    static AX_HD void load_pre(const R* q, Pre& p) {
        ld<R, D>(q, p.b);
        ld<R, DS>(q + D, p.C);
    }
    static AX_HD void apply(const Pre& p, const Full& e, Pre& o) { filter_apply<R, D>(p, e, o); }
    // inclusive prefix i  ->  filtered moments at time i + 1
    static AX_HD void write_out(const Args& a, int s, int i, const Pre& p) {
        const long long r = a.d.rec(s, (long long)i + 1);
        st<R, D>((R*)a.ms + r * D, p.b);
        R Pd[D * D];
        symunpack<R, D>(p.C, Pd);
        st<R, D * D>((R*)a.Ps + r * D * D, Pd);
    }
};

// ---- sampler ------------------------------------------------------------------------------------------
struct SampleArgs {
    KDims d;
    Arr Fs, Qs, bs;
    const void* ms;   // dense
    const void* Ps;   // dense
    const void* eps;  // dense (C,T,B,D)
    void* xs;         // dense (C,T,B,D)
    void* elem;       // [S][T][SampElem::NPAD], scan position j = T-1-t
};

template <typename R, int D> AX_HD void body_sample_init(const SampleArgs& a, int s, int j) {
    const int c = s / a.d.B, b = s % a.d.B;
    const int T = a.d.T;
    const long long t = (long long)T - 1 - j;
    const long long r = a.d.rec(s, t);
    R m[D], Pd[D * D], eps[D];
    ld<R, D>((const R*)a.ms + r * D, m);
    ld<R, D * D>((const R*)a.Ps + r * D * D, Pd);
    ld<R, D>((const R*)a.eps + r * D, eps);
    SampElem<R, D> e;
    if (j == 0) {
        sample_last<R, D>(m, Pd, eps, e);
    } else {
        R F[D * D], Q[D * D], bd[D];
        ld<R, D * D>(at<R>(a.Fs, c, t, b), F);
        ld<R, D * D>(at<R>(a.Qs, c, t, b), Q);
        ld<R, D>(at<R>(a.bs, c, t, b), bd);
        sample_elem<R, D>(F, Q, bd, m, Pd, eps, e);
    }
    R* p = (R*)a.elem + ((long long)s * T + j) * SampElem<R, D>::NPAD;
    st<R, D * D>(p, e.G);
    st<R, D>(p + D * D, e.e);
}

template <typename R_, int D> struct SampleOp {
    using R = R_;
    using Full = SampElem<R, D>;
    using Pre = SampPre<R, D>;
    using Args = SampleArgs;
    static AX_HD int length(const Args& a) { return a.d.T; }
    static AX_HD void load_rec(const R* p, Full& e) {
        ld<R, D * D>(p, e.G);
        ld<R, D>(p + D * D, e.e);
    }
    static AX_HD void store_rec(R* p, const Full& e) {
        st<R, D * D>(p, e.G);
        st<R, D>(p + D * D, e.e);
    }
    static AX_HD void load(const Args& a, int s, int j, Full& e) {
        load_rec((const R*)a.elem + ((long long)s * a.d.T + j) * Full::NPAD, e);
    }
    static AX_HD void identity(Full& e) {
#pragma unroll
        for (int i = 0; i < D * D; ++i) e.G[i] = (i / D == i % D) ? (R)1 : (R)0;
#pragma unroll
        for (int i = 0; i < D; ++i) e.e[i] = 0;
    }
    static AX_HD void combine(const Full& a1, const Full& a2, Full& o) { sample_combine<R, D>(a1, a2, o); }
    static AX_HD void to_pre(const Full& f, Pre& p) {
#pragma unroll
        for (int i = 0; i < D; ++i) p.e[i] = f.e[i];
    }
    static AX_HD void store_pre(R* q, const Pre& p) { st<R, D>(q, p.e); }
    static AX_HD void load_pre(const R* q, Pre& p) { ld<R, D>(q, p.e); }
    static AX_HD void apply(const Pre& p, const Full& e, Pre& o) { sample_apply<R, D>(p, e, o); }
    static AX_HD void write_out(const Args& a, int s, int j, const Pre& p) {
        const long long r = a.d.rec(s, (long long)a.d.T - 1 - j);
        st<R, D>((R*)a.xs + r * D, p.e);
    }
};

// ---- joint log-density of a trajectory: log_likelihood + prior_logpdf (base.py:99-166) ----------------
struct LogpdfArgs {
    KDims d;
    Arr m0, P0, Fs, Qs, bs, Hs, Rs, cs, ys, xs;
    int nan_policy;  // 0 reference, 1 masked
};

template <typename R, int D, int P> AX_HD R body_joint_logpdf(const LogpdfArgs& a, int s, int t) {
    const int c = s / a.d.B, b = s % a.d.B;
    R x[D];
    ld<R, D>(at<R>(a.xs, c, t, b), x);
    R out = 0;
    {  // observation term
        R H[P * D], cv[P], y[P], res[P];
        ld<R, P * D>(at<R>(a.Hs, c, t, b), H);
        ld<R, P>(at<R>(a.cs, c, t, b), cv);
        ld<R, P>(at<R>(a.ys, c, t, b), y);
        bool skip[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            R pr = cv[k];
#pragma unroll
            for (int j = 0; j < D; ++j) pr += H[k * D + j] * x[j];
            res[k] = y[k] - pr;
            skip[k] = (a.nan_policy == 1) && !finite_(y[k]);
        }
        out += gauss_logpdf<R, P>(res, at<R>(a.Rs, c, t, b), a.nan_policy == 1 ? skip : nullptr);
    }
    {  // transition / initial term
        R res[D];
        if (t == 0) {
            R m0[D];
            ld<R, D>(at<R>(a.m0, c, 0, b), m0);
#pragma unroll
            for (int k = 0; k < D; ++k) res[k] = x[k] - m0[k];
            out += gauss_logpdf<R, D>(res, at<R>(a.P0, c, 0, b), nullptr);
        } else {
            R xp[D], F[D * D], bd[D], pr[D];
            ld<R, D>(at<R>(a.xs, c, t - 1, b), xp);
            ld<R, D * D>(at<R>(a.Fs, c, t - 1, b), F);
            ld<R, D>(at<R>(a.bs, c, t - 1, b), bd);
            mv<R, D, D>(F, xp, pr);
#pragma unroll
            for (int k = 0; k < D; ++k) res[k] = x[k] - (pr[k] + bd[k]);
            out += gauss_logpdf<R, D>(res, at<R>(a.Qs, c, t - 1, b), nullptr);
        }
    }
    return out;
}


// ---- all log-densities of one auxiliary-Kalman sweep of the LG_CONCAT device model in one pass ---------------------
// For the current state x and the proposal xp (kalman/generic.py:64-70):
//   out[0] += loglik_concat(xp) + prior(xp)   (posterior_logpdf + ell of the proposal, base.py:72-96)
//   out[1] += loglik_concat(x)  + prior(x)
//   out[2] += loglik_obs(xp)    + prior(xp)   (log_likelihood_fn(x_prop), generic.py:89)
//   out[3] += loglik_obs(x)     + prior(x)
//   out[4] += ((xp-u)^2 - (x-u)^2) / delta    (generic.py:103-105)
// loglik_concat uses R = blkdiag(delta/2 I, Robs): its Cholesky is block diagonal, so the term is the auxiliary block
// (diagonal) plus the observation block -- the same arithmetic as the dense 2d x 2d factorisation without the zeros.
struct SweepLogpdfArgs {
    KDims d;
    Arr m0, P0, Fs, Qs, bs, Hs, Rs, cs, ys;  // dynamics + REAL observation model, ys = yobs
    const void* x;                           // dense (C,T,D)
    const void* xp;
    const void* u;
    double delta;
    int nan_policy;
};

template <typename R, int D, int PO> AX_HD void body_sweep_logpdf(const SweepLogpdfArgs& a, int c, int t, R* out5) {
    const long long r = (long long)c * a.d.T + t;
    R x[D], xp[D], u[D];
    ld<R, D>((const R*)a.x + r * D, x);
    ld<R, D>((const R*)a.xp + r * D, xp);
    ld<R, D>((const R*)a.u + r * D, u);
    // observation block
    R ob_x, ob_p;
    bool badobs_x = false, badobs_p = false;
    {
        R H[PO * D], cv[PO], y[PO], r1[PO], r2[PO];
        bool skip[PO];
        ld<R, PO * D>(at<R>(a.Hs, c, t, 0), H);
        ld<R, PO>(at<R>(a.cs, c, t, 0), cv);
        ld<R, PO>(at<R>(a.ys, c, t, 0), y);
#pragma unroll
        for (int k = 0; k < PO; ++k) {
            R p1 = cv[k], p2 = cv[k];
#pragma unroll
            for (int j = 0; j < D; ++j) p1 += H[k * D + j] * xp[j], p2 += H[k * D + j] * x[j];
            r1[k] = y[k] - p1;
            r2[k] = y[k] - p2;
            skip[k] = (a.nan_policy == 1) && !finite_(y[k]);
            badobs_p = badobs_p || (!skip[k] && !finite_(r1[k]));
            badobs_x = badobs_x || (!skip[k] && !finite_(r2[k]));
        }
        gauss_logpdf2<R, PO>(r1, r2, at<R>(a.Rs, c, t, 0), a.nan_policy == 1 ? skip : nullptr, ob_p, ob_x);
    }
    // auxiliary block: log N(u; x, delta/2 I), and the MH correction
    R ax_x, ax_p, corr = 0;
    bool bad_aux_p, bad_aux_x;
    {
        const R hd = (R)(0.5 * a.delta);
        const R sd = sqrt_(hd);
        R q1 = 0, q2 = 0;
        bool b1 = false, b2 = false;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const R d1 = u[k] - xp[k], d2 = u[k] - x[k];
            b1 = b1 || !finite_(d1);
            b2 = b2 || !finite_(d2);
            const R z1 = d1 / sd, z2 = d2 / sd;
            q1 += z1 * z1;
            q2 += z2 * z2;
            const R e1 = xp[k] - u[k], e2 = x[k] - u[k];
            corr += (e1 * e1 - e2 * e2) / (R)a.delta;
        }
        const R cst = -(R)D * log_(sd) - (R)(0.5 * LOG_2PI) * (R)D;
        ax_p = (R)-0.5 * q1 + cst;
        ax_x = (R)-0.5 * q2 + cst;
        if (b1) ax_p = 0;
        if (b2) ax_x = 0;
        bad_aux_p = b1;
        bad_aux_x = b2;
    }
    // transition / initial term
    R pr_x, pr_p;
    {
        R r1[D], r2[D];
        if (t == 0) {
            R m0[D];
            ld<R, D>(at<R>(a.m0, c, 0, 0), m0);
#pragma unroll
            for (int k = 0; k < D; ++k) r1[k] = xp[k] - m0[k], r2[k] = x[k] - m0[k];
            gauss_logpdf2<R, D>(r1, r2, at<R>(a.P0, c, 0, 0), nullptr, pr_p, pr_x);
        } else {
            R xq[D], xpq[D], F[D * D], bd[D], m1[D], m2[D];
            ld<R, D>((const R*)a.x + (r - 1) * D, xq);
            ld<R, D>((const R*)a.xp + (r - 1) * D, xpq);
            ld<R, D * D>(at<R>(a.Fs, c, t - 1, 0), F);
            ld<R, D>(at<R>(a.bs, c, t - 1, 0), bd);
            mv<R, D, D>(F, xpq, m1);
            mv<R, D, D>(F, xq, m2);
#pragma unroll
            for (int k = 0; k < D; ++k) r1[k] = xp[k] - (m1[k] + bd[k]), r2[k] = x[k] - (m2[k] + bd[k]);
            gauss_logpdf2<R, D>(r1, r2, at<R>(a.Qs, c, t - 1, 0), nullptr, pr_p, pr_x);
        }
    }
    // reference policy (jnp.nansum over per-step logpdfs): a non-finite component anywhere in the stacked residual
    // [u - x ; y - H x - c] drops the whole step of the concatenated model; the target only sees the observation block.
    const bool ref = a.nan_policy == 0;
    const R cc_p = (ref && (bad_aux_p || badobs_p)) ? (R)0 : ax_p + ob_p;
    const R cc_x = (ref && (bad_aux_x || badobs_x)) ? (R)0 : ax_x + ob_x;
    out5[0] = cc_p + pr_p;
    out5[1] = cc_x + pr_x;
    out5[2] = ob_p + pr_p;
    out5[3] = ob_x + pr_x;
    out5[4] = corr;
}

}  // namespace ax

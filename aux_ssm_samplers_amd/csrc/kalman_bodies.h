// kalman_bodies.h -- per-lane bodies of the Kalman kernels and the two scan operators.
//
// Lanes of a wave work on 64 CONSECUTIVE time steps of one sequence (s = chain c * B + batch b).  Every global read
// goes through an I/O policy:
//   * DirectIO   -- each lane reads its own record (host single-stepper in tests/hostsim, and the fallback for strides
//                   that are not record-dense);
//   * WaveIO     -- (kernels.hip.h) the wave copies the 64 records it needs as one contiguous, fully coalesced stream
//                   into LDS and each lane then picks its record from LDS.  Per-lane record walks are what made the first
//                   version of these kernels TA-bound (TA_BUSY 88-95 %, profiles/r01_b_pmc_*): a scattered lane access
//                   costs the address unit ~8 B/clk/CU whatever its width.
// A body is entered by ALL lanes of the wave (`valid` = this lane's index is in range) so that cooperative loads are never
// executed under divergence; stores and arithmetic are predicated on `valid`.
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

// ---- scan-element layout ------------------------------------------------------------------------------------------------
// The scan kernels give one lane a CHUNK of E consecutive elements; in iteration k a wave needs element k of 64 consecutive
// chunks.  Elements are therefore stored lane-interleaved: element i = (chunk, k) = (i / E, i % E), chunk = 64 g + l, lives at
// record (g E + k) W + l of its sequence (W = 64, or the chunk count if smaller), so those W records are one contiguous run.
struct ScanLayout {
    int E, nchunk, ngrp, W;  // W = chunks interleaved per row = min(64, nchunk); ngrp = ceil(nchunk / W)
    AX_HD long long seq_records() const { return (long long)ngrp * E * W; }
    AX_HD long long pos(int i) const {
        const int ch = i / E, k = i - ch * E;
        const int g = ch / W, l = ch - g * W;
        return ((long long)g * E + k) * W + l;
    }
    AX_HD long long row(int grp, int k) const { return ((long long)grp * E + k) * W; }
};

// Two-phase protocol so that the global-memory latency of all the arrays a body needs overlaps:
//   fetch<N>(lane_ptr, lane_stride, valid, buf)  -- issue the global reads into buf (no barrier);
//   finish<N>(lane_stride, valid, buf)           -- turn buf into this lane's record (WaveIO: transposition through LDS).
// lane_ptr = this lane's record; lane_stride = distance in reals between consecutive lanes' records.
struct DirectIO {
    template <typename R, int N> AX_HD void fetch(const R* lane_ptr, long long /*lane_stride*/, bool valid, R* buf) const {
        if (valid) {
            ld<R, N>(lane_ptr, buf);
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) buf[i] = 0;
        }
    }
    template <typename R, int N> AX_HD void finish(long long /*lane_stride*/, bool /*valid*/, R* /*buf*/) const {}
};

struct FilterArgs {
    KDims d;
    Arr m0, P0, Fs, Qs, bs, Hs, Rs, cs, ys;
    void* ms;       // dense (C,T,B,D)
    void* Ps;       // dense (C,T,B,D,D)
    void* elem;     // [S][lay.seq_records()][FiltElem::NPAD]
    void* ell0;     // [S]
    ScanLayout lay;
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
template <typename R, int D, int P, class IO>
AX_HD void body_filter_init(const FilterArgs& a, IO& io, int s, int i, bool valid) {
    const int c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    R F[D * D], bd[D], m_[D], P_[D * D], H[P * D], cv[P], y[P], Rm[P * P];
    io.template fetch<R, D * D>(at<R>(a.Fs, c, i, b), a.Fs.st, valid, F);
    io.template fetch<R, D>(at<R>(a.bs, c, i, b), a.bs.st, valid, bd);
    io.template fetch<R, D * D>(at<R>(a.Qs, c, i, b), a.Qs.st, valid, P_);
    io.template fetch<R, P * D>(at<R>(a.Hs, c, t, b), a.Hs.st, valid, H);
    io.template fetch<R, P>(at<R>(a.cs, c, t, b), a.cs.st, valid, cv);
    io.template fetch<R, P>(at<R>(a.ys, c, t, b), a.ys.st, valid, y);
    io.template fetch<R, P * P>(at<R>(a.Rs, c, t, b), a.Rs.st, valid, Rm);
    io.template finish<R, D * D>(a.Fs.st, valid, F);
    io.template finish<R, D>(a.bs.st, valid, bd);
    io.template finish<R, D * D>(a.Qs.st, valid, P_);
    io.template finish<R, P * D>(a.Hs.st, valid, H);
    io.template finish<R, P>(a.cs.st, valid, cv);
    io.template finish<R, P>(a.ys.st, valid, y);
    io.template finish<R, P * P>(a.Rs.st, valid, Rm);
    if (!valid) return;
    if (i == 0) {
        // first transition: built around predict(m0+, P0+)  (m_ = F m + b, P_ = F P F^T + Q, not symmetrised: filtering.py:200-201)
        const long long r = a.d.rec(s, 0);
        R m0p[D], P0p[D * D], tm[D], FP[D * D], Pn[D * D];
        ld<R, D>((const R*)a.ms + r * D, m0p);
        ld<R, D * D>((const R*)a.Ps + r * D * D, P0p);
        mv<R, D, D>(F, m0p, tm);
        mm<R, D, D, D>(F, P0p, FP);
        mmt<R, D, D, D>(FP, F, Pn);
#pragma unroll
        for (int k = 0; k < D; ++k) m_[k] = tm[k] + bd[k];
#pragma unroll
        for (int k = 0; k < D * D; ++k) P_[k] = Pn[k] + P_[k];
    } else {
        // (m, P) = (0, 0): m_ = b, P_ = Q exactly (filtering.py:190-191)
#pragma unroll
        for (int k = 0; k < D; ++k) m_[k] = bd[k];
    }
    FiltElem<R, D> e;
    filter_elem<R, D, P>(F, bd, m_, P_, H, cv, Rm, y, e);
    fe_store<R, D>((R*)a.elem + ((long long)s * a.lay.seq_records() + a.lay.pos(i)) * FiltElem<R, D>::NPAD, e);
}

// ---- log-likelihood increment of step i+1 from the filtered moments at i (filtering.py:60) ---------
template <typename R, int D, int P, class IO>
AX_HD R body_filter_ell(const FilterArgs& a, IO& io, int s, int i, bool valid) {
    const int c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    const long long r = a.d.rec(s, i);
    const long long dstride = a.d.B;  // dense arrays: consecutive time steps are B records apart
    R m[D], Pd[D * D], F[D * D], bd[D], Q[D * D], H[P * D], cv[P], y[P], Rm[P * P];
    io.template fetch<R, D>((const R*)a.ms + r * D, dstride * D, valid, m);
    io.template fetch<R, D * D>((const R*)a.Ps + r * D * D, dstride * D * D, valid, Pd);
    io.template fetch<R, D * D>(at<R>(a.Fs, c, i, b), a.Fs.st, valid, F);
    io.template fetch<R, D>(at<R>(a.bs, c, i, b), a.bs.st, valid, bd);
    io.template fetch<R, D * D>(at<R>(a.Qs, c, i, b), a.Qs.st, valid, Q);
    io.template fetch<R, P * D>(at<R>(a.Hs, c, t, b), a.Hs.st, valid, H);
    io.template fetch<R, P>(at<R>(a.cs, c, t, b), a.cs.st, valid, cv);
    io.template fetch<R, P>(at<R>(a.ys, c, t, b), a.ys.st, valid, y);
    io.template fetch<R, P * P>(at<R>(a.Rs, c, t, b), a.Rs.st, valid, Rm);
    io.template finish<R, D>(dstride * D, valid, m);
    io.template finish<R, D * D>(dstride * D * D, valid, Pd);
    io.template finish<R, D * D>(a.Fs.st, valid, F);
    io.template finish<R, D>(a.bs.st, valid, bd);
    io.template finish<R, D * D>(a.Qs.st, valid, Q);
    io.template finish<R, P * D>(a.Hs.st, valid, H);
    io.template finish<R, P>(a.cs.st, valid, cv);
    io.template finish<R, P>(a.ys.st, valid, y);
    io.template finish<R, P * P>(a.Rs.st, valid, Rm);
    if (!valid) return (R)0;
    kalman_predict<R, D>(m, Pd, F, bd, Q);
    return kalman_ell_inc<R, D, P>(m, Pd, H, cv, Rm, y);
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
    static AX_HD const ScanLayout& layout(const Args& a) { return a.lay; }
    // 64 consecutive records of row (grp, k): record of lane l at base + l * NPAD
    static AX_HD const R* row_ptr(const Args& a, int s, int grp, int k) {
        return (const R*)a.elem + ((long long)s * a.lay.seq_records() + a.lay.row(grp, k)) * Full::NPAD;
    }
    static AX_HD void unpack(const R* t, Full& e) {
#pragma unroll
        for (int i = 0; i < D * D; ++i) e.A[i] = t[i];
#pragma unroll
        for (int i = 0; i < D; ++i) e.b[i] = t[D * D + i], e.eta[i] = t[D * D + D + DS + i];
#pragma unroll
        for (int i = 0; i < DS; ++i) e.C[i] = t[D * D + D + i], e.J[i] = t[D * D + 2 * D + DS + i];
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
        R t[D + DS];
#pragma unroll
        for (int i = 0; i < D; ++i) t[i] = p.b[i];
#pragma unroll
        for (int i = 0; i < DS; ++i) t[D + i] = p.C[i];
        stv<R, D + DS>(q, t);
    }
    static AX_HD void load_pre(const R* q, Pre& p) {
        R t[D + DS];
        ldv<R, D + DS>(q, t);
#pragma unroll
        for (int i = 0; i < D; ++i) p.b[i] = t[i];
#pragma unroll
        for (int i = 0; i < DS; ++i) p.C[i] = t[D + i];
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
    void* elem;       // [S][lay.seq_records()][SampElem::NPAD], scan position j = T-1-t
    ScanLayout lay;
};

template <typename R_, int D> struct SampleOp;

// scan element j = jp + 1 <-> time t = T-2-jp, jp in [0, T-1)   (_sampling_init_one, sampling.py:108-112)
// lanes walk DOWN in time, hence the negative lane strides.
template <typename R, int D, class IO> AX_HD void body_sample_init(const SampleArgs& a, IO& io, int s, int jp, bool valid) {
    const int c = s / a.d.B, b = s % a.d.B;
    const int T = a.d.T;
    const long long t = (long long)T - 2 - jp;
    const long long r = a.d.rec(s, t);
    const long long ds = a.d.B;
    R m[D], Pd[D * D], eps[D], F[D * D], Q[D * D], bd[D];
    io.template fetch<R, D>((const R*)a.ms + r * D, -ds * D, valid, m);
    io.template fetch<R, D * D>((const R*)a.Ps + r * D * D, -ds * D * D, valid, Pd);
    io.template fetch<R, D>((const R*)a.eps + r * D, -ds * D, valid, eps);
    io.template fetch<R, D * D>(at<R>(a.Fs, c, t, b), -a.Fs.st, valid, F);
    io.template fetch<R, D * D>(at<R>(a.Qs, c, t, b), -a.Qs.st, valid, Q);
    io.template fetch<R, D>(at<R>(a.bs, c, t, b), -a.bs.st, valid, bd);
    io.template finish<R, D>(-ds * D, valid, m);
    io.template finish<R, D * D>(-ds * D * D, valid, Pd);
    io.template finish<R, D>(-ds * D, valid, eps);
    io.template finish<R, D * D>(-a.Fs.st, valid, F);
    io.template finish<R, D * D>(-a.Qs.st, valid, Q);
    io.template finish<R, D>(-a.bs.st, valid, bd);
    if (!valid) return;
    SampElem<R, D> e;
    sample_elem<R, D>(F, Q, bd, m, Pd, eps, e);
    SampleOp<R, D>::store_rec((R*)a.elem + ((long long)s * a.lay.seq_records() + a.lay.pos(jp + 1)) * SampElem<R, D>::NPAD, e);
}
// scan element 0 <-> t = T-1   (_sample_last_step, sampling.py:115-124)
template <typename R, int D> AX_HD void body_sample_last(const SampleArgs& a, int s) {
    const long long r = a.d.rec(s, (long long)a.d.T - 1);
    R m[D], Pd[D * D], eps[D];
    ld<R, D>((const R*)a.ms + r * D, m);
    ld<R, D * D>((const R*)a.Ps + r * D * D, Pd);
    ld<R, D>((const R*)a.eps + r * D, eps);
    SampElem<R, D> e;
    sample_last<R, D>(m, Pd, eps, e);
    SampleOp<R, D>::store_rec((R*)a.elem + ((long long)s * a.lay.seq_records() + a.lay.pos(0)) * SampElem<R, D>::NPAD, e);
}

template <typename R_, int D> struct SampleOp {
    using R = R_;
    using Full = SampElem<R, D>;
    using Pre = SampPre<R, D>;
    using Args = SampleArgs;
    static AX_HD int length(const Args& a) { return a.d.T; }
    static AX_HD const ScanLayout& layout(const Args& a) { return a.lay; }
    static AX_HD const R* row_ptr(const Args& a, int s, int grp, int k) {
        return (const R*)a.elem + ((long long)s * a.lay.seq_records() + a.lay.row(grp, k)) * Full::NPAD;
    }
    static AX_HD void unpack(const R* t, Full& e) {
#pragma unroll
        for (int i = 0; i < D * D; ++i) e.G[i] = t[i];
#pragma unroll
        for (int i = 0; i < D; ++i) e.e[i] = t[D * D + i];
    }
    static AX_HD void load_rec(const R* p, Full& e) {
        R t[D * D + D];
        ldv<R, D * D + D>(p, t);
        unpack(t, e);
    }
    static AX_HD void store_rec(R* p, const Full& e) {
        R t[D * D + D];
#pragma unroll
        for (int i = 0; i < D * D; ++i) t[i] = e.G[i];
#pragma unroll
        for (int i = 0; i < D; ++i) t[D * D + i] = e.e[i];
        stv<R, D * D + D>(p, t);
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
    static AX_HD void store_pre(R* q, const Pre& p) { stv<R, D>(q, p.e); }
    static AX_HD void load_pre(const R* q, Pre& p) { ldv<R, D>(q, p.e); }
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

// observation term at time t + transition term into t (t >= 1); lanes are indexed by i = t - 1
template <typename R, int D, int P, class IO> AX_HD R body_joint_logpdf(const LogpdfArgs& a, IO& io, int s, int i, bool valid) {
    const int c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    R x[D], xp[D], H[P * D], cv[P], y[P], Rm[P * P], F[D * D], bd[D], Q[D * D];
    io.template fetch<R, D>(at<R>(a.xs, c, t, b), a.xs.st, valid, x);
    io.template fetch<R, D>(at<R>(a.xs, c, i, b), a.xs.st, valid, xp);
    io.template fetch<R, P * D>(at<R>(a.Hs, c, t, b), a.Hs.st, valid, H);
    io.template fetch<R, P>(at<R>(a.cs, c, t, b), a.cs.st, valid, cv);
    io.template fetch<R, P>(at<R>(a.ys, c, t, b), a.ys.st, valid, y);
    io.template fetch<R, P * P>(at<R>(a.Rs, c, t, b), a.Rs.st, valid, Rm);
    io.template fetch<R, D * D>(at<R>(a.Fs, c, i, b), a.Fs.st, valid, F);
    io.template fetch<R, D>(at<R>(a.bs, c, i, b), a.bs.st, valid, bd);
    io.template fetch<R, D * D>(at<R>(a.Qs, c, i, b), a.Qs.st, valid, Q);
    io.template finish<R, D>(a.xs.st, valid, x);
    io.template finish<R, D>(a.xs.st, valid, xp);
    io.template finish<R, P * D>(a.Hs.st, valid, H);
    io.template finish<R, P>(a.cs.st, valid, cv);
    io.template finish<R, P>(a.ys.st, valid, y);
    io.template finish<R, P * P>(a.Rs.st, valid, Rm);
    io.template finish<R, D * D>(a.Fs.st, valid, F);
    io.template finish<R, D>(a.bs.st, valid, bd);
    io.template finish<R, D * D>(a.Qs.st, valid, Q);
    if (!valid) return (R)0;
    R out = 0;
    {
        R res[P];
        bool skip[P];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            R pr = cv[k];
#pragma unroll
            for (int j = 0; j < D; ++j) pr += H[k * D + j] * x[j];
            res[k] = y[k] - pr;
            skip[k] = (a.nan_policy == 1) && !finite_(y[k]);
        }
        out += gauss_logpdf<R, P>(res, Rm, a.nan_policy == 1 ? skip : nullptr);
    }
    {
        R res[D], pr[D];
        mv<R, D, D>(F, xp, pr);
#pragma unroll
        for (int k = 0; k < D; ++k) res[k] = x[k] - (pr[k] + bd[k]);
        out += gauss_logpdf<R, D>(res, Q, nullptr);
    }
    return out;
}
// t = 0: observation term + initial-state term (one lane per sequence)
template <typename R, int D, int P> AX_HD R body_joint_logpdf_head(const LogpdfArgs& a, int s) {
    const int c = s / a.d.B, b = s % a.d.B;
    R x[D], H[P * D], cv[P], y[P], m0[D];
    ld<R, D>(at<R>(a.xs, c, 0, b), x);
    ld<R, P * D>(at<R>(a.Hs, c, 0, b), H);
    ld<R, P>(at<R>(a.cs, c, 0, b), cv);
    ld<R, P>(at<R>(a.ys, c, 0, b), y);
    ld<R, D>(at<R>(a.m0, c, 0, b), m0);
    R out = 0;
    R res[P];
    bool skip[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        R pr = cv[k];
#pragma unroll
        for (int j = 0; j < D; ++j) pr += H[k * D + j] * x[j];
        res[k] = y[k] - pr;
        skip[k] = (a.nan_policy == 1) && !finite_(y[k]);
    }
    out += gauss_logpdf<R, P>(res, at<R>(a.Rs, c, 0, b), a.nan_policy == 1 ? skip : nullptr);
    R r0[D];
#pragma unroll
    for (int k = 0; k < D; ++k) r0[k] = x[k] - m0[k];
    out += gauss_logpdf<R, D>(r0, at<R>(a.P0, c, 0, b), nullptr);
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

// the observation + auxiliary blocks at one time step for (xp, x); returns via references
template <typename R, int D, int PO>
AX_HD void sweep_obs_terms(const SweepLogpdfArgs& a, const R* x, const R* xp, const R* u, const R* H, const R* cv, const R* y,
                           const R* Rm, R& cc_p, R& cc_x, R& ob_p, R& ob_x, R& corr) {
    bool badobs_x = false, badobs_p = false;
    {
        R r1[PO], r2[PO];
        bool skip[PO];
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
        gauss_logpdf2<R, PO>(r1, r2, Rm, a.nan_policy == 1 ? skip : nullptr, ob_p, ob_x);
    }
    R ax_x, ax_p;
    bool b1 = false, b2 = false;
    corr = 0;
    {
        const R hd = (R)(0.5 * a.delta);
        const R sd = sqrt_(hd);
        R q1 = 0, q2 = 0;
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
        ax_p = b1 ? (R)0 : (R)-0.5 * q1 + cst;
        ax_x = b2 ? (R)0 : (R)-0.5 * q2 + cst;
    }
    // reference policy (jnp.nansum over per-step logpdfs): a non-finite component anywhere in the stacked residual
    // [u - x ; y - H x - c] drops the whole step of the concatenated model; the target only sees the observation block.
    const bool ref = a.nan_policy == 0;
    cc_p = (ref && (b1 || badobs_p)) ? (R)0 : ax_p + ob_p;
    cc_x = (ref && (b2 || badobs_x)) ? (R)0 : ax_x + ob_x;
}

// lanes indexed by i = t - 1 (t >= 1)
template <typename R, int D, int PO, class IO>
AX_HD void body_sweep_logpdf(const SweepLogpdfArgs& a, IO& io, int c, int i, bool valid, R* out5) {
    const long long t = (long long)i + 1;
    const long long r = (long long)c * a.d.T + t;
    R x[D], xp[D], u[D], xq[D], xpq[D], H[PO * D], cv[PO], y[PO], Rm[PO * PO], F[D * D], bd[D], Q[D * D];
    io.template fetch<R, D>((const R*)a.x + r * D, D, valid, x);
    io.template fetch<R, D>((const R*)a.xp + r * D, D, valid, xp);
    io.template fetch<R, D>((const R*)a.u + r * D, D, valid, u);
    io.template fetch<R, D>((const R*)a.x + (r - 1) * D, D, valid, xq);
    io.template fetch<R, D>((const R*)a.xp + (r - 1) * D, D, valid, xpq);
    io.template fetch<R, PO * D>(at<R>(a.Hs, c, t, 0), a.Hs.st, valid, H);
    io.template fetch<R, PO>(at<R>(a.cs, c, t, 0), a.cs.st, valid, cv);
    io.template fetch<R, PO>(at<R>(a.ys, c, t, 0), a.ys.st, valid, y);
    io.template fetch<R, PO * PO>(at<R>(a.Rs, c, t, 0), a.Rs.st, valid, Rm);
    io.template fetch<R, D * D>(at<R>(a.Fs, c, i, 0), a.Fs.st, valid, F);
    io.template fetch<R, D>(at<R>(a.bs, c, i, 0), a.bs.st, valid, bd);
    io.template fetch<R, D * D>(at<R>(a.Qs, c, i, 0), a.Qs.st, valid, Q);
    io.template finish<R, D>(D, valid, x);
    io.template finish<R, D>(D, valid, xp);
    io.template finish<R, D>(D, valid, u);
    io.template finish<R, D>(D, valid, xq);
    io.template finish<R, D>(D, valid, xpq);
    io.template finish<R, PO * D>(a.Hs.st, valid, H);
    io.template finish<R, PO>(a.cs.st, valid, cv);
    io.template finish<R, PO>(a.ys.st, valid, y);
    io.template finish<R, PO * PO>(a.Rs.st, valid, Rm);
    io.template finish<R, D * D>(a.Fs.st, valid, F);
    io.template finish<R, D>(a.bs.st, valid, bd);
    io.template finish<R, D * D>(a.Qs.st, valid, Q);
#pragma unroll
    for (int k = 0; k < 5; ++k) out5[k] = 0;
    if (!valid) return;
    R cc_p, cc_x, ob_p, ob_x, corr;
    sweep_obs_terms<R, D, PO>(a, x, xp, u, H, cv, y, Rm, cc_p, cc_x, ob_p, ob_x, corr);
    R pr_p, pr_x;
    {
        R r1[D], r2[D], m1[D], m2[D];
        mv<R, D, D>(F, xpq, m1);
        mv<R, D, D>(F, xq, m2);
#pragma unroll
        for (int k = 0; k < D; ++k) r1[k] = xp[k] - (m1[k] + bd[k]), r2[k] = x[k] - (m2[k] + bd[k]);
        gauss_logpdf2<R, D>(r1, r2, Q, nullptr, pr_p, pr_x);
    }
    out5[0] = cc_p + pr_p;
    out5[1] = cc_x + pr_x;
    out5[2] = ob_p + pr_p;
    out5[3] = ob_x + pr_x;
    out5[4] = corr;
}
// t = 0 terms (one lane per chain)
template <typename R, int D, int PO> AX_HD void body_sweep_logpdf_head(const SweepLogpdfArgs& a, int c, R* out5) {
    const long long r = (long long)c * a.d.T;
    R x[D], xp[D], u[D], H[PO * D], cv[PO], y[PO], m0[D];
    ld<R, D>((const R*)a.x + r * D, x);
    ld<R, D>((const R*)a.xp + r * D, xp);
    ld<R, D>((const R*)a.u + r * D, u);
    ld<R, PO * D>(at<R>(a.Hs, c, 0, 0), H);
    ld<R, PO>(at<R>(a.cs, c, 0, 0), cv);
    ld<R, PO>(at<R>(a.ys, c, 0, 0), y);
    ld<R, D>(at<R>(a.m0, c, 0, 0), m0);
    R cc_p, cc_x, ob_p, ob_x, corr;
    sweep_obs_terms<R, D, PO>(a, x, xp, u, H, cv, y, at<R>(a.Rs, c, 0, 0), cc_p, cc_x, ob_p, ob_x, corr);
    R r1[D], r2[D], pr_p, pr_x;
#pragma unroll
    for (int k = 0; k < D; ++k) r1[k] = xp[k] - m0[k], r2[k] = x[k] - m0[k];
    gauss_logpdf2<R, D>(r1, r2, at<R>(a.P0, c, 0, 0), nullptr, pr_p, pr_x);
    out5[0] = cc_p + pr_p;
    out5[1] = cc_x + pr_x;
    out5[2] = ob_p + pr_p;
    out5[3] = ob_x + pr_x;
    out5[4] = corr;
}

}  // namespace ax

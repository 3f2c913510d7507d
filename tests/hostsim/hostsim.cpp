// hostsim.cpp -- TEST INFRASTRUCTURE ONLY.  Compiles the product's per-lane kernel bodies
// (aux_ssm_samplers_amd/csrc/kalman_bodies.h, AX_HD) for the host and runs them in plain loops, with the
// same chunked three-pass scan structure as kernels.hip.h (chunk reduce -> aggregate scan -> cheap re-walk),
// so that the math and the indexing of the HIP path can be checked against the oracle in the build container,
// which has no GPU.  It is never loaded by the product package: the product fails loudly without libauxssm.so.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../aux_ssm_samplers_amd/csrc/kalman_bodies.h"

using namespace ax;

struct HsArr { const void* ptr; long long sc, st, sb; };
static Arr cv(const HsArr& a) { return Arr{a.ptr, a.sc, a.st, a.sb, 1}; }

static ScanLayout make_layout_host(int n, int E, int cm, int S) {
    if (E <= 0 || E > n) E = n > 0 ? n : 1;
    const int nchunk = n > 0 ? (n + E - 1) / E : 1;
    const int W = nchunk < 64 ? nchunk : 64;
    return ScanLayout{E, nchunk, (nchunk + W - 1) / W, W, cm, S};
}
// chain-minor <-> dense copies of a (C,T,B,rec) buffer (host test helper)
template <typename R> static void cm_to_dense(const std::vector<R>& cmb, const KDims& d, int rec, R* dense) {
    const Arr a = cm_arr(cmb.data(), d, rec), o = dense_arr(dense, d, rec);
    for (int c = 0; c < d.C; ++c)
        for (int t = 0; t < d.T; ++t)
            for (int b = 0; b < d.B; ++b)
                for (int e = 0; e < rec; ++e) const_cast<R*>(at<R>(o, c, t, b))[e * o.se] = at<R>(a, c, t, b)[e * a.se];
}
template <typename R> static void dense_to_cm(const R* dense, const KDims& d, int rec, std::vector<R>& cmb) {
    const Arr a = cm_arr(cmb.data(), d, rec), o = dense_arr(dense, d, rec);
    for (int c = 0; c < d.C; ++c)
        for (int t = 0; t < d.T; ++t)
            for (int b = 0; b < d.B; ++b)
                for (int e = 0; e < rec; ++e) const_cast<R*>(at<R>(a, c, t, b))[e * a.se] = at<R>(o, c, t, b)[e * o.se];
}

template <class Op> static void scan_host(typename Op::Args& a, int S, int n) {
    using R = typename Op::R;
    using Full = typename Op::Full;
    using Pre = typename Op::Pre;
    if (n <= 0) return;
    const ScanLayout lay = Op::layout(a);
    const int E = lay.E, nchunk = lay.nchunk;
    auto load = [&](int s, int i, Full& e) { Op::load_elem(a, s, i, e); };
    std::vector<Full> agg((size_t)S * nchunk);
    std::vector<Pre> pre((size_t)S * nchunk);
    for (int s = 0; s < S; ++s)
        for (int ch = 0; ch < nchunk; ++ch) {
            const int i0 = ch * E, i1 = std::min(n, i0 + E);
            Full acc;
            load(s, i0, acc);
            for (int i = i0 + 1; i < i1; ++i) {
                Full e, o;
                load(s, i, e);
                Op::combine(acc, e, o);
                acc = o;
            }
            agg[(size_t)s * nchunk + ch] = acc;
        }
    for (int s = 0; s < S; ++s) {
        Full ex;
        Op::identity(ex);
        for (int ch = 0; ch < nchunk; ++ch) {
            Op::to_pre(ex, pre[(size_t)s * nchunk + ch]);
            Full o;
            Op::combine(ex, agg[(size_t)s * nchunk + ch], o);
            ex = o;
        }
    }
    for (int s = 0; s < S; ++s)
        for (int ch = 0; ch < nchunk; ++ch) {
            const int i0 = ch * E, i1 = std::min(n, i0 + E);
            Pre p = pre[(size_t)s * nchunk + ch];
            for (int i = i0; i < i1; ++i) {
                Full e;
                Pre o;
                load(s, i, e);
                Op::apply(p, e, o);
                p = o;
                Op::write_out(a, s, i, p);
            }
        }
}

template <typename R, int D, int P>
static int filter_T(int C, int T, int B, const HsArr* g, const HsArr* ys, int E, void* ms, void* Ps, void* ell, int pblk) {
    FilterArgs a;
    a.d = KDims{C, T, B};
    a.m0 = cv(g[0]); a.P0 = cv(g[1]); a.Fs = cv(g[2]); a.Qs = cv(g[3]); a.bs = cv(g[4]);
    a.Hs = cv(g[5]); a.Rs = cv(g[6]); a.cs = cv(g[7]); a.ys = cv(*ys);
    const int S = C * B, n = T - 1;
    const int cm = E < 0 ? 1 : 0;  // E < 0: chain-minor internal buffers with chunk |E|
    if (cm) E = -E;
    a.lay = make_layout_host(n, E, cm, S);
    std::vector<R> msc, Psc;
    if (cm) {
        msc.resize((size_t)S * T * D);
        Psc.resize((size_t)S * T * D * D);
        a.ms = cm_arr(msc.data(), a.d, D);
        a.Ps = cm_arr(Psc.data(), a.d, D * D);
    } else {
        a.ms = dense_arr(ms, a.d, D);
        a.Ps = dense_arr(Ps, a.d, D * D);
    }
    std::vector<R> elem((size_t)a.lay.total_reals(n, S, FiltElem<R, D>::NPAD) + 16), ell0(S), ellz(S, (R)0);
    a.elem = elem.data();
    a.ell0 = ell0.data();
    a.ellz = ellz.data();
    DirectIO io;
    for (int s = 0; s < S; ++s) body_filter_t0<R, D, P>(a, s);
    a.pblk = pblk;
    constexpr int P1 = (P > D) ? D : 0;
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < n; ++i) {
            if (pblk == D && P1 > 0) body_filter_init<R, D, P, DirectIO, P1>(a, io, s, i, true);
            else body_filter_init<R, D, P>(a, io, s, i, true);
        }
    scan_host<FilterOp<R, D>>(a, S, n);
    for (int c = 0; c < C; ++c) {
        R tot = 0;
        for (int b = 0; b < B; ++b) {
            const int s = c * B + b;
            tot += ell0[s] + (n > 0 ? ellz[s] : (R)0);  // the scan's log-scale is the marginal log-likelihood of t = 1..T-1
        }
        ((R*)ell)[c] = tot;
    }
    if (cm) {
        cm_to_dense<R>(msc, a.d, D, (R*)ms);
        cm_to_dense<R>(Psc, a.d, D * D, (R*)Ps);
    }
    return 0;
}

template <typename R, int D>
static int sample_T(int C, int T, int B, const HsArr* g, const void* ms, const void* Ps, const void* eps, int E, void* xs) {
    SampleArgs a;
    a.d = KDims{C, T, B};
    a.Fs = cv(g[2]); a.Qs = cv(g[3]); a.bs = cv(g[4]);
    const int S = C * B;
    const int cm = E < 0 ? 1 : 0;
    if (cm) E = -E;
    a.lay = make_layout_host(T, E, cm, S);
    std::vector<R> msc, Psc, xsc;
    if (cm) {
        msc.resize((size_t)S * T * D);
        Psc.resize((size_t)S * T * D * D);
        xsc.resize((size_t)S * T * D);
        dense_to_cm<R>((const R*)ms, a.d, D, msc);
        dense_to_cm<R>((const R*)Ps, a.d, D * D, Psc);
        a.ms = cm_arr(msc.data(), a.d, D);
        a.Ps = cm_arr(Psc.data(), a.d, D * D);
        a.xs = cm_arr(xsc.data(), a.d, D);
    } else {
        a.ms = dense_arr(ms, a.d, D);
        a.Ps = dense_arr(Ps, a.d, D * D);
        a.xs = dense_arr(xs, a.d, D);
    }
    a.eps = dense_arr(eps, a.d, D);
    std::vector<R> elem((size_t)a.lay.total_reals(T, S, SampElem<R, D>::NPAD) + 16);
    a.elem = elem.data();
    DirectIO io;
    for (int s = 0; s < S; ++s) {
        body_sample_last<R, D>(a, s);
        for (int jp = 0; jp < T - 1; ++jp) body_sample_init<R, D>(a, io, s, jp, true);
    }
    scan_host<SampleOp<R, D>>(a, S, T);
    if (cm) cm_to_dense<R>(xsc, a.d, D, (R*)xs);
    return 0;
}

template <typename R, int D, int P>
static int logpdf_T(int C, int T, int B, const HsArr* g, const HsArr* ys, const HsArr* xs, int pol, void* out) {
    LogpdfArgs a;
    a.d = KDims{C, T, B};
    a.m0 = cv(g[0]); a.P0 = cv(g[1]); a.Fs = cv(g[2]); a.Qs = cv(g[3]); a.bs = cv(g[4]);
    a.Hs = cv(g[5]); a.Rs = cv(g[6]); a.cs = cv(g[7]); a.ys = cv(*ys); a.xs = cv(*xs);
    a.nan_policy = pol;
    DirectIO io;
    for (int c = 0; c < C; ++c) {
        R tot = 0;
        for (int b = 0; b < B; ++b) {
            tot += body_joint_logpdf_head<R, D, P>(a, c * B + b);
            for (int i = 0; i < T - 1; ++i) tot += body_joint_logpdf<R, D, P>(a, io, c * B + b, i, true);
        }
        ((R*)out)[c] = tot;
    }
    return 0;
}

// the SV sweep's fused log-density pass (kalman_bodies.h::body_sv_logpdf), dense or chain-minor views of the same dense inputs
template <typename R, int D>
static int sv_logpdf_T(int C, int T, const HsArr* g, const void* yobs, const void* x, const void* xp, const void* u, const void* ys1,
                       const void* ys2, const void* R1, const void* R2, double delta, int cm, void* out) {
    SvLogpdfArgs a;
    a.d = KDims{C, T, 1};
    a.m0 = cv(g[0]); a.P0 = cv(g[1]); a.Fs = cv(g[2]); a.Qs = cv(g[3]); a.bs = cv(g[4]);
    a.yobs = Arr{yobs, 0, D, 0, 1};
    a.delta = delta;
    std::vector<std::vector<R>> keep;
    auto view = [&](const void* p, int rec) -> Arr {
        if (!p) return Arr{nullptr, 0, 0, 0, 1};
        if (!cm) return dense_arr(p, a.d, rec);
        keep.emplace_back((size_t)C * T * rec);
        dense_to_cm<R>((const R*)p, a.d, rec, keep.back());
        return cm_arr(keep.back().data(), a.d, rec);
    };
    a.x = view(x, D); a.xp = view(xp, D); a.u = view(u, D); a.ys1 = view(ys1, D); a.ys2 = view(ys2, D);
    a.R1 = view(R1, D * D); a.R2 = view(R2, D * D);
    for (int c = 0; c < C; ++c) {   // (as k_sv_logpdf_cm: the determinants multiplied up over the chain's steps, one logarithm per sum)
        R tot[5], f[4];
        LogProd<R> lp[4];
        body_sv_logpdf_head<R, D>(a, c, tot, f);
        for (int k = 0; k < 4; ++k) lp[k].mul(f[k]);
        for (int i = 0; i < T - 1; ++i) {
            R w[5];
            body_sv_logpdf<R, D>(a, c, i, true, w, f);
            for (int k = 0; k < 5; ++k) tot[k] += w[k];
            for (int k = 0; k < 4; ++k) lp[k].mul(f[k]);
        }
        for (int k = 0; k < 4; ++k) tot[k] += (R)0.5 * lp[k].log();
        for (int k = 0; k < 5; ++k) ((R*)out)[(size_t)k * C + c] = tot[k];
    }
    return 0;
}

// the Lorenz sweep's fused log-density pass (kalman_bodies.h::body_lorenz_logpdf), dense or chain-minor views of the same dense inputs
template <typename R, int PO>
static int lorenz_logpdf_T(int C, int T, const HsArr* g, const void* yobs, const void* x, const void* xp, const void* u, const void* par, int psc,
                           double delta, int pol, int cm, void* out) {
    SweepLogpdfArgs a;
    a.d = KDims{C, T, 1};
    a.dx = 3; a.po = PO;
    a.m0 = cv(g[0]); a.P0 = cv(g[1]); a.Qs = cv(g[3]);
    a.Fs = Arr{nullptr, 0, 0, 0, 1}; a.bs = Arr{nullptr, 0, 0, 0, 1};
    a.Hs = cv(g[5]); a.Rs = cv(g[6]); a.cs = cv(g[7]);
    a.ys = Arr{yobs, 0, PO, 0, 1};
    a.delta = delta; a.nan_policy = pol;
    a.lor_par = par; a.lor_psc = psc;
    std::vector<std::vector<R>> keep;
    auto view = [&](const void* p) -> Arr {
        if (!cm) return dense_arr(p, a.d, 3);
        keep.emplace_back((size_t)C * T * 3);
        dense_to_cm<R>((const R*)p, a.d, 3, keep.back());
        return cm_arr(keep.back().data(), a.d, 3);
    };
    a.x = view(x); a.xp = view(xp); a.u = view(u);
    for (int c = 0; c < C; ++c) {
        R tot[5];
        body_lorenz_logpdf_head<R, PO>(a, c, tot);
        if (cm) {   // the chain-minor kernel's form: determinants multiplied up, one logarithm per sum (k_lorenz_logpdf_cm)
            LogProd<R> lp[4];
            for (int i = 0; i < T - 1; ++i) {
                R w[5], f[4];
                body_lorenz_logpdf<R, PO>(a, c, i, true, w, f);
                for (int k = 0; k < 5; ++k) tot[k] += w[k];
                for (int k = 0; k < 4; ++k) lp[k].mul(f[k]);
            }
            for (int k = 0; k < 4; ++k) tot[k] += (R)0.5 * lp[k].log();
        } else {
            for (int i = 0; i < T - 1; ++i) {
                R w[5];
                body_lorenz_logpdf<R, PO>(a, c, i, true, w);
                for (int k = 0; k < 5; ++k) tot[k] += w[k];
            }
        }
        for (int k = 0; k < 5; ++k) ((R*)out)[(size_t)k * C + c] = tot[k];
    }
    return 0;
}

#define HS_P_SWITCH(CALL, R, D)                                           \
    switch (P) {                                                          \
        case 1: return CALL(R, D, 1); case 2: return CALL(R, D, 2);       \
        case 3: return CALL(R, D, 3); case 4: return CALL(R, D, 4);       \
        case 5: return CALL(R, D, 5); case 6: return CALL(R, D, 6);       \
        case 7: return CALL(R, D, 7); case 8: return CALL(R, D, 8);       \
        default: return -2;                                               \
    }
#define HS_D_SWITCH(CALL, R)                                              \
    switch (D) {                                                          \
        case 1: HS_P_SWITCH(CALL, R, 1) case 2: HS_P_SWITCH(CALL, R, 2)   \
        case 3: HS_P_SWITCH(CALL, R, 3) case 4: HS_P_SWITCH(CALL, R, 4)   \
        default: return -2;                                               \
    }


// fold-vs-combine identity of the general path (kalman_math.h): prefix (+) element(step) built from the step's observation information and
// combined with filter_combine, against filter_fold_step / filter_apply_step on the same inputs.  Returns the largest absolute difference.
template <int D> static double fold_check_T(const double* F, const double* Q, const double* bd, const double* Lam, const double* g0, double q0, double ldR,
                                            double dim, const double* acc_in) {
    constexpr int DS = symsize(D);
    FiltElem<double, D> acc, e2, o, f;
    const double* p = acc_in;
    for (int i = 0; i < D * D; ++i) acc.A[i] = *p++;
    for (int i = 0; i < D; ++i) acc.b[i] = *p++;
    for (int i = 0; i < DS; ++i) acc.C[i] = *p++;
    for (int i = 0; i < D; ++i) acc.eta[i] = *p++;
    for (int i = 0; i < DS; ++i) acc.J[i] = *p++;
    acc.z = *p++;
    // the step's own element: built around (m_, P_) = (b_dyn, Q)  (_filtering_init_one for t >= 1)
    double gm[D], q = q0;
    for (int i = 0; i < D; ++i) {
        double lm = 0;
        for (int k = 0; k < D; ++k) lm += Lam[sidx(D, i, k)] * bd[k];
        gm[i] = g0[i] - lm;
        q += bd[i] * (lm - 2 * g0[i]);
    }
    filter_elem_from_lam<double, D>(F, bd, Q, Lam, gm, gm, q, ldR, dim, true, e2);
    filter_combine<double, D>(acc, e2, o);
    StepInfo<double, D> si;
    for (int i = 0; i < DS; ++i) si.Lam[i] = Lam[i];
    for (int i = 0; i < D; ++i) si.g0[i] = g0[i], si.u[i] = 0;
    si.inv_hd = 0;  // (no separate auxiliary block: Lam / g0 / q0 are the whole observation)
    si.q0 = q0, si.ldR = ldR, si.dim = dim, si.ok = true;
    f = acc;
    filter_fold_step<double, D>(F, Q, bd, si, f);
    FiltPre<double, D> pre;
    for (int i = 0; i < D; ++i) pre.b[i] = acc.b[i];
    for (int i = 0; i < DS; ++i) pre.C[i] = acc.C[i];
    pre.z = acc.z;
    filter_apply_step<double, D>(F, Q, bd, si, pre);
    double m = 0;
    auto upd = [&](double a, double b) { m = std::max(m, std::abs(a - b)); };
    for (int i = 0; i < D * D; ++i) upd(o.A[i], f.A[i]);
    for (int i = 0; i < D; ++i) upd(o.b[i], f.b[i]), upd(o.eta[i], f.eta[i]), upd(o.b[i], pre.b[i]);
    for (int i = 0; i < DS; ++i) upd(o.C[i], f.C[i]), upd(o.J[i], f.J[i]), upd(o.C[i], pre.C[i]);
    upd(o.z, f.z);
    upd(o.z, pre.z);
    // the same two steps with the log-determinant DEFERRED into a running product (LogProd: what the chunk-serial passes do, one logarithm per chunk), the
    // observation's own log-determinant handed over as the factor rdet = exp(-2 ldR) instead of the additive ldR
    {
        StepInfo<double, D> sd = si;
        sd.ldR = 0;
        sd.rdet = std::exp(-2 * ldR);
        LogProd<double> lf, lw;
        lf.mul(3.5), lw.mul(0.02);  // (a chunk's product so far)
        FiltElem<double, D> fd = acc;
        filter_fold_step<double, D, true>(F, Q, bd, sd, fd, &lf);
        FiltPre<double, D> pd;
        for (int i = 0; i < D; ++i) pd.b[i] = acc.b[i];
        for (int i = 0; i < DS; ++i) pd.C[i] = acc.C[i];
        pd.z = acc.z;
        filter_apply_step<double, D, true>(F, Q, bd, sd, pd, &lw);
        upd(o.z, fd.z + 0.5 * (lf.log() - std::log(3.5)));
        upd(o.z, pd.z + 0.5 * (lw.log() - std::log(0.02)));
        for (int i = 0; i < D; ++i) upd(f.b[i], fd.b[i]), upd(pre.b[i], pd.b[i]);
        for (int i = 0; i < DS; ++i) upd(f.C[i], fd.C[i]), upd(f.J[i], fd.J[i]);
    }
    return m;
}

// the log-likelihood increment of one step (filter_apply_step on a prefix with z = 0) with the auxiliary block y = u, H = I, R = hd I given (a) folded into the
// information form around the origin (Lam + I / hd, g0 + u / hd, q0 + |u|^2 / hd; inv_hd = 0: how the general path formed it up to round 3) and (b) apart
// (StepInfo::u / inv_hd: evaluated around the predicted mean), in precision R.  out = {zinc (a), zinc (b), max |b' (a) - b' (b)|, max |C' (a) - C' (b)|}
template <typename R, int D> static void fold_aux_T(const double* F, const double* Q, const double* bd, const double* Lobs, const double* gobs, double qobs,
                                                    const double* u, double inv_hd, double ldR, double dim, const double* b0, const double* C0, double* out) {
    constexpr int DS = symsize(D);
    R Ff[D * D], Qf[D * D], bdf[D];
    for (int i = 0; i < D * D; ++i) Ff[i] = (R)F[i], Qf[i] = (R)Q[i];
    for (int i = 0; i < D; ++i) bdf[i] = (R)bd[i];
    StepInfo<R, D> sa, sb;
    double q0 = qobs;
    for (int i = 0; i < DS; ++i) sa.Lam[i] = sb.Lam[i] = (R)Lobs[i];
    for (int i = 0; i < D; ++i) {
        sa.Lam[sidx_u(D, i, i)] = (R)((double)sa.Lam[sidx_u(D, i, i)] + inv_hd);
        sa.g0[i] = (R)(gobs[i] + u[i] * inv_hd);
        sa.u[i] = 0;
        q0 += u[i] * u[i] * inv_hd;
        sb.g0[i] = (R)gobs[i];
        sb.u[i] = (R)u[i];
    }
    sa.inv_hd = 0, sa.q0 = (R)q0;
    sb.inv_hd = (R)inv_hd, sb.q0 = (R)qobs;
    sa.ldR = sb.ldR = (R)ldR, sa.dim = sb.dim = (R)dim, sa.ok = sb.ok = true;
    FiltPre<R, D> pa, pb;
    for (int i = 0; i < D; ++i) pa.b[i] = pb.b[i] = (R)b0[i];
    for (int i = 0; i < DS; ++i) pa.C[i] = pb.C[i] = (R)C0[i];
    pa.z = pb.z = 0;
    filter_apply_step<R, D>(Ff, Qf, bdf, sa, pa);
    filter_apply_step<R, D>(Ff, Qf, bdf, sb, pb);
    out[0] = (double)pa.z, out[1] = (double)pb.z, out[2] = 0, out[3] = 0;
    for (int i = 0; i < D; ++i) out[2] = std::max(out[2], std::abs((double)pa.b[i] - (double)pb.b[i]));
    for (int i = 0; i < DS; ++i) out[3] = std::max(out[3], std::abs((double)pa.C[i] - (double)pb.C[i]));
}

extern "C" {

int hs_fold_aux(int f32, int D, const double* F, const double* Q, const double* bd, const double* Lobs, const double* gobs, double qobs, const double* u,
                double inv_hd, double ldR, double dim, const double* b0, const double* C0, double* out) {
#define HS_FA(RR)                                                                                              \
    switch (D) {                                                                                               \
        case 1: fold_aux_T<RR, 1>(F, Q, bd, Lobs, gobs, qobs, u, inv_hd, ldR, dim, b0, C0, out); return 0;      \
        case 2: fold_aux_T<RR, 2>(F, Q, bd, Lobs, gobs, qobs, u, inv_hd, ldR, dim, b0, C0, out); return 0;      \
        case 3: fold_aux_T<RR, 3>(F, Q, bd, Lobs, gobs, qobs, u, inv_hd, ldR, dim, b0, C0, out); return 0;      \
        case 4: fold_aux_T<RR, 4>(F, Q, bd, Lobs, gobs, qobs, u, inv_hd, ldR, dim, b0, C0, out); return 0;      \
    }
    if (f32) { HS_FA(float) } else { HS_FA(double) }
#undef HS_FA
    return -1;
}

double hs_fold_check(int D, const double* F, const double* Q, const double* bd, const double* Lam, const double* g0, double q0, double ldR, double dim,
                     const double* acc) {
    switch (D) {
        case 1: return fold_check_T<1>(F, Q, bd, Lam, g0, q0, ldR, dim, acc);
        case 2: return fold_check_T<2>(F, Q, bd, Lam, g0, q0, ldR, dim, acc);
        case 3: return fold_check_T<3>(F, Q, bd, Lam, g0, q0, ldR, dim, acc);
        case 4: return fold_check_T<4>(F, Q, bd, Lam, g0, q0, ldR, dim, acc);
    }
    return -1;
}

int hs_filter(int dtype, int D, int P, int C, int T, int B, const HsArr* g, const HsArr* ys, int E, void* ms, void* Ps, void* ell, int pblk) {
#define CALL(R, D, P) filter_T<R, D, P>(C, T, B, g, ys, E, ms, Ps, ell, pblk)
    if (dtype == 0) { HS_D_SWITCH(CALL, float) } else { HS_D_SWITCH(CALL, double) }
#undef CALL
}

int hs_sample(int dtype, int D, int C, int T, int B, const HsArr* g, const void* ms, const void* Ps, const void* eps, int E, void* xs) {
#define CALLS(R)                                                                        \
    switch (D) {                                                                        \
        case 1: return sample_T<R, 1>(C, T, B, g, ms, Ps, eps, E, xs);                  \
        case 2: return sample_T<R, 2>(C, T, B, g, ms, Ps, eps, E, xs);                  \
        case 3: return sample_T<R, 3>(C, T, B, g, ms, Ps, eps, E, xs);                  \
        case 4: return sample_T<R, 4>(C, T, B, g, ms, Ps, eps, E, xs);                  \
        default: return -2;                                                             \
    }
    if (dtype == 0) { CALLS(float) } else { CALLS(double) }
#undef CALLS
}

int hs_logpdf(int dtype, int D, int P, int C, int T, int B, const HsArr* g, const HsArr* ys, const HsArr* xs, int pol, void* out) {
#define CALL(R, D, P) logpdf_T<R, D, P>(C, T, B, g, ys, xs, pol, out)
    if (dtype == 0) { HS_D_SWITCH(CALL, float) } else { HS_D_SWITCH(CALL, double) }
#undef CALL
}

int hs_sv_logpdf(int dtype, int D, int C, int T, const HsArr* g, const void* yobs, const void* x, const void* xp, const void* u, const void* ys1,
                 const void* ys2, const void* R1, const void* R2, double delta, int cm, void* out) {
#define CALLS(R)                                                                                              \
    switch (D) {                                                                                              \
        case 1: return sv_logpdf_T<R, 1>(C, T, g, yobs, x, xp, u, ys1, ys2, R1, R2, delta, cm, out);          \
        case 2: return sv_logpdf_T<R, 2>(C, T, g, yobs, x, xp, u, ys1, ys2, R1, R2, delta, cm, out);          \
        case 3: return sv_logpdf_T<R, 3>(C, T, g, yobs, x, xp, u, ys1, ys2, R1, R2, delta, cm, out);          \
        case 4: return sv_logpdf_T<R, 4>(C, T, g, yobs, x, xp, u, ys1, ys2, R1, R2, delta, cm, out);          \
        default: return -2;                                                                                   \
    }
    if (dtype == 0) { CALLS(float) } else { CALLS(double) }
#undef CALLS
}

int hs_lorenz_logpdf(int dtype, int PO, int C, int T, const HsArr* g, const void* yobs, const void* x, const void* xp, const void* u,
                     const void* par, int psc, double delta, int pol, int cm, void* out) {
#define CALLS(R)                                                                                       \
    switch (PO) {                                                                                      \
        case 1: return lorenz_logpdf_T<R, 1>(C, T, g, yobs, x, xp, u, par, psc, delta, pol, cm, out);  \
        case 2: return lorenz_logpdf_T<R, 2>(C, T, g, yobs, x, xp, u, par, psc, delta, pol, cm, out);  \
        case 3: return lorenz_logpdf_T<R, 3>(C, T, g, yobs, x, xp, u, par, psc, delta, pol, cm, out);  \
        default: return -2;                                                                            \
    }
    if (dtype == 0) { CALLS(float) } else { CALLS(double) }
#undef CALLS
}

}  // extern "C"

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

// Strided array view: element e of the record at (chain c, time t, batch b) lives at ptr + c sc + t st + b sb + e se.
// se = 1: the record is contiguous (user-facing dense / broadcast arrays).  se = S: "chain-minor" internal buffers
// [t][e][s] -- consecutive lanes = consecutive sequences read consecutive addresses for every component.
struct Arr {
    const void* ptr;
    long long sc, st, sb, se;
};
template <typename R> AX_HD const R* at(const Arr& a, int c, long long t, int b) {
    return (const R*)a.ptr + (long long)c * a.sc + t * a.st + (long long)b * a.sb;
}
template <typename R, int N> AX_HD void lds_(const R* __restrict__ p, long long se, R* out) {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = p[i * se];
}
template <typename R, int N> AX_HD void sts_(R* __restrict__ p, long long se, const R* v) {
#pragma unroll
    for (int i = 0; i < N; ++i) p[i * se] = v[i];
}
template <typename R, int N> AX_HD void rd(const Arr& a, int c, long long t, int b, R* out) { lds_<R, N>(at<R>(a, c, t, b), a.se, out); }
// only the upper triangle (row <= col) of a symmetric P x P record: the covariance consumers never touch the rest, and the
// 28 doubles (P = 8) it saves per lane are the difference between spilling and not in the p = 8 kernels
template <typename R, int P> AX_HD void lds_upper_(const R* __restrict__ p, long long se, R* out) {
#pragma unroll
    for (int k = 0; k < P; ++k)
#pragma unroll
        for (int l = k; l < P; ++l) out[k * P + l] = p[(long long)(k * P + l) * se];
}
template <typename R, int P> AX_HD void rd_upper(const Arr& a, int c, long long t, int b, R* out) { lds_upper_<R, P>(at<R>(a, c, t, b), a.se, out); }
template <typename R, int N> AX_HD void wr(const Arr& a, int c, long long t, int b, const R* v) {
    sts_<R, N>(const_cast<R*>(at<R>(a, c, t, b)), a.se, v);
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
    int cm, S;               // cm != 0: chain-minor element buffer [i][e][s] (S sequences), used when lanes map to sequences
    AX_HD long long seq_records() const { return (long long)ngrp * E * W; }
    AX_HD long long pos(int i) const {
        const int ch = i / E, k = i - ch * E;
        const int g = ch / W, l = ch - g * W;
        return ((long long)g * E + k) * W + l;
    }
    AX_HD long long row(int grp, int k) const { return ((long long)grp * E + k) * W; }
    // offset (in reals) of component 0 of element i of sequence s, and the component stride
    AX_HD long long eoff(int s, int i, int npad) const {
        return cm ? (long long)i * npad * S + s : ((long long)s * seq_records() + pos(i)) * npad;
    }
    AX_HD long long es() const { return cm ? S : 1; }
    AX_HD long long total_reals(int n, int Sq, int npad) const {
        return cm ? (long long)(n > 0 ? n : 1) * npad * Sq : (long long)Sq * seq_records() * npad;
    }
};

// Two-phase protocol so that the global-memory latency of all the arrays a body needs overlaps:
//   fetch<N>(lane_ptr, lane_stride, se, valid, buf)  -- issue the global reads into buf (no barrier); se = element stride
//   finish<N>(lane_stride, se, valid, buf)           -- turn buf into this lane's record (WaveIO: transposition through LDS).
// lane_ptr = this lane's record; lane_stride = distance in reals between consecutive lanes' records.
struct DirectIO {
    template <typename R, int N> AX_HD void fetch(const R* lane_ptr, long long /*lane_stride*/, long long se, bool valid, R* buf) const {
        if (valid) {
            lds_<R, N>(lane_ptr, se, buf);
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) buf[i] = 0;
        }
    }
    template <typename R, int N> AX_HD void finish(long long /*lane_stride*/, long long /*se*/, bool /*valid*/, R* /*buf*/) const {}
    // symmetric P x P record: upper triangle only, always a direct read
    template <typename R, int P> AX_HD void fetch_upper(const R* lane_ptr, long long se, bool valid, R* buf) const {
        if (valid) lds_upper_<R, P>(lane_ptr, se, buf);
    }
};

// the step size of a sweep: host scalars in the argument structs, or -- auxssm_kalman_sweep_dd -- a device block {delta, sqrt(delta / 2)}
// written by k_delta_block from the caller's device scalar (wave-uniform loads; nothing is read back to the host)
template <typename A> AX_HD double arg_delta(const A& a) { return a.dptr ? a.dptr[0] : a.delta; }
template <typename A> AX_HD double arg_shd(const A& a) { return a.dptr ? a.dptr[1] : a.shd; }
template <typename A> AX_HD double arg_aux_shd(const A& a) { return a.dptr ? a.dptr[1] : a.aux_shd; }
struct FilterArgs {
    KDims d;
    Arr m0, P0, Fs, Qs, bs, Hs, Rs, cs, ys;
    Arr ms;         // (C,T,B,D): dense, or chain-minor inside the fused sweep
    Arr Ps;         // (C,T,B,D,D)
    void* elem;     // scan elements, layout `lay`
    void* ell0;     // [S] log-likelihood term of the t = 0 update
    void* ellz;     // [S] log-scale of the whole scan = sum of the increments of t = 1..T-1 (written by the final pass)
    ScanLayout lay;
    int pblk;       // > 0: Rs is block diagonal, first block pblk x pblk (hint; enables the information form of kalman_math.h)
    const void* tab = nullptr;  // chain-shared parameters: one GainRow per transition (affine_shared.h; else null)
    void* pc = nullptr;         // wide-state chain-shared filter only (wide.hip::run_filter_shared): the caller's buffer for the per-step gain rows (wide_gain_tab_bytes);
                                // with tab_ready != 0 it HOLDS this model's rows (and Ps the matrix filter's covariances): the reverse filter of a sweep reuses both
    // Concatenated auxiliary observations built on the fly (sweep of the LG_CONCAT model, shared mode): for t >= 1 the observation
    // is y_t = [u_t ; yobs_t] with u = x + aux_shd * eps (kalman/generic.py:59-63); u is written to aux_u, ys holds row t = 0 only.
    int aux_on = 0;
    Arr aux_x{}, aux_eps{}, aux_u{}, aux_yobs{};
    double aux_shd = 0;
    // chain-shared filter with per-chain observations (SV first order): the observation MASK the gain table is built for comes from mask_ys (T, P; a
    // chain-independent carrier) when set, else from chain 0's ys; tab_ready: `tab` already holds this sweep's gain rows (the reverse filter of a sweep
    // reuses the proposal filter's: same model, same step size, same mask)
    Arr mask_ys{nullptr, 0, 0, 0, 1};
    int tab_ready = 0;
    int t0_keep_ps = 0;  // k_filter_t0: leave Ps[0] alone (chain-shared covariances: the matrix filter wrote the one slot all chains would write)
    const double* dptr = nullptr;  // device-resident step size: {delta, sqrt(delta / 2)} (auxssm_kalman_sweep_dd); null: the host values above
    // aux_gen != 0 (auxssm_kalman_sweep_keyed): rows t >= 1 of aux_eps are GENERATED by the reduce pass (the first reader) from the key and
    // written for the later readers (down pass, log-density); the values are those of auxssm_rng_normal(key, stream 0) at the same indices
    int aux_gen = 0;
    unsigned int gen_k0 = 0, gen_k1 = 0;
    // ps_packed != 0: Ps records are symmetric-packed (symsize(D) reals, upper storage) -- the internal chain-minor buffer of the general
    // fused sweep, which the pathwise sampler reads twice (SampleArgs::ps_packed)
    int ps_packed = 0;
    const void* obs_tab = nullptr;  // general chain-minor path with aux_on: one ObsInfoRow per transition (kalman_math.h), elements built on the fly
    // sv_order = 1 / 2 (chain-minor SV sweep, per-chain observation model): NO observation arrays exist -- the pseudo-observations of the SV factories
    // (examples/stochastic_volatility/auxiliary_kalman.py:28-46) are functions of the linearisation point aux_x, the auxiliary variable (aux_u, or
    // aux_x + aux_shd * aux_eps when aux_eps is set, then written to aux_u by the final pass) and the data aux_yobs, and both scan passes form them step by
    // step in information form (FilterOpFlySV below).  sv_delta = the step size; no_moments: the pass is run for its log-likelihood only (the reverse filter
    // of a sweep): ms / Ps rows t >= 1 are not written.
    int sv_order = 0;
    double sv_delta = 0;
    int no_moments = 0;
    int dx = 0, dy = 0;  // runtime sizes, read by the wide-state path (wide.hip) only
    // MODEL-STAGE MEMO (ctx.h::SideStage, round 4): non-null inside a memoised stage; *memo == 0 says "the stage's inputs are byte for byte the ones this slab's
    // tables were built from" and every stage kernel returns at once (memo_skip); non-zero: rebuild.  Null everywhere else.
    const int* memo = nullptr;
};
template <typename A> AX_HD bool memo_skip(const A& a) { return a.memo != nullptr && *a.memo == 0; }
AX_HD bool memo_skip_p(const int* memo) { return memo != nullptr && *memo == 0; }
AX_HD Arr dense_arr(const void* p, const KDims& d, long long rec) {
    return Arr{p, (long long)d.T * d.B * rec, (long long)d.B * rec, rec, 1};
}
// chain-minor view of an internal (S = C*B sequences, T, rec) buffer: [t][e][s]
AX_HD Arr cm_arr(const void* p, const KDims& d, long long rec) {
    const long long S = (long long)d.C * d.B;
    return Arr{p, (long long)d.B, rec * S, 1, S};
}

// a covariance record: full D x D, or symmetric-packed (ps_packed)
template <typename R, int D> AX_HD void rd_cov(const Arr& a, int c, long long t, int b, int packed, R* Pd) {
    if (packed) {
        R pk[symsize(D)];
        rd<R, symsize(D)>(a, c, t, b, pk);
        symunpack<R, D>(pk, Pd);
    } else {
        rd<R, D * D>(a, c, t, b, Pd);
    }
}
template <typename R, int D> AX_HD void wr_cov(const Arr& a, int c, long long t, int b, int packed, const R* Pd) {
    if (packed) {
        R pk[symsize(D)];
        sympack<R, D>(Pd, pk);
        wr<R, symsize(D)>(a, c, t, b, pk);
    } else {
        wr<R, D * D>(a, c, t, b, Pd);
    }
}
// ---- t = 0 measurement update (filtering.py:52) -------------------------------------------------
template <typename R, int D, int P> AX_HD void body_filter_t0(const FilterArgs& a, int s) {
    const int c = s / a.d.B, b = s % a.d.B;
    R m[D], Pd[D * D], H[P * D], cv[P], y[P], Rm[P * P];
    rd<R, D>(a.m0, c, 0, b, m);
    rd<R, D * D>(a.P0, c, 0, b, Pd);
    rd<R, P * D>(a.Hs, c, 0, b, H);
    rd<R, P>(a.cs, c, 0, b, cv);
    rd<R, P>(a.ys, c, 0, b, y);
    rd_upper<R, P>(a.Rs, c, 0, b, Rm);
    const R ell = kalman_update<R, D, P>(m, Pd, H, cv, Rm, y);
    wr<R, D>(a.ms, c, 0, b, m);
    if (!a.t0_keep_ps) wr_cov<R, D>(a.Ps, c, 0, b, a.ps_packed, Pd);
    ((R*)a.ell0)[s] = ell;
}

// A table row is read at a wave-uniform address and never written by the kernel that reads it: viewed through the constant
// address space the loads become scalar (s_load into SGPRs) instead of 64 identical vector loads.
#if defined(__HIP_DEVICE_COMPILE__)
template <typename R> using UniformRow = const __attribute__((address_space(4))) R*;
template <typename R> __device__ __forceinline__ UniformRow<R> uniform_row(const R* p) { return (UniformRow<R>)(unsigned long long)p; }
#else
template <typename R> using UniformRow = const R*;
template <typename R> AX_HD UniformRow<R> uniform_row(const R* p) { return p; }
#endif

template <typename R_, int D> struct FilterOp;
// ---- scan element for transition i -> i+1 (filtering.py:188-250) ---------------------------------
template <typename R, int D, int P, class IO, int P1 = 0>
AX_HD void filter_build_elem(const FilterArgs& a, IO& io, int s, int i, bool valid, FiltElem<R, D>& e) {
    const int c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    R F[D * D], bd[D], m_[D], P_[D * D], H[P * D], cv[P], y[P], Rm[P * P];
    io.template fetch<R, D * D>(at<R>(a.Fs, c, i, b), a.Fs.st, a.Fs.se, valid, F);
    io.template fetch<R, D>(at<R>(a.bs, c, i, b), a.bs.st, a.bs.se, valid, bd);
    io.template fetch<R, D * D>(at<R>(a.Qs, c, i, b), a.Qs.st, a.Qs.se, valid, P_);
    io.template fetch<R, P * D>(at<R>(a.Hs, c, t, b), a.Hs.st, a.Hs.se, valid, H);
    io.template fetch<R, P>(at<R>(a.cs, c, t, b), a.cs.st, a.cs.se, valid, cv);
    io.template fetch<R, P>(at<R>(a.ys, c, t, b), a.ys.st, a.ys.se, valid, y);
    io.template fetch_upper<R, P>(at<R>(a.Rs, c, t, b), a.Rs.se, valid, Rm);
    io.template finish<R, D * D>(a.Fs.st, a.Fs.se, valid, F);
    io.template finish<R, D>(a.bs.st, a.bs.se, valid, bd);
    io.template finish<R, D * D>(a.Qs.st, a.Qs.se, valid, P_);
    io.template finish<R, P * D>(a.Hs.st, a.Hs.se, valid, H);
    io.template finish<R, P>(a.cs.st, a.cs.se, valid, cv);
    io.template finish<R, P>(a.ys.st, a.ys.se, valid, y);
    if (!valid) return;
    if (i == 0) {
        // first transition: built around predict(m0+, P0+)  (m_ = F m + b, P_ = F P F^T + Q, not symmetrised: filtering.py:200-201)
        R m0p[D], P0p[D * D], tm[D], FP[D * D], Pn[D * D];
        rd<R, D>(a.ms, c, 0, b, m0p);
        rd_cov<R, D>(a.Ps, c, 0, b, a.ps_packed, P0p);
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
    if constexpr (P1 > 0 && P1 < P) filter_elem_blk<R, D, P, P1>(F, bd, m_, P_, H, cv, Rm, y, e);
    else filter_elem<R, D, P>(F, bd, m_, P_, H, cv, Rm, y, e);
}
template <typename R, int D, int P, class IO, int P1 = 0>
AX_HD void body_filter_init(const FilterArgs& a, IO& io, int s, int i, bool valid) {
    FiltElem<R, D> e;
    filter_build_elem<R, D, P, IO, P1>(a, io, s, i, valid, e);
    if (valid) FilterOp<R, D>::store_elem(a, s, i, e);
}

// ---- scan operator: parallel filter ----------------------------------------------------------------
struct ScanBufs {
    void* agg;   // [S][nchunk][Full::NPAD]
    void* pre;   // [S][nchunk][Pre::NPAD]
    const int* memo = nullptr;  // (model-stage memo: as FilterArgs::memo)
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
        e.z = t[Full::N - 1];
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
        p.z = f.z;
    }
    static AX_HD void store_pre(R* q, const Pre& p) {
        R t[D + DS + 1];
#pragma unroll
        for (int i = 0; i < D; ++i) t[i] = p.b[i];
#pragma unroll
        for (int i = 0; i < DS; ++i) t[D + i] = p.C[i];
        t[D + DS] = p.z;
        stv<R, D + DS + 1>(q, t);
    }
    static AX_HD void load_pre(const R* q, Pre& p) {
        R t[D + DS + 1];
        ldv<R, D + DS + 1>(q, t);
#pragma unroll
        for (int i = 0; i < D; ++i) p.b[i] = t[i];
#pragma unroll
        for (int i = 0; i < DS; ++i) p.C[i] = t[D + i];
        p.z = t[D + DS];
    }
    static AX_HD void apply(const Pre& p, const Full& e, Pre& o) { filter_apply<R, D>(p, e, o); }
    // inclusive prefix i  ->  filtered moments at time i + 1
    static AX_HD void write_out(const Args& a, int s, int i, const Pre& p) {
        const int c = s / a.d.B, b = s % a.d.B;
        wr<R, D>(a.ms, c, (long long)i + 1, b, p.b);
        if (a.ps_packed) {
            wr<R, symsize(D)>(a.Ps, c, (long long)i + 1, b, p.C);  // the prefix carries C packed already
        } else {
            R Pd[D * D];
            symunpack<R, D>(p.C, Pd);
            wr<R, D * D>(a.Ps, c, (long long)i + 1, b, Pd);
        }
        if (i == a.d.n() - 1 && a.ellz) ((R*)a.ellz)[s] = p.z;  // scale of the full product = log p(y_1..T-1 | y_0)
    }
    // element i of sequence s in either element layout
    static AX_HD void store_elem(const Args& a, int s, int i, const Full& e) {
        R* p = (R*)a.elem + a.lay.eoff(s, i, Full::NPAD);
        if (a.lay.cm) {
            R t[Full::N];
            pack(e, t);
            sts_<R, Full::N>(p, a.lay.es(), t);
        } else {
            fe_store<R, D>(p, e);
        }
    }
    static AX_HD void load_elem(const Args& a, int s, int i, Full& e) {
        const R* p = (const R*)a.elem + a.lay.eoff(s, i, Full::NPAD);
        if (a.lay.cm) {
            R t[Full::N];
            lds_<R, Full::N>(p, a.lay.es(), t);
            unpack(t, e);
        } else {
            fe_load<R, D>(p, e);
        }
    }
    static AX_HD void pack(const Full& e, R* t) {
#pragma unroll
        for (int i = 0; i < D * D; ++i) t[i] = e.A[i];
#pragma unroll
        for (int i = 0; i < D; ++i) t[D * D + i] = e.b[i], t[D * D + D + DS + i] = e.eta[i];
#pragma unroll
        for (int i = 0; i < DS; ++i) t[D * D + D + i] = e.C[i], t[D * D + 2 * D + DS + i] = e.J[i];
        t[Full::N - 1] = e.z;
    }
    // Two-step element access of the chain-minor scan kernels: load_raw issues the global reads of element i (they fly during the
    // previous element's combine), build turns them into the element.  Stored elements: the record itself.
    using Raw = Full;
    static AX_HD void load_raw(const Args& a, int s, int i, Raw& r) { load_elem(a, s, i, r); }
    static AX_HD void build(const Args&, int, int, const Raw& r, Full& e) { e = r; }
    static constexpr bool kFold = false;  // true: the operator folds raw steps onto the accumulated prefix itself (fold / walk)
};

// FilterOp that never materialises its elements (general chain-minor sweep, concatenated auxiliary observations): load_raw reads the
// chain's linearisation-point data of transition i -> i+1 -- x_t, eps_t (u = x + sqrt(delta/2) eps) and its dynamics F, Q, b, which
// are chain-strided (one linearisation per chain) or stride-0 broadcasts -- and build forms the element from them and the chain-shared
// ObsInfoRow of the time step (kalman_math.h::filter_elem_aux: d x d work only, no p x p factorisation per chain).  Both scan passes
// then stream 2d (+ per-chain dynamics) reals per step instead of writing and re-reading the (3d^2+2d)-real element, and the element
// kernel disappears.  WRITE_U: the pass that also stores u for the log-density kernel (the final pass).
#ifndef AUXSSM_FLY_DOWN_WAVES
#define AUXSSM_FLY_DOWN_WAVES 1  // 2 was measured: 10 spilled registers, filter scan 2.17 -> 2.73 ms at C2 / 256 chains
#endif
template <typename R_, int D, int P, bool WRITE_U> struct FilterOpFly : FilterOp<R_, D> {
    using R = R_;
    using Full = typename FilterOp<R_, D>::Full;
    using Args = FilterArgs;
    struct Raw {
        R F[D * D], Q[D * D], bd[D], x[D], eps[D];
    };
    static AX_HD void load_raw(const Args& a, int s, int i, Raw& r) {
        const int c = s / a.d.B, b = s % a.d.B;
        const long long t = (long long)i + 1;
        rd<R, D>(a.aux_x, c, t, b, r.x);
        rd<R, D>(a.aux_eps, c, t, b, r.eps);
        rd<R, D * D>(a.Fs, c, i, b, r.F);
        rd<R, D>(a.bs, c, i, b, r.bd);
        rd<R, D * D>(a.Qs, c, i, b, r.Q);
    }
    static AX_HD void build(const Args& a, int s, int i, const Raw& r, Full& e) {
        using T = ObsInfoRow<R, D>;
        const int c = s / a.d.B, b = s % a.d.B;
        R u[D], m_[D], P_[D * D];
#pragma unroll
        for (int k = 0; k < D; ++k) u[k] = r.x[k] + (R)arg_aux_shd(a) * r.eps[k];
        if constexpr (WRITE_U) wr<R, D>(a.aux_u, c, (long long)i + 1, b, u);
        if (i == 0) {  // built around predict(m0+, P0+) (filtering.py:188-192, :200-201)
            R m0p[D], P0p[D * D], tm[D], FP[D * D], Pn[D * D];
            rd<R, D>(a.ms, c, 0, b, m0p);
            rd_cov<R, D>(a.Ps, c, 0, b, a.ps_packed, P0p);
            mv<R, D, D>(r.F, m0p, tm);
            mm<R, D, D, D>(r.F, P0p, FP);
            mmt<R, D, D, D>(FP, r.F, Pn);
#pragma unroll
            for (int k = 0; k < D; ++k) m_[k] = tm[k] + r.bd[k];
#pragma unroll
            for (int k = 0; k < D * D; ++k) P_[k] = Pn[k] + r.Q[k];
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) m_[k] = r.bd[k];
#pragma unroll
            for (int k = 0; k < D * D; ++k) P_[k] = r.Q[k];
        }
        const UniformRow<R> row = uniform_row<R>((const R*)a.obs_tab + (long long)i * T::NPAD);
        const R inv_hd = (R)1 / ((R)arg_aux_shd(a) * (R)arg_aux_shd(a));
        filter_elem_aux<R, D>(r.F, r.bd, m_, P_, u, row, inv_hd, i == 0, e);
    }
    static AX_HD void load_elem(const Args& a, int s, int i, Full& e) {
        Raw r;
        load_raw(a, s, i, r);
        build(a, s, i, r, e);
    }
    // ---- chunk-serial passes: fold the raw step onto the prefix (kalman_math.h::filter_fold_step / filter_apply_step) ----
    static constexpr bool kFold = true;
    static constexpr int kDownWaves = AUXSSM_FLY_DOWN_WAVES;  // the (b, C, z) walk needs 264 unified registers at fp64 d = 4: 8 over two waves per SIMD (forcing it costs more than it gains)
    using Pre = typename FilterOp<R_, D>::Pre;
    static constexpr int DS = symsize(D);
    // the prefix entering position 0 of the whole scan is the t = 0 posterior (m0+, P0+): a constant map (A = 0)
    static AX_HD void init_acc(const Args& a, int s, int ch, Full& acc) {
        fe_identity<R, D>(acc);
        if (ch == 0) {
            R P0p[D * D];
#pragma unroll
            for (int k = 0; k < D * D; ++k) acc.A[k] = 0;
            rd<R, D>(a.ms, s / a.d.B, 0, s % a.d.B, acc.b);
            rd_cov<R, D>(a.Ps, s / a.d.B, 0, s % a.d.B, a.ps_packed, P0p);
            sympack<R, D>(P0p, acc.C);
        }
    }
    static AX_HD void init_pre(const Args& a, int s, Pre& p) {
        R P0p[D * D];
        rd<R, D>(a.ms, s / a.d.B, 0, s % a.d.B, p.b);
        rd_cov<R, D>(a.Ps, s / a.d.B, 0, s % a.d.B, a.ps_packed, P0p);
        sympack<R, D>(P0p, p.C);
        p.z = 0;
    }
    static AX_HD void step_info(const Args& a, int s, int i, const Raw& r, StepInfo<R, D>& si) {
        using T = ObsInfoRow<R, D>;
        const UniformRow<R> row = uniform_row<R>((const R*)a.obs_tab + (long long)i * T::NPAD);
        const R inv_hd = (R)1 / ((R)arg_aux_shd(a) * (R)arg_aux_shd(a));
        R u[D];
#pragma unroll
        for (int k = 0; k < D; ++k) u[k] = r.x[k] + (R)arg_aux_shd(a) * r.eps[k];
        if constexpr (WRITE_U) wr<R, D>(a.aux_u, s / a.d.B, (long long)i + 1, s % a.d.B, u);
#pragma unroll
        for (int k = 0; k < DS; ++k) si.Lam[k] = row[T::oL + k];
        if constexpr (sizeof(R) == 4) {
            // fp32: the auxiliary block stays apart and step_predict_solve evaluates it around the predicted mean (StepInfo: the folded form below cancels
            // ~|x|^2 / hd down to the innovation, every digit of an fp32 log-likelihood increment at Lorenz-63 scale)
#pragma unroll
            for (int k = 0; k < D; ++k) si.u[k] = u[k], si.g0[k] = row[T::oG + k];
            si.inv_hd = inv_hd;
            si.q0 = row[T::oK];
        } else {
            // fp64: folded into the information form around the origin (8 of 16 digits to spare; 16 fewer multiply-adds and D fewer live values per step in a
            // kernel at its register cap: the general filter scan is 13 % faster this way)
            R q0 = row[T::oK];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                si.g0[k] = u[k] * inv_hd + row[T::oG + k];
                si.u[k] = 0;
                q0 += u[k] * u[k] * inv_hd;
                si.Lam[sidx_u(D, k, k)] += inv_hd;
            }
            si.inv_hd = 0;
            si.q0 = q0;
        }
        si.ldR = row[T::oLd];
        si.dim = row[T::oDim];
        si.ok = row[T::oOk] != (R)0;
    }
    // (the down pass hands the log-likelihood back chunk by chunk: k_scan_down_cm)
    static AX_HD void write_out(const Args& a, int s, int i, const Pre& p) {
        const int c = s / a.d.B, b = s % a.d.B;
        wr<R, D>(a.ms, c, (long long)i + 1, b, p.b);
        if (a.ps_packed) {
            wr<R, symsize(D)>(a.Ps, c, (long long)i + 1, b, p.C);
        } else {
            R Pd[D * D];
            symunpack<R, D>(p.C, Pd);
            wr<R, D * D>(a.Ps, c, (long long)i + 1, b, Pd);
        }
    }
    static AX_HD void write_zpart(const Args& a, int s, int ch, int nchunk, R z) { ((R*)a.ellz)[(long long)s * nchunk + ch] = z; }
    // Carry: what a chunk's walk keeps beside the prefix -- the running product of the steps' 1 / |det W| (kalman_math.h::LogProd), turned into the chunk's
    // log-determinant sum by ONE logarithm when the chunk ends (carry_flush)
    using Carry = LogProd<R>;
    static AX_HD void carry_flush(const Carry& cy, R& z) { z += (R)0.5 * cy.log(); }
    static AX_HD void fold(const Args& a, int s, int i, const Raw& r, Full& acc, Carry& cy) {
        StepInfo<R, D> si;
        step_info(a, s, i, r, si);
        filter_fold_step<R, D, true>(r.F, r.Q, r.bd, si, acc, &cy);
    }
    static AX_HD void walk(const Args& a, int s, int i, const Raw& r, Pre& p, Carry& cy) {
        StepInfo<R, D> si;
        step_info(a, s, i, r, si);
        filter_apply_step<R, D, true>(r.F, r.Q, r.bd, si, p, &cy);
    }
};
// table row of transition i -> i + 1 (time t = i + 1) from the concatenated model arrays and the data
template <typename R, int D, int P> AX_HD void body_obs_info_tab(const FilterArgs& a, int i) {
    using T = ObsInfoRow<R, D>;
    const long long t = (long long)i + 1;
    R H[P * D], cv[P], y[P], Rm[P * P], row[T::NPAD];
    rd<R, P * D>(a.Hs, 0, t, 0, H);
    rd<R, P>(a.cs, 0, t, 0, cv);
    rd_upper<R, P>(a.Rs, 0, t, 0, Rm);
#pragma unroll
    for (int k = 0; k < P; ++k) y[k] = k < D ? (R)0 : at<R>(a.aux_yobs, 0, t, 0)[k - D];
#pragma unroll
    for (int k = T::N; k < T::NPAD; ++k) row[k] = 0;
    obs_info_row<R, D, P>(H, cv, Rm, y, (R)arg_aux_shd(a) * (R)arg_aux_shd(a), row);
    stv<R, T::NPAD>((R*)a.obs_tab + (long long)i * T::NPAD, row);
}

// ---- the SV factories without observation arrays (FilterArgs::sv_order) -----------------------------------------------------------------
// Per component k (the factories are diagonal: H = I, c = 0, R = diag): w = y^2 e^{-x}, grad = nan_to_num((w - 1) / 2), hess = -w / 2, a = 2 / delta.
//   first order  (auxiliary_kalman.py:28-35): ys = u + (delta / 2) grad, R = delta / 2          -> precision a,        information a ys
//   second order (:37-46):                    Om^-1 = -hess + a, ys = Om (a u + grad - hess x)  -> precision -hess + a, information a u + (grad - hess x)
// which is exactly StepInfo's split (kalman_math.h): an auxiliary block (u', a) evaluated around the predicted mean plus a "model" block (Lam, g0) around the
// origin: first order u' = ys, Lam = 0; second order u' = u, Lam = diag(-hess), g0 = grad - hess x, q0 = ys^2 / Om - a u^2 in a form without the |u|^2 a
// cancellation, ldR = -log(-hess + a) / 2.  A NaN datum (second order) makes that component's ys NaN = unobserved (k_sv_obs + the element path do the same
// component by component): the step then takes the folded form with that component's precision and information zero.
template <typename R, int D> AX_HD void sv_step_info(const FilterArgs& a, const R* x, const R* u, const R* y, StepInfo<R, D>& si) {
    const R delta = (R)a.sv_delta, av = (R)2 / delta;
    const bool second = a.sv_order != 1;
    R lam[D], g[D], grad[D];
    bool miss[D], any = false;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R w = y[k] * y[k] * exp_(-x[k]);
        grad[k] = nan_to_num<R>((R)0.5 * (w - (R)1));
        lam[k] = second ? (R)0.5 * w : (R)0;
        g[k] = second ? grad[k] + lam[k] * x[k] : (R)0;
        miss[k] = second && !finite_(w);   // (k_sv_obs: ys = om (...) is NaN exactly when w is NaN or infinite)
        any = any || miss[k];
    }
    // (every field is assigned exactly once, outside the branches: written per branch, the struct went through scratch memory)
    R q0 = 0, rd = 1;   // rd = 1 / det Om = prod (lam_k + a): log(rd) / 2 = -log(det Om) / 2, taken with the chunk's other determinants (StepInfo::rdet)
    int dim = 0;
    R Ld[D], G0[D], U[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R tot = lam[k] + av;
        if (!second) {
            Ld[k] = 0, G0[k] = 0;
            U[k] = u[k] + (R)0.5 * delta * grad[k];
            dim += 1;
        } else if (!any) {
            const R e = u[k] - x[k];
            Ld[k] = lam[k], G0[k] = g[k], U[k] = u[k];
            // (a u + g)^2 / (lam + a) - a u^2 = [2 a u grad + a lam (x^2 - (u - x)^2) + g^2] / (lam + a)
            q0 += ((R)2 * av * u[k] * grad[k] + av * lam[k] * (x[k] * x[k] - e * e) + g[k] * g[k]) / tot;
            rd *= tot;
            dim += 1;
        } else {
            const R gi = av * u[k] + g[k];
            Ld[k] = miss[k] ? (R)0 : tot;
            G0[k] = miss[k] ? (R)0 : gi;
            U[k] = 0;
            q0 += miss[k] ? (R)0 : gi * gi / tot;
            rd *= miss[k] ? (R)1 : tot;
            dim += miss[k] ? 0 : 1;
        }
    }
#pragma unroll
    for (int k = 0; k < symsize(D); ++k) si.Lam[k] = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) si.Lam[sidx_u(D, k, k)] = Ld[k], si.g0[k] = G0[k], si.u[k] = U[k];
    si.inv_hd = (second && any) ? (R)0 : av;
    si.q0 = q0;
    si.ldR = second ? (R)0 : (R)D * log_(sqrt_((R)0.5 * delta));
    si.rdet = rd;
    si.dim = (R)dim;
    si.ok = true;
}
// the t = 0 update of the same model: the pseudo-observation of (x_0, u_0) as k_sv_obs forms it, then the plain update (filtering.py:52)
template <typename R, int D> AX_HD void body_filter_t0_sv(const FilterArgs& a, int s) {
    const int c = s / a.d.B, b = s % a.d.B;
    R m[D], Pd[D * D], H[D * D], cv[D], ys[D], Rm[D * D], x[D], u[D], y[D];
    rd<R, D>(a.m0, c, 0, b, m);
    rd<R, D * D>(a.P0, c, 0, b, Pd);
    rd<R, D>(a.aux_x, c, 0, b, x);
    rd<R, D>(a.aux_yobs, 0, 0, 0, y);
    if (a.aux_eps.ptr) {
        R eps[D];
        rd<R, D>(a.aux_eps, c, 0, b, eps);
#pragma unroll
        for (int k = 0; k < D; ++k) u[k] = x[k] + (R)a.aux_shd * eps[k];
        wr<R, D>(a.aux_u, c, 0, b, u);
    } else {
        rd<R, D>(a.aux_u, c, 0, b, u);
    }
    const R delta = (R)a.sv_delta;
#pragma unroll
    for (int k = 0; k < D * D; ++k) H[k] = (k / D == k % D) ? (R)1 : (R)0, Rm[k] = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R w = y[k] * y[k] * exp_(-x[k]);
        const R grad = nan_to_num<R>((R)0.5 * (w - (R)1));
        cv[k] = 0;
        if (a.sv_order == 1) {
            ys[k] = u[k] + (R)0.5 * delta * grad;
            Rm[k * D + k] = (R)0.5 * delta;
        } else {
            const R hess = (R)-0.5 * w;
            const R om = (R)1 / (-hess + (R)2 / delta);
            ys[k] = om * ((R)2 * u[k] / delta + grad - hess * x[k]);
            Rm[k * D + k] = om;
        }
    }
    const R ell = kalman_update<R, D, D>(m, Pd, H, cv, Rm, ys);
    wr<R, D>(a.ms, c, 0, b, m);
    wr_cov<R, D>(a.Ps, c, 0, b, a.ps_packed, Pd);
    ((R*)a.ell0)[s] = ell;
}
// The folding operator: FilterOpFly's passes (init_acc / init_pre / write_zpart are its own) with the step's information built by sv_step_info.
template <typename R_, int D, bool WRITE_U> struct FilterOpFlySV : FilterOpFly<R_, D, D + 1, false> {
    using R = R_;
    using Base = FilterOpFly<R_, D, D + 1, false>;
    using Full = typename Base::Full;
    using Pre = typename Base::Pre;
    using Args = FilterArgs;
    struct Raw {
        R F[D * D], Q[D * D], bd[D], x[D], ue[D], y[D];
    };
    static AX_HD void load_raw(const Args& a, int s, int i, Raw& r) {
        const int c = s / a.d.B, b = s % a.d.B;
        const long long t = (long long)i + 1;
        rd<R, D>(a.aux_x, c, t, b, r.x);
        if (a.aux_eps.ptr) rd<R, D>(a.aux_eps, c, t, b, r.ue);
        else rd<R, D>(a.aux_u, c, t, b, r.ue);
        rd<R, D>(a.aux_yobs, 0, t, 0, r.y);
        rd<R, D * D>(a.Fs, c, i, b, r.F);
        rd<R, D>(a.bs, c, i, b, r.bd);
        rd<R, D * D>(a.Qs, c, i, b, r.Q);
    }
    static AX_HD void step_info(const Args& a, int s, int i, const Raw& r, StepInfo<R, D>& si) {
        R u[D];
#pragma unroll
        for (int k = 0; k < D; ++k) u[k] = a.aux_eps.ptr ? r.x[k] + (R)a.aux_shd * r.ue[k] : r.ue[k];
        if constexpr (WRITE_U) {
            if (a.aux_eps.ptr) wr<R, D>(a.aux_u, s / a.d.B, (long long)i + 1, s % a.d.B, u);
        }
        sv_step_info<R, D>(a, r.x, u, r.y, si);
    }
    static AX_HD void write_out(const Args& a, int s, int i, const Pre& p) {
        if (!a.no_moments) Base::write_out(a, s, i, p);
    }
    using Carry = typename Base::Carry;
    static AX_HD void fold(const Args& a, int s, int i, const Raw& r, Full& acc, Carry& cy) {
        StepInfo<R, D> si;
        step_info(a, s, i, r, si);
        filter_fold_step<R, D, true>(r.F, r.Q, r.bd, si, acc, &cy);
    }
    static AX_HD void walk(const Args& a, int s, int i, const Raw& r, Pre& p, Carry& cy) {
        StepInfo<R, D> si;
        step_info(a, s, i, r, si);
        filter_apply_step<R, D, true>(r.F, r.Q, r.bd, si, p, &cy);
    }
};

// ---- many sequences, any model (lanes <-> sequences, the arrays read through their strides: e.g. the dense (C, T, B, .) layout with the batch index fastest:
// the reference's spatial example is B = 64 scalar LGSSMs, examples/spatial/model.py:103-112) -----------------------------------------------------------------
// No element buffer.  Reduce pass: a chunk's composite from elements built in registers (filter_build_elem: the dense S = H P H^T + R form, any R) and combined
// as they come.  Final pass: from the chunk's prefix -- the filtering distribution at the chunk's first step -- the reference's SEQUENTIAL recursion itself
// (predict + sequential_update, filtering.py:66-130), so moments and log-likelihood increments are the sequential filter's given the prefix; the increments are
// summed per chunk (nansum, filtering.py:62).  One chunk (parallel = 0, or few steps): exactly the sequential filter, one sequence per lane.
template <typename R_, int D, int P, int P1> struct FilterOpBuildCm : FilterOp<R_, D> {
    using R = R_;
    using Full = typename FilterOp<R_, D>::Full;
    using Args = FilterArgs;
    using Raw = Full;
    static AX_HD void load_elem(const Args& a, int s, int i, Full& e) {
        DirectIO io;
        filter_build_elem<R, D, P, DirectIO, P1>(a, io, s, i, true, e);
    }
    static AX_HD void load_raw(const Args& a, int s, int i, Raw& r) { load_elem(a, s, i, r); }
    static AX_HD void build(const Args&, int, int, const Raw& r, Full& e) { e = r; }
};
template <typename R_, int D, int P> struct FilterOpSeqWalk : FilterOp<R_, D> {
    using R = R_;
    using Full = typename FilterOp<R_, D>::Full;
    using Pre = typename FilterOp<R_, D>::Pre;
    using Args = FilterArgs;
    static constexpr bool kFold = true;
    struct Raw {
        R F[D * D], Q[D * D], bd[D], H[P * D], cv[P], y[P], Rm[P * P];
    };
    struct Carry {};
    static AX_HD void carry_flush(const Carry&, R&) {}
    static AX_HD void load_raw(const Args& a, int s, int i, Raw& r) {
        const int c = s / a.d.B, b = s % a.d.B;
        const long long t = (long long)i + 1;
        rd<R, D * D>(a.Fs, c, i, b, r.F);
        rd<R, D>(a.bs, c, i, b, r.bd);
        rd<R, D * D>(a.Qs, c, i, b, r.Q);
        rd<R, P * D>(a.Hs, c, t, b, r.H);
        rd<R, P>(a.cs, c, t, b, r.cv);
        rd<R, P>(a.ys, c, t, b, r.y);
        rd_upper<R, P>(a.Rs, c, t, b, r.Rm);
    }
    static AX_HD void init_pre(const Args& a, int s, Pre& p) {
        R P0p[D * D];
        rd<R, D>(a.ms, s / a.d.B, 0, s % a.d.B, p.b);
        rd_cov<R, D>(a.Ps, s / a.d.B, 0, s % a.d.B, a.ps_packed, P0p);
        sympack<R, D>(P0p, p.C);
        p.z = 0;
    }
    static AX_HD void walk(const Args&, int, int, const Raw& r, Pre& p, Carry&) {
        R Pd[D * D], m_[D], FP[D * D], Pn[D * D];
        symunpack<R, D>(p.C, Pd);
        mv<R, D, D>(r.F, p.b, m_);
        mm<R, D, D, D>(r.F, Pd, FP);
        mmt<R, D, D, D>(FP, r.F, Pn);
#pragma unroll
        for (int k = 0; k < D; ++k) m_[k] += r.bd[k];
#pragma unroll
        for (int k = 0; k < D * D; ++k) Pn[k] += r.Q[k];
        const R ell = kalman_update<R, D, P>(m_, Pn, r.H, r.cv, r.Rm, r.y);
#pragma unroll
        for (int k = 0; k < D; ++k) p.b[k] = m_[k];
        sympack<R, D>(Pn, p.C);
        p.z += isnan_(ell) ? (R)0 : ell;
    }
    static AX_HD void write_out(const Args& a, int s, int i, const Pre& p) {
        const int c = s / a.d.B, b = s % a.d.B;
        wr<R, D>(a.ms, c, (long long)i + 1, b, p.b);
        R Pd[D * D];
        symunpack<R, D>(p.C, Pd);
        wr_cov<R, D>(a.Ps, c, (long long)i + 1, b, a.ps_packed, Pd);
    }
    static AX_HD void write_zpart(const Args& a, int s, int ch, int nchunk, R z) { ((R*)a.ellz)[(long long)s * nchunk + ch] = z; }
};

// ---- sampler ------------------------------------------------------------------------------------------
struct SampleArgs {
    KDims d;
    Arr Fs, Qs, bs;
    Arr ms;           // (C,T,B,D) dense or chain-minor
    Arr Ps;
    Arr eps;          // (C,T,B,D)
    Arr xs;           // (C,T,B,D) output
    void* elem;       // scan elements (layout `lay`), scan position j = T-1-t
    ScanLayout lay;
    int dx = 0;       // runtime size (wide.hip only)
    int ps_shared = 0;        // != 0: Ps (and Fs, Qs, bs) do not depend on the chain (filter ran on chain-shared parameters)
    int ps_packed = 0;        // != 0: Ps records are symmetric-packed (as FilterArgs::ps_packed)
    int eps_gen = 0;          // != 0: eps is generated by the reduce pass from (gen_k0, gen_k1) and written for the down pass (as FilterArgs::aux_gen)
    unsigned int gen_k0 = 0, gen_k1 = 0;
    const void* tab = nullptr;  // then: one SampShared row per time step
    const int* memo = nullptr;  // model-stage memo (FilterArgs::memo)
};

template <typename R_, int D> struct SampleOp;

// scan element j = jp + 1 <-> time t = T-2-jp, jp in [0, T-1)   (_sampling_init_one, sampling.py:108-112)
// lanes walk DOWN in time, hence the negative lane strides.
template <typename R, int D, class IO> AX_HD void body_sample_init(const SampleArgs& a, IO& io, int s, int jp, bool valid) {
    const int c = s / a.d.B, b = s % a.d.B;
    const int T = a.d.T;
    const long long t = (long long)T - 2 - jp;
    R m[D], Pd[D * D], eps[D], F[D * D], Q[D * D], bd[D];
    io.template fetch<R, D>(at<R>(a.ms, c, t, b), -a.ms.st, a.ms.se, valid, m);
    io.template fetch<R, D * D>(at<R>(a.Ps, c, t, b), -a.Ps.st, a.Ps.se, valid, Pd);
    io.template fetch<R, D>(at<R>(a.eps, c, t, b), -a.eps.st, a.eps.se, valid, eps);
    io.template fetch<R, D * D>(at<R>(a.Fs, c, t, b), -a.Fs.st, a.Fs.se, valid, F);
    io.template fetch<R, D * D>(at<R>(a.Qs, c, t, b), -a.Qs.st, a.Qs.se, valid, Q);
    io.template fetch<R, D>(at<R>(a.bs, c, t, b), -a.bs.st, a.bs.se, valid, bd);
    io.template finish<R, D>(-a.ms.st, a.ms.se, valid, m);
    io.template finish<R, D * D>(-a.Ps.st, a.Ps.se, valid, Pd);
    io.template finish<R, D>(-a.eps.st, a.eps.se, valid, eps);
    io.template finish<R, D * D>(-a.Fs.st, a.Fs.se, valid, F);
    io.template finish<R, D * D>(-a.Qs.st, a.Qs.se, valid, Q);
    io.template finish<R, D>(-a.bs.st, a.bs.se, valid, bd);
    if (!valid) return;
    SampElem<R, D> e;
    sample_elem<R, D>(F, Q, bd, m, Pd, eps, e);
    SampleOp<R, D>::store_elem(a, s, jp + 1, e);
}
// scan element 0 <-> t = T-1   (_sample_last_step, sampling.py:115-124)
template <typename R, int D> AX_HD void body_sample_last(const SampleArgs& a, int s) {
    const int c = s / a.d.B, b = s % a.d.B;
    const long long tl = (long long)a.d.T - 1;
    R m[D], Pd[D * D], eps[D];
    rd<R, D>(a.ms, c, tl, b, m);
    rd_cov<R, D>(a.Ps, c, tl, b, a.ps_packed, Pd);
    rd<R, D>(a.eps, c, tl, b, eps);
    SampElem<R, D> e;
    sample_last<R, D>(m, Pd, eps, e);
    SampleOp<R, D>::store_elem(a, s, 0, e);
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
        wr<R, D>(a.xs, s / a.d.B, (long long)a.d.T - 1 - j, s % a.d.B, p.e);
    }
    static AX_HD void pack(const Full& e, R* t) {
#pragma unroll
        for (int i = 0; i < D * D; ++i) t[i] = e.G[i];
#pragma unroll
        for (int i = 0; i < D; ++i) t[D * D + i] = e.e[i];
    }
    static AX_HD void store_elem(const Args& a, int s, int j, const Full& e) {
        R* p = (R*)a.elem + a.lay.eoff(s, j, Full::NPAD);
        if (a.lay.cm) {
            R t[Full::N];
            pack(e, t);
            sts_<R, Full::N>(p, a.lay.es(), t);
        } else {
            store_rec(p, e);
        }
    }
    static AX_HD void load_elem(const Args& a, int s, int j, Full& e) {
        const R* p = (const R*)a.elem + a.lay.eoff(s, j, Full::NPAD);
        if (a.lay.cm) {
            R t[Full::N];
            lds_<R, Full::N>(p, a.lay.es(), t);
            unpack(t, e);
        } else {
            load_rec(p, e);
        }
    }
    using Raw = Full;
    static AX_HD void load_raw(const Args& a, int s, int j, Raw& r) { load_elem(a, s, j, r); }
    static AX_HD void build(const Args&, int, int, const Raw& r, Full& e) { e = r; }
    static constexpr bool kFold = false;
};

// SampleOp that never materialises its elements: load_elem builds (G_t, inc_t) from the filtered moments, the dynamics and the
// noise (sampling.py:60-124) each time a pass needs it.  Both scan passes then stream ms / Ps / eps (d^2 + 2d reals per step)
// instead of writing and re-reading the (d^2 + d)-real element buffer, and the sample-init launch disappears; the extra
// arithmetic (one d x d Cholesky + SPD solve per step and pass) is small next to the saved HBM traffic.  Chain-minor mode only.
template <typename R_, int D> struct SampleOpFly : SampleOp<R_, D> {
    using R = R_;
    using Full = typename SampleOp<R_, D>::Full;
    struct Raw {
        R m[D], Pd[D * D], eps[D], F[D * D], Q[D * D], bd[D];
    };
    static AX_HD void load_raw(const SampleArgs& a, int s, int j, Raw& r) {
        const int c = s / a.d.B, b = s % a.d.B;
        const long long t = (long long)a.d.T - 1 - j;
        rd<R, D>(a.ms, c, t, b, r.m);
        rd_cov<R, D>(a.Ps, c, t, b, a.ps_packed, r.Pd);
        rd<R, D>(a.eps, c, t, b, r.eps);
        if (j != 0) {
            rd<R, D * D>(a.Fs, c, t, b, r.F);
            rd<R, D * D>(a.Qs, c, t, b, r.Q);
            rd<R, D>(a.bs, c, t, b, r.bd);
        }
    }
    static AX_HD void build(const SampleArgs&, int, int j, const Raw& r, Full& e) {
        if (j == 0) sample_last<R, D>(r.m, r.Pd, r.eps, e);
        else sample_elem<R, D>(r.F, r.Q, r.bd, r.m, r.Pd, r.eps, e);
    }
    static AX_HD void load_elem(const SampleArgs& a, int s, int j, Full& e) {
        Raw r;
        load_raw(a, s, j, r);
        build(a, s, j, r, e);
    }
};

// chain-shared covariances: gain / Cholesky table row of time t (chain 0's copy of the shared Ps), and the op that reads it
template <typename R, int D> AX_HD void body_sample_shared_tab(const SampleArgs& a, int t) {
    using T = SampShared<R, D>;
    R Pd[D * D], F[D * D], Q[D * D], bd[D], row[T::N];
    rd<R, D * D>(a.Ps, 0, t, 0, Pd);
    const bool last = t == a.d.T - 1;
    if (!last) {
        rd<R, D * D>(a.Fs, 0, t, 0, F);
        rd<R, D * D>(a.Qs, 0, t, 0, Q);
        rd<R, D>(a.bs, 0, t, 0, bd);
    }
    sample_shared_row<R, D>(F, Q, bd, Pd, last, row);
    stv<R, T::N>((R*)a.tab + (long long)t * T::NPAD, row);
}
// ---- joint log-density of a trajectory: log_likelihood + prior_logpdf (base.py:99-166) ----------------
struct LogpdfArgs {
    KDims d;
    Arr m0, P0, Fs, Qs, bs, Hs, Rs, cs, ys, xs;
    int nan_policy;  // 0 reference, 1 masked
    int dx = 0, dy = 0;  // runtime sizes (wide.hip only)
};

// observation term at time t + transition term into t (t >= 1); lanes are indexed by i = t - 1
template <typename R, int D, int P, class IO> AX_HD R body_joint_logpdf(const LogpdfArgs& a, IO& io, int s, int i, bool valid) {
    const int c = s / a.d.B, b = s % a.d.B;
    const long long t = (long long)i + 1;
    R x[D], xp[D], H[P * D], cv[P], y[P], Rm[P * P], F[D * D], bd[D], Q[D * D];
    io.template fetch<R, D>(at<R>(a.xs, c, t, b), a.xs.st, a.xs.se, valid, x);
    io.template fetch<R, D>(at<R>(a.xs, c, i, b), a.xs.st, a.xs.se, valid, xp);
    io.template fetch<R, P * D>(at<R>(a.Hs, c, t, b), a.Hs.st, a.Hs.se, valid, H);
    io.template fetch<R, P>(at<R>(a.cs, c, t, b), a.cs.st, a.cs.se, valid, cv);
    io.template fetch<R, P>(at<R>(a.ys, c, t, b), a.ys.st, a.ys.se, valid, y);
    io.template fetch_upper<R, P>(at<R>(a.Rs, c, t, b), a.Rs.se, valid, Rm);
    io.template fetch<R, D * D>(at<R>(a.Fs, c, i, b), a.Fs.st, a.Fs.se, valid, F);
    io.template fetch<R, D>(at<R>(a.bs, c, i, b), a.bs.st, a.bs.se, valid, bd);
    io.template fetch<R, D * D>(at<R>(a.Qs, c, i, b), a.Qs.st, a.Qs.se, valid, Q);
    io.template finish<R, D>(a.xs.st, a.xs.se, valid, x);
    io.template finish<R, D>(a.xs.st, a.xs.se, valid, xp);
    io.template finish<R, P * D>(a.Hs.st, a.Hs.se, valid, H);
    io.template finish<R, P>(a.cs.st, a.cs.se, valid, cv);
    io.template finish<R, P>(a.ys.st, a.ys.se, valid, y);
    io.template finish<R, D * D>(a.Fs.st, a.Fs.se, valid, F);
    io.template finish<R, D>(a.bs.st, a.bs.se, valid, bd);
    io.template finish<R, D * D>(a.Qs.st, a.Qs.se, valid, Q);
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
    rd<R, D>(a.xs, c, 0, b, x);
    rd<R, P * D>(a.Hs, c, 0, b, H);
    rd<R, P>(a.cs, c, 0, b, cv);
    rd<R, P>(a.ys, c, 0, b, y);
    rd<R, D>(a.m0, c, 0, b, m0);
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
    {
        R Rm[P * P];
        rd_upper<R, P>(a.Rs, c, 0, b, Rm);
        out += gauss_logpdf<R, P>(res, Rm, a.nan_policy == 1 ? skip : nullptr);
    }
    R r0[D], P0m[D * D];
    rd<R, D * D>(a.P0, c, 0, b, P0m);
#pragma unroll
    for (int k = 0; k < D; ++k) r0[k] = x[k] - m0[k];
    out += gauss_logpdf<R, D>(r0, P0m, nullptr);
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
    Arr x;                                   // (C,T,D)
    Arr xp;
    Arr u;
    double delta;
    int nan_policy;
    int dx = 0, po = 0;  // runtime sizes (wide.hip only)
    const void* tab = nullptr;  // chain-shared parameters: per time step the Cholesky rows of Q_{t-1} and Robs_t (else null)
    int tab_semi = 0;           // the table is built for the SEMI-shared pass (Q, H, R, c, y chain-shared; F, b per chain or rebuilt): its WF / wb entries are not formed
    const void* lor_par = nullptr;  // Lorenz-63 sweep: rows [theta1, theta2, theta3, dt], chain stride lor_psc (Fs / bs are not read then)
    long long lor_psc = 0;
    // u_fly != 0: rows t >= 1 of u are not materialised (the filter built them on the fly, FilterArgs::aux_*): u_t = x_t + shd eps_t
    int u_fly = 0;
    Arr eps_aux{};
    double shd = 0;
    const double* dptr = nullptr;  // device-resident {delta, sqrt(delta / 2)}; null: the host values
    const int* memo = nullptr;     // model-stage memo (FilterArgs::memo)
    // wide-state shared form only (wide_shared.h::wk_lp_cols): ys may be per chain (chain stride != 0; reference NaN policy only), and ys_x -- when set -- is the
    // observation array scored against x while ys is scored against xp (the SV sweep's two pseudo-observation sets: one launch gives both joint densities)
    Arr ys_x{nullptr, 0, 0, 0, 1};
};
// the auxiliary variable of chain c at time t >= 1
template <typename R, int D> AX_HD void sweep_u(const SweepLogpdfArgs& a, int c, long long t, const R* x, R* u) {
    if (a.u_fly) {
        R e[D];
        rd<R, D>(a.eps_aux, c, t, 0, e);
#pragma unroll
        for (int k = 0; k < D; ++k) u[k] = x[k] + (R)arg_shd(a) * e[k];
    } else {
        rd<R, D>(a.u, c, t, 0, u);
    }
}
// table row of the chain-shared log-density pass, WHITENED: with W = chol(.)^-1 (lower triangular; a deleted component's row and column are zero)
//   transition t-1 -> t:  z = WQ x_t - WF x_{t-1} - wb,   WF = WQ F, wb = WQ b          log N(x_t; F x_{t-1} + b, Q)     = -|z|^2 / 2 + cQ
//   observation at t:     z = yw - WH x_t,                WH = WR H, yw = WR (y - c)     log N(y_t; H x_t + c, R) (kept) = -|z|^2 / 2 + cR
// built once per time step by the model stage; a chain's step reads 52 scalars at d = po = 4 instead of the 74 of (F, b, H, c, y, two factors).
template <typename R, int D, int PO> struct LogShared {
    static constexpr int oWQ = 0, oWF = symsize(D), oWb = oWF + D * D, oCQ = oWb + D, oWH = oCQ + 1, oYw = oWH + PO * D, oCR = oYw + PO, N = oCR + 1;
    static constexpr int VEC = 16 / sizeof(R);
    static constexpr int NPAD = (N + VEC - 1) / VEC * VEC;
};

// the observation + auxiliary blocks at one time step for (xp, x); returns via references
// fo (optional): the log-determinant of the observation block is NOT taken; fo[0..3] receive the factors whose log / 2 ADDS to cc_p, cc_x, ob_p, ob_x (1 where
// the nansum rule dropped the term) -- the chain-minor kernels multiply them up over a time tile (kalman_math.h::LogProd: one logarithm per tile and sum)
template <typename R, int D, int PO>
AX_HD void sweep_obs_terms(const SweepLogpdfArgs& a, const R* x, const R* xp, const R* u, const R* H, const R* cv, const R* y,
                           const R* Rm, R& cc_p, R& cc_x, R& ob_p, R& ob_x, R& corr, R* fo = nullptr) {
    bool badobs_x = false, badobs_p = false;
    R fR = 1;
    bool kp = true, kx = true;
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
        if (fo) gauss_logpdf2_lp<R, PO>(r1, r2, Rm, ob_p, ob_x, fR, kp, kx, a.nan_policy == 1 ? skip : nullptr);
        else gauss_logpdf2<R, PO>(r1, r2, Rm, a.nan_policy == 1 ? skip : nullptr, ob_p, ob_x);
    }
    R ax_x, ax_p;
    bool b1 = false, b2 = false;
    corr = 0;
    {
        // (x - u)^2 / delta and the N(u; x, delta/2 I) terms share the squared distances: one reciprocal of delta, no division per component
        const R hd = (R)(0.5 * arg_delta(a));
        const R inv_delta = (R)1 / (R)arg_delta(a);
        R q1 = 0, q2 = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const R d1 = u[k] - xp[k], d2 = u[k] - x[k];
            b1 = b1 || !finite_(d1);
            b2 = b2 || !finite_(d2);
            q1 += d1 * d1;
            q2 += d2 * d2;
        }
        corr = (q1 - q2) * inv_delta;
        const R cst = (R)-0.5 * (R)D * log_(hd) - (R)(0.5 * LOG_2PI) * (R)D;
        ax_p = b1 ? (R)0 : -q1 * inv_delta + cst;
        ax_x = b2 ? (R)0 : -q2 * inv_delta + cst;
    }
    // reference policy (jnp.nansum over per-step logpdfs): a non-finite component anywhere in the stacked residual
    // [u - x ; y - H x - c] drops the whole step of the concatenated model; the target only sees the observation block.
    const bool ref = a.nan_policy == 0;
    const bool dp = ref && (b1 || badobs_p), dx_ = ref && (b2 || badobs_x);
    cc_p = dp ? (R)0 : ax_p + ob_p;
    cc_x = dx_ ? (R)0 : ax_x + ob_x;
    if (fo) {
        fo[2] = kp ? fR : (R)1;
        fo[3] = kx ? fR : (R)1;
        fo[0] = dp ? (R)1 : fo[2];
        fo[1] = dx_ ? (R)1 : fo[3];
    }
}

// the five terms of one time step t >= 1 from values in registers: x, xp at t; u_in = u_t (or eps_t when a.u_fly); xq, xpq at t - 1;
// the real observation model at t (Rm: upper triangle) and the transition t - 1 -> t
// the prior pair's factor joins the four sums' factors (fac[0..3]; sweep_obs_terms)
template <typename R> AX_HD void sweep_join_prior(R fq, bool kp, bool kx, R* fac) {
    fac[0] *= kp ? fq : (R)1;
    fac[1] *= kx ? fq : (R)1;
    fac[2] *= kp ? fq : (R)1;
    fac[3] *= kx ? fq : (R)1;
}
template <typename R, int D, int PO>
AX_HD void sweep_logpdf_core(const SweepLogpdfArgs& a, const R* x, const R* xp, const R* u_in, const R* xq, const R* xpq, const R* H, const R* cv,
                             const R* y, const R* Rm, const R* F, const R* bd, const R* Q, R* out5, R* fac = nullptr) {
    R u[D];
#pragma unroll
    for (int k = 0; k < D; ++k) u[k] = a.u_fly ? x[k] + (R)arg_shd(a) * u_in[k] : u_in[k];
    R cc_p, cc_x, ob_p, ob_x, corr;
    sweep_obs_terms<R, D, PO>(a, x, xp, u, H, cv, y, Rm, cc_p, cc_x, ob_p, ob_x, corr, fac);
    R pr_p, pr_x;
    {
        R r1[D], r2[D], m1[D], m2[D];
        mv<R, D, D>(F, xpq, m1);
        mv<R, D, D>(F, xq, m2);
#pragma unroll
        for (int k = 0; k < D; ++k) r1[k] = xp[k] - (m1[k] + bd[k]), r2[k] = x[k] - (m2[k] + bd[k]);
        if (fac) {
            R fq;
            bool kp, kx;
            gauss_logpdf2_lp<R, D>(r1, r2, Q, pr_p, pr_x, fq, kp, kx);
            sweep_join_prior<R>(fq, kp, kx, fac);
        } else {
            gauss_logpdf2<R, D>(r1, r2, Q, nullptr, pr_p, pr_x);
        }
    }
    out5[0] = cc_p + pr_p;
    out5[1] = cc_x + pr_x;
    out5[2] = ob_p + pr_p;
    out5[3] = ob_x + pr_x;
    out5[4] = corr;
}

// lanes indexed by i = t - 1 (t >= 1)
template <typename R, int D, int PO, class IO>
AX_HD void body_sweep_logpdf(const SweepLogpdfArgs& a, IO& io, int c, int i, bool valid, R* out5) {
    const long long t = (long long)i + 1;
    R x[D], xp[D], u[D], xq[D], xpq[D], H[PO * D], cv[PO], y[PO], Rm[PO * PO], F[D * D], bd[D], Q[D * D];
    io.template fetch<R, D>(at<R>(a.x, c, t, 0), a.x.st, a.x.se, valid, x);
    io.template fetch<R, D>(at<R>(a.xp, c, t, 0), a.xp.st, a.xp.se, valid, xp);
    const Arr& ua = a.u_fly ? a.eps_aux : a.u;
    io.template fetch<R, D>(at<R>(ua, c, t, 0), ua.st, ua.se, valid, u);
    io.template fetch<R, D>(at<R>(a.x, c, i, 0), a.x.st, a.x.se, valid, xq);
    io.template fetch<R, D>(at<R>(a.xp, c, i, 0), a.xp.st, a.xp.se, valid, xpq);
    io.template fetch<R, PO * D>(at<R>(a.Hs, c, t, 0), a.Hs.st, a.Hs.se, valid, H);
    io.template fetch<R, PO>(at<R>(a.cs, c, t, 0), a.cs.st, a.cs.se, valid, cv);
    io.template fetch<R, PO>(at<R>(a.ys, c, t, 0), a.ys.st, a.ys.se, valid, y);
    io.template fetch_upper<R, PO>(at<R>(a.Rs, c, t, 0), a.Rs.se, valid, Rm);
    io.template fetch<R, D * D>(at<R>(a.Fs, c, i, 0), a.Fs.st, a.Fs.se, valid, F);
    io.template fetch<R, D>(at<R>(a.bs, c, i, 0), a.bs.st, a.bs.se, valid, bd);
    io.template fetch<R, D * D>(at<R>(a.Qs, c, i, 0), a.Qs.st, a.Qs.se, valid, Q);
    io.template finish<R, D>(a.x.st, a.x.se, valid, x);
    io.template finish<R, D>(a.xp.st, a.xp.se, valid, xp);
    io.template finish<R, D>(ua.st, ua.se, valid, u);
    io.template finish<R, D>(a.x.st, a.x.se, valid, xq);
    io.template finish<R, D>(a.xp.st, a.xp.se, valid, xpq);
    io.template finish<R, PO * D>(a.Hs.st, a.Hs.se, valid, H);
    io.template finish<R, PO>(a.cs.st, a.cs.se, valid, cv);
    io.template finish<R, PO>(a.ys.st, a.ys.se, valid, y);
    io.template finish<R, D * D>(a.Fs.st, a.Fs.se, valid, F);
    io.template finish<R, D>(a.bs.st, a.bs.se, valid, bd);
    io.template finish<R, D * D>(a.Qs.st, a.Qs.se, valid, Q);
#pragma unroll
    for (int k = 0; k < 5; ++k) out5[k] = 0;
    if (!valid) return;
    sweep_logpdf_core<R, D, PO>(a, x, xp, u, xq, xpq, H, cv, y, Rm, F, bd, Q, out5);
}
// t = 0 terms (one lane per chain)
template <typename R, int D, int PO> AX_HD void body_sweep_logpdf_head(const SweepLogpdfArgs& a, int c, R* out5) {
    R x[D], xp[D], u[D], H[PO * D], cv[PO], y[PO], m0[D], Rm[PO * PO], P0m[D * D];
    rd<R, D>(a.x, c, 0, 0, x);
    rd<R, D>(a.xp, c, 0, 0, xp);
    rd<R, D>(a.u, c, 0, 0, u);
    rd_upper<R, PO>(a.Rs, c, 0, 0, Rm);
    rd<R, D * D>(a.P0, c, 0, 0, P0m);
    rd<R, PO * D>(a.Hs, c, 0, 0, H);
    rd<R, PO>(a.cs, c, 0, 0, cv);
    rd<R, PO>(a.ys, c, 0, 0, y);
    rd<R, D>(a.m0, c, 0, 0, m0);
    R cc_p, cc_x, ob_p, ob_x, corr;
    sweep_obs_terms<R, D, PO>(a, x, xp, u, H, cv, y, Rm, cc_p, cc_x, ob_p, ob_x, corr);
    R r1[D], r2[D], pr_p, pr_x;
#pragma unroll
    for (int k = 0; k < D; ++k) r1[k] = xp[k] - m0[k], r2[k] = x[k] - m0[k];
    gauss_logpdf2<R, D>(r1, r2, P0m, nullptr, pr_p, pr_x);
    out5[0] = cc_p + pr_p;
    out5[1] = cc_x + pr_x;
    out5[2] = ob_p + pr_p;
    out5[3] = ob_x + pr_x;
    out5[4] = corr;
}

// ---- the Lorenz-63 sweep's log-densities in one pass (examples/lorenz/auxiliary_kalman.py:14-52 + kalman/generic.py:88-89, :98-106) ----
// mean(x) = x + dt (phi_0(x) + theta * phi(x)) (model.py:10-25); its Jacobian in closed form (the reference's jacfwd, linearisation.py:11-44)
template <typename R> AX_HD void lorenz_mean(const R* th, R dt, const R* x, R* mu) {
    mu[0] = x[0] + dt * (th[0] * (x[1] - x[0]));
    mu[1] = x[1] + dt * (th[1] * x[0] - x[1] - x[0] * x[2]);
    mu[2] = x[2] + dt * (x[0] * x[1] - th[2] * x[2]);
}
template <typename R> AX_HD void lorenz_lin_F(const R* th, R dt, const R* x, R* F) {
    const R J[9] = {-th[0], th[0], 0, th[1] - x[2], (R)-1, -x[0], x[1], x[0], -th[2]};
#pragma unroll
    for (int k = 0; k < 9; ++k) F[k] = ((k / 3 == k % 3) ? (R)1 : (R)0) + dt * J[k];
}
// per chain the five sums  [0] joint of the auxiliary LGSSM linearised at x, evaluated at x'   [1] the one linearised at x', evaluated at x
//                          [2] target(x')   [3] target(x)   [4] the MH correction;  the linearised transition F_t z + b_t is rebuilt from the
// linearisation point (F_t = I + dt J(x_t), b_t = mean(x_t) - F_t x_t), so no F / b array is read.  lanes indexed by i = t - 1.
template <typename R, int PO> AX_HD void body_lorenz_logpdf(const SweepLogpdfArgs& a, int c, int i, bool valid, R* out5, R* fac = nullptr) {
    constexpr int D = 3;
#pragma unroll
    for (int k = 0; k < 5; ++k) out5[k] = 0;
    if (fac) fac[0] = fac[1] = fac[2] = fac[3] = 1;
    if (!valid) return;
    const long long t = (long long)i + 1;
    const R* par = (const R*)a.lor_par + (long long)c * a.lor_psc;
    const R th[3] = {par[0], par[1], par[2]};
    const R dt = par[3];
    R x[D], xp[D], u[D], xq[D], xpq[D], H[PO * D], cv[PO], y[PO], Rm[PO * PO], Q[D * D];
    rd<R, D>(a.x, c, t, 0, x);
    rd<R, D>(a.xp, c, t, 0, xp);
    sweep_u<R, D>(a, c, t, x, u);
    rd<R, D>(a.x, c, i, 0, xq);
    rd<R, D>(a.xp, c, i, 0, xpq);
    rd<R, PO * D>(a.Hs, c, t, 0, H);
    rd<R, PO>(a.cs, c, t, 0, cv);
    rd<R, PO>(a.ys, c, t, 0, y);
    rd_upper<R, PO>(a.Rs, c, t, 0, Rm);
    rd<R, D * D>(a.Qs, c, i, 0, Q);
    R cc_p, cc_x, ob_p, ob_x, corr;
    sweep_obs_terms<R, D, PO>(a, x, xp, u, H, cv, y, Rm, cc_p, cc_x, ob_p, ob_x, corr, fac);
    R mx[D], mp[D], F1[D * D], F2[D * D], dl[D], f1[D], f2[D];
    lorenz_mean<R>(th, dt, xq, mx);
    lorenz_mean<R>(th, dt, xpq, mp);
    lorenz_lin_F<R>(th, dt, xq, F1);
    lorenz_lin_F<R>(th, dt, xpq, F2);
#pragma unroll
    for (int k = 0; k < D; ++k) dl[k] = xpq[k] - xq[k];
    mv<R, D, D>(F1, dl, f1);
    mv<R, D, D>(F2, dl, f2);
    R r1[D], r2[D], rp[D], rx[D], l1, l2, tp, tx;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        r1[k] = xp[k] - (mx[k] + f1[k]);  // x'_t - (F1 x'_{t-1} + b1)
        r2[k] = x[k] - (mp[k] - f2[k]);   // x_t  - (F2 x_{t-1}  + b2)
        rp[k] = xp[k] - mp[k];
        rx[k] = x[k] - mx[k];
    }
    if (fac) {
        R fq, fq2;
        bool k1, k2, k3, k4;
        gauss_logpdf2_lp<R, D>(r1, r2, Q, l1, l2, fq, k1, k2);
        gauss_logpdf2_lp<R, D>(rp, rx, Q, tp, tx, fq2, k3, k4);   // (the same factorisation: the compiler merges the two)
        fac[0] *= k1 ? fq : (R)1;
        fac[1] *= k2 ? fq : (R)1;
        fac[2] *= k3 ? fq2 : (R)1;
        fac[3] *= k4 ? fq2 : (R)1;
    } else {
        gauss_logpdf2<R, D>(r1, r2, Q, nullptr, l1, l2);
        gauss_logpdf2<R, D>(rp, rx, Q, nullptr, tp, tx);
    }
    out5[0] = cc_p + l1;
    out5[1] = cc_x + l2;
    out5[2] = ob_p + tp;
    out5[3] = ob_x + tx;
    out5[4] = corr;
}
template <typename R, int PO> AX_HD void body_lorenz_logpdf_head(const SweepLogpdfArgs& a, int c, R* out5) {
    body_sweep_logpdf_head<R, 3, PO>(a, c, out5);  // at t = 0 the linear-Gaussian and the Lorenz models agree: N(m0, P0) + observations
}

// ---- the stochastic-volatility sweep's log-densities in one pass (examples/stochastic_volatility/auxiliary_kalman.py:22-48 +
// kalman/generic.py:88-89, :98-106): per chain the five sums
//   [0] log N(ys1; x', R1) + prior(x')   (joint of the proposal's auxiliary model at x')       [1] log N(ys2; x, R2) + prior(x)
//   [2] log g(x') + prior(x')            (target at x')                                        [3] log g(x) + prior(x)
//   [4] sum ((x' - u)^2 - (x - u)^2) / delta
// with g the SV potential, R1 / R2 the (diagonal) auxiliary observation covariances of the two linearisation points (null: delta/2 I)
// and a NaN auxiliary term dropped per time step (the reference's nansum).  Any layout: every array is read through its strides.
struct SvLogpdfArgs {
    KDims d;
    Arr m0, P0, Fs, Qs, bs;  // linear-Gaussian dynamics
    Arr yobs;                // (T, D), chain-shared
    Arr x, xp, u, ys1, ys2;  // (C, T, D)
    Arr R1, R2;              // (C, T, D, D), ptr null for the first-order factory
    double delta;
    const double* dptr = nullptr;  // device-resident {delta, sqrt(delta / 2)}; null: the host value
    int fly_order = 0;             // 1 / 2: ys1, ys2, R1, R2 do not exist (FilterArgs::sv_order): the pseudo-observations are re-formed from (x, xp, u, yobs)
};
// Kernels call resolve_step(a) on their by-value argument struct first: a device-resident step size (dptr) is read ONCE, into the host fields, and the
// pointer cleared, so that arg_delta / arg_shd / arg_aux_shd are loop-invariant scalars in every per-step body.  (Left to the bodies, the conditional
// load sat inside the time loops with an `s_waitcnt vmcnt(0)` behind it -- which also waited for the next step's prefetched reads.)
template <typename A> AX_HD void resolve_step(A&) {}
AX_HD void resolve_step(FilterArgs& a) {
    const double* p = a.dptr;
    a.dptr = nullptr;
    if (p) a.aux_shd = p[1], a.sv_delta = p[0];
}
AX_HD void resolve_step(SweepLogpdfArgs& a) {
    const double* p = a.dptr;
    a.dptr = nullptr;
    if (p) a.delta = p[0], a.shd = p[1];
}
AX_HD void resolve_step(SvLogpdfArgs& a) {
    const double* p = a.dptr;
    a.dptr = nullptr;
    if (p) a.delta = p[0];
}
// The logarithms of the step (the two auxiliary variances, the transition's determinant) are NOT taken here: fac[k] > 0 is the factor whose log / 2 ADDS to
// o5[k] (k < 4; 1 where the term was dropped by the nansum rule) -- the chain-minor kernel multiplies them up over its time tile (kalman_math.h::LogProd), the
// one-step-per-lane kernel takes the logarithm at once.
template <typename R, int D>
AX_HD void sv_step_terms(const SvLogpdfArgs& a, int c, long long t, const R* x, const R* xp, R* o5, R* fac) {
    R u[D], y[D], y1[D], y2[D];
    rd<R, D>(a.u, c, t, 0, u);
    rd<R, D>(a.yobs, 0, t, 0, y);
    if (!a.fly_order) {
        rd<R, D>(a.ys1, c, t, 0, y1);
        rd<R, D>(a.ys2, c, t, 0, y2);
    }
    const R delta = (R)arg_delta(a), a2 = (R)2 / delta;
    R pp = 0, px = 0, l1 = 0, l2 = 0, cr = 0, f1 = 1, f2 = 1;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R av = xp[k], bv = x[k];
        // the two exponentials serve the potentials and (fly) the pseudo-observations: 0.5 * (y y e) is (0.5 y) y e bit for bit (scaling by a power of two)
        const R wb = y[k] * y[k] * exp_(-bv), wa = y[k] * y[k] * exp_(-av);
        pp += nan_to_num<R>((R)(-0.5 * LOG_2PI) - (R)0.5 * av - (R)0.5 * wa);
        px += nan_to_num<R>((R)(-0.5 * LOG_2PI) - (R)0.5 * bv - (R)0.5 * wb);
        R t1, t2;   // the auxiliary precisions 1 / R1_kk, 1 / R2_kk
        if (a.fly_order) {
            // k_sv_obs's pseudo-observations, re-formed: the proposal's are linearised at x (and scored at xp), the reverse move's at xp
            const R gb = nan_to_num<R>((R)0.5 * (wb - (R)1)), ga = nan_to_num<R>((R)0.5 * (wa - (R)1));
            if (a.fly_order == 1) {
                y1[k] = u[k] + (R)0.5 * delta * gb;
                y2[k] = u[k] + (R)0.5 * delta * ga;
                t1 = t2 = a2;
            } else {
                const R hb = (R)-0.5 * wb, ha = (R)-0.5 * wa;
                t1 = -hb + a2;
                t2 = -ha + a2;
                const R iv = (R)1 / (t1 * t2);   // one division for the two variances Om = 1 / t
                y1[k] = (t2 * iv) * (a2 * u[k] + gb - hb * bv);
                y2[k] = (t1 * iv) * (a2 * u[k] + ga - ha * av);
            }
        } else {
            t1 = a.R1.ptr ? (R)1 / at<R>(a.R1, c, t, 0)[(long long)(k * D + k) * a.R1.se] : a2;
            t2 = a.R2.ptr ? (R)1 / at<R>(a.R2, c, t, 0)[(long long)(k * D + k) * a.R2.se] : a2;
        }
        const R d1 = y1[k] - av, d2 = y2[k] - bv;
        l1 += (R)-0.5 * d1 * d1 * t1 - (R)(0.5 * LOG_2PI);
        l2 += (R)-0.5 * d2 * d2 * t2 - (R)(0.5 * LOG_2PI);
        f1 *= t1;
        f2 *= t2;
        const R e1 = av - u[k], e2 = bv - u[k];
        cr += (e1 * e1 - e2 * e2) * ((R)0.5 * a2);
    }
    // (a non-positive or non-finite variance made log(sqrt(R)) NaN in the term-by-term form: the step is dropped, as there)
    const bool k1 = !isnan_(l1) && f1 > (R)0 && finite_(f1), k2 = !isnan_(l2) && f2 > (R)0 && finite_(f2);
    o5[0] = k1 ? l1 : (R)0;
    o5[1] = k2 ? l2 : (R)0;
    o5[2] = pp;
    o5[3] = px;
    o5[4] = cr;
    fac[0] = k1 ? f1 : (R)1;
    fac[1] = k2 ? f2 : (R)1;
    fac[2] = 1;
    fac[3] = 1;
}
template <typename R> AX_HD void sv_add_prior(R pr_p, R pr_x, R fq, bool kp, bool kx, R* out5, R* fac) {
    out5[0] += pr_p;
    out5[1] += pr_x;
    out5[2] += pr_p;
    out5[3] += pr_x;
    fac[0] *= kp ? fq : (R)1;
    fac[1] *= kx ? fq : (R)1;
    fac[2] *= kp ? fq : (R)1;
    fac[3] *= kx ? fq : (R)1;
}
// lanes indexed by i = t - 1 (t >= 1)
template <typename R, int D> AX_HD void body_sv_logpdf(const SvLogpdfArgs& a, int c, int i, bool valid, R* out5, R* fac) {
#pragma unroll
    for (int k = 0; k < 5; ++k) out5[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) fac[k] = 1;
    if (!valid) return;
    const long long t = (long long)i + 1;
    R x[D], xp[D], xq[D], xpq[D], F[D * D], bd[D], Q[D * D];
    rd<R, D>(a.x, c, t, 0, x);
    rd<R, D>(a.xp, c, t, 0, xp);
    rd<R, D>(a.x, c, i, 0, xq);
    rd<R, D>(a.xp, c, i, 0, xpq);
    rd<R, D * D>(a.Fs, c, i, 0, F);
    rd<R, D>(a.bs, c, i, 0, bd);
    rd<R, D * D>(a.Qs, c, i, 0, Q);
    sv_step_terms<R, D>(a, c, t, x, xp, out5, fac);
    R r1[D], r2[D], m1[D], m2[D], pr_p, pr_x, fq;
    bool kp, kx;
    mv<R, D, D>(F, xpq, m1);
    mv<R, D, D>(F, xq, m2);
#pragma unroll
    for (int k = 0; k < D; ++k) r1[k] = xp[k] - (m1[k] + bd[k]), r2[k] = x[k] - (m2[k] + bd[k]);
    gauss_logpdf2_lp<R, D>(r1, r2, Q, pr_p, pr_x, fq, kp, kx);
    sv_add_prior<R>(pr_p, pr_x, fq, kp, kx, out5, fac);
}
template <typename R, int D> AX_HD void body_sv_logpdf_head(const SvLogpdfArgs& a, int c, R* out5, R* fac) {
    R x[D], xp[D], m0[D], P0m[D * D];
    rd<R, D>(a.x, c, 0, 0, x);
    rd<R, D>(a.xp, c, 0, 0, xp);
    rd<R, D>(a.m0, c, 0, 0, m0);
    rd<R, D * D>(a.P0, c, 0, 0, P0m);
    sv_step_terms<R, D>(a, c, 0, x, xp, out5, fac);
    R r1[D], r2[D], pr_p, pr_x, fq;
    bool kp, kx;
#pragma unroll
    for (int k = 0; k < D; ++k) r1[k] = xp[k] - m0[k], r2[k] = x[k] - m0[k];
    gauss_logpdf2_lp<R, D>(r1, r2, P0m, pr_p, pr_x, fq, kp, kx);
    sv_add_prior<R>(pr_p, pr_x, fq, kp, kx, out5, fac);
}

// ---- the same pass with chain-shared parameters: Cholesky factors and log-determinants of Q_{t-1}, Robs_t once per time step ---
template <typename R, int N> AX_HD void chol_inverse_lower(const R* L, const R* invd, const bool* skip, R* W) {  // W = L^-1, lower-packed (lidx)
#pragma unroll
    for (int k = 0; k < N; ++k) {
        R e[N];
#pragma unroll
        for (int l = 0; l < N; ++l) e[l] = (l == k) ? (R)1 : (R)0;
        lsolve<R, N>(L, invd, e);
#pragma unroll
        for (int l = k; l < N; ++l) W[lidx(l, k)] = (skip && (skip[k] || skip[l])) ? (R)0 : e[l];
    }
}
template <typename R, int D, int PO> AX_HD void body_sweep_logpdf_tab(const SweepLogpdfArgs& a, int i) {
    using T = LogShared<R, D, PO>;
    using CQ = CholRow<R, D>;
    using CR = CholRow<R, PO>;
    const long long t = (long long)i + 1;
    R Q[D * D], Rm[PO * PO], y[PO], H[PO * D], cv[PO], F[D * D], bd[D], row[T::N], cq[CQ::SZ], cr[CR::SZ];
    rd<R, D * D>(a.Qs, 0, i, 0, Q);
    rd_upper<R, PO>(a.Rs, 0, t, 0, Rm);
    rd<R, PO>(a.ys, 0, t, 0, y);
    rd<R, PO * D>(a.Hs, 0, t, 0, H);
    rd<R, PO>(a.cs, 0, t, 0, cv);
    if (a.tab_semi) {
#pragma unroll
        for (int k = 0; k < D * D; ++k) F[k] = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) bd[k] = 0;
    } else {
        rd<R, D * D>(a.Fs, 0, i, 0, F);
        rd<R, D>(a.bs, 0, i, 0, bd);
    }
    bool skip[PO];
#pragma unroll
    for (int k = 0; k < PO; ++k) skip[k] = (a.nan_policy == 1) && !finite_(y[k]);
    chol_row<R, D>(Q, nullptr, cq);
    chol_row<R, PO>(Rm, a.nan_policy == 1 ? skip : nullptr, cr);
    R* WQ = row + T::oWQ;
    chol_inverse_lower<R, D>(cq + CQ::oL, cq + CQ::oI, nullptr, WQ);
#pragma unroll
    for (int r = 0; r < D; ++r) {
        R wb = 0;
#pragma unroll
        for (int l = 0; l <= r; ++l) wb += WQ[lidx(r, l)] * bd[l];
        row[T::oWb + r] = wb;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R v = 0;
#pragma unroll
            for (int l = 0; l <= r; ++l) v += WQ[lidx(r, l)] * F[l * D + j];
            row[T::oWF + r * D + j] = v;
        }
    }
    row[T::oCQ] = cq[CQ::oC];
    R WR[symsize(PO)];
    chol_inverse_lower<R, PO>(cr + CR::oL, cr + CR::oI, a.nan_policy == 1 ? skip : nullptr, WR);
#pragma unroll
    for (int r = 0; r < PO; ++r) {
        R yw = 0;
#pragma unroll
        for (int l = 0; l <= r; ++l) yw += skip[l] ? (R)0 : WR[lidx(r, l)] * (y[l] - cv[l]);  // (a NaN observation the policy keeps makes the row NaN: dropped below)
        row[T::oYw + r] = yw;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R v = 0;
#pragma unroll
            for (int l = 0; l <= r; ++l) v += WR[lidx(r, l)] * H[l * D + j];
            row[T::oWH + r * D + j] = v;
        }
    }
    row[T::oCR] = cr[CR::oC];
    stv<R, T::N>((R*)a.tab + (long long)i * T::NPAD, row);
}
// x, xp, u: the chain's values at time t = i + 1 (u: eps_t when a.u_fly); xq, xpq: at time t - 1.  The caller streams them (the
// previous step's x / xp stay in registers, the next step's reads are issued before this one is evaluated).
template <typename R, int D, int PO>
AX_HD void body_sweep_logpdf_shared(const SweepLogpdfArgs& a, int i, const R* x, const R* xp, const R* u_in, const R* xq, const R* xpq, R* out5) {
    using T = LogShared<R, D, PO>;
    const UniformRow<R> row = uniform_row<R>((const R*)a.tab + (long long)i * T::NPAD);
    R u[D];
#pragma unroll
    for (int k = 0; k < D; ++k) u[k] = a.u_fly ? x[k] + (R)arg_shd(a) * u_in[k] : u_in[k];
    // observation block, whitened rows (LogShared): z = yw - WH x; a deleted component's row is zero; a non-finite kept residual <=> a non-finite z
    // (W is triangular with a non-zero diagonal) drops the term, as does a failed factorisation (cR = NaN) -- the reference's nansum
    R ob_p, ob_x;
    bool badobs_x = false, badobs_p = false;
    {
        R q1 = 0, q2 = 0;
#pragma unroll
        for (int k = 0; k < PO; ++k) {
            R z1 = row[T::oYw + k], z2 = z1;
#pragma unroll
            for (int j = 0; j < D; ++j) z1 -= row[T::oWH + k * D + j] * xp[j], z2 -= row[T::oWH + k * D + j] * x[j];
            badobs_p = badobs_p || !finite_(z1);
            badobs_x = badobs_x || !finite_(z2);
            q1 += z1 * z1;
            q2 += z2 * z2;
        }
        ob_p = (R)-0.5 * q1 + row[T::oCR];
        ob_x = (R)-0.5 * q2 + row[T::oCR];
        if (badobs_p || isnan_(ob_p)) ob_p = 0;
        if (badobs_x || isnan_(ob_x)) ob_x = 0;
    }
    R ax_x, ax_p, corr = 0;
    bool b1 = false, b2 = false;
    {
        // (x - u)^2 / delta and the N(u; x, delta/2 I) terms share the squared distances: one reciprocal of delta, no division per component
        const R hd = (R)(0.5 * arg_delta(a));
        const R inv_delta = (R)1 / (R)arg_delta(a);
        R q1 = 0, q2 = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const R d1 = u[k] - xp[k], d2 = u[k] - x[k];
            b1 = b1 || !finite_(d1);
            b2 = b2 || !finite_(d2);
            q1 += d1 * d1;
            q2 += d2 * d2;
        }
        corr = (q1 - q2) * inv_delta;
        const R cst = (R)-0.5 * (R)D * log_(hd) - (R)(0.5 * LOG_2PI) * (R)D;
        ax_p = b1 ? (R)0 : -q1 * inv_delta + cst;
        ax_x = b2 ? (R)0 : -q2 * inv_delta + cst;
    }
    const bool ref = a.nan_policy == 0;
    const R cc_p = (ref && (b1 || badobs_p)) ? (R)0 : ax_p + ob_p;
    const R cc_x = (ref && (b2 || badobs_x)) ? (R)0 : ax_x + ob_x;
    R pr_p, pr_x;
    {
        // transition, whitened: z = WQ x_t - WF x_{t-1} - wb
        R q1 = 0, q2 = 0;
        bool bad1 = false, bad2 = false;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            R z1 = -row[T::oWb + k], z2 = z1;
#pragma unroll
            for (int l = 0; l <= k; ++l) z1 += row[T::oWQ + lidx(k, l)] * xp[l], z2 += row[T::oWQ + lidx(k, l)] * x[l];
#pragma unroll
            for (int j = 0; j < D; ++j) z1 -= row[T::oWF + k * D + j] * xpq[j], z2 -= row[T::oWF + k * D + j] * xq[j];
            bad1 = bad1 || !finite_(z1);
            bad2 = bad2 || !finite_(z2);
            q1 += z1 * z1;
            q2 += z2 * z2;
        }
        pr_p = (R)-0.5 * q1 + row[T::oCQ];
        pr_x = (R)-0.5 * q2 + row[T::oCQ];
        if (bad1 || isnan_(pr_p)) pr_p = 0;
        if (bad2 || isnan_(pr_x)) pr_x = 0;
    }
    out5[0] = cc_p + pr_p;
    out5[1] = cc_x + pr_x;
    out5[2] = ob_p + pr_p;
    out5[3] = ob_x + pr_x;
    out5[4] = corr;
}

// ---- SEMI-shared pass ("drop the per-chain Cholesky of Q and R ... when those are chain-shared even though F and b are not: the SV and
// Lorenz case"): the covariances, the observation model and the data are the chains' common ones -- their whitening rows come from the LogShared table -- while the
// transition's mean is the chain's own (per-chain F_t, b_t, or the Lorenz-63 step rebuilt from the chain's state).  Same rules as body_sweep_logpdf_shared.
template <typename R, int D, int PO, typename RowP>
AX_HD void semi_obs_terms(const SweepLogpdfArgs& a, RowP row, const R* x, const R* xp, const R* u, R& cc_p, R& cc_x, R& ob_p, R& ob_x, R& corr) {
    using T = LogShared<R, D, PO>;
    bool badobs_x = false, badobs_p = false;
    {
        R q1 = 0, q2 = 0;
#pragma unroll
        for (int k = 0; k < PO; ++k) {
            R z1 = row[T::oYw + k], z2 = z1;
#pragma unroll
            for (int j = 0; j < D; ++j) z1 -= row[T::oWH + k * D + j] * xp[j], z2 -= row[T::oWH + k * D + j] * x[j];
            badobs_p = badobs_p || !finite_(z1);
            badobs_x = badobs_x || !finite_(z2);
            q1 += z1 * z1;
            q2 += z2 * z2;
        }
        ob_p = (R)-0.5 * q1 + row[T::oCR];
        ob_x = (R)-0.5 * q2 + row[T::oCR];
        if (badobs_p || isnan_(ob_p)) ob_p = 0;
        if (badobs_x || isnan_(ob_x)) ob_x = 0;
    }
    R ax_x, ax_p;
    bool b1 = false, b2 = false;
    {
        const R hd = (R)(0.5 * arg_delta(a));
        const R inv_delta = (R)1 / (R)arg_delta(a);
        R q1 = 0, q2 = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const R d1 = u[k] - xp[k], d2 = u[k] - x[k];
            b1 = b1 || !finite_(d1);
            b2 = b2 || !finite_(d2);
            q1 += d1 * d1;
            q2 += d2 * d2;
        }
        corr = (q1 - q2) * inv_delta;
        const R cst = (R)-0.5 * (R)D * log_(hd) - (R)(0.5 * LOG_2PI) * (R)D;
        ax_p = b1 ? (R)0 : -q1 * inv_delta + cst;
        ax_x = b2 ? (R)0 : -q2 * inv_delta + cst;
    }
    const bool ref = a.nan_policy == 0;
    cc_p = (ref && (b1 || badobs_p)) ? (R)0 : ax_p + ob_p;
    cc_x = (ref && (b2 || badobs_x)) ? (R)0 : ax_x + ob_x;
}
// log N(r; 0, Q) of two residuals from the table's W_Q = chol(Q)^-1 and c_Q
template <typename R, int D, int PO, typename RowP> AX_HD void semi_prior(RowP row, const R* r1, const R* r2, R& o1, R& o2) {
    using T = LogShared<R, D, PO>;
    R q1 = 0, q2 = 0;
    bool bad1 = false, bad2 = false;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        R z1 = 0, z2 = 0;
#pragma unroll
        for (int l = 0; l <= k; ++l) z1 += row[T::oWQ + lidx(k, l)] * r1[l], z2 += row[T::oWQ + lidx(k, l)] * r2[l];
        bad1 = bad1 || !finite_(z1);
        bad2 = bad2 || !finite_(z2);
        q1 += z1 * z1;
        q2 += z2 * z2;
    }
    o1 = (R)-0.5 * q1 + row[T::oCQ];
    o2 = (R)-0.5 * q2 + row[T::oCQ];
    if (bad1 || isnan_(o1)) o1 = 0;
    if (bad2 || isnan_(o2)) o2 = 0;
}
// linear transitions with the chain's own F, b
template <typename R, int D, int PO>
AX_HD void body_sweep_logpdf_semi(const SweepLogpdfArgs& a, int i, const R* x, const R* xp, const R* u_in, const R* xq, const R* xpq, const R* F, const R* bd, R* out5) {
    using T = LogShared<R, D, PO>;
    const UniformRow<R> row = uniform_row<R>((const R*)a.tab + (long long)i * T::NPAD);
    R u[D];
#pragma unroll
    for (int k = 0; k < D; ++k) u[k] = a.u_fly ? x[k] + (R)arg_shd(a) * u_in[k] : u_in[k];
    R cc_p, cc_x, ob_p, ob_x, corr, pr_p, pr_x;
    semi_obs_terms<R, D, PO>(a, row, x, xp, u, cc_p, cc_x, ob_p, ob_x, corr);
    R r1[D], r2[D], m1[D], m2[D];
    mv<R, D, D>(F, xpq, m1);
    mv<R, D, D>(F, xq, m2);
#pragma unroll
    for (int k = 0; k < D; ++k) r1[k] = xp[k] - (m1[k] + bd[k]), r2[k] = x[k] - (m2[k] + bd[k]);
    semi_prior<R, D, PO>(row, r1, r2, pr_p, pr_x);
    out5[0] = cc_p + pr_p;
    out5[1] = cc_x + pr_x;
    out5[2] = ob_p + pr_p;
    out5[3] = ob_x + pr_x;
    out5[4] = corr;
}
// the Lorenz-63 sweep (body_lorenz_logpdf with the table's rows)
template <typename R, int PO> AX_HD void body_lorenz_logpdf_semi(const SweepLogpdfArgs& a, int c, int i, const R* x, const R* xp, const R* u_in, const R* xq, const R* xpq, R* out5) {
    constexpr int D = 3;
    using T = LogShared<R, D, PO>;
    const UniformRow<R> row = uniform_row<R>((const R*)a.tab + (long long)i * T::NPAD);
    const R* par = (const R*)a.lor_par + (long long)c * a.lor_psc;
    const R th[3] = {par[0], par[1], par[2]};
    const R dt = par[3];
    R u[D];
#pragma unroll
    for (int k = 0; k < D; ++k) u[k] = a.u_fly ? x[k] + (R)arg_shd(a) * u_in[k] : u_in[k];
    R cc_p, cc_x, ob_p, ob_x, corr;
    semi_obs_terms<R, D, PO>(a, row, x, xp, u, cc_p, cc_x, ob_p, ob_x, corr);
    R mx[D], mp[D], F1[D * D], F2[D * D], dl[D], f1[D], f2[D];
    lorenz_mean<R>(th, dt, xq, mx);
    lorenz_mean<R>(th, dt, xpq, mp);
    lorenz_lin_F<R>(th, dt, xq, F1);
    lorenz_lin_F<R>(th, dt, xpq, F2);
#pragma unroll
    for (int k = 0; k < D; ++k) dl[k] = xpq[k] - xq[k];
    mv<R, D, D>(F1, dl, f1);
    mv<R, D, D>(F2, dl, f2);
    R r1[D], r2[D], rp[D], rx[D], l1, l2, tp, tx;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        r1[k] = xp[k] - (mx[k] + f1[k]);
        r2[k] = x[k] - (mp[k] - f2[k]);
        rp[k] = xp[k] - mp[k];
        rx[k] = x[k] - mx[k];
    }
    semi_prior<R, D, PO>(row, r1, r2, l1, l2);
    semi_prior<R, D, PO>(row, rp, rx, tp, tx);
    out5[0] = cc_p + l1;
    out5[1] = cc_x + l2;
    out5[2] = ob_p + tp;
    out5[3] = ob_x + tx;
    out5[4] = corr;
}

}  // namespace ax

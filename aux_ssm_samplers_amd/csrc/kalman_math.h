// kalman_math.h -- per-lane bodies of the auxiliary-Kalman hot path (one lane = one time step or one
// scan element).  Reference semantics: aux_samplers/_primitives/kalman/{filtering,sampling,base}.py and
// _primitives/math/mvn/base.py; the citations below are relative to /root/reference/aux_samplers.
#pragma once
#include "smallmat.h"

namespace ax {

// ------------------------------------------------------------------------------------------------
// Scan element of the parallel filter: (A, b, C, eta, J), C and J symmetric-packed.
// Memory record: [A (D*D) | b (D) | C (DS) | eta (D) | J (DS) | z], padded to NPAD reals.
// ------------------------------------------------------------------------------------------------
// z = log of the element's scale factor: the element is the function  exp(z) N(x_k; A x_{k-1} + b, C) exp(eta^T x_{k-1} - x_{k-1}^T J x_{k-1} / 2)
// of (x_{k-1}, x_k) (Sarkka & Garcia-Fernandez 2021, eq. 12-13, with the normaliser the reference drops).  Carrying it through the
// scan makes the marginal log-likelihood the z of the total product, so the reference's second pass over the filtered moments
// (filtering.py:60-62) is not needed: same number, no extra memory pass.
template <typename R, int D> struct FiltElem {
    static constexpr int DS = symsize(D);
    static constexpr int N = D * D + 2 * D + 2 * DS + 1;
    static constexpr int VEC = 16 / sizeof(R);
    static constexpr int NPAD = (N + VEC - 1) / VEC * VEC;
    R A[D * D];
    R b[D];
    R C[DS];
    R eta[D];
    R J[DS];
    R z;
};
// Reduced prefix that the final pass carries: only (b, C) = filtered (mean, cov) (filtering.py:55 discards the rest)
template <typename R, int D> struct FiltPre {
    static constexpr int DS = symsize(D);
    static constexpr int N = D + DS + 1;
    static constexpr int VEC = 16 / sizeof(R);
    static constexpr int NPAD = (N + VEC - 1) / VEC * VEC;
    R b[D];
    R C[DS];
    R z;  // log-scale of the prefix = log p(y_1..k) (plus the t = 0 term added by the caller)
};

// records are 16-byte aligned (NPAD, workspace carve): whole-record 16-byte vector loads/stores
template <typename R, int D> AX_HD void fe_store(R* __restrict__ p, const FiltElem<R, D>& e) {
    constexpr int DS = symsize(D);
    constexpr int N = FiltElem<R, D>::N;
    R t[N];
#pragma unroll
    for (int i = 0; i < D * D; ++i) t[i] = e.A[i];
#pragma unroll
    for (int i = 0; i < D; ++i) t[D * D + i] = e.b[i], t[D * D + D + DS + i] = e.eta[i];
#pragma unroll
    for (int i = 0; i < DS; ++i) t[D * D + D + i] = e.C[i], t[D * D + 2 * D + DS + i] = e.J[i];
    t[N - 1] = e.z;
    stv<R, N>(p, t);
}
template <typename R, int D> AX_HD void fe_load(const R* __restrict__ p, FiltElem<R, D>& e) {
    constexpr int DS = symsize(D);
    constexpr int N = FiltElem<R, D>::N;
    R t[N];
    ldv<R, N>(p, t);
#pragma unroll
    for (int i = 0; i < D * D; ++i) e.A[i] = t[i];
#pragma unroll
    for (int i = 0; i < D; ++i) e.b[i] = t[D * D + i], e.eta[i] = t[D * D + D + DS + i];
#pragma unroll
    for (int i = 0; i < DS; ++i) e.C[i] = t[D * D + D + i], e.J[i] = t[D * D + 2 * D + DS + i];
    e.z = t[N - 1];
}
template <typename R, int D> AX_HD void fe_identity(FiltElem<R, D>& e) {
#pragma unroll
    for (int i = 0; i < D * D; ++i) e.A[i] = (i / D == i % D) ? (R)1 : (R)0;
#pragma unroll
    for (int i = 0; i < D; ++i) e.b[i] = 0, e.eta[i] = 0;
#pragma unroll
    for (int i = 0; i < symsize(D); ++i) e.C[i] = 0, e.J[i] = 0;
    e.z = 0;
}

// ------------------------------------------------------------------------------------------------
// Observation masking (filtering.py:89-100, :204-213): missing components (non-finite y) lose their rows
// of H and c; the matching rows/cols of S are treated as deleted by chol_packed(skip=nan).
// ------------------------------------------------------------------------------------------------
template <typename R, int D, int P>
AX_HD bool obs_mask(const R* y, const R* H, const R* c, bool* nan, R* H_, R* c_) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        nan[k] = !finite_(y[k]);
        any = any || !nan[k];
        c_[k] = nan[k] ? (R)0 : c[k];
#pragma unroll
        for (int j = 0; j < D; ++j) H_[k * D + j] = nan[k] ? (R)0 : H[k * D + j];
    }
    return any;
}

// S = H_ P H_^T + R_ (LOWER-packed, lidx), PHt = P H_^T.  Rm is read from memory entry by entry (keeps R out of VGPRs).
template <typename R, int D, int P>
AX_HD void innovation_cov(const R* Pd, const R* H_, const R* __restrict__ Rm, const bool* nan, R* PHt, R* S) {
    mmt<R, D, D, P>(Pd, H_, PHt);  // PHt[i][k] = sum_j P[i][j] H_[k][j]
#pragma unroll
    for (int k = 0; k < P; ++k)
#pragma unroll
        for (int l = k; l < P; ++l) {
            R s = 0;
#pragma unroll
            for (int j = 0; j < D; ++j) s += H_[k * D + j] * PHt[j * P + l];
            const R r = (nan[k] || nan[l]) ? (R)0 : Rm[k * P + l];
            S[lidx(l, k)] = s + r;
        }
}

// sequential_update (filtering.py:83-130).  m, Pd (dense) updated in place; returns ell increment.
template <typename R, int D, int P>
AX_HD R kalman_update(R* m, R* Pd, const R* H, const R* c, const R* __restrict__ Rm, const R* y) {
    bool nan[P];
    R H_[P * D], c_[P];
    const bool any = obs_mask<R, D, P>(y, H, c, nan, H_, c_);
    if (!any) return (R)0;  // _passthrough :127-130
    R yd[P];
    int dim = 0;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        R yh = c_[k];
#pragma unroll
        for (int j = 0; j < D; ++j) yh += H_[k * D + j] * m[j];
        yd[k] = nan[k] ? (R)0 : y[k] - yh;
        dim += nan[k] ? 0 : 1;
    }
    R PHt[D * P], L[symsize(P)], invd[P];
    innovation_cov<R, D, P>(Pd, H_, Rm, nan, PHt, L);
    R G[D * P];
    R ell;
    if constexpr (P == 1) {  // scalar branch :108-111
        const R s = L[0];
        const R sd = sqrt_(s);
        const R z = yd[0] / sd;
        ell = (R)-0.5 * z * z - log_(sd) - (R)(0.5 * LOG_2PI);
#pragma unroll
        for (int i = 0; i < D; ++i) G[i] = PHt[i] / s;
    } else {
        const bool ok = chol_inplace<R, P>(L, invd, nan);
        R z[P];
        R logdet = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            z[k] = yd[k];
            logdet += nan[k] ? (R)0 : log_(L[lidx(k, k)]);
        }
        lsolve<R, P>(L, invd, z);
        R q = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) q += z[k] * z[k];
        ell = (R)-0.5 * q - logdet - (R)(0.5 * LOG_2PI) * (R)dim;
        if (!ok) ell = r_nan<R>();
#pragma unroll
        for (int i = 0; i < D; ++i) {
            R g[P];
#pragma unroll
            for (int k = 0; k < P; ++k) g[k] = PHt[i * P + k];
            cho_solve<R, P>(L, invd, g);
#pragma unroll
            for (int k = 0; k < P; ++k) G[i * P + k] = ok ? g[k] : r_nan<R>();
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R s = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) s += G[i * P + k] * yd[k];
        m[i] += s;
    }
    // P - G S0 G^T with S0 G^T = (H_ P) on the observed block  (== PHt^T)
    R Pn[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R s = 0;
#pragma unroll
            for (int k = 0; k < P; ++k) s += G[i * P + k] * PHt[j * P + k];
            Pn[i * D + j] = Pd[i * D + j] - s;
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Pd[i * D + j] = (i == j) ? Pn[i * D + i] : (R)0.5 * (Pn[i * D + j] + Pn[j * D + i]);
    return isnan_(ell) ? (R)0 : ell;
}

// sequential_predict (filtering.py:134-139)
template <typename R, int D> AX_HD void kalman_predict(R* m, R* Pd, const R* F, const R* b, const R* Q) {
    R t[D], FP[D * D], Pn[D * D];
    mv<R, D, D>(F, m, t);
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = t[i] + b[i];
    mm<R, D, D, D>(F, Pd, FP);
    mmt<R, D, D, D>(FP, F, Pn);
#pragma unroll
    for (int i = 0; i < D * D; ++i) Pn[i] += Q[i];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Pd[i * D + j] = (i == j) ? Pn[i * D + i] : (R)0.5 * (Pn[i * D + j] + Pn[j * D + i]);
}

// element from the information quantities M = H^T S^-1 H (packed sym), vm = H^T S^-1 (y - H m_ - c), vb = H^T S^-1 (y - H b - c):
//   A = F - P_ M F;  b = m_ + P_ vm;  C = P_ - P_ M P_;  eta = F^T vb;  J = F^T M F
template <typename R, int D>
AX_HD void filter_elem_from_info(const R* F, const R* m_, const R* P_, const R* M, const R* vm, const R* vb, FiltElem<R, D>& e) {
    R PM[D * D], MF[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < D; ++k) s1 += P_[i * D + k] * M[sidx(D, k, j)], s2 += M[sidx(D, i, k)] * F[k * D + j];
            PM[i * D + j] = s1;
            MF[i * D + j] = s2;
        }
    R Cd[D * D], Jd[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R sb = m_[i], se = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R sa = F[i * D + j], sc = P_[i * D + j], sj = 0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                sa -= PM[i * D + k] * F[k * D + j];
                sc -= PM[i * D + k] * P_[k * D + j];
                sj += F[k * D + i] * MF[k * D + j];
            }
            e.A[i * D + j] = sa;
            Cd[i * D + j] = sc;
            Jd[i * D + j] = sj;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) sb += P_[i * D + k] * vm[k], se += F[k * D + i] * vb[k];
        e.b[i] = sb;
        e.eta[i] = se;
    }
    sympack<R, D>(Cd, e.C);
    sympack<R, D>(Jd, e.J);
}

// _filtering_init_one (filtering.py:196-250).  (m_, P_) are the *predicted* moments the element is built
// around: predict(m0+, P0+) for the first transition, (b, Q) for every other one (:188-192).
template <typename R, int D, int P>
AX_HD void filter_elem(const R* F, const R* bdyn, const R* m_, const R* P_, const R* H, const R* c,
                       const R* __restrict__ Rm, const R* y, FiltElem<R, D>& e) {
    bool nan[P];
    R H_[P * D], c_[P];
    const bool any = obs_mask<R, D, P>(y, H, c, nan, H_, c_);
    if (!any) {  // _passthrough :239-248
#pragma unroll
        for (int i = 0; i < D * D; ++i) e.A[i] = F[i];
#pragma unroll
        for (int i = 0; i < D; ++i) e.b[i] = m_[i], e.eta[i] = 0;
        sympack<R, D>(P_, e.C);
#pragma unroll
        for (int i = 0; i < symsize(D); ++i) e.J[i] = 0;
        e.z = 0;
        return;
    }
    // Information form of the same element.  With L L^T = S and W = L^-1 H_ (p x d):
    //   M = H_^T S^-1 H_ = W^T W,  v(r) = H_^T S^-1 r = W^T (L^-1 r)
    //   K H_ = P_ M,  K r = P_ v(r),  K S0 K^T = P_ M P_      (S0 = S on the observed block)
    //   A = F - P_ M F;  b = m_ + P_ v(y - H_ m_ - c_);  C = P_ - P_ M P_;  eta = F^T v(y - H_ b - c_);  J = F^T M F
    // i.e. filtering.py:224-236 with the p x d intermediates (S_invH_T, K, temp) contracted away: only d x d products
    // remain after one Cholesky and d + 2 forward substitutions.
    R L[symsize(P)], invd[P];
    {
        R PHt[D * P];
        innovation_cov<R, D, P>(P_, H_, Rm, nan, PHt, L);
    }
    const bool ok = chol_inplace<R, P>(L, invd, nan);
    R rm[P], rb[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        R hm = c_[k], hb = c_[k];
#pragma unroll
        for (int j = 0; j < D; ++j) hm += H_[k * D + j] * m_[j], hb += H_[k * D + j] * bdyn[j];
        rm[k] = nan[k] ? (R)0 : y[k] - hm;
        rb[k] = nan[k] ? (R)0 : y[k] - hb;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) lsolve_col<R, P, D>(L, invd, H_, j);  // H_ <- W = L^-1 H_
    lsolve<R, P>(L, invd, rm);
    lsolve<R, P>(L, invd, rb);
    {  // log N(y; H_ m_ + c_, S): the element's scale (== the ell increment of sequential_update, filtering.py:106-114)
        R q = 0, logdet = 0;
        int dim = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            q += rm[k] * rm[k];
            logdet += nan[k] ? (R)0 : log_(L[lidx(k, k)]);
            dim += nan[k] ? 0 : 1;
        }
        e.z = ok ? (R)-0.5 * q - logdet - (R)(0.5 * LOG_2PI) * (R)dim : r_nan<R>();
    }
    R M[symsize(D)], vm[D], vb[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R s1 = 0, s2 = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) s1 += H_[k * D + i] * rm[k], s2 += H_[k * D + i] * rb[k];
        vm[i] = ok ? s1 : r_nan<R>();
        vb[i] = ok ? s2 : r_nan<R>();
#pragma unroll
        for (int j = i; j < D; ++j) {
            R s = 0;
#pragma unroll
            for (int k = 0; k < P; ++k) s += H_[k * D + i] * H_[k * D + j];
            M[sidx_u(D, i, j)] = ok ? s : r_nan<R>();
        }
    }
    filter_elem_from_info<R, D>(F, m_, P_, M, vm, vb, e);
}

// ------------------------------------------------------------------------------------------------
// Information form for a BLOCK-DIAGONAL observation covariance R = blkdiag(R_1 (P1 x P1), R_2 (P-P1 x P-P1)), e.g. the
// auxiliary observations concatenated with the real ones (R = blkdiag(delta/2 I, Robs)).  With
//     Lam = H^T R^-1 H = sum_b H_b^T R_b^-1 H_b,   g(r) = H^T R^-1 r = sum_b H_b^T R_b^-1 r_b,
// the push-through / Woodbury / determinant identities give, for S = H P H^T + R,
//     H^T S^-1 H = (I + Lam P)^-1 Lam,   H^T S^-1 r = (I + Lam P)^-1 g(r),
//     r^T S^-1 r = r^T R^-1 r - g^T P (I + Lam P)^-1 g,   log|S| = log|R| + log|I + Lam P|,
// i.e. the same M, v and log-likelihood as the dense p x p Cholesky of S (filtering.py:106-117, :214-236) from two small
// Cholesky factors and one d x d LU.  Missing components are deleted inside their block (skip), as in the dense path.
// Needs R_b positive definite; the dense path stays the default and the only one for general R.
// ------------------------------------------------------------------------------------------------
// One block: rows [O, O+PB) of H (P x D), c, y and the diagonal block of R (upper entries of the P x P record).
// Accumulates Lam (packed sym D), g1 = H_b^T R_b^-1 r1, g2 (second residual, may alias the first), q1 = r1^T R_b^-1 r1,
// logdet += sum log L_kk, dim += #observed.
template <typename R, int D, int P, int O, int PB>
AX_HD bool info_block(const R* H, const R* Rm, const bool* nan, const R* r1, const R* r2, R* Lam, R* g1, R* g2, R& q1, R& logdet, int& dim) {
    R L[symsize(PB)], invd[PB], W[PB * D], z1[PB], z2[PB];
    bool sk[PB];
#pragma unroll
    for (int k = 0; k < PB; ++k) {
        sk[k] = nan[O + k];
        z1[k] = sk[k] ? (R)0 : r1[O + k];
        z2[k] = sk[k] ? (R)0 : r2[O + k];
        dim += sk[k] ? 0 : 1;
#pragma unroll
        for (int l = 0; l <= k; ++l) L[lidx(k, l)] = (sk[k] || sk[l]) ? (R)0 : Rm[(O + l) * P + (O + k)];
#pragma unroll
        for (int j = 0; j < D; ++j) W[k * D + j] = sk[k] ? (R)0 : H[(O + k) * D + j];
    }
    const bool ok = chol_inplace<R, PB>(L, invd, sk);
#pragma unroll
    for (int k = 0; k < PB; ++k) logdet += sk[k] ? (R)0 : log_(L[lidx(k, k)]);
#pragma unroll
    for (int j = 0; j < D; ++j) lsolve_col<R, PB, D>(L, invd, W, j);
    lsolve<R, PB>(L, invd, z1);
    lsolve<R, PB>(L, invd, z2);
#pragma unroll
    for (int k = 0; k < PB; ++k) q1 += z1[k] * z1[k];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R s1 = 0, s2 = 0;
#pragma unroll
        for (int k = 0; k < PB; ++k) s1 += W[k * D + i] * z1[k], s2 += W[k * D + i] * z2[k];
        g1[i] += s1;
        g2[i] += s2;
#pragma unroll
        for (int j = i; j < D; ++j) {
            R s = 0;
#pragma unroll
            for (int k = 0; k < PB; ++k) s += W[k * D + i] * W[k * D + j];
            Lam[sidx_u(D, i, j)] += s;
        }
    }
    return ok;
}

// The element from the information quantities of the observation model (block-diagonal or not):
//   Lam = H^T R^-1 H (packed sym), gm = H^T R^-1 (y - H m_ - c), gb = H^T R^-1 (y - H b_dyn - c), q = r_m^T R^-1 r_m,
//   logdetR = log|R|/2 (sum of the log Cholesky diagonals), dim = number of observed components.
// W2 = I + Lam P_;  [M | vm | vb] = W2^-1 [Lam | gm | gb];  scale = log N(r_m; 0, S) by the determinant / Woodbury identities
// (log|S| = log|R| + log|I + Lam P|, r^T S^-1 r = r^T R^-1 r - g^T P (I + Lam P)^-1 g).
template <typename R, int D>
AX_HD void filter_elem_from_lam(const R* F, const R* m_, const R* P_, const R* Lam, const R* gm, const R* gb, R q, R logdetR, R dim, bool ok,
                                FiltElem<R, D>& e) {
    constexpr int NR = D + 2;
    R W2[D * D], B[D * NR];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R s = (i == j) ? (R)1 : (R)0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += Lam[sidx(D, i, k)] * P_[k * D + j];
            W2[i * D + j] = s;
            B[i * NR + j] = Lam[sidx(D, i, j)];
        }
        B[i * NR + D] = gm[i];
        B[i * NR + D + 1] = gb[i];
    }
    const R ldw = lu_solve_logdet<R, D, NR>(W2, B);
    {
        R Pg[D];
        mv<R, D, D>(P_, gm, Pg);
        R corr = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) corr += Pg[i] * B[i * NR + D];
        e.z = ok ? (R)-0.5 * (q - corr) - logdetR - (R)0.5 * ldw - (R)(0.5 * LOG_2PI) * dim : r_nan<R>();
    }
    R M[symsize(D)], vm[D], vb[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        vm[i] = ok ? B[i * NR + D] : r_nan<R>();
        vb[i] = ok ? B[i * NR + D + 1] : r_nan<R>();
#pragma unroll
        for (int j = i; j < D; ++j) M[sidx_u(D, i, j)] = ok ? ((i == j) ? B[i * NR + i] : (R)0.5 * (B[i * NR + j] + B[j * NR + i])) : r_nan<R>();
    }
    filter_elem_from_info<R, D>(F, m_, P_, M, vm, vb, e);
}

// Observation-side information of the CONCATENATED auxiliary model y = [u ; yobs], H = [I ; Hobs], c = [0 ; cobs],
// R = blkdiag(hd I, Robs), hd = delta / 2 (examples/lorenz/auxiliary_kalman.py:26-35): the real observation model and the data are
// shared by the chains, so per time step
//   Lobs = Hobs^T Robs^-1 Hobs,  gam = Hobs^T Robs^-1 (yobs - cobs),  kap = (yobs - cobs)^T Robs^-1 (yobs - cobs),
//   ldR = D log sqrt(hd) + sum log chol(Robs)_kk,  dim = D + #observed      (missing components deleted inside the block)
// come from a table and a chain's part is closed form in u and the predicted mean m:
//   Lam = Lobs + I / hd;  g(m) = (u - m) / hd + gam - Lobs m;  q(m) = |u - m|^2 / hd + kap - 2 gam.m + m^T Lobs m.
template <typename R, int D> struct ObsInfoRow {
    static constexpr int DS = symsize(D);
    static constexpr int oL = 0, oG = DS, oK = DS + D, oLd = oK + 1, oDim = oLd + 1, oOk = oDim + 1, N = oOk + 1;
    static constexpr int VEC = 16 / sizeof(R);
    static constexpr int NPAD = (N + VEC - 1) / VEC * VEC;
};
// H, Rm, c, y: the concatenated (P = D + PO) records of the time step (only the observation block, rows D.., is read)
template <typename R, int D, int P> AX_HD void obs_info_row(const R* H, const R* c, const R* __restrict__ Rm, const R* y, R hd, R* row) {
    using T = ObsInfoRow<R, D>;
    static_assert(P > D, "concatenated model");
    bool nan[P];
    R r[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        nan[k] = k < D ? false : !finite_(y[k]);
        r[k] = k < D ? (R)0 : y[k] - c[k];
    }
    R Lam[symsize(D)], g1[D], g2[D];
#pragma unroll
    for (int i = 0; i < symsize(D); ++i) Lam[i] = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) g1[i] = 0, g2[i] = 0;
    R q = 0, logdet = 0;
    int dim = 0;
    const bool ok = info_block<R, D, P, D, P - D>(H, Rm, nan, r, r, Lam, g1, g2, q, logdet, dim);
#pragma unroll
    for (int i = 0; i < symsize(D); ++i) row[T::oL + i] = Lam[i];
#pragma unroll
    for (int i = 0; i < D; ++i) row[T::oG + i] = g1[i];
    row[T::oK] = q;
    row[T::oLd] = logdet + (R)D * log_(sqrt_(hd));
    row[T::oDim] = (R)(dim + D);
    row[T::oOk] = ok ? (R)1 : (R)0;
}
// element of transition i -> i + 1 for one chain from the table row (any pointer-like `row`), the chain's u and its dynamics
template <typename R, int D, typename RowP>
AX_HD void filter_elem_aux(const R* F, const R* bdyn, const R* m_, const R* P_, const R* u, RowP row, R inv_hd, bool first, FiltElem<R, D>& e) {
    using T = ObsInfoRow<R, D>;
    constexpr int DS = symsize(D);
    R Lam[DS], gm[D], gb[D];
#pragma unroll
    for (int i = 0; i < DS; ++i) Lam[i] = row[T::oL + i];
    R q = row[T::oK];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R lm = 0, lb = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) lm += Lam[sidx(D, i, k)] * m_[k], lb += Lam[sidx(D, i, k)] * bdyn[k];
        const R dm = u[i] - m_[i];
        gm[i] = dm * inv_hd + row[T::oG + i] - lm;
        gb[i] = first ? (u[i] - bdyn[i]) * inv_hd + row[T::oG + i] - lb : gm[i];
        q += dm * dm * inv_hd - (R)2 * row[T::oG + i] * m_[i] + m_[i] * lm;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) Lam[sidx_u(D, i, i)] += inv_hd;
    filter_elem_from_lam<R, D>(F, m_, P_, Lam, gm, gb, q, row[T::oLd], row[T::oDim], row[T::oOk] != (R)0, e);
}

// scan element, block-diagonal R (same outputs as filter_elem)
template <typename R, int D, int P, int P1>
AX_HD void filter_elem_blk(const R* F, const R* bdyn, const R* m_, const R* P_, const R* H, const R* c, const R* Rm, const R* y,
                           FiltElem<R, D>& e) {
    bool nan[P];
    bool any = false;
    R rm[P], rb[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        nan[k] = !finite_(y[k]);
        any = any || !nan[k];
        R hm = c[k], hb = c[k];
#pragma unroll
        for (int j = 0; j < D; ++j) hm += H[k * D + j] * m_[j], hb += H[k * D + j] * bdyn[j];
        rm[k] = y[k] - hm;
        rb[k] = y[k] - hb;
    }
    if (!any) {  // _passthrough :239-248
#pragma unroll
        for (int i = 0; i < D * D; ++i) e.A[i] = F[i];
#pragma unroll
        for (int i = 0; i < D; ++i) e.b[i] = m_[i], e.eta[i] = 0;
        sympack<R, D>(P_, e.C);
#pragma unroll
        for (int i = 0; i < symsize(D); ++i) e.J[i] = 0;
        e.z = 0;
        return;
    }
    R Lam[symsize(D)], gm[D], gb[D];
#pragma unroll
    for (int i = 0; i < symsize(D); ++i) Lam[i] = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) gm[i] = 0, gb[i] = 0;
    R q = 0, logdet = 0;
    int dim = 0;
    bool ok = info_block<R, D, P, 0, P1>(H, Rm, nan, rm, rb, Lam, gm, gb, q, logdet, dim);
    ok = info_block<R, D, P, P1, P - P1>(H, Rm, nan, rm, rb, Lam, gm, gb, q, logdet, dim) && ok;
    filter_elem_from_lam<R, D>(F, m_, P_, Lam, gm, gb, q, logdet, (R)dim, ok, e);
}

// ------------------------------------------------------------------------------------------------
// _filtering_op_impl (filtering.py:163-183), a1 = earlier prefix, a2 = later element.
// The reference forms M = A2 (I + C1 J2)^-1 and Nn = A1^T (I + J2 C1)^-1 with two LU solves.  Since
// (I + J2 C1) = (I + C1 J2)^T for symmetric C1, J2, one factorisation of W = I + C1 J2 serves both:
//     X = W^-1 A1,  Y = W^-1 C1,  z = W^-1 (b1 + C1 eta2)
//     A = A2 X;  b = A2 z + b2;  C = A2 Y A2^T + C2;  eta = X^T (eta2 - J2 b1) + eta1;  J = X^T (J2 A1) + J1
// (same values, rounding-level differences).  d = 1 uses divisions (:171-173).
// ------------------------------------------------------------------------------------------------
template <typename R, int D>
AX_HD void filter_combine(const FiltElem<R, D>& a1, const FiltElem<R, D>& a2, FiltElem<R, D>& o) {
    if constexpr (D == 1) {
        const R w = (R)1 + a1.C[0] * a2.J[0];
        const R M = a2.A[0] / w, Nn = a1.A[0] / w;
        o.A[0] = M * a1.A[0];
        o.b[0] = M * (a1.b[0] + a1.C[0] * a2.eta[0]) + a2.b[0];
        o.C[0] = M * (a1.C[0] * a2.A[0]) + a2.C[0];
        o.eta[0] = Nn * (a2.eta[0] - a2.J[0] * a1.b[0]) + a1.eta[0];
        o.J[0] = Nn * (a2.J[0] * a1.A[0]) + a1.J[0];
        const R u = a1.b[0] / w, v = a1.C[0] * a2.eta[0] / w;
        o.z = a1.z + a2.z - (R)0.5 * log_(abs_(w)) + a2.eta[0] * u + (R)0.5 * a2.eta[0] * v - (R)0.5 * (a2.J[0] * a1.b[0]) * u;
    } else {
        constexpr int NR = 2 * D + 1;
        R W[D * D], B[D * NR];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                R s = (i == j) ? (R)1 : (R)0;
#pragma unroll
                for (int k = 0; k < D; ++k) s += a1.C[sidx(D, i, k)] * a2.J[sidx(D, k, j)];
                W[i * D + j] = s;
            }
        R v[D];
        symv<R, D>(a1.C, a2.eta, v);
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                B[i * NR + j] = a1.A[i * D + j];
                B[i * NR + D + j] = a1.C[sidx(D, i, j)];
            }
            B[i * NR + 2 * D] = a1.b[i] + v[i];
        }
        const R ldw = lu_solve_logdet<R, D, NR>(W, B);  // B = [X | Y | z]
        R JA[D * D];
        symm<R, D, D>(a2.J, a1.A, JA);
        R w[D];
        symv<R, D>(a2.J, a1.b, w);
        {  // scale of the product: z1 + z2 - log|W|/2 + eta2.u + eta2.v/2 - (J2 b1).u/2,  u = W^-1 b1,  v = W^-1 C1 eta2 = Y eta2
            R t = 0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                R vi = 0;
#pragma unroll
                for (int j = 0; j < D; ++j) vi += B[i * NR + D + j] * a2.eta[j];
                const R ui = B[i * NR + 2 * D] - vi;
                t += a2.eta[i] * ui + (R)0.5 * a2.eta[i] * vi - (R)0.5 * w[i] * ui;
            }
            o.z = a1.z + a2.z - (R)0.5 * ldw + t;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) w[i] = a2.eta[i] - w[i];
        R AY[D * D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            R sb = a2.b[i];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                R sa = 0, sy = 0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    sa += a2.A[i * D + k] * B[k * NR + j];
                    sy += a2.A[i * D + k] * B[k * NR + D + j];
                }
                o.A[i * D + j] = sa;
                AY[i * D + j] = sy;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) sb += a2.A[i * D + k] * B[k * NR + 2 * D];
            o.b[i] = sb;
        }
        R Cd[D * D], Jd[D * D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            R se = a1.eta[i];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                R sc = a2.C[sidx(D, i, j)], sj = a1.J[sidx(D, i, j)];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    sc += AY[i * D + k] * a2.A[j * D + k];
                    sj += B[k * NR + i] * JA[k * D + j];
                }
                Cd[i * D + j] = sc;
                Jd[i * D + j] = sj;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) se += B[k * NR + i] * w[k];
            o.eta[i] = se;
        }
        sympack<R, D>(Cd, o.C);
        sympack<R, D>(Jd, o.J);
    }
}

// The (b, C) half of the same combine: all the final pass needs, because (b, C) of a1 (+) a2 depend on a1
// only through (b1, C1).
template <typename R, int D>
AX_HD void filter_apply(const FiltPre<R, D>& p, const FiltElem<R, D>& a2, FiltPre<R, D>& o) {
    if constexpr (D == 1) {
        const R w = (R)1 + p.C[0] * a2.J[0];
        const R M = a2.A[0] / w;
        o.b[0] = M * (p.b[0] + p.C[0] * a2.eta[0]) + a2.b[0];
        o.C[0] = M * (p.C[0] * a2.A[0]) + a2.C[0];
        const R u = p.b[0] / w, v = p.C[0] * a2.eta[0] / w;
        o.z = p.z + a2.z - (R)0.5 * log_(abs_(w)) + a2.eta[0] * u + (R)0.5 * a2.eta[0] * v - (R)0.5 * (a2.J[0] * p.b[0]) * u;
    } else {
        constexpr int NR = D + 1;
        R W[D * D], B[D * NR];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                R s = (i == j) ? (R)1 : (R)0;
#pragma unroll
                for (int k = 0; k < D; ++k) s += p.C[sidx(D, i, k)] * a2.J[sidx(D, k, j)];
                W[i * D + j] = s;
            }
        R v[D];
        symv<R, D>(p.C, a2.eta, v);
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
            for (int j = 0; j < D; ++j) B[i * NR + j] = p.C[sidx(D, i, j)];
            B[i * NR + D] = p.b[i] + v[i];
        }
        const R ldw = lu_solve_logdet<R, D, NR>(W, B);  // [Y | z]
        {
            R Jb[D], t = 0;
            symv<R, D>(a2.J, p.b, Jb);
#pragma unroll
            for (int i = 0; i < D; ++i) {
                R vi = 0;
#pragma unroll
                for (int j = 0; j < D; ++j) vi += B[i * NR + j] * a2.eta[j];
                const R ui = B[i * NR + D] - vi;
                t += a2.eta[i] * ui + (R)0.5 * a2.eta[i] * vi - (R)0.5 * Jb[i] * ui;
            }
            o.z = p.z + a2.z - (R)0.5 * ldw + t;
        }
        R AY[D * D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            R sb = a2.b[i];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                R sy = 0;
#pragma unroll
                for (int k = 0; k < D; ++k) sy += a2.A[i * D + k] * B[k * NR + j];
                AY[i * D + j] = sy;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) sb += a2.A[i * D + k] * B[k * NR + D];
            o.b[i] = sb;
        }
        R Cd[D * D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) {
                R sc = a2.C[sidx(D, i, j)];
#pragma unroll
                for (int k = 0; k < D; ++k) sc += AY[i * D + k] * a2.A[j * D + k];
                Cd[i * D + j] = sc;
            }
        sympack<R, D>(Cd, o.C);
    }
}

// ------------------------------------------------------------------------------------------------
// Folding ONE filtering step onto an accumulated prefix without materialising the step's element.
// With acc = (A, b, C, eta, J, z) (given x_0: x_k | y_1:k ~ N(A x_0 + b, C); p(y_1:k | x_0) = e^z N_I(x_0; eta, J)) and a step with
// dynamics (F, Q, b_dyn) and observation information (Lam = H^T R^-1 H, g0 = H^T R^-1 (y - c), q0 = (y - c)^T R^-1 (y - c)):
//   FA = F A,  mb = F b + b_dyn,  Pp = F C F^T + Q                                  (predict, filtering.py:134-139)
//   W = I + Lam Pp,  [M | v] = W^-1 [Lam | g0 - Lam mb]                             (M = H^T S^-1 H, v = H^T S^-1 (y - H mb - c))
//   A' = FA - Pp M FA,  b' = mb + Pp v,  C' = Pp - Pp M Pp                          (update, filtering.py:83-130 in information form)
//   eta' = eta + FA^T v,  J' = J + FA^T M FA,  z' = z + log N(y; H mb + c, S)
// which IS acc (+) element(step) of filtering.py:163-183 with the element's own (A2, b2, C2, eta2, J2) contracted away: 9 d^3 + one
// (d + 1)-column LU instead of building the element (5 d^3 + LU) and a general combine (19 d^3 + a (2d + 1)-column LU).  The general
// combine stays what the aggregate scan runs; this is the chunk-serial part of the scan (kernels.hip.h: k_scan_reduce_cm / k_scan_down_cm
// of the on-the-fly operator).
// ------------------------------------------------------------------------------------------------
// The observation block (Lam, g0, q0) is in information form around the ORIGIN -- harmless for the real observations (|y|^2 / R is a moderate number) -- but the
// auxiliary block y = u, H = I, R = hd I is kept apart and evaluated around the predicted mean: its information-form pieces |u|^2 / hd, u.m / hd and m.m / hd are
// ~|x|^2 / hd each (3.7e7 at Lorenz-63 scale with delta = 1e-4) and cancel down to the innovation |u - m|^2 / hd ~ dim; formed separately in fp32 that cancellation
// cost a bias of +0.024 per step in the log-likelihood (ell of C4's 16 384 steps off by +390, log alpha by ~20: round 4, tools/c4_fp32_diag.py).
// Running product of positive factors as (mantissa in [0.5, 1), exponent): the log-determinant terms of a chunk's steps are multiplied up and ONE logarithm is
// taken per chunk (an fp64 log is ~55 instructions, a step of the d = 1 filter ~120 without it); the product's rounding error grows like sqrt(steps) ulp, below
// that of the sum of as many rounded logarithms.  A zero, negative, infinite or NaN factor ends in the logarithm of the same (-inf / NaN), as the sum would.
template <typename R> struct LogProd {
    R m = 1;
    int e = 0;
    AX_HD void mul(R f) {
        m *= f;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (sizeof(R) == 8) {
            e += __builtin_amdgcn_frexp_exp(m);
            m = __builtin_amdgcn_frexp_mant(m);
        } else {
            e += __builtin_amdgcn_frexp_expf(m);
            m = __builtin_amdgcn_frexp_mantf(m);
        }
#else
        if (m - m == 0 && m != 0) {  // (the hardware's frexp passes inf / NaN / 0 through with exponent 0)
            int k;
            m = (R)frexp((double)m, &k);
            e += k;
        }
#endif
    }
    AX_HD R log() const { return log_(m) + (R)e * (R)0.6931471805599453094; }
};
template <typename R, int D> struct StepInfo {
    R Lam[symsize(D)];  // the observation block WITHOUT the auxiliary block's I / hd
    R g0[D];
    R u[D];             // the auxiliary observation of the step (ignored when inv_hd == 0)
    R inv_hd = 0;       // 1 / (delta / 2); 0: no auxiliary block (Lam, g0, q0 are the whole observation)
    R q0, ldR, dim;
    R rdet = 1;         // DEFER only: a positive factor whose logarithm / 2 ADDS to the step's log-likelihood (1 / det R of a per-step observation covariance)
    bool ok;
};
// shared front end: predict + solve; returns through references.  Cp = packed C of the prefix.
// DEFER: the -log|W| / 2 of the step is NOT in zinc; 1 / |det W| (times si.rdet) is multiplied into *lp instead (LogProd: one logarithm per chunk)
template <typename R, int D, bool DEFER = false>
AX_HD void step_predict_solve(const R* F, const R* Q, const R* bd, const StepInfo<R, D>& si, const R* bprev, const R* Cp, R* mb, R* Pp, R* M, R* v, R& zinc,
                              LogProd<R>* lp = nullptr) {
    R FC[D * D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R sm = bd[i];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R sfc = 0;
#pragma unroll
            for (int k = 0; k < D; ++k) sfc += F[i * D + k] * Cp[sidx(D, k, j)];
            FC[i * D + j] = sfc;
            sm += F[i * D + j] * bprev[j];
        }
        mb[i] = sm;
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = i; j < D; ++j) {
            // F C F^T is symmetric for symmetric C: the upper entry alone, mirrored (exactly symmetric by construction, half the products)
            R s1 = 0;
#pragma unroll
            for (int k = 0; k < D; ++k) s1 += FC[i * D + k] * F[j * D + k];
            const R vv = (i == j) ? s1 + Q[i * D + i] : s1 + (R)0.5 * (Q[i * D + j] + Q[j * D + i]);
            Pp[i * D + j] = vv;
            Pp[j * D + i] = vv;
        }
    constexpr int NR = D + 1;
    R W[D * D], B[D * NR], g[D];
    R q = si.q0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R lm = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) lm += si.Lam[sidx(D, i, k)] * mb[k];
        const R dm = si.u[i] - mb[i];  // the auxiliary block around the predicted mean (no cancellation of |u|^2 / hd against u.m / hd)
        g[i] = (si.g0[i] - lm) + dm * si.inv_hd;
        q += mb[i] * (lm - (R)2 * si.g0[i]) + dm * dm * si.inv_hd;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R s = ((i == j) ? (R)1 : (R)0) + si.inv_hd * Pp[i * D + j];
#pragma unroll
            for (int k = 0; k < D; ++k) s += si.Lam[sidx(D, i, k)] * Pp[k * D + j];
            W[i * D + j] = s;
            B[i * NR + j] = si.Lam[sidx(D, i, j)] + ((i == j) ? si.inv_hd : (R)0);
        }
        B[i * NR + D] = g[i];
    }
    R ldw = 0;
    if constexpr (DEFER) {
        R pr[(D + 1) / 2];
        lu_solve_logdet<R, D, NR, true>(W, B, pr);
        pr[0] *= si.rdet;
#pragma unroll
        for (int k = 0; k < (D + 1) / 2; ++k) lp->mul(pr[k]);
    } else {
        ldw = lu_solve_logdet<R, D, NR>(W, B);
    }
    R corr = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R pg = 0;
#pragma unroll
        for (int k = 0; k < D; ++k) pg += Pp[i * D + k] * g[k];
        corr += pg * B[i * NR + D];
    }
    zinc = si.ok ? (R)-0.5 * (q - corr) - si.ldR - (R)0.5 * ldw - (R)(0.5 * LOG_2PI) * si.dim : r_nan<R>();
#pragma unroll
    for (int i = 0; i < D; ++i) {
        v[i] = si.ok ? B[i * NR + D] : r_nan<R>();
#pragma unroll
        for (int j = 0; j < D; ++j) M[i * D + j] = si.ok ? ((i == j) ? B[i * NR + i] : (R)0.5 * (B[i * NR + j] + B[j * NR + i])) : r_nan<R>();
    }
}
template <typename R, int D, bool DEFER = false>
AX_HD void filter_fold_step(const R* F, const R* Q, const R* bd, const StepInfo<R, D>& si, FiltElem<R, D>& acc, LogProd<R>* lp = nullptr) {
    R mb[D], Pp[D * D], M[D * D], v[D], zinc;
    step_predict_solve<R, D, DEFER>(F, Q, bd, si, acc.b, acc.C, mb, Pp, M, v, zinc, lp);
    R FA[D * D], PM[D * D], MFA[D * D];
    mm<R, D, D, D>(F, acc.A, FA);
    mm<R, D, D, D>(Pp, M, PM);
    mm<R, D, D, D>(M, FA, MFA);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R sb = mb[i], se = acc.eta[i];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R sa = FA[i * D + j];
#pragma unroll
            for (int k = 0; k < D; ++k) sa -= PM[i * D + k] * FA[k * D + j];
            acc.A[i * D + j] = sa;  // (A itself was consumed by FA)
        }
        // C' = Pp - Pp M Pp and FA^T M FA are symmetric (M, Pp symmetric): upper entries only, packed directly
#pragma unroll
        for (int j = i; j < D; ++j) {
            R sc = Pp[i * D + j], sj = 0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                sc -= PM[i * D + k] * Pp[k * D + j];
                sj += FA[k * D + i] * MFA[k * D + j];
            }
            acc.C[sidx_u(D, i, j)] = sc;
            acc.J[sidx_u(D, i, j)] += sj;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) sb += Pp[i * D + k] * v[k], se += FA[k * D + i] * v[k];
        acc.b[i] = sb;
        acc.eta[i] = se;
    }
    acc.z += zinc;
}
// the (b, C, z) half: one Kalman step in information form
template <typename R, int D, bool DEFER = false>
AX_HD void filter_apply_step(const R* F, const R* Q, const R* bd, const StepInfo<R, D>& si, FiltPre<R, D>& p, LogProd<R>* lp = nullptr) {
    R mb[D], Pp[D * D], M[D * D], v[D], zinc;
    step_predict_solve<R, D, DEFER>(F, Q, bd, si, p.b, p.C, mb, Pp, M, v, zinc, lp);
    R PM[D * D];
    mm<R, D, D, D>(Pp, M, PM);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R sb = mb[i];
#pragma unroll
        for (int j = i; j < D; ++j) {  // symmetric: upper entries only
            R sc = Pp[i * D + j];
#pragma unroll
            for (int k = 0; k < D; ++k) sc -= PM[i * D + k] * Pp[k * D + j];
            p.C[sidx_u(D, i, j)] = sc;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) sb += Pp[i * D + k] * v[k];
        p.b[i] = sb;
    }
    p.z += zinc;
}

// ------------------------------------------------------------------------------------------------
// Pathwise sampler (sampling.py).  Scan element (G, e): memory record [G (D*D) | e (D)] padded.
// ------------------------------------------------------------------------------------------------
template <typename R, int D> struct SampElem {
    static constexpr int N = D * D + D;
    static constexpr int VEC = 16 / sizeof(R);
    static constexpr int NPAD = (N + VEC - 1) / VEC * VEC;
    R G[D * D];
    R e[D];
};
template <typename R, int D> struct SampPre {
    static constexpr int N = D;
    static constexpr int VEC = 16 / sizeof(R);
    static constexpr int NPAD = (N + VEC - 1) / VEC * VEC;
    R e[D];
};

template <typename R> AX_HD R nan_to_num(R x) {
    if (isnan_(x)) return (R)0;
    if (!finite_(x)) return x > 0 ? (R)(sizeof(R) == 4 ? 3.4028234663852886e38 : 1.7976931348623157e308)
                                  : (R)(sizeof(R) == 4 ? -3.4028234663852886e38 : -1.7976931348623157e308);
    return x;
}

// lower Cholesky of a packed-symmetric matrix into a DENSE lower-triangular D x D, jnp semantics followed by
// nan_to_num (sampling.py:100-104, :117-121): a failed factorisation is all-NaN, hence all-zero.
template <typename R, int D> AX_HD void chol_nan_to_num(const R* Spacked, R* Ld) {
    if constexpr (D == 1) {
        Ld[0] = nan_to_num(sqrt_(Spacked[0]));
    } else {
        R L[symsize(D)], invd[D];
        const bool ok = chol_packed<R, D>(Spacked, L, invd, nullptr);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) Ld[i * D + j] = (j <= i && ok) ? nan_to_num(L[lidx(i, j)]) : (R)0;
    }
}

// mean_and_chol (sampling.py:60-105) for t < T-1: gain G = P (S^-1 F)^T, S = F P F^T + Q, and Lc = nan_to_num(chol(P - G S G^T))
template <typename R, int D>
AX_HD void sample_gain_chol(const R* F, const R* Q, const R* Pd, R* G, R* Lc) {
    R FP[D * D], Sd[D * D], S[symsize(D)];
    mm<R, D, D, D>(F, Pd, FP);
    mmt<R, D, D, D>(FP, F, Sd);
#pragma unroll
    for (int i = 0; i < D * D; ++i) Sd[i] += Q[i];
    sympack<R, D>(Sd, S);
    if constexpr (D == 1) {
        G[0] = Pd[0] * F[0] / S[0];
    } else {
        R L[symsize(D)], invd[D];
        const bool ok = chol_packed<R, D>(S, L, invd, nullptr);
        R X[D * D];  // S^-1 F
#pragma unroll
        for (int i = 0; i < D * D; ++i) X[i] = F[i];
#pragma unroll
        for (int j = 0; j < D; ++j) cho_solve_col<R, D, D>(L, invd, X, j);
        mmt<R, D, D, D>(Pd, X, G);  // gain = P X^T
        if (!ok) {
#pragma unroll
            for (int i = 0; i < D * D; ++i) G[i] = r_nan<R>();
        }
    }
    // inc_Sig = sym(P - gain S gain^T)
    R gS[D * D], Sg[D * D], Sig[symsize(D)];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            R s = 0;
#pragma unroll
            for (int k = 0; k < D; ++k) s += G[i * D + k] * S[sidx(D, k, j)];
            gS[i * D + j] = s;
        }
    mmt<R, D, D, D>(gS, G, Sg);
#pragma unroll
    for (int i = 0; i < D * D; ++i) Sg[i] = Pd[i] - Sg[i];
    sympack<R, D>(Sg, Sig);
    chol_nan_to_num<R, D>(Sig, Lc);
}
// + _sampling_init_one (sampling.py:108-112): inc = m - G (F m + b) + Lc eps
template <typename R, int D>
AX_HD void sample_elem(const R* F, const R* Q, const R* b, const R* m, const R* Pd, const R* eps, SampElem<R, D>& e) {
    R Lc[D * D];
    sample_gain_chol<R, D>(F, Q, Pd, e.G, Lc);
    R pm[D], t[D];
    mv<R, D, D>(F, m, pm);
#pragma unroll
    for (int i = 0; i < D; ++i) pm[i] += b[i];
    mv<R, D, D>(e.G, pm, t);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R s = m[i] - t[i];
#pragma unroll
        for (int k = 0; k <= i; ++k) s += Lc[i * D + k] * eps[k];
        e.e[i] = s;
    }
}
// Chain-shared parameters AND covariances (the filtered P_t of a linear-Gaussian model do not depend on the data): the gain G_t,
// M1 = I - G F, gb = G b and the Cholesky factor Lc of the increment covariance are the same for every chain; a chain's increment is
// e = M1 m - gb + Lc eps (sampling.py:108-112 rearranged).  Row layout: [G D*D | M1 D*D | gb D | Lc D*D (dense lower)].
template <typename R, int D> struct SampShared {
    static constexpr int oG = 0, oM = D * D, oGb = 2 * D * D, oL = 2 * D * D + D, N = 3 * D * D + D;
    static constexpr int VEC = 16 / sizeof(R);
    static constexpr int NPAD = (N + VEC - 1) / VEC * VEC;
};
template <typename R, int D>
AX_HD void sample_shared_row(const R* F, const R* Q, const R* b, const R* Pd, bool last, R* row) {
    using T = SampShared<R, D>;
    R Lc[D * D];
    if (last) {  // _sample_last_step: G = 0, e = m + chol(P) eps
        R S[symsize(D)];
        sympack<R, D>(Pd, S);
        chol_nan_to_num<R, D>(S, Lc);
#pragma unroll
        for (int i = 0; i < D * D; ++i) row[T::oG + i] = 0, row[T::oM + i] = (i / D == i % D) ? (R)1 : (R)0, row[T::oL + i] = Lc[i];
#pragma unroll
        for (int i = 0; i < D; ++i) row[T::oGb + i] = 0;
        return;
    }
    R G[D * D];
    sample_gain_chol<R, D>(F, Q, Pd, G, Lc);
    R GF[D * D], gb[D];
    mm<R, D, D, D>(G, F, GF);
    mv<R, D, D>(G, b, gb);
#pragma unroll
    for (int i = 0; i < D * D; ++i) row[T::oG + i] = G[i], row[T::oM + i] = ((i / D == i % D) ? (R)1 : (R)0) - GF[i], row[T::oL + i] = Lc[i];
#pragma unroll
    for (int i = 0; i < D; ++i) row[T::oGb + i] = gb[i];
}

// _sample_last_step (sampling.py:115-124)
template <typename R, int D> AX_HD void sample_last(const R* m, const R* Pd, const R* eps, SampElem<R, D>& e) {
    R S[symsize(D)], Lc[D * D];
    sympack<R, D>(Pd, S);
    chol_nan_to_num<R, D>(S, Lc);
#pragma unroll
    for (int i = 0; i < D * D; ++i) e.G[i] = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R s = m[i];
#pragma unroll
        for (int k = 0; k <= i; ++k) s += Lc[i * D + k] * eps[k];
        e.e[i] = s;
    }
}
// _sampling_op_impl (sampling.py:51-55): acc = later times already composed, cur = this time step
template <typename R, int D>
AX_HD void sample_combine(const SampElem<R, D>& acc, const SampElem<R, D>& cur, SampElem<R, D>& o) {
    mm<R, D, D, D>(cur.G, acc.G, o.G);
    R t[D];
    mv<R, D, D>(cur.G, acc.e, t);
#pragma unroll
    for (int i = 0; i < D; ++i) o.e[i] = t[i] + cur.e[i];
}
template <typename R, int D>
AX_HD void sample_apply(const SampPre<R, D>& p, const SampElem<R, D>& cur, SampPre<R, D>& o) {
    R t[D];
    mv<R, D, D>(cur.G, p.e, t);
#pragma unroll
    for (int i = 0; i < D; ++i) o.e[i] = t[i] + cur.e[i];
}

// ------------------------------------------------------------------------------------------------
// mvn.logpdf of a residual r with covariance given densely (math/mvn/base.py:15-58), N = dimension.
//   drop_nonfinite: a non-finite residual component makes the reference's result NaN, which its nansum
//   then drops -> return 0.  `skip` (may be null) deletes components (MASKED policy).
// ------------------------------------------------------------------------------------------------
template <typename R, int N>
AX_HD R gauss_logpdf(const R* r, const R* __restrict__ cov, const bool* skip) {
    R res[N];
    int dim = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool sk = skip ? skip[k] : false;
        res[k] = sk ? (R)0 : r[k];
        bad = bad || !finite_(res[k]);
        dim += sk ? 0 : 1;
    }
    R out;
    if constexpr (N == 1) {
        const R sd = sqrt_(cov[0]);
        const R z = res[0] / sd;
        out = (R)-0.5 * z * z - log_(sd) - (R)(0.5 * LOG_2PI);
        if (skip && skip[0]) out = 0;
    } else {
        R L[symsize(N)], invd[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = i; j < N; ++j) L[lidx(j, i)] = cov[i * N + j];
        const bool ok = chol_inplace<R, N>(L, invd, skip);
        R logdet = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) logdet += (skip && skip[k]) ? (R)0 : log_(L[lidx(k, k)]);
        lsolve<R, N>(L, invd, res);
        R q = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) q += res[k] * res[k];
        out = (R)-0.5 * q - logdet - (R)(0.5 * LOG_2PI) * (R)dim;
        if (!ok) out = r_nan<R>();
    }
    if (bad || isnan_(out)) return (R)0;
    return out;
}


// Same as gauss_logpdf for two residuals sharing one covariance (one Cholesky, two triangular solves).
template <typename R, int N>
AX_HD void gauss_logpdf2(const R* r1, const R* r2, const R* __restrict__ cov, const bool* skip, R& o1, R& o2) {
    R a[N], b[N];
    int dim = 0;
    bool bad1 = false, bad2 = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool sk = skip ? skip[k] : false;
        a[k] = sk ? (R)0 : r1[k];
        b[k] = sk ? (R)0 : r2[k];
        bad1 = bad1 || !finite_(a[k]);
        bad2 = bad2 || !finite_(b[k]);
        dim += sk ? 0 : 1;
    }
    if constexpr (N == 1) {
        const R sd = sqrt_(cov[0]);
        const R z1 = a[0] / sd, z2 = b[0] / sd;
        const R cst = -log_(sd) - (R)(0.5 * LOG_2PI);
        o1 = (R)-0.5 * z1 * z1 + cst;
        o2 = (R)-0.5 * z2 * z2 + cst;
        if (skip && skip[0]) o1 = o2 = 0;
    } else {
        R L[symsize(N)], invd[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = i; j < N; ++j) L[lidx(j, i)] = cov[i * N + j];
        const bool ok = chol_inplace<R, N>(L, invd, skip);
        R logdet = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) logdet += (skip && skip[k]) ? (R)0 : log_(L[lidx(k, k)]);
        lsolve<R, N>(L, invd, a);
        lsolve<R, N>(L, invd, b);
        R q1 = 0, q2 = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) q1 += a[k] * a[k], q2 += b[k] * b[k];
        const R cst = -logdet - (R)(0.5 * LOG_2PI) * (R)dim;
        o1 = ok ? (R)-0.5 * q1 + cst : r_nan<R>();
        o2 = ok ? (R)-0.5 * q2 + cst : r_nan<R>();
    }
    if (bad1 || isnan_(o1)) o1 = 0;
    if (bad2 || isnan_(o2)) o2 = 0;
}

// gauss_logpdf2 with the log-determinant DEFERRED (LogProd): o1 / o2 lack the -sum log L_kk, which is log(fac), fac = prod 1 / L_kk; keep1 / keep2 say whether the
// value survived the nansum rule (a dropped one is 0 and must not receive the determinant either)
template <typename R, int N>
AX_HD void gauss_logpdf2_lp(const R* r1, const R* r2, const R* __restrict__ cov, R& o1, R& o2, R& fac, bool& keep1, bool& keep2, const bool* skip = nullptr) {
    R a[N], b[N];
    int dim = 0;
    bool bad1 = false, bad2 = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool sk = skip ? skip[k] : false;
        a[k] = sk ? (R)0 : r1[k];
        b[k] = sk ? (R)0 : r2[k];
        bad1 = bad1 || !finite_(a[k]);
        bad2 = bad2 || !finite_(b[k]);
        dim += sk ? 0 : 1;
    }
    bool ok = true;
    if constexpr (N == 1) {
        const R iv = (R)1 / cov[0];   // (z^2 = r^2 / cov; -log sqrt(cov) = log(1 / cov) / 2: the factor carries the square)
        o1 = (R)-0.5 * a[0] * a[0] * iv - (R)(0.5 * LOG_2PI);
        o2 = (R)-0.5 * b[0] * b[0] * iv - (R)(0.5 * LOG_2PI);
        fac = iv;
        if (skip && skip[0]) o1 = o2 = 0, fac = 1;   // (gauss_logpdf2: a masked component contributes 0 -- and is kept, as a 0)
    } else {
        R L[symsize(N)], invd[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = i; j < N; ++j) L[lidx(j, i)] = cov[i * N + j];
        ok = chol_inplace<R, N>(L, invd, skip);
        R f = 1;
#pragma unroll
        for (int k = 0; k < N; ++k) f *= (skip && skip[k]) ? (R)1 : invd[k];
        lsolve<R, N>(L, invd, a);
        lsolve<R, N>(L, invd, b);
        R q1 = 0, q2 = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) q1 += a[k] * a[k], q2 += b[k] * b[k];
        o1 = (R)-0.5 * q1 - (R)(0.5 * LOG_2PI) * (R)dim;
        o2 = (R)-0.5 * q2 - (R)(0.5 * LOG_2PI) * (R)dim;
        fac = f * f;   // (N > 1: the factor is squared so that every caller adds log(fac) / 2, as for N = 1)
    }
    const bool fbad = !(fac > (R)0) || !finite_(fac) || !ok;   // (a failed factorisation, sqrt of a negative: NaN in gauss_logpdf2)
    keep1 = !(bad1 || isnan_(o1) || fbad);
    keep2 = !(bad2 || isnan_(o2) || fbad);
    if (!keep1) o1 = 0;
    if (!keep2) o2 = 0;
}

// Two residuals against one covariance whose Cholesky factor (packed lower L, reciprocal diagonal invd) and additive constant
// cst = -sum log L_kk - dim/2 log 2 pi come from a table (chain-shared parameters).  Same semantics as gauss_logpdf2: a failed
// factorisation is cst = NaN; a non-finite kept residual or a NaN result is 0 (the reference's nansum).
template <typename R, int N, typename LP>
AX_HD void gauss_logpdf2_fact(const R* r1, const R* r2, LP L, LP invd, R cst, const bool* skip, R& o1, R& o2) {
    R a[N], b[N];
    bool bad1 = false, bad2 = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool sk = skip ? skip[k] : false;
        a[k] = sk ? (R)0 : r1[k];
        b[k] = sk ? (R)0 : r2[k];
        bad1 = bad1 || !finite_(a[k]);
        bad2 = bad2 || !finite_(b[k]);
    }
    R q1 = 0, q2 = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        R s1 = a[i], s2 = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s1 -= L[lidx(i, k)] * a[k], s2 -= L[lidx(i, k)] * b[k];
        a[i] = s1 * invd[i];
        b[i] = s2 * invd[i];
        q1 += a[i] * a[i];
        q2 += b[i] * b[i];
    }
    o1 = (R)-0.5 * q1 + cst;
    o2 = (R)-0.5 * q2 + cst;
    if (bad1 || isnan_(o1)) o1 = 0;
    if (bad2 || isnan_(o2)) o2 = 0;
}
// table part for one covariance: [L packed (symsize N) | invd (N) | cst]; `skip` deletes components (L_kk = 1)
template <typename R, int N> struct CholRow {
    static constexpr int oL = 0, oI = symsize(N), oC = symsize(N) + N, SZ = symsize(N) + N + 1;
};
template <typename R, int N> AX_HD void chol_row(const R* __restrict__ cov, const bool* skip, R* out) {
    using T = CholRow<R, N>;
    R L[symsize(N)], invd[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = i; j < N; ++j) L[lidx(j, i)] = cov[i * N + j];
    bool ok;
    if constexpr (N == 1) {
        ok = L[0] > (R)0;
        L[0] = (skip && skip[0]) ? (R)1 : sqrt_(L[0]);
        invd[0] = (R)1 / L[0];
    } else {
        ok = chol_inplace<R, N>(L, invd, skip);
    }
    R logdet = 0;
    int dim = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const bool sk = skip ? skip[k] : false;
        logdet += sk ? (R)0 : log_(L[lidx(k, k)]);
        dim += sk ? 0 : 1;
    }
#pragma unroll
    for (int i = 0; i < symsize(N); ++i) out[T::oL + i] = L[i];
#pragma unroll
    for (int i = 0; i < N; ++i) out[T::oI + i] = invd[i];
    out[T::oC] = ok ? -logdet - (R)(0.5 * LOG_2PI) * (R)dim : r_nan<R>();
}

}  // namespace ax

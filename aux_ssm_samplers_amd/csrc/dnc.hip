// dnc.hip -- the reference's divide-and-conquer pathwise sampler of an LGSSM (aux_samplers/_primitives/kalman/dnc_sampling.py:17-186) on the device.
//
// The reference calls it a proof of concept (:38-41) and points callers to the parallel sampler; it is built here so that the module is an implementation and not a
// delegate.  Algorithm, as the reference's:
//   leaves   (E_t, g_t, L_t), t < T - 1: x_t | x_{t+1} ~ N(E_t x_{t+1} + g_t, L_t) from the filtered moments (_init_elems :128-137)
//   up       pairs of neighbouring intervals [a, b], [b, c] -> [a, c]: E = E1 E2, g = g1 + E1 g2, L = L1 + E1 L2 E1^T, and the mid-point's conditional
//            x_b | x_a, x_c ~ N(G x_a + Gamma x_c + w, V): G = (L^-1 E1 L2)^T, Gamma = E2 - G E, w = g2 - G g, V = L2 - G L G^T (_combination_operator_impl :104-118); an
//            odd interval count carries its last interval up unchanged (_combine_elements :140-169)
//   top      x_{T-1} ~ N(m_{T-1}, P_{T-1}); x_0 | x_{T-1} from the root interval (:53-68)
//   down     level by level, every mid-point from its two sampled neighbours (:70-76)
// One launch per tree level in either direction (2 ceil(log2(T - 1)) + 3 launches), one lane per (chain, pair); the index plan of the tree (left / mid / right time
// index of every pair) is built on the host once per horizon and cached on the handle.  Noise: explicit eps (C, T, dx) -- the time index t is sampled exactly once, with
// eps[t] (the reference splits its key per level instead; the parity contract is on explicit noise).  dx <= 4, unbatched (B = 1), as the reference (:42-43).
#include "ctx.h"

namespace ax {

template <typename R, int D> struct DncElem {  // [E (D*D) | g (D) | L (D*D, full)]
    static constexpr int N = 2 * D * D + D;
};
template <typename R, int D> struct DncAux {  // [G (D*D) | Gamma (D*D) | w (D) | chol V (D*D, dense lower)]
    static constexpr int N = 3 * D * D + D;
};

// dense lower Cholesky factor of a full symmetric matrix; NaN on failure (jnp.linalg.cholesky)
template <typename R, int D> __device__ __forceinline__ void dnc_chol(const R* A, R* Ld) {
    if constexpr (D == 1) {
        Ld[0] = sqrt_(A[0]);
    } else {
        R S[symsize(D)], L[symsize(D)], invd[D];
        sympack<R, D>(A, S);
        const bool ok = chol_packed<R, D>(S, L, invd, nullptr);
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j = 0; j < D; ++j) Ld[i * D + j] = j <= i ? (ok ? L[lidx(i, j)] : r_nan<R>()) : (R)0;
    }
}
// X = S^-1 B for SPD S (full storage), B (D x D); NaN on failure
template <typename R, int D> __device__ __forceinline__ void dnc_spd_solve(const R* Sd, R* X) {
    if constexpr (D == 1) {
        X[0] = X[0] / Sd[0];
    } else {
        R S[symsize(D)], L[symsize(D)], invd[D];
        sympack<R, D>(Sd, S);
        const bool ok = chol_packed<R, D>(S, L, invd, nullptr);
#pragma unroll
        for (int j = 0; j < D; ++j) cho_solve_col<R, D, D>(L, invd, X, j);
        if (!ok) {
#pragma unroll
            for (int i = 0; i < D * D; ++i) X[i] = r_nan<R>();
        }
    }
}

// leaves (_init_elems): E = (S^-1 F P)^T, S = F P F^T + Q; g = m - E (F m + b); L = P - E F P
template <typename R, int D> __global__ void __launch_bounds__(128) k_dnc_init(int C, int n, Arr ms, Arr Ps, Arr Fs, Arr Qs, Arr bs, R* __restrict__ elem) {
    const long long g_ = (long long)blockIdx.x * 128 + threadIdx.x;
    if (g_ >= (long long)C * n) return;
    const int c = (int)(g_ / n), t = (int)(g_ % n);
    R m[D], P[D * D], F[D * D], Q[D * D], b[D];
    ld<R, D>(at<R>(ms, c, t, 0), m);
    ld<R, D * D>(at<R>(Ps, c, t, 0), P);
    ld<R, D * D>(at<R>(Fs, c, t, 0), F);
    ld<R, D * D>(at<R>(Qs, c, t, 0), Q);
    ld<R, D>(at<R>(bs, c, t, 0), b);
    R FP[D * D], S[D * D], X[D * D], E[D * D], EF[D * D], EFP[D * D], pm[D], Epm[D];
    mm<R, D, D, D>(F, P, FP);
    mmt<R, D, D, D>(FP, F, S);
#pragma unroll
    for (int i = 0; i < D * D; ++i) S[i] += Q[i], X[i] = FP[i];
    dnc_spd_solve<R, D>(S, X);  // S^-1 F P
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) E[i * D + j] = X[j * D + i];
    mv<R, D, D>(F, m, pm);
#pragma unroll
    for (int i = 0; i < D; ++i) pm[i] += b[i];
    mv<R, D, D>(E, pm, Epm);
    mm<R, D, D, D>(E, F, EF);
    mm<R, D, D, D>(EF, P, EFP);
    R* o = elem + g_ * DncElem<R, D>::N;
#pragma unroll
    for (int i = 0; i < D * D; ++i) o[i] = E[i], o[D * D + D + i] = P[i] - EFP[i];
#pragma unroll
    for (int i = 0; i < D; ++i) o[D * D + i] = m[i] - Epm[i];
}

// one level up (_combination_operator_impl): pairs (2p, 2p + 1) of the ne intervals of a chain -> interval p of the next level + the mid-point's aux record; an odd
// last interval is copied up
template <typename R, int D>
__global__ void __launch_bounds__(128) k_dnc_combine(int C, int ne, const R* __restrict__ ein, R* __restrict__ eout, R* __restrict__ aux) {
    using TE = DncElem<R, D>;
    using TA = DncAux<R, D>;
    const int npairs = ne / 2, nout = npairs + (ne & 1);
    const long long g_ = (long long)blockIdx.x * 128 + threadIdx.x;
    if (g_ >= (long long)C * nout) return;
    const int c = (int)(g_ / nout), p = (int)(g_ % nout);
    const R* in = ein + (long long)c * ne * TE::N;
    R* out = eout + ((long long)c * nout + p) * TE::N;
    if (p == npairs) {  // the remainder
#pragma unroll
        for (int i = 0; i < TE::N; ++i) out[i] = in[(long long)(ne - 1) * TE::N + i];
        return;
    }
    R E1[D * D], g1[D], L1[D * D], E2[D * D], g2[D], L2[D * D];
    const R* a = in + (long long)(2 * p) * TE::N;
    const R* b = a + TE::N;
    ld<R, D * D>(a, E1); ld<R, D>(a + D * D, g1); ld<R, D * D>(a + D * D + D, L1);
    ld<R, D * D>(b, E2); ld<R, D>(b + D * D, g2); ld<R, D * D>(b + D * D + D, L2);
    R E[D * D], g[D], L[D * D], t1[D * D], t2[D * D], e1g2[D];
    mm<R, D, D, D>(E1, E2, E);
    mv<R, D, D>(E1, g2, e1g2);
    mm<R, D, D, D>(E1, L2, t1);      // E1 L2
    mmt<R, D, D, D>(t1, E1, t2);     // E1 L2 E1^T
#pragma unroll
    for (int i = 0; i < D; ++i) g[i] = g1[i] + e1g2[i];
#pragma unroll
    for (int i = 0; i < D * D; ++i) L[i] = L1[i] + t2[i];
    R X[D * D], G[D * D], GE[D * D], Gg[D], GL[D * D], GLG[D * D], V[D * D], cV[D * D];
#pragma unroll
    for (int i = 0; i < D * D; ++i) X[i] = t1[i];
    dnc_spd_solve<R, D>(L, X);       // L^-1 E1 L2
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) G[i * D + j] = X[j * D + i];
    mm<R, D, D, D>(G, E, GE);
    mv<R, D, D>(G, g, Gg);
    mm<R, D, D, D>(G, L, GL);
    mmt<R, D, D, D>(GL, G, GLG);
#pragma unroll
    for (int i = 0; i < D * D; ++i) V[i] = L2[i] - GLG[i];
    dnc_chol<R, D>(V, cV);
#pragma unroll
    for (int i = 0; i < D * D; ++i) out[i] = E[i], out[D * D + D + i] = L[i];
#pragma unroll
    for (int i = 0; i < D; ++i) out[D * D + i] = g[i];
    R* q = aux + ((long long)c * npairs + p) * TA::N;
#pragma unroll
    for (int i = 0; i < D * D; ++i) q[i] = G[i], q[D * D + i] = E2[i] - GE[i], q[2 * D * D + D + i] = cV[i];
#pragma unroll
    for (int i = 0; i < D; ++i) q[2 * D * D + i] = g2[i] - Gg[i];
}

// the two ends: x_{T-1} = m_{T-1} + chol(P_{T-1}) eps_{T-1};  x_0 = E x_{T-1} + g + chol(L) eps_0 from the root interval
template <typename R, int D> __global__ void __launch_bounds__(128) k_dnc_root(int C, int T, Arr ms, Arr Ps, const R* __restrict__ top, const R* __restrict__ eps,
                                                                               R* __restrict__ xs) {
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= C) return;
    R m[D], P[D * D], cP[D * D], e[D], xT[D];
    ld<R, D>(at<R>(ms, c, T - 1, 0), m);
    ld<R, D * D>(at<R>(Ps, c, T - 1, 0), P);
    dnc_chol<R, D>(P, cP);
    ld<R, D>(eps + ((long long)c * T + (T - 1)) * D, e);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R v = m[i];
#pragma unroll
        for (int j = 0; j <= i; ++j) v += cP[i * D + j] * e[j];
        xT[i] = v;
    }
    st<R, D>(xs + ((long long)c * T + (T - 1)) * D, xT);
    if (T < 2) return;
    const R* r = top + (long long)c * DncElem<R, D>::N;
    R E[D * D], g[D], L[D * D], cL[D * D], x0[D];
    ld<R, D * D>(r, E); ld<R, D>(r + D * D, g); ld<R, D * D>(r + D * D + D, L);
    dnc_chol<R, D>(L, cL);
    ld<R, D>(eps + (long long)c * T * D, e);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R v = g[i];
#pragma unroll
        for (int j = 0; j < D; ++j) v += E[i * D + j] * xT[j];
#pragma unroll
        for (int j = 0; j <= i; ++j) v += cL[i * D + j] * e[j];
        x0[i] = v;
    }
    st<R, D>(xs + (long long)c * T * D, x0);
}

// one level down (_sample :78-86): mid-point p of the level from its sampled neighbours
template <typename R, int D>
__global__ void __launch_bounds__(128) k_dnc_sample(int C, int T, int npairs, const R* __restrict__ aux, const int32_t* __restrict__ left, const int32_t* __restrict__ mid,
                                                     const int32_t* __restrict__ right, const R* __restrict__ eps, R* __restrict__ xs) {
    using TA = DncAux<R, D>;
    const long long g_ = (long long)blockIdx.x * 128 + threadIdx.x;
    if (g_ >= (long long)C * npairs) return;
    const int c = (int)(g_ / npairs), p = (int)(g_ % npairs);
    const R* q = aux + g_ * TA::N;
    R G[D * D], Gm[D * D], w[D], cV[D * D], x1[D], x2[D], e[D], x[D];
    ld<R, D * D>(q, G); ld<R, D * D>(q + D * D, Gm); ld<R, D>(q + 2 * D * D, w); ld<R, D * D>(q + 2 * D * D + D, cV);
    const long long base = (long long)c * T;
    ld<R, D>(xs + (base + left[p]) * D, x1);
    ld<R, D>(xs + (base + right[p]) * D, x2);
    ld<R, D>(eps + (base + mid[p]) * D, e);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R v = w[i];
#pragma unroll
        for (int j = 0; j < D; ++j) v += G[i * D + j] * x1[j] + Gm[i * D + j] * x2[j];
#pragma unroll
        for (int j = 0; j <= i; ++j) v += cV[i * D + j] * e[j];
        x[i] = v;
    }
    st<R, D>(xs + (base + mid[p]) * D, x);
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------------
// the tree's index plan for T states (make_dnc_tree :172-186): per level (bottom first) the pairs' (left, mid, right) time indices, flattened; levels[l] = first pair of level l
struct DncPlan {
    std::vector<int32_t> left, mid, right;
    std::vector<int> first, count, ne;  // per level: offset into the arrays, pairs, intervals entering the level
};
static void dnc_plan(int T, DncPlan& pl) {
    std::vector<std::pair<int, int>> iv;
    for (int t = 0; t + 1 < T; ++t) iv.push_back({t, t + 1});
    while (iv.size() > 1) {
        const int ne = (int)iv.size(), np = ne / 2;
        pl.first.push_back((int)pl.left.size());
        pl.count.push_back(np);
        pl.ne.push_back(ne);
        std::vector<std::pair<int, int>> nx;
        for (int p = 0; p < np; ++p) {
            pl.left.push_back(iv[2 * p].first);
            pl.mid.push_back(iv[2 * p].second);
            pl.right.push_back(iv[2 * p + 1].second);
            nx.push_back({iv[2 * p].first, iv[2 * p + 1].second});
        }
        if (ne & 1) nx.push_back(iv[ne - 1]);
        iv.swap(nx);
    }
}

template <typename R, int D> static int run_dnc_t(auxssm_ctx* h, int C, int T, const Arr& Fs, const Arr& Qs, const Arr& bs, const Arr& ms, const Arr& Ps, const R* eps, R* xs) {
    using TE = DncElem<R, D>;
    using TA = DncAux<R, D>;
    const int n = T - 1;
    DncPlan pl;
    if (n >= 1) dnc_plan(T, pl);
    const size_t npairs_tot = pl.left.size();
    size_t need = (size_t)3 * (npairs_tot + 1) * sizeof(int32_t) + 1024 + (size_t)2 * C * std::max(n, 1) * TE::N * sizeof(R) + (size_t)C * (npairs_tot + 1) * TA::N * sizeof(R) + 4096;
    int rc = ws_reserve(h, need);
    if (rc) return rc;
    int32_t* idx = (int32_t*)ws_take(h, (size_t)3 * (npairs_tot + 1) * sizeof(int32_t));
    R* e0 = (R*)ws_take(h, (size_t)C * std::max(n, 1) * TE::N * sizeof(R));
    R* e1 = (R*)ws_take(h, (size_t)C * std::max(n, 1) * TE::N * sizeof(R));
    R* aux = (R*)ws_take(h, (size_t)C * (npairs_tot + 1) * TA::N * sizeof(R));
    if (!idx || !e0 || !e1 || !aux) return AUXSSM_ERR_NOMEM;
    if (npairs_tot) {  // (a primitive, not a sweep: the plan is uploaded per call behind the stream's tail -- pageable source, so the copy is complete on return)
        AX_HIP(hipMemcpyAsync(idx, pl.left.data(), npairs_tot * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        AX_HIP(hipMemcpyAsync(idx + npairs_tot, pl.mid.data(), npairs_tot * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        AX_HIP(hipMemcpyAsync(idx + 2 * npairs_tot, pl.right.data(), npairs_tot * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        AX_HIP(hipStreamSynchronize(h->stream));
    }
    auto grid = [](long long lanes) { return dim3((unsigned)((lanes + 127) / 128)); };
    R* cur = e0;
    R* nxt = e1;
    if (n >= 1) hipLaunchKernelGGL((k_dnc_init<R, D>), grid((long long)C * n), dim3(128), 0, h->stream, C, n, ms, Ps, Fs, Qs, bs, cur);
    const int nlev = (int)pl.count.size();
    for (int l = 0; l < nlev; ++l) {
        const int ne = pl.ne[l], nout = ne / 2 + (ne & 1);
        // aux records of level l: [chain][pair] blocks, levels back to back by their first pair (scaled by C)
        hipLaunchKernelGGL((k_dnc_combine<R, D>), grid((long long)C * nout), dim3(128), 0, h->stream, C, ne, (const R*)cur, nxt, aux + (size_t)C * pl.first[l] * TA::N);
        std::swap(cur, nxt);
    }
    hipLaunchKernelGGL((k_dnc_root<R, D>), grid(C), dim3(128), 0, h->stream, C, T, ms, Ps, (const R*)cur, eps, xs);
    for (int l = nlev - 1; l >= 0; --l) {
        const int np = pl.count[l];
        hipLaunchKernelGGL((k_dnc_sample<R, D>), grid((long long)C * np), dim3(128), 0, h->stream, C, T, np, (const R*)(aux + (size_t)C * pl.first[l] * TA::N),
                           (const int32_t*)(idx + pl.first[l]), (const int32_t*)(idx + npairs_tot + pl.first[l]), (const int32_t*)(idx + 2 * npairs_tot + pl.first[l]), eps, xs);
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

int run_dnc(auxssm_ctx* h, int dtype, int C, int T, int D, const Arr& Fs, const Arr& Qs, const Arr& bs, const Arr& ms, const Arr& Ps, const void* eps, void* xs) {
#define AX_DNC(R)                                                                                     \
    switch (D) {                                                                                      \
        case 1: return run_dnc_t<R, 1>(h, C, T, Fs, Qs, bs, ms, Ps, (const R*)eps, (R*)xs);           \
        case 2: return run_dnc_t<R, 2>(h, C, T, Fs, Qs, bs, ms, Ps, (const R*)eps, (R*)xs);           \
        case 3: return run_dnc_t<R, 3>(h, C, T, Fs, Qs, bs, ms, Ps, (const R*)eps, (R*)xs);           \
        case 4: return run_dnc_t<R, 4>(h, C, T, Fs, Qs, bs, ms, Ps, (const R*)eps, (R*)xs);           \
    }
    if (dtype == AUXSSM_F32) { AX_DNC(float) } else { AX_DNC(double) }
#undef AX_DNC
    return AUXSSM_ERR_UNSUPPORTED;
}

}  // namespace ax

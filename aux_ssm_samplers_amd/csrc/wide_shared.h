// wide_shared.h -- included inside namespace ax::wide of wide.hip (after the filter kernels, before the host side).
//
// The wide-state FILTER for S >= 2 sequences that share EVERY model parameter (P0, Fs, Qs, bs, Hs, Rs, cs with chain and batch stride 0 -- what
// jax.vmap leaves unbatched when only ys is mapped over, filtering.py:18-46) and the same missing-observation pattern.  The covariance recursion, the
// gains and the innovation covariances then do not depend on the sequence (filtering.py:89-130: P_t, K_t, S_t are functions of the parameters and
// of WHICH components of y_t are observed, never of their values), so they are computed ONCE -- by the one-sequence matrix filter of this file
// (the parallel scan of filtering.py:163-183) on sequence 0 -- and every sequence is left with the affine mean recursion of the gain form
//
//      m_t = A_t m_{t-1} + K_t y_t + g_t,     A_t = (I - K_t H_t) F_{t-1},   g_t = b_{t-1} - K_t (H_t b_{t-1} + c_t)
//      v_t = y_t - H_t F_{t-1} m_{t-1} - (H_t b_{t-1} + c_t),   ell_t = -1/2 v_t^T S_t^-1 v_t - 1/2 log|S_t| - (#observed / 2) log 2 pi
//
// which is a parallel scan over time of d x d times d x S products: the S means ride as the COLUMNS of one matrix (a 64 x 64 x 16 MFMA product per
// step at C5's sizes instead of sixteen 64^3 combines with a pivoted LU each).  Three launches, chunked like every scan here: chunk composites
// (A-products shared, one d x S offset block per chunk), a short sequential pass over the chunk composites, and the down pass that re-walks each chunk
// from its true start, writes the means and accumulates the log-likelihood.  Same C2-style hoist as affine_shared.h, for d up to the wide path's limit.
//
// Table row of transition i -> i + 1 (GRow; every matrix is stored as its padded LDS image, leading dimension ldp_(.), so a row is staged with
// straight linear copies):   [ H F (p x d) | A (d x d) | X = S^-1 H P^- = K^T (p x d) | S^-1 (p x p) | g (d) | hb (p) | mask (p) | ellc | ok ].

struct GRow {
    int d, p, ldd, ldp;
    long long oHF, oA, oX, oS, oV, size;
    __host__ __device__ GRow(int d_, int p_) : d(d_), p(p_), ldd(ldp_(d_)), ldp(ldp_(p_)) {
        oHF = 0;
        oA = oHF + (long long)p * ldd;
        oX = oA + (long long)d * ldd;
        oS = oX + (long long)p * ldd;
        oV = oS + (long long)p * ldp;
        size = (oV + d + 2 * p + 2 + 3) & ~3ll;
    }
};

// one padded matrix image (<= 85 x 85 reals) in flight: fetched a step ahead into registers, dropped into LDS at the top of its step
constexpr int SH_NR = 8;
template <typename R, int NRI = SH_NR> struct Img {  // NRI = 5 holds 71 x 71 (C5's 64 x 65 images): three registers per image less than the general 8
    R v[NRI];
};
template <typename R, int NRI> __device__ __forceinline__ void img_fetch(Img<R, NRI>& g, const R* __restrict__ src, int count, int tid) {
#pragma unroll
    for (int u = 0; u < NRI; ++u) {
        const int e = tid + u * NT;
        g.v[u] = e < count ? src[e] : (R)0;
    }
}
template <typename R, int NRI> __device__ __forceinline__ void img_drop(const Img<R, NRI>& g, R* dst, int count, int tid) {
#pragma unroll
    for (int u = 0; u < NRI; ++u) {
        const int e = tid + u * NT;
        if (e < count) dst[e] = g.v[u];
    }
}
// the observations of one time step of a block of CB sequences: y[k][sl] (p x CB, ld ldc), missing entries as 0 (the mask row deletes them)
constexpr int SH_NY = 4;  // p * CB <= 4 NT (the host picks CB accordingly)
template <typename R> struct YTile {
    const R* base[SH_NY];  // &ys[sequence, t = 0, k], null outside the block / beyond S
    int off[SH_NY];        // k * ldc + sl
    R v[SH_NY];
};
template <typename R> __device__ __forceinline__ void ytile_init(YTile<R>& y, const FilterArgs& a, int s0, int CB, int S, int p, int ldc, int tid) {
#pragma unroll
    for (int u = 0; u < SH_NY; ++u) {
        const int e = tid + u * NT;
        const int sl = e / p, k = e - sl * p, s = s0 + sl;
        const bool v = sl < CB && s < S;
        y.base[u] = v ? at<R>(a.ys, s / a.d.B, 0, s % a.d.B) + (long long)k * a.ys.se : nullptr;
        y.off[u] = v ? k * ldc + sl : -1;
        y.v[u] = 0;
    }
}
template <typename R> __device__ __forceinline__ void ytile_fetch(YTile<R>& y, long long tst) {
#pragma unroll
    for (int u = 0; u < SH_NY; ++u) y.v[u] = y.base[u] ? y.base[u][tst] : (R)0;
}
template <typename R> __device__ __forceinline__ void ytile_drop(const YTile<R>& y, R* Y) {
#pragma unroll
    for (int u = 0; u < SH_NY; ++u)
        if (y.off[u] >= 0) Y[y.off[u]] = finite_(y.v[u]) ? y.v[u] : (R)0;
}

// ---- every sequence has the missing-observation pattern of sequence 0? (flag |= 1 otherwise) ------------------------------------------------
template <typename R> __global__ void __launch_bounds__(256) wk_mask_check(FilterArgs a, int* __restrict__ flag) {
    const int p = a.dy, S = a.d.S();
    const long long total = (long long)a.d.T * p;
    int bad = 0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long t = e / p;
        const int k = (int)(e - t * p);
        const bool f0 = finite_(at<R>(a.ys, 0, t, 0)[(long long)k * a.ys.se]);
        for (int s = 1; s < S; ++s) bad |= finite_(at<R>(a.ys, s / a.d.B, t, s % a.d.B)[(long long)k * a.ys.se]) != f0;
    }
    if (bad) atomicOr(flag, 1);
}

// ---- the gain-form row of transition i -> i + 1 from the filtered covariance P_i of the matrix filter; one workgroup per transition -------------
static size_t lds_gain_tab(size_t s, int d, int p) {
    const size_t ldd = ldp_(d), nct = 2 * (size_t)p + d, ldz = ldp_((int)nct);
    return 3 * al16(d * ldd * s) + 3 * al16(p * ldd * s) + al16(p * ldz * s) + 2 * al16(d * s) + 4 * al16(p * s) + al16((2 * (nct + 1) + NWV) * s) + al16(p) + 128;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_gain_tab(FilterArgs a, R* __restrict__ tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, p = a.dy, i = blockIdx.x;
    const long long t = (long long)i + 1;
    const int ldd = ldp_(d), ldpp = ldp_(p), nct = 2 * p + d, ldz = ldp_(nct);
    const GRow g(d, p);
    R* row = tab + (long long)i * g.size;
    Bump L{smem};
    R* F = L.take<R>(d * ldd);
    R* P = L.take<R>(d * ldd);
    R* T1 = L.take<R>(d * ldd);
    Obs<R> o;
    o.H_ = L.take<R>(p * ldd);
    R* HP = L.take<R>(p * ldd);
    R* HF = L.take<R>(p * ldd);
    R* Z = L.take<R>(p * ldz);  // [S | H_ P^- | I] -> [. | X | S^-1]
    R* bd = L.take<R>(d);
    R* g0 = L.take<R>(d);
    o.c_ = L.take<R>(p);
    o.y = L.take<R>(p);
    R* hb = L.take<R>(p);
    R* piv = L.take<R>(p);
    R* rowbuf = L.take<R>(2 * (nct + 1) + NWV);
    o.nan = L.take<unsigned char>(p);
    o.cnt = L.take<int>(1);
    // every global operand of the step is requested before anything waits: one 1024-lane workgroup fills a CU, so nothing else hides the latency
    Img<R> rF, rP, rQ, rH, rR;
    img_fetch(rF, at<R>(a.Fs, 0, i, 0), d * d, tid);
    img_fetch(rP, at<R>(a.Ps, 0, i, 0), d * d, tid);
    img_fetch(rH, at<R>(a.Hs, 0, t, 0), p * d, tid);
    img_fetch(rQ, at<R>(a.Qs, 0, i, 0), d * d, tid);
    img_fetch(rR, at<R>(a.Rs, 0, t, 0), p * p, tid);
    const R vb_ = tid < d ? at<R>(a.bs, 0, i, 0)[tid] : (R)0;
    // the observation pattern the row is built for: sequence 0's, or the caller's chain-independent carrier (FilterArgs::mask_ys: a sweep whose pseudo-observations
    // are finite by construction vouches for the pattern, and one diverged chain must not delete rows of everybody's gains)
    const Arr& my = a.mask_ys.ptr ? a.mask_ys : a.ys;
    const R vy_ = tid < p ? at<R>(my, 0, t, 0)[(long long)tid * my.se] : (R)0;
    const R vc_ = tid < p ? at<R>(a.cs, 0, t, 0)[tid] : (R)0;
    if (tid == 0) *o.cnt = 0;
    __syncthreads();
    if (tid < d) bd[tid] = vb_;
    if (tid < p) {  // load_obs (filtering.py:89-100): a missing component deletes its row of H and its offset
        const bool nn = !finite_(vy_);
        o.nan[tid] = nn;
        o.y[tid] = vy_;
        o.c_[tid] = nn ? (R)0 : vc_;
        if (!nn) atomicAdd(o.cnt, 1);
    }
#pragma unroll
    for (int u = 0; u < SH_NR; ++u) {
        const int e = tid + u * NT;
        if (e < d * d) {
            const int r = e / d, q = e - r * d;
            F[r * ldd + q] = rF.v[u];
            P[r * ldd + q] = rP.v[u];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SH_NR; ++u) {
        const int e = tid + u * NT;
        if (e < p * d) {
            const int r = e / d, q = e - r * d;
            o.H_[r * ldd + q] = o.nan[r] ? (R)0 : rH.v[u];
        }
    }
#if defined(AUXSSM_GT_PHASE)
    if (AUXSSM_GT_PHASE == 1) return;
#endif
    // P^- = F P F^T + Q (not symmetrised, filtering.py:200-201 / predict)
    gemm<false, false>(d, d, d, F, ldd, P, ldd, T1, ldd, (R)1, (R)0, tid);
    gemm<false, true>(d, d, d, T1, ldd, F, ldd, P, ldd, (R)1, (R)0, tid);
#pragma unroll
    for (int u = 0; u < SH_NR; ++u) {
        const int e = tid + u * NT;
        if (e < d * d) {
            const int r = e / d, q = e - r * d;
            P[r * ldd + q] += rQ.v[u];
        }
    }
    __syncthreads();
#if defined(AUXSSM_GT_PHASE)
    if (AUXSSM_GT_PHASE == 2) return;
#endif
    // S = H_ P^- H_^T + R_ ; right-hand sides H_ P^- and the identity.  A step with nothing observed needs no special case: H_ = 0, every row deleted -> X = 0, S^-1
    // rows 0, half log-determinant 0, i.e. A = F, g = b, ell_t = 0 (_passthrough, filtering.py:239-248)
    gemm<false, false>(p, d, d, o.H_, ldd, P, ldd, HP, ldd, (R)1, (R)0, tid);
    gemm<false, true>(p, p, d, HP, ldd, o.H_, ldd, Z, ldz, (R)1, (R)0, tid);
    {
        // the value the reference computes for the upper entry (q, r), q <= r: (H_ P^- H_^T)[r][q] + R[q][r], mirrored (as wk_filter_t0)
#pragma unroll
        for (int u = 0; u < SH_NR; ++u) {
            const int e = tid + u * NT;
            if (e < p * p) {
                const int q = e / p, r = e - q * p;
                if (q <= r) {
                    const R v = Z[r * ldz + q] + ((o.nan[r] || o.nan[q]) ? (R)0 : rR.v[u]);
                    Z[r * ldz + q] = v;
                    Z[q * ldz + r] = v;
                }
            }
        }
        for (int r = tid / 64; r < p; r += NWV) {
            for (int q = tid & 63; q < d; q += 64) Z[r * ldz + p + q] = HP[r * ldd + q];
            for (int q = tid & 63; q < p; q += 64) Z[r * ldz + p + d + q] = r == q ? (R)1 : (R)0;
        }
    }
    __syncthreads();
#if defined(AUXSSM_GT_PHASE)
    if (AUXSSM_GT_PHASE == 3) return;
#endif
    R hl;
    const bool ok = spd_split_fits(p, nct, ldz) ? spd_solve_split<R>(Z, ldz, p, nct, o.nan, rowbuf, piv, &hl, tid)
                                                : spd_solve<R>(Z, ldz, p, nct, o.nan, rowbuf, piv, &hl, tid, true);
#if defined(AUXSSM_GT_PHASE)
    if (AUXSSM_GT_PHASE == 4) return;
#endif
    // H F;  A = F - X^T (H F);  hb = H_ b + c_;  g = b - X^T hb
    gemm<false, false>(p, d, d, o.H_, ldd, F, ldd, HF, ldd, (R)1, (R)0, tid);
    gemm<true, false>(d, d, p, Z + p, ldz, HF, ldd, T1, ldd, (R)-1, (R)1, tid, F, ldd);
    gemv<R, false>(p, d, o.H_, ldd, bd, hb, (R)1, (R)0, tid);
    for (int k = tid; k < p; k += NT) hb[k] = o.nan[k] ? (R)0 : hb[k] + o.c_[k];
    __syncthreads();
    gemv<R, true>(d, p, Z + p, ldz, hb, g0, (R)1, (R)0, tid);
#if defined(AUXSSM_GT_PHASE)
    if (AUXSSM_GT_PHASE == 5) return;
#endif
    const R bad = r_nan<R>();
    for (int e = tid; e < p * ldd; e += NT) {
        const int r = e / ldd, q = e - r * ldd;
        row[g.oHF + e] = q < d ? HF[e] : (R)0;
        row[g.oX + e] = q < d ? Z[r * ldz + p + q] : (R)0;
    }
    for (int e = tid; e < d * ldd; e += NT) {
        const int r = e / ldd, q = e - r * ldd;
        row[g.oA + e] = q < d ? (ok ? T1[e] : bad) : (R)0;  // a failed factorisation poisons the means from here on, as the reference's NaN gain does
    }
    for (int e = tid; e < p * ldpp; e += NT) {
        const int r = e / ldpp, q = e - r * ldpp;
        row[g.oS + e] = q < p ? Z[r * ldz + p + d + q] : (R)0;
    }
    R* vec = row + g.oV;
    for (int k = tid; k < d; k += NT) vec[k] = bd[k] - g0[k];
    for (int k = tid; k < p; k += NT) {
        vec[d + k] = hb[k];
        vec[d + p + k] = o.nan[k] ? (R)0 : (R)1;
    }
    if (tid == 0) {
        vec[d + 2 * p] = ok ? -hl - (R)(0.5 * LOG_2PI) * (R)*o.cnt : bad;
        vec[d + 2 * p + 1] = ok ? (R)1 : (R)0;
    }
}

// ---- chunk composites: G_j (d x S) = the chunk's recursion started from 0, Abar_j = the product of its A's (block 0 of the sequences computes it) ----
static size_t lds_mean_reduce(size_t s, int d, int p, int CB) {
    const size_t ldd = ldp_(d), ldg = ldp_(CB + d), ldc = ldp_(CB);
    return al16(d * ldd * s) + al16(p * ldd * s) + 2 * al16(d * ldg * s) + al16(p * ldc * s) + al16(d * s) + 64;
}
template <typename R, int NRI>
__global__ void __launch_bounds__(NT) wk_mean_reduce(FilterArgs a, const R* __restrict__ tab, R* __restrict__ aggA, R* __restrict__ aggG, int E, int ncb, int CB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, p = a.dy, n = a.d.T - 1, S = a.d.S();
    const int j = blockIdx.x / ncb, cb = blockIdx.x - j * ncb, s0 = cb * CB;
    const int i0 = j * E, i1 = i0 + E < n ? i0 + E : n;
    const int ldd = ldp_(d), ldg = ldp_(CB + d), ldc = ldp_(CB), Spad = ncb * CB;
    const bool doA = cb == 0;
    const GRow g(d, p);
    Bump L{smem};
    R* Al = L.take<R>(d * ldd);
    R* Xl = L.take<R>(p * ldd);
    R* GA[2] = {L.take<R>(d * ldg), L.take<R>(d * ldg)};  // [G (d x CB) | Abar (d x d)]
    R* Yl = L.take<R>(p * ldc);
    R* gl = L.take<R>(d);
    for (int e = tid; e < d * ldg; e += NT) {
        const int r = e / ldg, q = e - r * ldg;
        GA[0][e] = (q >= CB && q - CB == r) ? (R)1 : (R)0;
        GA[1][e] = 0;
    }
    YTile<R> yt;
    ytile_init<R>(yt, a, s0, CB, S, p, ldc, tid);
    Img<R, NRI> ia, ix;
    R gv = 0;
    {
        const R* row = tab + (long long)i0 * g.size;
        img_fetch(ia, row + g.oA, d * ldd, tid);
        img_fetch(ix, row + g.oX, p * ldd, tid);
        if (tid < d) gv = row[g.oV + tid];
        ytile_fetch<R>(yt, (long long)(i0 + 1) * a.ys.st);
    }
    int cur = 0;
    for (int i = i0; i < i1; ++i) {
        __syncthreads();  // the previous step's products have read Al / Xl / Yl / gl; its offsets are complete
        img_drop(ia, Al, d * ldd, tid);
        img_drop(ix, Xl, p * ldd, tid);
        ytile_drop<R>(yt, Yl);
        if (tid < d) gl[tid] = gv;
        __syncthreads();
        if (i + 1 < i1) {
            const R* row = tab + (long long)(i + 1) * g.size;
            img_fetch(ia, row + g.oA, d * ldd, tid);
            img_fetch(ix, row + g.oX, p * ldd, tid);
            if (tid < d) gv = row[g.oV + tid];
            ytile_fetch<R>(yt, (long long)(i + 2) * a.ys.st);
        }
        R* c0 = GA[cur];
        R* c1 = GA[cur ^ 1];
        gemm<false, false>(d, doA ? CB + d : CB, d, Al, ldd, c0, ldg, c1, ldg, (R)1, (R)0, tid);
        gemm<true, false>(d, CB, p, Xl, ldd, Yl, ldc, c1, ldg, (R)1, (R)1, tid);
        for (int e = tid; e < d * CB; e += NT) {
            const int r = e / CB, q = e - r * CB;
            c1[r * ldg + q] += gl[r];
        }
        cur ^= 1;
    }
    __syncthreads();
    const R* fin = GA[cur];
    for (int e = tid; e < d * CB; e += NT) {
        const int r = e / CB, q = e - r * CB;
        aggG[((long long)j * d + r) * Spad + s0 + q] = fin[r * ldg + q];
    }
    if (doA)
        for (int e = tid; e < d * ldd; e += NT) {
            const int r = e / ldd, q = e - r * ldd;
            aggA[(long long)j * d * ldd + e] = q < d ? fin[r * ldg + CB + q] : (R)0;
        }
}

// ---- composites of GROUPS of G consecutive chunk composites (the second level of the scan: with 256 chunks the plain sequential pass below is 256 dependent
// products on ONE workgroup, 0.46 ms at C5; groups of 16 make it 16 + 16 + 16) ----------------------------------------------------------------------
static size_t lds_mean_group(size_t s, int d, int CB) {
    const size_t ldd = ldp_(d), ldg = ldp_(CB + d), ldc = ldp_(CB);
    return al16(d * ldd * s) + 2 * al16(d * ldg * s) + al16(d * ldc * s) + 64;
}
template <typename R, int NRI>
__global__ void __launch_bounds__(NT) wk_mean_group(FilterArgs a, const R* __restrict__ aggA, const R* __restrict__ aggG, R* __restrict__ supA, R* __restrict__ supG, int nchunk_all,
                                                    int ncb, int CB, int G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, g = blockIdx.x / ncb, cb = blockIdx.x - g * ncb, s0 = cb * CB;
    const int ldd = ldp_(d), ldg = ldp_(CB + d), ldc = ldp_(CB), Spad = ncb * CB;
    const int j0 = g * G, j1 = j0 + G < nchunk_all ? j0 + G : nchunk_all;
    const bool doA = cb == 0;
    Bump L{smem};
    R* Ab = L.take<R>(d * ldd);
    R* GA[2] = {L.take<R>(d * ldg), L.take<R>(d * ldg)};  // [offset block (d x CB) | product of the A-composites (d x d)]
    R* Gl = L.take<R>(d * ldc);
    for (int e = tid; e < d * ldg; e += NT) {
        const int r = e / ldg, q = e - r * ldg;
        GA[0][e] = (q >= CB && q - CB == r) ? (R)1 : (R)0;
        GA[1][e] = 0;
    }
    Img<R, NRI> ia;
    Img<R> ig;
    auto fetch = [&](int j) {
        img_fetch(ia, aggA + (long long)j * d * ldd, d * ldd, tid);
#pragma unroll
        for (int u = 0; u < SH_NR; ++u) {
            const int e = tid + u * NT, r = e / CB, q = e - r * CB;
            ig.v[u] = e < d * CB ? aggG[((long long)j * d + r) * Spad + s0 + q] : (R)0;
        }
    };
    fetch(j0);
    int cur = 0;
    for (int j = j0; j < j1; ++j) {
        __syncthreads();
        img_drop(ia, Ab, d * ldd, tid);
#pragma unroll
        for (int u = 0; u < SH_NR; ++u) {
            const int e = tid + u * NT, r = e / CB, q = e - r * CB;
            if (e < d * CB) Gl[r * ldc + q] = ig.v[u];
        }
        __syncthreads();
        if (j + 1 < j1) fetch(j + 1);
        R* c0 = GA[cur];
        R* c1 = GA[cur ^ 1];
        gemm<false, false>(d, doA ? CB + d : CB, d, Ab, ldd, c0, ldg, c1, ldg, (R)1, (R)0, tid);
        for (int e = tid; e < d * CB; e += NT) {
            const int r = e / CB, q = e - r * CB;
            c1[r * ldg + q] += Gl[r * ldc + q];
        }
        cur ^= 1;
    }
    __syncthreads();
    const R* fin = GA[cur];
    for (int e = tid; e < d * CB; e += NT) {
        const int r = e / CB, q = e - r * CB;
        supG[((long long)g * d + r) * Spad + s0 + q] = fin[r * ldg + q];
    }
    if (doA)
        for (int e = tid; e < d * ldd; e += NT) {
            const int r = e / ldd, q = e - r * ldd;
            supA[(long long)g * d * ldd + e] = q < d ? fin[r * ldg + CB + q] : (R)0;
        }
}

// ---- the means at the chunk starts: M_0 = the t = 0 update (wk_filter_t0), M_{j+1} = Abar_j M_j + G_j; one workgroup per block of sequences ----
static size_t lds_mean_aggs(size_t s, int d, int CB) {
    const size_t ldd = ldp_(d), ldc = ldp_(CB);
    return al16(d * ldd * s) + 3 * al16(d * ldc * s) + 64;
}
template <typename R, int NRI>
__global__ void __launch_bounds__(NT) wk_mean_aggs(FilterArgs a, const R* __restrict__ aggA, const R* __restrict__ aggG, R* __restrict__ pre, int nchunk_all, int ncb, int CB,
                                                   int G, const R* __restrict__ start) {
    // workgroup (g, cb): the composites j0 = g G .. j1 - 1 of column block cb, started from start[g] (the means at the start of group g) or, start == null, from the
    // t = 0 update.  One group of all composites = the plain sequential pass; groups of G run in parallel once a pass over the GROUP composites (wk_mean_group)
    // has produced their starts.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, S = a.d.S(), g = blockIdx.x / ncb, cb = blockIdx.x - g * ncb, s0 = cb * CB;
    const int ldd = ldp_(d), ldc = ldp_(CB), Spad = ncb * CB;
    const int j0 = g * G, nchunk = (j0 + G < nchunk_all ? j0 + G : nchunk_all) - j0;
    aggA += (long long)j0 * d * ldd;
    aggG += (long long)j0 * d * Spad;
    pre += (long long)j0 * d * Spad;
    Bump L{smem};
    R* Ab = L.take<R>(d * ldd);
    R* M[2] = {L.take<R>(d * ldc), L.take<R>(d * ldc)};
    R* Gl = L.take<R>(d * ldc);
    if (start) {
        for (int e = tid; e < d * CB; e += NT) {
            const int r = e / CB, q = e - r * CB;
            M[0][r * ldc + q] = start[((long long)g * d + r) * Spad + s0 + q];
        }
    } else {
        for (int e = tid; e < d * CB; e += NT) {
            const int sl = e / d, r = e - sl * d, s = s0 + sl;
            M[0][r * ldc + sl] = s < S ? at<R>(a.ms, s / a.d.B, 0, s % a.d.B)[(long long)r * a.ms.se] : (R)0;
        }
    }
    Img<R, NRI> ia;
    Img<R> ig;  // d * CB <= 85 * 64 < 8 NT
    if (nchunk > 1) {
        img_fetch(ia, aggA, d * ldd, tid);
#pragma unroll
        for (int u = 0; u < SH_NR; ++u) {
            const int e = tid + u * NT, r = e / CB, q = e - r * CB;
            ig.v[u] = e < d * CB ? aggG[(long long)r * Spad + s0 + q] : (R)0;
        }
    }
    int cur = 0;
    for (int j = 0; j < nchunk; ++j) {
        __syncthreads();
        const R* m = M[cur];
        for (int e = tid; e < d * CB; e += NT) {
            const int r = e / CB, q = e - r * CB;
            pre[((long long)j * d + r) * Spad + s0 + q] = m[r * ldc + q];
        }
        if (j + 1 == nchunk) break;
        img_drop(ia, Ab, d * ldd, tid);
#pragma unroll
        for (int u = 0; u < SH_NR; ++u) {
            const int e = tid + u * NT, r = e / CB, q = e - r * CB;
            if (e < d * CB) Gl[r * ldc + q] = ig.v[u];
        }
        __syncthreads();
        if (j + 2 < nchunk) {
            img_fetch(ia, aggA + (long long)(j + 1) * d * ldd, d * ldd, tid);
#pragma unroll
            for (int u = 0; u < SH_NR; ++u) {
                const int e = tid + u * NT, r = e / CB, q = e - r * CB;
                ig.v[u] = e < d * CB ? aggG[((long long)(j + 1) * d + r) * Spad + s0 + q] : (R)0;
            }
        }
        gemm<false, false>(d, CB, d, Ab, ldd, m, ldc, M[cur ^ 1], ldc, (R)1, (R)1, tid, Gl, ldc);
        cur ^= 1;
    }
}

// ---- down pass: each chunk re-walked from its true start; means written, log-likelihood increments accumulated per sequence -------------------
static size_t lds_mean_down(size_t s, int d, int p, int CB) {
    const size_t ldd = ldp_(d), ldpp = ldp_(p), ldc = ldp_(CB);
    return al16((p + d) * ldd * s) + al16(p * ldd * s) + al16(p * ldpp * s) + 2 * al16(d * ldc * s) + al16((p + d) * ldc * s) + 2 * al16(p * ldc * s) +
           al16((d + 2 * p + 2) * s) + al16(NWV * CB * s) + 64;
}
template <typename R, int NRI>
__global__ void __launch_bounds__(NT) wk_mean_down(FilterArgs a, const R* __restrict__ tab, const R* __restrict__ pre, R* __restrict__ ellpart, int E, int nchunk, int ncb, int CB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, d = a.dx, p = a.dy, n = a.d.T - 1, S = a.d.S();
    const int j = blockIdx.x / ncb, cb = blockIdx.x - j * ncb, s0 = cb * CB;
    const int i0 = j * E, i1 = i0 + E < n ? i0 + E : n;
    const int ldd = ldp_(d), ldpp = ldp_(p), ldc = ldp_(CB), Spad = ncb * CB;
    const GRow g(d, p);
    Bump L{smem};
    R* HA = L.take<R>((p + d) * ldd);  // [H F ; A]
    R* Xl = L.take<R>(p * ldd);
    R* Sl = L.take<R>(p * ldpp);
    R* M[2] = {L.take<R>(d * ldc), L.take<R>(d * ldc)};
    R* VM = L.take<R>((p + d) * ldc);  // [H F m ; A m] -> rows < p become the innovations v
    R* Yl = L.take<R>(p * ldc);
    R* Ul = L.take<R>(p * ldc);
    R* vec = L.take<R>(d + 2 * p + 2);
    R* qpart = L.take<R>(NWV * CB);
    for (int e = tid; e < d * CB; e += NT) {
        const int r = e / CB, q = e - r * CB;
        M[0][r * ldc + q] = pre[((long long)j * d + r) * Spad + s0 + q];
    }
    YTile<R> yt;
    ytile_init<R>(yt, a, s0, CB, S, p, ldc, tid);
    Img<R, NRI> ih, ia, ix, is;
    R vv[3] = {0, 0, 0};  // d + 2 p + 2 <= 3 NT
    auto fetch = [&](int i) {
        const R* row = tab + (long long)i * g.size;
        img_fetch(ih, row + g.oHF, p * ldd, tid);
        img_fetch(ia, row + g.oA, d * ldd, tid);
        img_fetch(ix, row + g.oX, p * ldd, tid);
        img_fetch(is, row + g.oS, p * ldpp, tid);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = tid + u * NT;
            vv[u] = e < d + 2 * p + 2 ? row[g.oV + e] : (R)0;
        }
        ytile_fetch<R>(yt, (long long)(i + 1) * a.ys.st);
    };
    fetch(i0);
    R ellacc = 0;
    int cur = 0;
    for (int i = i0; i < i1; ++i) {
        __syncthreads();
        img_drop(ih, HA, p * ldd, tid);
        img_drop(ia, HA + p * ldd, d * ldd, tid);
        img_drop(ix, Xl, p * ldd, tid);
        img_drop(is, Sl, p * ldpp, tid);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = tid + u * NT;
            if (e < d + 2 * p + 2) vec[e] = vv[u];
        }
        ytile_drop<R>(yt, Yl);
        __syncthreads();
        if (i + 1 < i1) fetch(i + 1);
        const R* m = M[cur];
        R* mn = M[cur ^ 1];
        gemm<false, false>(p + d, CB, d, HA, ldd, m, ldc, VM, ldc, (R)1, (R)0, tid);
        // v = mask (y - H F m - hb)
        for (int e = tid; e < p * CB; e += NT) {
            const int k = e / CB, q = e - k * CB;
            VM[k * ldc + q] = vec[d + p + k] != (R)0 ? Yl[k * ldc + q] - VM[k * ldc + q] - vec[d + k] : (R)0;
        }
        __syncthreads();
        gemm<false, false>(p, CB, p, Sl, ldpp, VM, ldc, Ul, ldc, (R)1, (R)0, tid, nullptr, 0, false);
        gemm<true, false>(d, CB, p, Xl, ldd, Yl, ldc, mn, ldc, (R)1, (R)1, tid, VM + p * ldc, ldc);  // A m + K y  (trailing barrier: Ul complete too)
        // q_s = v_s^T S^-1 v_s: lane's sequence is tid % CB (NT % CB == 0), partial over its rows, then over the lanes / waves of that sequence
        {
            R acc = 0;
            for (int e = tid; e < p * CB; e += NT) {
                const int k = e / CB, q = e - k * CB;
                acc += VM[k * ldc + q] * Ul[k * ldc + q];
            }
            for (int off = CB; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
            if (lane < CB) qpart[wv * CB + lane] = acc;
        }
        // m_t = A m + K y + g: finished in place and written out, element (sequence sl, component r) -> ms[s, t, r] (indices on the fly: registers are what
        // this kernel is short of)
        for (int e = tid; e < d * CB; e += NT) {
            const int sl = e / d, r = e - sl * d, s = s0 + sl;
            const R val = mn[r * ldc + sl] + vec[r];
            mn[r * ldc + sl] = val;
            if (s < S) const_cast<R*>(at<R>(a.ms, s / a.d.B, (long long)i + 1, s % a.d.B))[(long long)r * a.ms.se] = val;
        }
        __syncthreads();
        if (tid < CB) {
            R q = 0;
            for (int w = 0; w < NWV; ++w) q += qpart[w * CB + tid];
            const R e = (R)-0.5 * q + vec[d + 2 * p];
            ellacc += isnan_(e) ? (R)0 : e;  // the reference's nansum (filtering.py:62)
        }
        cur ^= 1;
    }
    if (tid < CB && s0 + tid < S) ellpart[(long long)(s0 + tid) * nchunk + j] = ellacc;
}

// ---- the covariances of sequence 0 (the matrix filter's) copied to the other sequences' output slots, t >= 1; one workgroup per time step ------
template <typename R> __global__ void __launch_bounds__(NT) wk_ps_bcast(FilterArgs a) {
    const int tid = threadIdx.x, d = a.dx, S = a.d.S();
    const long long t = (long long)blockIdx.x + 1;
    const R* src = at<R>(a.Ps, 0, t, 0);
    R v[SH_NR];
#pragma unroll
    for (int u = 0; u < SH_NR; ++u) {
        const int e = tid + u * NT;
        v[u] = e < d * d ? src[e] : (R)0;
    }
    for (int s = 1; s < S; ++s) {
        R* dst = const_cast<R*>(at<R>(a.Ps, s / a.d.B, t, s % a.d.B));
#pragma unroll
        for (int u = 0; u < SH_NR; ++u) {
            const int e = tid + u * NT;
            if (e < d * d) dst[e] = v[u];
        }
    }
}

// =================================================================================================================================================
// The pathwise SAMPLER of sequences that share the model and the filtered covariances (sampling.py:60-124): the gains
// G_t = P_t (S_t^-1 F)^T and the factors Lc_t = chol(P_t - G_t S_t G_t^T) depend on neither the sequence's means nor its noise, so wk_sample_init runs ONCE
// per time step (on sequence 0: its [G | e] records are the G table, Lc goes to a second table) instead of once per (sequence, time step), the products of
// the gains over a chunk are formed once (wk_sscan_reduce on that one table), and the sequences carry only d-vectors: their increments
// e_t = m_t - G_t (F m_t + b) + Lc_t eps_t and the recursion x_t = G_t x_{t+1} + e_t, with the sequences as the COLUMNS of d x CB matrices -- every step of
// every pass is one d x d by d x CB product on the matrix cores.  Scan position j <-> time T - 1 - j, as in wide.hip.  ec: [T][S][d].
// =================================================================================================================================================
static size_t lds_samp_evec(size_t s, int d, int CB) { return 3 * al16(d * (size_t)ldp_(d) * s) + 5 * al16(d * (size_t)ldp_(CB) * s) + al16(d * s) + 64; }
template <typename R> __global__ void __launch_bounds__(NT) wk_samp_evec(SampleArgs a, const R* __restrict__ gtab, const R* __restrict__ ltab, R* __restrict__ ec, int ncb, int CB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, T = a.d.T, S = a.d.S();
    const int j = blockIdx.x / ncb, cb = blockIdx.x - j * ncb, s0 = cb * CB, nc = min(CB, S - s0);
    const long long t = (long long)T - 1 - j;
    const int ldd = ldp_(d), ldc = ldp_(CB);
    const long long ne = (long long)d * d + d;
    Bump L{smem};
    R* G = L.take<R>(d * ldd);
    R* Lc = L.take<R>(d * ldd);
    R* F = L.take<R>(d * ldd);
    R* M = L.take<R>(d * ldc);
    R* Ee = L.take<R>(d * ldc);
    R* PM = L.take<R>(d * ldc);
    R* TV = L.take<R>(d * ldc);
    R* LE = L.take<R>(d * ldc);
    R* bd = L.take<R>(d);
    // the sequences' means and noise of this time step as columns
    for (int e = tid; e < d * nc; e += NT) {
        const int q = e / d, k = e - q * d, sq = s0 + q;
        M[k * ldc + q] = at<R>(a.ms, sq / a.d.B, t, sq % a.d.B)[k];
        Ee[k * ldc + q] = at<R>(a.eps, sq / a.d.B, t, sq % a.d.B)[k];
    }
    load_mat<R>(Lc, ldd, ltab + (long long)j * d * d, d, d, tid);
    gemm<false, false>(d, nc, d, Lc, ldd, Ee, ldc, LE, ldc, (R)1, (R)0, tid);  // Lc eps (the factor's upper part is zero)
    R* out = ec + ((long long)j * S + s0) * d;
    if (j == 0) {  // _sample_last_step :115-124: e = m + Lc eps
        for (int e = tid; e < d * nc; e += NT) {
            const int q = e / d, k = e - q * d;
            out[(long long)q * d + k] = M[k * ldc + q] + LE[k * ldc + q];
        }
        return;
    }
    load_mat<R>(G, ldd, gtab + (long long)j * ne, d, d, tid);
    load_mat<R>(F, ldd, at<R>(a.Fs, 0, t, 0), d, d, tid);
    load_vec<R>(bd, at<R>(a.bs, 0, t, 0), d, tid);
    // inc = m - G (F m + b) + Lc eps  (:108-112)
    gemm<false, false>(d, nc, d, F, ldd, M, ldc, PM, ldc, (R)1, (R)0, tid);
    for (int e = tid; e < d * nc; e += NT) {
        const int q = e / d, k = e - q * d;
        PM[k * ldc + q] += bd[k];
    }
    __syncthreads();
    gemm<false, false>(d, nc, d, G, ldd, PM, ldc, TV, ldc, (R)1, (R)0, tid);
    for (int e = tid; e < d * nc; e += NT) {
        const int q = e / d, k = e - q * d;
        out[(long long)q * d + k] = M[k * ldc + q] - TV[k * ldc + q] + LE[k * ldc + q];
    }
}
// E (d x nc) <- the increments of position i, sequences s0 .. s0 + nc
template <typename R> __device__ __forceinline__ void samp_load_cols(R* E, int ldc, const R* __restrict__ src, int d, int nc, int tid) {
    for (int e = tid; e < d * nc; e += NT) {
        const int q = e / d, k = e - q * d;
        E[k * ldc + q] = src[(long long)q * d + k];
    }
    __syncthreads();
}
static size_t lds_samp_scan(size_t s, int d, int CB) { return al16(d * (size_t)ldp_(d) * s) + 3 * al16(d * (size_t)ldp_(CB) * s) + 64; }
// a chunk's composite increment of every sequence: E <- G_i E + e_i over the chunk (the product of the gains is the shared table's aggregate)
template <typename R> __global__ void __launch_bounds__(NT) wk_samp_reduce(const R* __restrict__ gtab, const R* __restrict__ ec, R* __restrict__ eagg, int n, int E_, int nchunk, int d, int S,
                                                                          int ncb, int CB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, ch = blockIdx.x / ncb, cb = blockIdx.x - ch * ncb, s0 = cb * CB, nc = min(CB, S - s0);
    const int ldd = ldp_(d), ldc = ldp_(CB);
    const long long ne = (long long)d * d + d;
    Bump L{smem};
    R* G = L.take<R>(d * ldd);
    R* Ea = L.take<R>(d * ldc);
    R* Eb = L.take<R>(d * ldc);
    R* Ec = L.take<R>(d * ldc);
    const int i0 = ch * E_, i1 = min(n, i0 + E_);
    samp_load_cols<R>(Ea, ldc, ec + ((long long)i0 * S + s0) * d, d, nc, tid);
    for (int i = i0 + 1; i < i1; ++i) {
        load_mat<R>(G, ldd, gtab + (long long)i * ne, d, d, tid);
        samp_load_cols<R>(Ec, ldc, ec + ((long long)i * S + s0) * d, d, nc, tid);
        gemm<false, false>(d, nc, d, G, ldd, Ea, ldc, Eb, ldc, (R)1, (R)1, tid, Ec, ldc);
        R* sw = Ea;
        Ea = Eb;
        Eb = sw;
    }
    R* out = eagg + ((long long)ch * S + s0) * d;
    for (int e = tid; e < d * nc; e += NT) {
        const int q = e / d, k = e - q * d;
        out[(long long)q * d + k] = Ea[k * ldc + q];
    }
}
// the exclusive prefixes of the chunk composites, every sequence of a column block in one workgroup: pre[ch] for ch >= 1 (gagg: the shared gain products)
template <typename R> __global__ void __launch_bounds__(NT) wk_samp_aggs(const R* __restrict__ gagg, const R* __restrict__ eagg, R* __restrict__ pre, int nchunk, int d, int S, int CB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, s0 = blockIdx.x * CB, nc = min(CB, S - s0);
    const int ldd = ldp_(d), ldc = ldp_(CB);
    const long long ne = (long long)d * d + d;
    Bump L{smem};
    R* G = L.take<R>(d * ldd);
    R* Ea = L.take<R>(d * ldc);
    R* Eb = L.take<R>(d * ldc);
    R* Ec = L.take<R>(d * ldc);
    samp_load_cols<R>(Ea, ldc, eagg + (long long)s0 * d, d, nc, tid);
    for (int ch = 1; ch < nchunk; ++ch) {
        R* out = pre + ((long long)ch * S + s0) * d;
        for (int e = tid; e < d * nc; e += NT) {
            const int q = e / d, k = e - q * d;
            out[(long long)q * d + k] = Ea[k * ldc + q];
        }
        if (ch + 1 < nchunk) {
            load_mat<R>(G, ldd, gagg + (long long)ch * ne, d, d, tid);
            samp_load_cols<R>(Ec, ldc, eagg + ((long long)ch * S + s0) * d, d, nc, tid);
            gemm<false, false>(d, nc, d, G, ldd, Ea, ldc, Eb, ldc, (R)1, (R)1, tid, Ec, ldc);
            R* sw = Ea;
            Ea = Eb;
            Eb = sw;
        }
    }
}
// the trajectories: from the chunk's prefix, x <- G_i x + e_i position by position; position i is time T - 1 - i
template <typename R> __global__ void __launch_bounds__(NT) wk_samp_down(SampleArgs a, const R* __restrict__ gtab, const R* __restrict__ ec, const R* __restrict__ pre, int E_, int nchunk,
                                                                        int ncb, int CB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, n = a.d.T, S = a.d.S();
    const int ch = blockIdx.x / ncb, cb = blockIdx.x - ch * ncb, s0 = cb * CB, nc = min(CB, S - s0);
    const int ldd = ldp_(d), ldc = ldp_(CB);
    const long long ne = (long long)d * d + d;
    Bump L{smem};
    R* G = L.take<R>(d * ldd);
    R* Ea = L.take<R>(d * ldc);
    R* Eb = L.take<R>(d * ldc);
    R* Ec = L.take<R>(d * ldc);
    const int i0 = ch * E_, i1 = min(n, i0 + E_);
    samp_load_cols<R>(Ea, ldc, (ch == 0 ? ec : pre + (long long)ch * S * d) + (long long)s0 * d, d, nc, tid);
    for (int i = i0; i < i1; ++i) {
        if (!(ch == 0 && i == 0)) {
            load_mat<R>(G, ldd, gtab + (long long)i * ne, d, d, tid);
            samp_load_cols<R>(Ec, ldc, ec + ((long long)i * S + s0) * d, d, nc, tid);
            gemm<false, false>(d, nc, d, G, ldd, Ea, ldc, Eb, ldc, (R)1, (R)1, tid, Ec, ldc);
            R* sw = Ea;
            Ea = Eb;
            Eb = sw;
        }
        for (int e = tid; e < d * nc; e += NT) {
            const int q = e / d, k = e - q * d, sq = s0 + q;
            const_cast<R*>(at<R>(a.xs, sq / a.d.B, (long long)n - 1 - i, sq % a.d.B))[k] = Ea[k * ldc + q];
        }
        __syncthreads();
    }
}

// =================================================================================================================================================
// The sweep's LOG-DENSITIES for chains that share the model (kalman/generic.py:88-89, :98-106; wk_sweep_logpdf factorises Q_{t-1} and R_t once per (chain, time
// step)): per time step ONE table row -- Q_{t-1}^-1 (P0^-1 at t = 0), R_t^-1 over the kept components, their half log-determinants -- and the chains as the
// columns of d x CB matrices: residuals by products with F and H, quadratic forms r^T (S^-1 r) by one more product and a column-wise dot.
// Row layout: [Qinv d*d | Rinv po*po | hlQ, hlR, dimR, okQ, okR, anynan, 0, 0].
// =================================================================================================================================================
__host__ __device__ inline size_t lp_row(int d, int po) { return (size_t)d * d + (size_t)po * po + 8; }
static size_t lds_lp_tab(size_t s, int d, int po) {
    const int n = std::max(d, po);
    return al16(n * (size_t)ldp_(2 * n) * s) + al16(n * s) + al16((2 * (2 * n + 1) + NWV) * s) + al16(n) + 256;
}
template <typename R> __global__ void __launch_bounds__(NT) wk_lp_tab(SweepLogpdfArgs a, R* __restrict__ tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, po = a.po, t = blockIdx.x, nmax = d > po ? d : po;
    Bump L{smem};
    const int ldz = ldp_(2 * nmax);
    R* Z = L.take<R>(nmax * ldz);
    R* piv = L.take<R>(nmax);
    R* rowbuf = L.take<R>(2 * (2 * nmax + 1) + NWV);
    unsigned char* skip = L.take<unsigned char>(nmax);
    __shared__ int s_any;
    R* row = tab + (size_t)t * lp_row(d, po);
    // transition covariance (P0 at t = 0): Z = [Q | I] -> [. | Q^-1]
    const R* cov = t == 0 ? at<R>(a.P0, 0, 0, 0) : at<R>(a.Qs, 0, (long long)t - 1, 0);
    for (int i = tid / 64; i < d; i += NWV)
        for (int j = tid & 63; j < 2 * d; j += 64) Z[i * ldz + j] = j < d ? cov[(long long)(i >= j ? i : j) * d + (i >= j ? j : i)] : (j - d == i ? (R)1 : (R)0);
    __syncthreads();
    R hl = 0;
    const bool okq = spd_solve<R>(Z, ldz, d, 2 * d, nullptr, rowbuf, piv, &hl, tid, true);
    for (int i = tid / 64; i < d; i += NWV)
        for (int j = tid & 63; j < d; j += 64) row[(size_t)i * d + j] = Z[i * ldz + d + j];
    __syncthreads();
    // observation covariance over the kept components
    const R* yg = at<R>(a.ys, 0, t, 0);
    if (tid == 0) s_any = 0;
    __syncthreads();
    int dm = 0;
    for (int k = tid; k < po; k += NT) {
        const bool nanv = a.ys.sc == 0 && !finite_(yg[k]);   // (per-chain observations come with the reference policy: nothing is deleted)
        skip[k] = (a.nan_policy == 1) && nanv;
        if (nanv) atomicOr(&s_any, 1);
    }
    __syncthreads();
    const R* Rg = at<R>(a.Rs, 0, t, 0);
    for (int i = tid / 64; i < po; i += NWV)
        for (int j = tid & 63; j < 2 * po; j += 64) {
            R v;
            if (j < po) v = (skip[i] || skip[j]) ? (R)0 : Rg[(long long)(i >= j ? i : j) * po + (i >= j ? j : i)];
            else v = (j - po == i && !skip[i]) ? (R)1 : (R)0;
            Z[i * ldz + j] = v;
        }
    __syncthreads();
    R hlr = 0;
    const bool okr = spd_solve<R>(Z, ldz, po, 2 * po, a.nan_policy == 1 ? skip : nullptr, rowbuf, piv, &hlr, tid, true);
    R* rinv = row + (size_t)d * d;
    for (int i = tid / 64; i < po; i += NWV)
        for (int j = tid & 63; j < po; j += 64) rinv[(size_t)i * po + j] = (skip[i] || skip[j]) ? (R)0 : Z[i * ldz + po + j];
    for (int k = tid; k < po; k += NT) dm += skip[k] ? 0 : 1;
    const R dimr = block_sum<R>((R)dm, rowbuf, tid);
    if (tid == 0) {
        R* sc = rinv + (size_t)po * po;
        sc[0] = hl, sc[1] = hlr, sc[2] = dimr, sc[3] = okq ? (R)1 : (R)0, sc[4] = okr ? (R)1 : (R)0, sc[5] = s_any ? (R)1 : (R)0, sc[6] = 0, sc[7] = 0;
    }
}
static size_t lds_lp_cols(size_t s, int d, int po, int CB) {
    const int n = std::max(d, po);
    return 2 * al16(n * (size_t)ldp_(n) * s) + 9 * al16(n * (size_t)ldp_(CB) * s) + 2 * al16(n * s) + al16(10 * CB * s) + 256;
}
// column-wise sum_k A[k][q] B[k][q] for q < nc into out[q] (every lane-group of 16 lanes owns a column: 64 columns per pass)
template <typename R> __device__ __forceinline__ void col_dots(const R* A, const R* B, int ldc, int n, int nc, R* out, int tid) {
    const int q = tid >> 4, l = tid & 15;
    if (q < nc) {
        R sacc = 0;
        for (int k = l; k < n; k += 16) sacc += A[k * ldc + q] * B[k * ldc + q];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) sacc += __shfl_xor(sacc, off, 64);
        if (l == 0) out[q] = sacc;
    }
    __syncthreads();
}
// any non-finite entry in column q of A (n x nc)?  flags[q]
template <typename R> __device__ __forceinline__ void col_bad(const R* A, int ldc, int n, int nc, R* flags, int tid) {
    const int q = tid >> 4, l = tid & 15;
    if (q < nc) {
        int b = 0;
        for (int k = l; k < n; k += 16) b |= finite_(A[k * ldc + q]) ? 0 : 1;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) b |= __shfl_xor(b, off, 64);
        if (l == 0) flags[q] = b ? (R)1 : (R)0;
    }
    __syncthreads();
}
template <typename R> __global__ void __launch_bounds__(NT) wk_lp_cols(SweepLogpdfArgs a, const R* __restrict__ tab, R* __restrict__ part, int ncb, int CB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, d = a.dx, po = a.po, T = a.d.T, C = a.d.C, nmax = d > po ? d : po;
    const int t = blockIdx.x / ncb, cb = blockIdx.x - t * ncb, c0 = cb * CB, nc = min(CB, C - c0);
    const int ldn = ldp_(nmax), ldc = ldp_(CB);
    Bump L{smem};
    R* W = L.take<R>(nmax * ldn);    // Qinv / Rinv
    R* FH = L.take<R>(nmax * ldn);   // F / H
    R* X = L.take<R>(nmax * ldc);
    R* XP = L.take<R>(nmax * ldc);
    R* U = L.take<R>(nmax * ldc);
    R* XQ = L.take<R>(nmax * ldc);
    R* XPQ = L.take<R>(nmax * ldc);
    R* R1 = L.take<R>(nmax * ldc);
    R* R2 = L.take<R>(nmax * ldc);
    R* S1 = L.take<R>(nmax * ldc);
    R* S2 = L.take<R>(nmax * ldc);
    R* v1 = L.take<R>(nmax);
    R* v2 = L.take<R>(nmax);
    R* sc = L.take<R>(10 * CB);  // per column: q_ob1, q_ob2, bad_ob1, bad_ob2, q_pr1, q_pr2, bad_pr1, bad_pr2, (2 spare)
    const R* row = tab + (size_t)t * lp_row(d, po);
    const R* tsc = row + (size_t)d * d + (size_t)po * po;
    const R hlq = tsc[0], hlr = tsc[1], dimr = tsc[2];
    const bool okq = tsc[3] != (R)0, okr = tsc[4] != (R)0;
    // the chains' values of this time step (and the one before) as columns
    for (int e = tid; e < d * nc; e += NT) {
        const int q = e / d, k = e - q * d, c = c0 + q;
        X[k * ldc + q] = at<R>(a.x, c, t, 0)[k];
        XP[k * ldc + q] = at<R>(a.xp, c, t, 0)[k];
        U[k * ldc + q] = at<R>(a.u, c, t, 0)[k];
        if (t > 0) {
            XQ[k * ldc + q] = at<R>(a.x, c, (long long)t - 1, 0)[k];
            XPQ[k * ldc + q] = at<R>(a.xp, c, (long long)t - 1, 0)[k];
        }
    }
    // ---- observation block: r = y - (H x + c) over the kept components; q = r^T Rinv r
    load_mat<R>(FH, ldn, at<R>(a.Hs, 0, t, 0), po, d, tid);
    const bool ypc = a.ys.sc != 0;                       // per-chain observations (reference NaN policy: a non-finite residual drops the term, nothing is deleted)
    const Arr& ysx = a.ys_x.ptr ? a.ys_x : a.ys;          // the observations scored against x
    for (int k = tid; k < po; k += NT) v1[k] = ypc ? (R)0 : at<R>(a.ys, 0, t, 0)[k], v2[k] = at<R>(a.cs, 0, t, 0)[k];
    if (ypc) {
        for (int e = tid; e < po * nc; e += NT) {
            const int q = e / po, k = e - q * po, c = c0 + q;
            S1[k * ldc + q] = at<R>(a.ys, c, t, 0)[k];
            S2[k * ldc + q] = at<R>(ysx, c, t, 0)[k];
        }
    }
    __syncthreads();
    gemm<false, false>(po, nc, d, FH, ldn, XP, ldc, R1, ldc, (R)1, (R)0, tid);
    gemm<false, false>(po, nc, d, FH, ldn, X, ldc, R2, ldc, (R)1, (R)0, tid);
    for (int e = tid; e < po * nc; e += NT) {
        const int k = e / nc, q = e - k * nc;
        const bool sk = !ypc && (a.nan_policy == 1) && !finite_(v1[k]);
        const R y1 = ypc ? S1[k * ldc + q] : v1[k], y2 = ypc ? S2[k * ldc + q] : v1[k];
        R1[k * ldc + q] = sk ? (R)0 : y1 - (v2[k] + R1[k * ldc + q]);
        R2[k * ldc + q] = sk ? (R)0 : y2 - (v2[k] + R2[k * ldc + q]);
    }
    __syncthreads();
    col_bad<R>(R1, ldc, po, nc, sc + 2 * CB, tid);
    col_bad<R>(R2, ldc, po, nc, sc + 3 * CB, tid);
    load_mat<R>(W, ldn, row + (size_t)d * d, po, po, tid);
    gemm<false, false>(po, nc, po, W, ldn, R1, ldc, S1, ldc, (R)1, (R)0, tid);
    gemm<false, false>(po, nc, po, W, ldn, R2, ldc, S2, ldc, (R)1, (R)0, tid);
    col_dots<R>(R1, S1, ldc, po, nc, sc, tid);
    col_dots<R>(R2, S2, ldc, po, nc, sc + CB, tid);
    // ---- transition block: r = x_t - (F x_{t-1} + b)  (t = 0: x_0 - m0); q = r^T Qinv r
    if (t == 0) {
        for (int k = tid; k < d; k += NT) v1[k] = at<R>(a.m0, 0, 0, 0)[k];
        __syncthreads();
        for (int e = tid; e < d * nc; e += NT) {
            const int k = e / nc, q = e - k * nc;
            R1[k * ldc + q] = XP[k * ldc + q] - v1[k];
            R2[k * ldc + q] = X[k * ldc + q] - v1[k];
        }
        __syncthreads();
    } else {
        load_mat<R>(FH, ldn, at<R>(a.Fs, 0, (long long)t - 1, 0), d, d, tid);
        for (int k = tid; k < d; k += NT) v1[k] = at<R>(a.bs, 0, (long long)t - 1, 0)[k];
        __syncthreads();
        gemm<false, false>(d, nc, d, FH, ldn, XPQ, ldc, R1, ldc, (R)1, (R)0, tid);
        gemm<false, false>(d, nc, d, FH, ldn, XQ, ldc, R2, ldc, (R)1, (R)0, tid);
        for (int e = tid; e < d * nc; e += NT) {
            const int k = e / nc, q = e - k * nc;
            R1[k * ldc + q] = XP[k * ldc + q] - (R1[k * ldc + q] + v1[k]);
            R2[k * ldc + q] = X[k * ldc + q] - (R2[k * ldc + q] + v1[k]);
        }
        __syncthreads();
    }
    col_bad<R>(R1, ldc, d, nc, sc + 6 * CB, tid);
    col_bad<R>(R2, ldc, d, nc, sc + 7 * CB, tid);
    load_mat<R>(W, ldn, row, d, d, tid);
    gemm<false, false>(d, nc, d, W, ldn, R1, ldc, S1, ldc, (R)1, (R)0, tid);
    gemm<false, false>(d, nc, d, W, ldn, R2, ldc, S2, ldc, (R)1, (R)0, tid);
    col_dots<R>(R1, S1, ldc, d, nc, sc + 4 * CB, tid);
    col_dots<R>(R2, S2, ldc, d, nc, sc + 5 * CB, tid);
    // ---- per chain: the auxiliary block N(u; x, delta / 2 I), the MH correction (generic.py:103-105), the five sums' terms (wk_sweep_logpdf's own rules)
    if (tid < nc) {
        const int q = tid, c = c0 + q;
        const R hd = (R)(0.5 * arg_delta(a)), sd = sqrt_(hd);
        R q1 = 0, q2 = 0, corr = 0;
        bool b1 = false, b2 = false;
        for (int k = 0; k < d; ++k) {
            const R e1 = U[k * ldc + q] - XP[k * ldc + q], e2 = U[k * ldc + q] - X[k * ldc + q];
            b1 = b1 || !finite_(e1);
            b2 = b2 || !finite_(e2);
            const R z1 = e1 / sd, z2 = e2 / sd;
            q1 += z1 * z1;
            q2 += z2 * z2;
            const R f1 = XP[k * ldc + q] - U[k * ldc + q], f2 = X[k * ldc + q] - U[k * ldc + q];
            corr += (f1 * f1 - f2 * f2) / (R)arg_delta(a);
        }
        const R cst = -(R)d * log_(sd) - (R)(0.5 * LOG_2PI) * (R)d;
        const R ax_p = b1 ? (R)0 : (R)-0.5 * q1 + cst, ax_x = b2 ? (R)0 : (R)-0.5 * q2 + cst;
        const bool badobs_p = sc[2 * CB + q] != (R)0, badobs_x = sc[3 * CB + q] != (R)0;
        const R cr = -hlr - (R)(0.5 * LOG_2PI) * dimr, cq = -hlq - (R)(0.5 * LOG_2PI) * (R)d;
        R ob_p = okr ? (R)-0.5 * sc[q] + cr : r_nan<R>(), ob_x = okr ? (R)-0.5 * sc[CB + q] + cr : r_nan<R>();
        if (badobs_p || isnan_(ob_p)) ob_p = 0;
        if (badobs_x || isnan_(ob_x)) ob_x = 0;
        R pr_p = okq ? (R)-0.5 * sc[4 * CB + q] + cq : r_nan<R>(), pr_x = okq ? (R)-0.5 * sc[5 * CB + q] + cq : r_nan<R>();
        if (sc[6 * CB + q] != (R)0 || isnan_(pr_p)) pr_p = 0;
        if (sc[7 * CB + q] != (R)0 || isnan_(pr_x)) pr_x = 0;
        const bool ref = a.nan_policy == 0;
        const R cc_p = (ref && (b1 || badobs_p)) ? (R)0 : ax_p + ob_p;
        const R cc_x = (ref && (b2 || badobs_x)) ? (R)0 : ax_x + ob_x;
        const long long CT = (long long)C * T, o = (long long)c * T + t;
        part[o] = cc_p + pr_p;
        part[CT + o] = cc_x + pr_x;
        part[2 * CT + o] = ob_p + pr_p;
        part[3 * CT + o] = ob_x + pr_x;
        part[4 * CT + o] = corr;
    }
}


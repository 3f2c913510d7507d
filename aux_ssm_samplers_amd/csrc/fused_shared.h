// fused_shared.h -- the chain-shared LG_CONCAT sweep in three streaming passes (round 3; included at the end of kernels.hip.h).
//
// What the chain-shared sweep of affine_shared.h moved per chain and time step (reals of d components; 8.06 GB at C2 / 256 chains, fp64):
//   filter reduce  r x, w eps_aux | filter down r x, r eps_aux, w ms | sampler reduce r ms, w eps_samp | sampler down r ms, r eps_samp, w x'
//   log-density r x, r x', r eps_aux | select r x', w x                                                  = 15 d
// Round 3 (7 d = 3.76 GB): A (r x, w u), C (r u, w inc), E (r inc, r u, w x').  Round 4 (6 d = 3.22 GB): passes A and C are ONE pass --
//   AC (k_fs_ac) r x            w u, inc0  draws eps_aux, u = x + sqrt(delta/2) eps; the MH terms of the CURRENT state x (prior, observation, auxiliary:
//                                         kalman/generic.py:88-89, :103-105); the filter's chunk fold h_t = Mb h_{t-1} + kc + K u_t FROM A ZERO START (the chunk's first
//                                         mean is not known yet), which is at the same time the chunk-LOCAL filtered mean: m_t = h_t + Phi_t m_start with the
//                                         chain-shared prefix product Phi_t of the chunk's Mb (k_fs_fprod); and, in the same walk, the sampler's LOCAL increments
//                                         inc0_t = M1_t h_t - gb_t + Lc_t eps_t (sampling.py:108-112; draws eps_samp) with their chunk aggregate
//                                         e0 = sum_t (G_ta ... G_{t-1}) inc0_t against the table of within-chunk gain products (k_fs_gpre)
//   --  (k_aff_aggs, k_fs_esfix)          m_start of every (chain, chunk) by the aggregate scan of the folds; the sampler's aggregates completed by the part that is
//                                         linear in m_start: e = e0 + Psi_chunk m_start, Psi = sum_t (G_ta ... G_{t-1}) M1_t Phi_t (chain-shared, k_fs_psi)
//   E  (k_fs_e)  r inc0, r u    w x'      sampler walk (descending) x'_t = G_t x'_{t+1} + inc0_t + N_t m_start, N_t = M1_t Phi_t (in the pass's coefficient row), and the MH
//                                         terms of the PROPOSAL x' where it is in registers
// The filtered means never go to memory, u is written once and read once, and the chain passes are two kernels bound by HBM (AC: 3 d at 615 instructions per
// chain-step; E: 3 d) instead of three.
// plus the two aggregate scans (k_aff_aggs, unchanged), one lane per chain for the t = 0 terms (k_fs_head) and the accept step (k_fs_accept).
// There is no select pass: with a `sel` array the state is LAZY -- chain c lives in buffer sel[c] of a ping-pong pair, reads its x from there, writes
// its proposal to the other buffer, and acceptance flips sel[c] (auxssm_kalman_sweep_lazy); without one, x' goes to a scratch buffer and the
// caller runs the usual select.  Same tables as affine_shared.h / kalman_bodies.h (GainRow, SampShared, LogShared), same per-term arithmetic and NaN
// policy as FilterMeanOp / SampleAffOp / body_sweep_logpdf_shared; the per-chain totals are summed per (chain, chunk) lane and then over chunks, so
// they agree with the unfused path to rounding, not bitwise (tests/test_gpu_fused.py: 1e-9 relative on x', 1e-7 absolute on log alpha at C2 size).
#pragma once
// (included inside namespace ax by kernels.hip.h)

struct FusedArgs {
    int C, T, E, nchunk;
    const void* xa;      // (T, D, C) chain-minor; chain c reads its state from xa (sel null or sel[c] == 0) or xb,
    void* xb;            //                         and writes its proposal to the other one
    const int32_t* sel;
    void* u;             // (T, D, C): row 0 written by the caller, rows >= 1 by k_fs_ac
    void* inc;           // (T, D, C)
    const void* gain;    // n rows GainRow<R, D, P>      (transition i -> i + 1)
    const void* samp;    // T rows SampShared<R, D>
    const void* logt;    // n rows LogShared<R, D, PO>
    const void* gpre;    // T rows D*D: product of the sampler gains of the chunk's earlier steps, G_ta ... G_{t-1} (I at the chunk's first step)
    const void* m0p;     // (D, C): filtered mean at t = 0
    void* agg_f; void* pre_f; void* agg_s; void* pre_s;  // chunk aggregates / exclusive prefixes of the two scans (k_aff_aggs layout)
    Acc* pa;             // (3, C, nchunk): sums over the chunk of [q(x | u) terms, target(x) terms, |x - u|^2 / delta]
    Acc* pe;             // (3, C, nchunk): the same of x'
    void* pell;          // (C, nchunk) -1/2 sum |eps_samp|^2 of the chunk: the data part of log q(x' | u) (R)
    const Acc* clog;     // sum_t (sum_i log Lc_t[i][i] + D/2 log 2 pi): the chain-shared part of log q(x' | u) (k_fs_clog, model stage)
    unsigned ka0, ka1, ks0, ks1;  // keys of eps_aux / eps_samp (stream 0 of auxssm_rng_normal at the (T, D, C) flat index)
    const void* eps0s;   // (D, C): row 0 of eps_samp (fill kernel)
    double delta, shd;
    const double* dptr;  // device-resident {delta, sqrt(delta / 2)} or null
    int nan_policy;
    const int* memo;     // model-stage memo (FilterArgs::memo): the stage kernels return at once when *memo == 0
    int pack, cp;        // few chains (PK instantiations of the two chain passes): one wave walks `pack` consecutive chunks side by side, lane = sub * cp + chain, cp = the
                         // chain count rounded up to a power of two; 1 / 0 otherwise
};
AX_HD void fs_resolve(FusedArgs& a) {
    const double* p = a.dptr;
    a.dptr = nullptr;
    if (p) a.delta = p[0], a.shd = p[1];
}

// the chunk [ch E, (ch + 1) E) of time steps and the (chain, chunk) lane: same block-id swizzle as the affine passes
__device__ __forceinline__ bool fs_decode(const FusedArgs& a, int& ch, int& s) { return decode_aff(a.C, a.nchunk, ch, s); }

// ---- model stage: chunk products of the filter's matrices; within-chunk prefix products + chunk products of the sampler's gains ------------
template <typename R, int D, int P> __global__ void __launch_bounds__(TB_CM) k_fs_fprod(FusedArgs a, R* __restrict__ cprod, R* __restrict__ fpre) {
    if (memo_skip(a)) return;
    using TG = GainRow<R, D, P>;
    const int ch = blockIdx.x * TB_CM + threadIdx.x;
    if (ch >= a.nchunk) return;
    const int t0 = ch * a.E, ta = max(1, t0), tb = min(a.T, t0 + a.E);
    R M[D * D];
#pragma unroll
    for (int k = 0; k < D * D; ++k) M[k] = (k / D == k % D) ? (R)1 : (R)0;
    if (ch == 0) {
#pragma unroll
        for (int k = 0; k < D * D; ++k) fpre[k] = 0;  // t = 0
    }
    for (int t = ta; t < tb; ++t) {
        const R* row = (const R*)a.gain + (long long)(t - 1) * TG::NPAD;
        R G[D * D], o[D * D];
#pragma unroll
        for (int k = 0; k < D * D; ++k) G[k] = row[TG::oM + k];
        mm<R, D, D, D>(G, M, o);
#pragma unroll
        for (int k = 0; k < D * D; ++k) M[k] = o[k];
        // Phi_t = Mb_{t-1} ... Mb_{ta-1}: what the chunk's (unknown) first mean contributes to the mean at t.  Chunk 0 starts from the true m_0: nothing to add.
#pragma unroll
        for (int k = 0; k < D * D; ++k) fpre[(long long)t * D * D + k] = ch == 0 ? (R)0 : M[k];
    }
#pragma unroll
    for (int k = 0; k < D * D; ++k) cprod[(long long)ch * D * D + k] = M[k];
}
template <typename R, int D> __global__ void __launch_bounds__(TB_CM) k_fs_gpre(FusedArgs a, R* __restrict__ gpre, R* __restrict__ cprod) {
    if (memo_skip(a)) return;  // (k_fs_psi below completes the chain-shared tables)
    using TS = SampShared<R, D>;
    const int ch = blockIdx.x * TB_CM + threadIdx.x;
    if (ch >= a.nchunk) return;
    const int ta = ch * a.E, tb = min(a.T, (ch + 1) * a.E);
    R M[D * D];
#pragma unroll
    for (int k = 0; k < D * D; ++k) M[k] = (k / D == k % D) ? (R)1 : (R)0;
    for (int t = ta; t < tb; ++t) {
#pragma unroll
        for (int k = 0; k < D * D; ++k) gpre[(long long)t * D * D + k] = M[k];
        const R* row = (const R*)a.samp + (long long)t * TS::NPAD;
        R G[D * D], o[D * D];
#pragma unroll
        for (int k = 0; k < D * D; ++k) G[k] = row[TS::oG + k];
        mm<R, D, D, D>(M, G, o);  // (G_ta ... G_{t-1}) G_t
#pragma unroll
        for (int k = 0; k < D * D; ++k) M[k] = o[k];
    }
    // scan position of the sampler = reversed time: chunk ch is aggregate nchunk - 1 - ch
#pragma unroll
    for (int k = 0; k < D * D; ++k) cprod[(long long)(a.nchunk - 1 - ch) * D * D + k] = M[k];
}

// N_t = M1_t Phi_t (the sampler increment's dependence on the chunk's first mean) and Psi_chunk = sum_t (G_ta ... G_{t-1}) N_t (the same for the sampler's chunk aggregate)
template <typename R, int D> __global__ void __launch_bounds__(TB_CM) k_fs_psi(FusedArgs a, const R* __restrict__ gpre, const R* __restrict__ fpre, R* __restrict__ ntab,
                                                                              R* __restrict__ psi) {
    if (memo_skip(a)) return;
    using TS = SampShared<R, D>;
    const int ch = blockIdx.x * TB_CM + threadIdx.x;
    if (ch >= a.nchunk) return;
    const int t0 = ch * a.E, tb = min(a.T, t0 + a.E);
    R Ps[D * D];
#pragma unroll
    for (int k = 0; k < D * D; ++k) Ps[k] = 0;
    for (int t = t0; t < tb; ++t) {
        const R* row = (const R*)a.samp + (long long)t * TS::NPAD;
        R M1[D * D], Ph[D * D], Gp[D * D], Nt[D * D], o[D * D];
#pragma unroll
        for (int k = 0; k < D * D; ++k) M1[k] = row[TS::oM + k], Ph[k] = fpre[(long long)t * D * D + k], Gp[k] = gpre[(long long)t * D * D + k];
        mm<R, D, D, D>(M1, Ph, Nt);
        mm<R, D, D, D>(Gp, Nt, o);
#pragma unroll
        for (int k = 0; k < D * D; ++k) ntab[(long long)t * D * D + k] = Nt[k], Ps[k] += o[k];
    }
#pragma unroll
    for (int k = 0; k < D * D; ++k) psi[(long long)ch * D * D + k] = Ps[k];
}
// the sampler's chunk aggregates completed: e(chain, chunk) += Psi_chunk m_start(chain, chunk) (the scan position of the sampler is reversed time)
template <typename R, int D> __global__ void __launch_bounds__(256) k_fs_esfix(int C, int nchunk, const R* __restrict__ psi, const R* __restrict__ pre_f, R* __restrict__ agg_s) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)C * nchunk) return;
    const int c = (int)(g / nchunk), ch = (int)(g % nchunk);  // (consecutive lanes: consecutive chunks of one chain -- the records of a chain are contiguous)
    if (ch == 0) return;  // (Psi_0 = 0)
    R m[D], e[D], Ps[D * D];
    ldv<R, D>(pre_f + ((long long)c * nchunk + ch) * SampPre<R, D>::NPAD, m);
    R* q = agg_s + ((long long)c * nchunk + (nchunk - 1 - ch)) * SampPre<R, D>::NPAD;
    ldv<R, D>(q, e);
#pragma unroll
    for (int k = 0; k < D * D; ++k) Ps[k] = psi[(long long)ch * D * D + k];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        R v = e[i];
#pragma unroll
        for (int k = 0; k < D; ++k) v += Ps[i * D + k] * m[k];
        e[i] = v;
    }
    stv<R, D>(q, e);
}

// ---- compact per-pass rows (model stage) ----------------------------------------------------------------------------------------------------
// The passes wait for their chain-shared coefficients more than they compute: read straight from the tables with scalar loads every chunk streams ~1 KB
// of rows per step through the scalar cache with no reuse, i.e. a dozen dependent L2 round trips per step (measured: pass C 0.92 ms against a VALU
// floor of 0.28).  So each pass gets its OWN row per time step, holding exactly what it reads, and a workgroup (all its chains x one chunk) copies the
// chunk's rows into LDS with one burst of coalesced loads; inside the time loop a coefficient is an LDS broadcast read.
template <typename R, int D, int PO> struct FsRows {
    static constexpr int P = D + PO;
    using TL = LogShared<R, D, PO>;
    static constexpr int VEC = 16 / sizeof(R);
    static constexpr int pad(int n) { return (n + VEC - 1) / VEC * VEC; }
    // pass AC, row t (t >= 0): [Mb | kc | K[:, :D] of transition t - 1 -> t | M1_t | gb_t | Lc_t (lower, packed) | gpre_t | LogShared row t - 1]; the filter and
    // log-density parts of row 0 are zero (no transition into t = 0)
    static constexpr int cM = 0, cKc = D * D, cK = cKc + D, cM1 = cK + D * D, cGb = cM1 + D * D, cL = cGb + D, cGp = cL + symsize(D), cLg = cGp + D * D,
                         NC = pad(cLg + TL::N);
    // pass E, row t: [G_t | N_t = M1_t Phi_t | LogShared row t (transition t -> t + 1, observation at t + 1; zero for t = T - 1)]
    static constexpr int eG = 0, eN = D * D, eL = 2 * D * D, NE = pad(eL + TL::N);
};
// where element k of a compact row comes from: {table (0 gain row t - 1, 1 sampler row t, 2 log-density row t - 1, 3 gpre row t, 4 log-density row t,
// 5 N row t, -1 zero), offset}.  Evaluated per element by one thread, so that consecutive threads write consecutive reals of the destination rows.
struct FsSrc { int tab, off; };
template <typename R, int D, int PO> __device__ __forceinline__ FsSrc fs_src_c(int k) {
    constexpr int P = D + PO;
    using F = FsRows<R, D, PO>; using TG = GainRow<R, D, P>; using TS = SampShared<R, D>; using TL = LogShared<R, D, PO>;
    if (k < F::cKc) return {0, TG::oM + k};
    if (k < F::cK) return {0, TG::oKc + (k - F::cKc)};
    if (k < F::cM1) { const int q = k - F::cK; return {0, TG::oK + (q / D) * P + (q % D)}; }
    if (k < F::cGb) return {1, TS::oM + (k - F::cM1)};
    if (k < F::cL) return {1, TS::oGb + (k - F::cGb)};
    if (k < F::cGp) {  // packed lower index -> (i, j) of the dense D x D factor
        const int q = k - F::cL;
        int i = 0;
        while (lidx(i + 1, 0) <= q) ++i;
        return {1, TS::oL + i * D + (q - lidx(i, 0))};
    }
    if (k < F::cLg) return {3, k - F::cGp};
    if (k < F::cLg + TL::N) return {2, k - F::cLg};
    return {-1, 0};
}
template <typename R, int D, int PO> __device__ __forceinline__ FsSrc fs_src_e(int k) {
    using F = FsRows<R, D, PO>; using TS = SampShared<R, D>; using TL = LogShared<R, D, PO>;
    if (k < F::eN) return {1, TS::oG + k};
    if (k < F::eL) return {5, k - F::eN};
    if (k < F::eL + TL::N) return {4, k - F::eL};
    return {-1, 0};
}
// one thread per destination element: blockIdx.y picks the row family (0 AC, 1 E), the flat index runs over (t, k)
template <typename R, int D, int PO> __global__ void __launch_bounds__(256) k_fs_rows(FusedArgs a, const R* __restrict__ ntab, R* __restrict__ rc, R* __restrict__ re) {
    if (memo_skip(a)) return;
    constexpr int P = D + PO;
    using F = FsRows<R, D, PO>; using TG = GainRow<R, D, P>; using TS = SampShared<R, D>; using TL = LogShared<R, D, PO>;
    const int fam = blockIdx.y;
    const int N = fam == 0 ? F::NC : F::NE;
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)a.T * N) return;
    const int t = (int)(g / N), k = (int)(g % N);
    const FsSrc sc = fam == 0 ? fs_src_c<R, D, PO>(k) : fs_src_e<R, D, PO>(k);
    R v = 0;
    if (sc.tab == 0) { if (t >= 1) v = ((const R*)a.gain)[(long long)(t - 1) * TG::NPAD + sc.off]; }
    else if (sc.tab == 1) v = ((const R*)a.samp)[(long long)t * TS::NPAD + sc.off];
    else if (sc.tab == 2) { if (t >= 1) v = ((const R*)a.logt)[(long long)(t - 1) * TL::NPAD + sc.off]; }
    else if (sc.tab == 3) v = ((const R*)a.gpre)[(long long)t * D * D + sc.off];
    else if (sc.tab == 4) { if (t + 1 < a.T) v = ((const R*)a.logt)[(long long)t * TL::NPAD + sc.off]; }
    else if (sc.tab == 5) v = ntab[(long long)t * D * D + sc.off];
    (fam == 0 ? rc : re)[g] = v;
}

// the chain-shared part of log q(x' | u): x'_t | x'_{t+1} = G_t x'_{t+1} + M1_t m_t - gb_t + Lc_t eps_t, so log q = sum_t (-1/2 |eps_t|^2 - sum_i log Lc_t[i][i] - D/2 log 2 pi).
// Two launches with a fixed shape (one lane per time step, a tree per workgroup, then one workgroup over the partial sums): deterministic, and a few microseconds
// on the model stage's chain of dependent launches (one workgroup walking all T steps took 0.6 ms there and made the stage the critical path).
template <typename R, int D> __global__ void __launch_bounds__(256) k_fs_clog_part(int T, const R* __restrict__ samp, Acc* __restrict__ part, const int* memo) {
    if (memo_skip_p(memo)) return;
    using TS = SampShared<R, D>;
    __shared__ Acc sh[256];
    const int t = blockIdx.x * 256 + threadIdx.x;
    R s = 0;
    if (t < T) {
        const R* row = samp + (long long)t * TS::NPAD;
#pragma unroll
        for (int i = 0; i < D; ++i) s += log_(row[TS::oL + i * D + i]);
    }
    sh[threadIdx.x] = (Acc)s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
template <int D> __global__ void __launch_bounds__(256) k_fs_clog_sum(int T, int nb, const Acc* __restrict__ part, Acc* __restrict__ out, const int* memo) {
    if (memo_skip_p(memo)) return;
    __shared__ Acc sh[256];
    Acc acc = 0;
    for (int b = threadIdx.x; b < nb; b += 256) acc += part[b];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0] + (Acc)T * (Acc)D * (Acc)(0.5 * LOG_2PI);
}

// workgroup = all (up to 256) chains of one chunk; consecutive blocks = the chain groups of one chunk
__device__ __forceinline__ void fs_block(const FusedArgs& a, int& ch, int& c) {
    const int groups = (a.C + (int)blockDim.x - 1) / (int)blockDim.x;
    ch = blockIdx.x / groups;
    c = (blockIdx.x % groups) * blockDim.x + threadIdx.x;
}
// copy rows [r0, r1) of a table with rows of N reals into LDS (N a multiple of 16 bytes): one burst of coalesced 16-byte loads
template <typename R, int N> __device__ __forceinline__ void fs_stage(const R* __restrict__ tab, int r0, int r1, R* __restrict__ lds) {
    using V = typename Vec16<R>::type;
    constexpr int W = Vec16<R>::W;
    const V* src = reinterpret_cast<const V*>(tab + (long long)r0 * N);
    V* dst = reinterpret_cast<V*>(lds);
    const int n = (r1 - r0) * (N / W), B = blockDim.x;
    // four loads in flight per trip (one load -> wait -> LDS write per trip left a one-wave workgroup a dozen serial memory latencies before its first step)
    for (int i = threadIdx.x; i < n; i += 4 * B) {
        const bool p1 = i + B < n, p2 = i + 2 * B < n, p3 = i + 3 * B < n;
        const V v0 = src[i], v1 = p1 ? src[i + B] : V{}, v2 = p2 ? src[i + 2 * B] : V{}, v3 = p3 ? src[i + 3 * B] : V{};
        dst[i] = v0;
        if (p1) dst[i + B] = v1;
        if (p2) dst[i + 2 * B] = v2;
        if (p3) dst[i + 3 * B] = v3;
    }
    __syncthreads();
}

// packed form (few chains): rows [r0, r1) = `pack` consecutive chunks of E rows, each chunk's block shifted by one 16-byte slot more than the one before it -- the lanes
// of different chunks read the SAME row offset of their own block at the same time, and E N reals is a multiple of the bank count, so without the shift every read
// would be a `pack`-way bank conflict
template <typename R, int N> __device__ __forceinline__ void fs_stage_pk(const R* __restrict__ tab, int r0, int r1, int E, R* __restrict__ lds) {
    using V = typename Vec16<R>::type;
    constexpr int W = Vec16<R>::W, NV = N / W;
    const V* src = reinterpret_cast<const V*>(tab + (long long)r0 * N);
    V* dst = reinterpret_cast<V*>(lds);
    const int n = (r1 - r0) * NV, B = blockDim.x;
    for (int i = threadIdx.x; i < n; i += 8 * B) {   // eight loads in flight per trip
        V v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = i + q * B < n ? src[i + q * B] : V{};
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i + q * B < n) dst[i + q * B + ((i + q * B) / NV) / E] = v[q];
    }
    __syncthreads();
}
template <typename R, int N> AX_HD size_t fs_pk_block(int E) { return (size_t)E * N + Vec16<R>::W; }  // reals per chunk block in LDS (packed form)

// the tables of the fp64 normal transform (rng.h) in the workgroup's LDS, behind the pass's coefficient rows; the barrier of fs_stage publishes them.
// fp32 draws use the hardware functions: no table, no LDS.
template <typename R> struct FsNormTabSel { using type = NormTabGlobal; static constexpr size_t BYTES = 0; };
template <> struct FsNormTabSel<double> { using type = NormTabLds; static constexpr size_t BYTES = (size_t)RNG_TAB_DOUBLES * sizeof(double); };
template <typename R> using FsNormTab = typename FsNormTabSel<R>::type;
template <typename R> __device__ __forceinline__ FsNormTab<R> fs_stage_normtab(R* behind_rows) {
    if constexpr (sizeof(R) == 8) {
        NormTabLds::stage((double*)behind_rows, false);
        return NormTabLds{(const double*)behind_rows};
    } else {
        return NormTabGlobal{};
    }
}

// the MH terms of one state v at time t = i + 1 given the state w at time t - 1 (v, w = x or x'), LogShared row i (`row`, in LDS) -- the per-state
// half of body_sweep_logpdf_shared, same operations: [q-term (concatenated likelihood + prior), target term (likelihood + prior), |v - u|^2 / delta]
template <typename R, int D, int PO>
__device__ __forceinline__ void fs_terms(const FusedArgs& a, const R* row, const R* v, const R* w, const R* u, R inv_delta, R cst, R* out3) {
    using TL = LogShared<R, D, PO>;
    R ob;
    bool badobs = false;
    {
        R q = 0;
#pragma unroll
        for (int k = 0; k < PO; ++k) {
            R z = row[TL::oYw + k];
#pragma unroll
            for (int j = 0; j < D; ++j) z -= row[TL::oWH + k * D + j] * v[j];
            badobs = badobs || !finite_(z);
            q += z * z;
        }
        ob = (R)-0.5 * q + row[TL::oCR];
        if (badobs || isnan_(ob)) ob = 0;
    }
    asm volatile("" ::: "memory");  // (the three blocks read disjoint parts of the row: keep their LDS reads apart)
    R ax, qa = 0;
    bool b = false;
    {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const R d = u[k] - v[k];
            b = b || !finite_(d);
            qa += d * d;
        }
        ax = b ? (R)0 : -qa * inv_delta + cst;
    }
    const R cc = (a.nan_policy == 0 && (b || badobs)) ? (R)0 : ax + ob;
    asm volatile("" ::: "memory");
    R pr;
    {
        R q = 0;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            R z = -row[TL::oWb + k];
#pragma unroll
            for (int l = 0; l <= k; ++l) z += row[TL::oWQ + lidx(k, l)] * v[l];
#pragma unroll
            for (int j = 0; j < D; ++j) z -= row[TL::oWF + k * D + j] * w[j];
            bad = bad || !finite_(z);
            q += z * z;
        }
        pr = (R)-0.5 * q + row[TL::oCQ];
        if (bad || isnan_(pr)) pr = 0;
    }
    out3[0] = cc + pr;
    out3[1] = ob + pr;
    out3[2] = qa * inv_delta;
}

// The same three sums with NO per-term policy: every residual kept, nothing tested.  Returns true when the lane's result is not finite -- a non-finite
// residual, difference or row constant makes the sum of the three terms non-finite -- in which case fs_terms (the per-term NaN policy: masked rows are
// zeros, a non-finite kept residual drops its term) has to redo the step; when everything is finite the two agree term by term.  The passes branch on the
// WAVE's ballot of that flag: the policy's fourteen class tests, their boolean algebra and the selects on doubles were a tenth of pass A's instructions,
// spent on a case (a NaN in the state or the data row) that almost no wave ever sees.
template <typename R, int D, int PO>
__device__ __forceinline__ bool fs_terms_fast(const R* row, const R* v, const R* w, const R* u, R inv_delta, R cst, R* out3) {
    using TL = LogShared<R, D, PO>;
    R qo = 0;
#pragma unroll
    for (int k = 0; k < PO; ++k) {
        R z = row[TL::oYw + k];
#pragma unroll
        for (int j = 0; j < D; ++j) z -= row[TL::oWH + k * D + j] * v[j];
        qo += z * z;
    }
    const R ob = (R)-0.5 * qo + row[TL::oCR];
    asm volatile("" ::: "memory");  // (the blocks read disjoint parts of the row: keep their LDS reads apart)
    R qa = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const R d = u[k] - v[k];
        qa += d * d;
    }
    const R ax = -qa * inv_delta + cst;
    asm volatile("" ::: "memory");
    R qp = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        R z = -row[TL::oWb + k];
#pragma unroll
        for (int l = 0; l <= k; ++l) z += row[TL::oWQ + lidx(k, l)] * v[l];
#pragma unroll
        for (int j = 0; j < D; ++j) z -= row[TL::oWF + k * D + j] * w[j];
        qp += z * z;
    }
    const R pr = (R)-0.5 * qp + row[TL::oCQ];
    out3[0] = (ax + ob) + pr;
    out3[1] = ob + pr;
    out3[2] = qa * inv_delta;
    return !finite_(out3[0]);
}
__device__ __forceinline__ bool fs_wave_any(bool flag) { return __builtin_amdgcn_ballot_w64(flag) != 0; }

// ---- pass AC --------------------------------------------------------------------------------------------------------------------------
#ifndef AUXSSM_FS_WPE_A
#define AUXSSM_FS_WPE_A 2
#endif
// PK (few chains, one wave per workgroup): lane = sub * cp + chain walks chunk blockIdx.x * pack + sub -- the time index is per lane, everything else is the same code
template <typename R, int D, int PO, bool PK = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AUXSSM_FS_WPE_A))) k_fs_ac(FusedArgs a, const R* __restrict__ rows) {
    using F = FsRows<R, D, PO>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    R* lds = (R*)smem;
    fs_resolve(a);
    int ch, c;
    if constexpr (PK) {
        const int sub = (int)threadIdx.x / a.cp;
        c = (int)threadIdx.x - sub * a.cp;
        ch = blockIdx.x * a.pack + sub;
    } else {
        fs_block(a, ch, c);
    }
    const long long C = a.C;
    const int t0 = ch * a.E, ta = max(1, t0), tb = min(a.T, t0 + a.E);
    const FsNormTab<R> ntab = fs_stage_normtab<R>(lds + (PK ? (size_t)a.pack * fs_pk_block<R, F::NC>(a.E) : (size_t)a.E * F::NC));
    if constexpr (PK) {
        const int c0 = blockIdx.x * a.pack;
        fs_stage_pk<R, F::NC>(rows, c0 * a.E, min(a.T, (c0 + a.pack) * a.E), a.E, lds);
        lds += (size_t)(ch - c0) * fs_pk_block<R, F::NC>(a.E);   // the lane's own chunk block
        if (ch >= a.nchunk) return;
    } else {
        fs_stage<R, F::NC>(rows, t0, tb, lds);
    }
    if (c >= a.C) return;
    const R* xr = (const R*)((a.sel && a.sel[c]) ? (const void*)a.xb : a.xa) + c;
    R* up = (R*)a.u + c;
    R* ip = (R*)a.inc + c;
    R h[D], xq[D], es[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        h[k] = ch == 0 ? ((const R*)a.m0p)[k * C + c] : (R)0;  // chunk 0 folds from the true m_0; the others from zero (their first mean comes from the aggregate scan)
        xq[k] = xr[((long long)(ta - 1) * D + k) * C];
        es[k] = 0;
    }
    const R shd = (R)a.shd, inv_delta = (R)1 / (R)a.delta;
    const R cst = (R)-0.5 * (R)D * log_((R)(0.5 * a.delta)) - (R)(0.5 * LOG_2PI) * (R)D;
    Acc v0 = 0, v1 = 0, v2 = 0;
    R se = 0;  // sum |eps_samp|^2 over the chunk's steps: the data part of log q(x' | u)
    // one LOCAL sampler increment: inc0 = M1 h - gb + Lc eps (SampleAffOp::step without the G term, with the chunk-local mean h), stored, and folded into the chunk aggregate
    auto emit = [&](int tu, const R* row, const R* eps) {
        R inc[D];
#pragma unroll
        for (int k = 0; k < D; ++k) se += eps[k] * eps[k];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            R v = -row[F::cGb + i];
#pragma unroll
            for (int k = 0; k < D; ++k) v += row[F::cM1 + i * D + k] * h[k];
#pragma unroll
            for (int k = 0; k <= i; ++k) v += row[F::cL + lidx(i, k)] * eps[k];
            inc[i] = v;
            ip[((long long)tu * D + i) * C] = v;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            R v = es[i];
#pragma unroll
            for (int k = 0; k < D; ++k) v += row[F::cGp + i * D + k] * inc[k];
            es[i] = v;
        }
    };
    if (ch == 0) {  // t = 0: the mean of k_filter_t0, the noise row of the fill kernel
        R eps[D];
#pragma unroll
        for (int k = 0; k < D; ++k) eps[k] = ((const R*)a.eps0s)[k * C + c];
        emit(0, lds, eps);
    }
    R xn[D];  // the next step's state, fetched one step ahead
#pragma unroll
    for (int k = 0; k < D; ++k) xn[k] = ta < tb ? xr[((long long)ta * D + k) * C] : (R)0;
#pragma unroll 1
    for (int t = ta; t < tb; ++t) {
        const int tu = PK ? t : opaque_uniform(t);
        const R* row = lds + (tu - t0) * F::NC;
        R x[D], ev[D], u[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = xn[k];
        if (t + 1 < tb) {
            const int tn = PK ? t + 1 : opaque_uniform(t + 1);
#pragma unroll
            for (int k = 0; k < D; ++k) xn[k] = xr[((long long)tn * D + k) * C];
        }
        normals_cm<R, D>(a.ka0, a.ka1, (long long)tu * D * C + c, C, ev, ntab);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            u[k] = x[k] + shd * ev[k];
            up[((long long)tu * D + k) * C] = u[k];
        }
        asm volatile("" ::: "memory");
        // the MH terms of the current state; the rare wave with a non-finite value anywhere redoes them under the per-term policy and masks u for the fold
        R w[3], um[D];
        const bool bad = fs_terms_fast<R, D, PO>(row + F::cLg, x, xq, u, inv_delta, cst, w);
#pragma unroll
        for (int k = 0; k < D; ++k) um[k] = u[k];
        if (fs_wave_any(bad)) {
            fs_terms<R, D, PO>(a, row + F::cLg, x, xq, u, inv_delta, cst, w);
#pragma unroll
            for (int k = 0; k < D; ++k) um[k] = finite_(u[k]) ? u[k] : (R)0;
        }
        v0 += (Acc)w[0];
        v1 += (Acc)w[1];
        v2 += (Acc)w[2];
        asm volatile("" ::: "memory");  // keep the fold's LDS reads behind the log-density row's: hoisted together they cost 200 registers
        {  // filter fold (FilterMeanOp::fold, folded rows): h <- Mb h + kc + K[:, :D] u -- the chunk aggregate at the end, the chunk-local filtered mean on the way
            R o[D];
#pragma unroll
            for (int r = 0; r < D; ++r) {
                R v = row[F::cKc + r];
#pragma unroll
                for (int k = 0; k < D; ++k) v += row[F::cM + r * D + k] * h[k];
#pragma unroll
                for (int k = 0; k < D; ++k) v += row[F::cK + r * D + k] * um[k];
                o[r] = v;
            }
#pragma unroll
            for (int r = 0; r < D; ++r) h[r] = o[r];
        }
        asm volatile("" ::: "memory");
        R eps[D];
        normals_cm<R, D>(a.ks0, a.ks1, (long long)tu * D * C + c, C, eps, ntab);
        emit(tu, row, eps);
#pragma unroll
        for (int k = 0; k < D; ++k) xq[k] = x[k];
    }
    stv<R, D>((R*)a.agg_f + ((long long)c * a.nchunk + ch) * SampPre<R, D>::NPAD, h);
    stv<R, D>((R*)a.agg_s + ((long long)c * a.nchunk + (a.nchunk - 1 - ch)) * SampPre<R, D>::NPAD, es);
    a.pa[((long long)0 * C + c) * a.nchunk + ch] = v0;
    a.pa[((long long)1 * C + c) * a.nchunk + ch] = v1;
    a.pa[((long long)2 * C + c) * a.nchunk + ch] = v2;
    ((R*)a.pell)[(long long)c * a.nchunk + ch] = (R)-0.5 * se;
}

// ---- pass E ---------------------------------------------------------------------------------------------------------------------------
template <typename R, int D, int PO, bool PK = false> __global__ void __launch_bounds__(256) k_fs_e(FusedArgs a, const R* __restrict__ rows) {
    using F = FsRows<R, D, PO>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    R* lds = (R*)smem;
    fs_resolve(a);
    int ch, c;
    if constexpr (PK) {
        const int sub = (int)threadIdx.x / a.cp;
        c = (int)threadIdx.x - sub * a.cp;
        ch = blockIdx.x * a.pack + sub;
    } else {
        fs_block(a, ch, c);
    }
    const long long C = a.C;
    const int t0 = ch * a.E, tb = min(a.T, t0 + a.E);
    if constexpr (PK) {
        const int c0 = blockIdx.x * a.pack;
        fs_stage_pk<R, F::NE>(rows, c0 * a.E, min(a.T, (c0 + a.pack) * a.E), a.E, lds);
        lds += (size_t)(ch - c0) * fs_pk_block<R, F::NE>(a.E);
        if (ch >= a.nchunk) return;
    } else {
        fs_stage<R, F::NE>(rows, t0, tb, lds);
    }
    if (c >= a.C) return;
    R* xw = (R*)((a.sel && a.sel[c]) ? const_cast<void*>(a.xa) : a.xb) + c;
    const R* up = (const R*)a.u + c;
    const R* ip = (const R*)a.inc + c;
    R h[D], uq[D], mst[D];  // mst: the chunk's first filtered mean (aggregate scan of the folds): the part of the increments pass AC could not know
    if (ch > 0) {
        ldv<R, D>((const R*)a.pre_f + ((long long)c * a.nchunk + ch) * SampPre<R, D>::NPAD, mst);
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) mst[k] = 0;
    }
    if (ch == a.nchunk - 1) {
#pragma unroll
        for (int k = 0; k < D; ++k) h[k] = 0, uq[k] = 0;  // G_{T-1} = 0: the first position ignores the incoming state
    } else {
        ldv<R, D>((const R*)a.pre_s + ((long long)c * a.nchunk + (a.nchunk - 1 - ch)) * SampPre<R, D>::NPAD, h);  // x'_{tb}
#pragma unroll
        for (int k = 0; k < D; ++k) uq[k] = up[((long long)tb * D + k) * C];
    }
    const R inv_delta = (R)1 / (R)a.delta;
    const R cst = (R)-0.5 * (R)D * log_((R)(0.5 * a.delta)) - (R)(0.5 * LOG_2PI) * (R)D;
    Acc v0 = 0, v1 = 0, v2 = 0;
    R incn[D], utn[D];  // the next (earlier) step's records, fetched one step ahead
#pragma unroll
    for (int k = 0; k < D; ++k) incn[k] = ip[((long long)(tb - 1) * D + k) * C], utn[k] = up[((long long)(tb - 1) * D + k) * C];
#pragma unroll 1
    for (int t = tb - 1; t >= t0; --t) {
        const int tu = PK ? t : opaque_uniform(t);
        const R* row = lds + (tu - t0) * F::NE;
        R inc[D], ut[D], xp[D];
#pragma unroll
        for (int k = 0; k < D; ++k) inc[k] = incn[k], ut[k] = utn[k];
        if (t > t0) {
            const int tn = PK ? t - 1 : opaque_uniform(t - 1);
#pragma unroll
            for (int k = 0; k < D; ++k) incn[k] = ip[((long long)tn * D + k) * C], utn[k] = up[((long long)tn * D + k) * C];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            R v = inc[i];
#pragma unroll
            for (int k = 0; k < D; ++k) v += row[F::eN + i * D + k] * mst[k];
#pragma unroll
            for (int k = 0; k < D; ++k) v += row[F::eG + i * D + k] * h[k];
            xp[i] = v;
            xw[((long long)tu * D + i) * C] = v;
        }
        if (tu + 1 < a.T) {  // (wave-uniform) LogShared row tu: x'_{tu+1} = h given x'_tu = xp, observation and auxiliary terms at tu + 1
            R w[3];
            if (fs_wave_any(fs_terms_fast<R, D, PO>(row + F::eL, h, xp, uq, inv_delta, cst, w))) fs_terms<R, D, PO>(a, row + F::eL, h, xp, uq, inv_delta, cst, w);
            v0 += (Acc)w[0];
            v1 += (Acc)w[1];
            v2 += (Acc)w[2];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) h[k] = xp[k], uq[k] = ut[k];
    }
    a.pe[((long long)0 * C + c) * a.nchunk + ch] = v0;
    a.pe[((long long)1 * C + c) * a.nchunk + ch] = v1;
    a.pe[((long long)2 * C + c) * a.nchunk + ch] = v2;
}

// ---- t = 0 terms (one lane per chain): body_sweep_logpdf_head on the lane's own pair of buffers -----------------------------------------------
template <typename R, int D, int PO> __global__ void __launch_bounds__(TB_ELEM) k_fs_head(SweepLogpdfArgs la, const void* xa, const void* xb, const int32_t* sel, R* __restrict__ head5) {
    resolve_step(la);
    const int c = blockIdx.x * TB_ELEM + threadIdx.x;
    if (c >= la.d.C) return;
    const bool sw = sel && sel[c];
    la.x.ptr = sw ? xb : xa;
    la.xp.ptr = sw ? xa : xb;
    R h5[5];
    body_sweep_logpdf_head<R, D, PO>(la, c, h5);
#pragma unroll
    for (int k = 0; k < 5; ++k) head5[(long long)k * la.d.C + c] = h5[k];
}

// ---- accept (generic.py:70-73, 98-106): one workgroup per chain sums its chunks' partial totals in a fixed order, forms log alpha in Acc
// exactly as k_accept does, draws the Bernoulli and -- lazy state -- flips the chain's buffer selector ------------------------------------------
template <typename R> __global__ void __launch_bounds__(TB_ELEM) k_fs_accept(FusedArgs a, const R* __restrict__ head5, const R* __restrict__ ell0,
                                                                            const R* __restrict__ u_acc, int32_t* __restrict__ accepted, R* __restrict__ logs,
                                                                            int32_t* __restrict__ sel) {
    __shared__ Acc sh[TB_ELEM];
    const int c = blockIdx.x;
    const long long C = a.C;
    Acc tot[7] = {0, 0, 0, 0, 0, 0, 0};
    const R* pl = (const R*)a.pell + (long long)c * a.nchunk;
    for (int j = threadIdx.x; j < a.nchunk; j += TB_ELEM) {  // (seven independent loads per trip)
#pragma unroll
        for (int k = 0; k < 3; ++k) tot[k] += a.pa[((long long)k * C + c) * a.nchunk + j], tot[3 + k] += a.pe[((long long)k * C + c) * a.nchunk + j];
        tot[6] += (Acc)pl[j];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) tot[k] = block_sum<Acc, TB_ELEM>(tot[k], sh);
    if (threadIdx.x != 0) return;
    // The marginal log-likelihood of the auxiliary model WITHOUT walking its innovations: for any path z, ell = log p(z, u, y) - log p(z | u, y), and for the sampled
    // proposal x' both are at hand -- the joint is the sum pass E forms anyway (jp_prop), the conditional is the density of the pathwise draw,
    // log q(x' | u) = sum_t (-1/2 |eps_t|^2 - log|Lc_t| - D/2 log 2 pi) (pass C accumulates the first term, k_fs_clog the rest).  Same quantity as filtering.py:55-62
    // to rounding; log alpha does not depend on it at all (it enters lp_prop and lp_rev alike), only the reported lp_prop / lp_rev do.
    (void)ell0;
    const Acc jp_prop = tot[3] + (Acc)head5[0 * C + c], jp_rev = tot[0] + (Acc)head5[1 * C + c];
    const Acc ell = jp_prop - (tot[6] - *a.clog);
    const Acc lt_prop = tot[4] + (Acc)head5[2 * C + c], lt_rev = tot[1] + (Acc)head5[3 * C + c];
    const Acc corr = (tot[5] - tot[2]) + (Acc)head5[4 * C + c];
    const Acc lp_prop = jp_prop - ell, lp_rev = jp_rev - ell;
    Acc la = lt_prop - lt_rev;
    la += lp_rev - lp_prop;
    la -= corr;
    const Acc alpha = exp_(la != la ? la : min_(la, (Acc)0));  // jnp.minimum(0, nan) = nan (generic.py:105): a NaN ratio rejects
    const int acc = ((Acc)u_acc[c] < alpha) ? 1 : 0;  // NaN alpha -> reject, as jax.random.bernoulli(key, nan)
    accepted[c] = acc;
    if (sel) sel[c] ^= acc;
    if (logs) {
        logs[c * 5 + 0] = (R)la;
        logs[c * 5 + 1] = (R)lp_prop;
        logs[c * 5 + 2] = (R)lp_rev;
        logs[c * 5 + 3] = (R)lt_prop;
        logs[c * 5 + 4] = (R)lt_rev;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------
// few chains (<= 128): how many chunks one workgroup of up to 256 lanes walks side by side (lanes = pack x cp), 1 = the plain mapping.  AUXSSM_FS_PACK=0 switches it off, n > 1 caps it (measurement).
// Measured at C2's sizes (profiles/r04_i_low_chain_layout.txt): 8 chains 18.2k -> 23.4k sweeps/s, 16 chains 40.7k -> 51.0k, 32 chains 78.0k -> 91.0k, 64 chains 133.7k -> 152.0k,
// 128 chains 189.4k -> 218.4k (the chunks of a workgroup share one staging of the normal tables and one barrier).
inline int fs_pack(int C, int* cp_out) {
    static const int on = [] { const char* e = getenv("AUXSSM_FS_PACK"); return e ? atoi(e) : 1; }();
    int cp = 2;
    while (cp < C) cp <<= 1;
    *cp_out = cp;
    static const int lanes = [] { const char* e = getenv("AUXSSM_FS_PACK_LANES"); const int v = e ? atoi(e) : 256; return v == 64 || v == 128 ? v : 256; }();   // lanes per packed workgroup (measured: 256 best from 32 chains on)
    if (!on || cp > lanes / 2) return 1;
    int pack = lanes / cp;
    const int cap = on > 1 ? on : 8;   // (LDS: pack blocks of E coefficient rows)
    return pack > cap ? cap : pack;
}
inline int fs_chunk_len(const auxssm_ctx* h, int C, int T) {
    static const int waves = [] { const char* e = getenv("AUXSSM_FS_WAVES"); const int v = e ? atoi(e) : 10; return v >= 1 && v <= 64 ? v : 10; }();
    static const int fixedE = [] { const char* e = getenv("AUXSSM_FS_E"); return e ? atoi(e) : 0; }();
    if (fixedE >= 2 && fixedE <= 64) return fixedE;
    const long long stiles = (C + TB_CM - 1) / TB_CM;
    long long want = (long long)h->num_cu * 4 * waves / stiles;
    if (want < 1) want = 1;
    long long E = (T + want - 1) / want;
    if (E < 16) E = 16;
    if (E > 32) E = 32;
    // measured at T = 65536, d = 4, fp64 once pass C stopped walking the innovations (rows of 88 / 84 / 68 reals: 64 steps = 45 KB of LDS at most): 64 chains 16 (81k
    // sweeps/s; 32: 76k), 96 chains 32 (100k; 16: 96k), 128 chains 32 (125k; 16: 118k), 256 chains 32 or 64 (175k; 24: 172k), 1024 chains 64 (228k; 32: 220k)
    if (C >= 1024) E = 64;
    else if (C >= 96) E = 32;
    if (E > T) E = T > 2 ? T : 2;
    return (int)E;
}
template <typename R, int D, int PO> size_t fused_ws(const auxssm_ctx* h, const KDims& d) {
    constexpr int P = D + PO;
    const int E = fs_chunk_len(h, d.C, d.T), nchunk = (d.T + E - 1) / E;
    size_t b = 0;
    b += (size_t)d.C * nchunk * (4 * SampPre<R, D>::NPAD * sizeof(R) + 6 * sizeof(Acc) + sizeof(R)) + 16 * 256;
    b += (size_t)d.C * (5 + 1 + D) * sizeof(R) + 4 * 256;
    // model stage (side slab when the stage overlaps, else this one): matrix filter, gain / sampler / log-density tables, chunk products
    b += filter_ws<R, D, P>(h, KDims{1, d.T, 1}, 1);
    b += (size_t)d.T * (SampShared<R, D>::NPAD + LogShared<R, D, PO>::NPAD + (size_t)D * D) * sizeof(R) + (size_t)2 * nchunk * D * D * sizeof(R) + 10 * 256 + ((size_t)d.T / 256 + 2) * sizeof(Acc);
    b += (size_t)d.T * (FsRows<R, D, PO>::NC + FsRows<R, D, PO>::NE + 2 * (size_t)D * D) * sizeof(R) + (size_t)nchunk * D * D * sizeof(R) + 8 * 256;
    return b;
}
template <typename R, int D, int PO> int run_fused_shared(auxssm_ctx* h, FusedHost& f) {
    constexpr int P = D + PO;
    const int C = f.fa.d.C, T = f.fa.d.T, n = T - 1;
    FusedArgs a{};
    a.C = C; a.T = T;
    a.E = fs_chunk_len(h, C, T);
    a.nchunk = (T + a.E - 1) / a.E;
    a.pack = fs_pack(C, &a.cp);
    {   // the packed workgroup stages `pack` chunk blocks of coefficient rows: keep them (and the normal tables) within 128 KB of LDS
        using FR = FsRows<R, D, PO>;
        constexpr int NMAX = FR::NC > FR::NE ? FR::NC : FR::NE;
        while (a.pack > 1 && (size_t)a.pack * fs_pk_block<R, NMAX>(a.E) * sizeof(R) + FsNormTabSel<R>::BYTES > (size_t)128 * 1024) a.pack >>= 1;
    }
    a.xa = f.xa; a.xb = f.xb; a.sel = f.sel; a.u = f.u; a.inc = f.inc;
    a.ka0 = f.keys[0]; a.ka1 = f.keys[1]; a.ks0 = f.keys[2]; a.ks1 = f.keys[3];
    a.eps0s = f.eps0s;
    a.delta = f.la.delta; a.shd = f.la.shd; a.dptr = f.la.dptr; a.nan_policy = f.la.nan_policy;
    a.memo = f.memo;
    f.fa.memo = f.sa.memo = f.la.memo = f.memo;  // the model stage below; cleared before the chain passes (k_filter_t0 serves both)
    R *cprod_f, *cprod_s, *rows_c, *rows_e, *psi;
    {
        // MODEL STAGE (ctx.h::SideStage when the sweep opened one): everything that reads the model and the step size only
        {
            const int rc = build_gain_table<R, D, P>(h, f.fa);
            if (rc) return rc;
        }
        SideScope side(h);
        f.sa.tab = ws_take(h, (size_t)T * SampShared<R, D>::NPAD * sizeof(R));
        f.la.tab = ws_take(h, (size_t)n * LogShared<R, D, PO>::NPAD * sizeof(R));
        R* gpre = (R*)ws_take(h, (size_t)T * D * D * sizeof(R));
        cprod_f = (R*)ws_take(h, (size_t)a.nchunk * D * D * sizeof(R));
        cprod_s = (R*)ws_take(h, (size_t)a.nchunk * D * D * sizeof(R));
        rows_c = (R*)ws_take(h, (size_t)T * FsRows<R, D, PO>::NC * sizeof(R));
        rows_e = (R*)ws_take(h, (size_t)T * FsRows<R, D, PO>::NE * sizeof(R));
        R* fpre = (R*)ws_take(h, (size_t)T * D * D * sizeof(R));
        R* ntab = (R*)ws_take(h, (size_t)T * D * D * sizeof(R));
        psi = (R*)ws_take(h, (size_t)a.nchunk * D * D * sizeof(R));
        const int nclb = (T + 255) / 256;
        Acc* clog = (Acc*)ws_take(h, 256 + (size_t)nclb * sizeof(Acc));  // [0] the sum, [32 ...) the partial sums
        if (!f.sa.tab || !f.la.tab || !gpre || !cprod_f || !cprod_s || !rows_c || !rows_e || !fpre || !ntab || !psi || !clog) return AUXSSM_ERR_NOMEM;
        a.clog = clog;
        a.gain = f.fa.tab; a.samp = f.sa.tab; a.logt = f.la.tab; a.gpre = gpre;
        {
            ProfScope ps(h, AUXSSM_K_SAMPLE_INIT);
            hipLaunchKernelGGL((k_sample_shared_tab<R, D>), dim3((T + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, f.sa);
            hipLaunchKernelGGL((k_fs_clog_part<R, D>), dim3(nclb), dim3(256), 0, h->stream, T, (const R*)f.sa.tab, clog + 32, f.memo);
            hipLaunchKernelGGL((k_fs_clog_sum<D>), dim3(1), dim3(256), 0, h->stream, T, nclb, (const Acc*)(clog + 32), clog, f.memo);
            hipLaunchKernelGGL((k_sweep_logpdf_tab<R, D, PO>), dim3((n + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, f.la);
            hipLaunchKernelGGL((k_fs_fprod<R, D, P>), dim3((a.nchunk + TB_CM - 1) / TB_CM), dim3(TB_CM), 0, h->stream, a, cprod_f, fpre);
            hipLaunchKernelGGL((k_fs_gpre<R, D>), dim3((a.nchunk + TB_CM - 1) / TB_CM), dim3(TB_CM), 0, h->stream, a, gpre, cprod_s);
            hipLaunchKernelGGL((k_fs_psi<R, D>), dim3((a.nchunk + TB_CM - 1) / TB_CM), dim3(TB_CM), 0, h->stream, a, (const R*)gpre, (const R*)fpre, ntab, psi);
            {
                using F = FsRows<R, D, PO>;
                const long long nmax = (long long)T * (F::NC > F::NE ? F::NC : F::NE);
                hipLaunchKernelGGL((k_fs_rows<R, D, PO>), dim3((unsigned)((nmax + 255) / 256), 2), dim3(256), 0, h->stream, a, (const R*)ntab, rows_c, rows_e);
            }
        }
    }
    {
        const int rc = side_close(h);
        if (rc) return rc;
    }
    a.memo = nullptr;
    f.fa.memo = f.sa.memo = f.la.memo = nullptr;  // (the chain passes are never skipped)
    const size_t npre = (size_t)C * a.nchunk * SampPre<R, D>::NPAD * sizeof(R);
    a.agg_f = ws_take(h, npre); a.pre_f = ws_take(h, npre); a.agg_s = ws_take(h, npre); a.pre_s = ws_take(h, npre);
    a.pa = (Acc*)ws_take(h, (size_t)3 * C * a.nchunk * sizeof(Acc));
    a.pe = (Acc*)ws_take(h, (size_t)3 * C * a.nchunk * sizeof(Acc));
    a.pell = ws_take(h, (size_t)C * a.nchunk * sizeof(R));
    R* head5 = (R*)ws_take(h, (size_t)5 * C * sizeof(R));
    f.fa.ell0 = ws_take(h, (size_t)C * sizeof(R));
    if (!a.agg_f || !a.pre_f || !a.agg_s || !a.pre_s || !a.pa || !a.pe || !a.pell || !head5 || !f.fa.ell0) return AUXSSM_ERR_NOMEM;
    a.m0p = f.fa.ms.ptr;
    // t = 0 update of every chain (reads the concatenated model: after the join)
    f.fa.t0_keep_ps = 1;
    hipLaunchKernelGGL((k_filter_t0<R, D, P>), dim3((C + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, f.fa);
    using F = FsRows<R, D, PO>;
    static const int tbf_env = getenv("AUXSSM_FS_TBF") ? atoi(getenv("AUXSSM_FS_TBF")) : 0;
    const int TBF = (tbf_env == 64 || tbf_env == 128 || tbf_env == 256) && C >= tbf_env ? tbf_env : (C >= 256 ? 256 : (C + 63) / 64 * 64);
    const unsigned grid = (unsigned)a.nchunk * (unsigned)((C + TBF - 1) / TBF);
    const unsigned grid_pk = (unsigned)((a.nchunk + a.pack - 1) / a.pack);
    const size_t lds = (size_t)TB_AGGS * SampElem<R, D>::NPAD * sizeof(R);
    if (a.pack > 1) {
        const size_t lds_ac = (size_t)a.pack * fs_pk_block<R, F::NC>(a.E) * sizeof(R) + FsNormTabSel<R>::BYTES;
        if (lds_ac > 64 * 1024) AX_HIP(hipFuncSetAttribute((const void*)k_fs_ac<R, D, PO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ac));
        ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
        hipLaunchKernelGGL((k_fs_ac<R, D, PO, true>), dim3(grid_pk), dim3(a.pack * a.cp), lds_ac, h->stream, a, (const R*)rows_c);
    } else {
        const size_t lds_ac = (size_t)a.E * F::NC * sizeof(R) + FsNormTabSel<R>::BYTES;
        if (lds_ac > 64 * 1024) AX_HIP(hipFuncSetAttribute((const void*)k_fs_ac<R, D, PO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ac));
        ProfScope ps(h, AUXSSM_K_FILTER_SCAN);
        hipLaunchKernelGGL((k_fs_ac<R, D, PO>), dim3(grid), dim3(TBF), lds_ac, h->stream, a, (const R*)rows_c);
    }
    {
        ProfScope ps(h, AUXSSM_K_SAMPLE_SCAN);  // the two aggregate scans: first means of the chunks, then (completed by Psi m_start) the sampler's chunk starts
        hipLaunchKernelGGL((k_aff_aggs<R, D>), dim3(C), dim3(TB_AGGS), lds, h->stream, ScanBufs{a.agg_f, a.pre_f}, (const R*)cprod_f, a.nchunk);
        hipLaunchKernelGGL((k_fs_esfix<R, D>), dim3((unsigned)(((long long)C * a.nchunk + 255) / 256)), dim3(256), 0, h->stream, C, a.nchunk, (const R*)psi, (const R*)a.pre_f,
                           (R*)a.agg_s);
        hipLaunchKernelGGL((k_aff_aggs<R, D>), dim3(C), dim3(TB_AGGS), lds, h->stream, ScanBufs{a.agg_s, a.pre_s}, (const R*)cprod_s, a.nchunk);
    }
    {
        ProfScope ps(h, AUXSSM_K_LOGPDF);
        if (a.pack > 1) {
            const size_t lds_e = (size_t)a.pack * fs_pk_block<R, F::NE>(a.E) * sizeof(R);
            if (lds_e > 64 * 1024) AX_HIP(hipFuncSetAttribute((const void*)k_fs_e<R, D, PO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_e));
            hipLaunchKernelGGL((k_fs_e<R, D, PO, true>), dim3(grid_pk), dim3(a.pack * a.cp), lds_e, h->stream, a, (const R*)rows_e);
        } else
            hipLaunchKernelGGL((k_fs_e<R, D, PO>), dim3(grid), dim3(TBF), (size_t)a.E * F::NE * sizeof(R), h->stream, a, (const R*)rows_e);
    }
    {
        ProfScope ps(h, AUXSSM_K_SELECT);
        hipLaunchKernelGGL((k_fs_head<R, D, PO>), dim3((C + TB_ELEM - 1) / TB_ELEM), dim3(TB_ELEM), 0, h->stream, f.la, f.xa, (const void*)f.xb, (const int32_t*)f.sel, head5);
        hipLaunchKernelGGL((k_fs_accept<R>), dim3(C), dim3(TB_ELEM), 0, h->stream, a, (const R*)head5, (const R*)f.fa.ell0, (const R*)f.u_acc, f.accepted,
                           (R*)f.logs, f.sel);
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}


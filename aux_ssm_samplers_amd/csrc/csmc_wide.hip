// csmc_wide.hip -- the conditional-SMC sweep for WIDE states with FEW particles: 4 < dx <= 32, N <= 64 (the reference's own timed stochastic-volatility
// protocol is D = 30, N = 25, T = 250: examples/stochastic_volatility/experiment.sh:1-10, experiment.py:38-55, auxiliary_csmc.py:14-46).
//
// Same algorithm, same sweep contract (csmc_dev.h: unnormalised weights shifted by a bound or the exact maximum, DPP-order cumsum, descent search,
// ballot-counted single draw, reciprocal Cholesky diagonals, det_exp / det_log, explicit fma) and the same oracle (oracle/csmc_ref.c, MAXD = 32) as the
// register kernels of csmc.hip -- what changes is where a particle lives (the register kernels keep R x[D], eps[D], mu[D], z[D] per lane and the model BY VALUE
// in the kernel arguments, which stops at dx = 4): here a particle's dx components sit ACROSS the 32 lanes of a half-wave, two particles per wave, 16 waves per
// chain while there are CUs to spare (8 beyond), the model's matrices in LDS.  Linear-Gaussian transitions, every potential / proposal of the family,
// gradient-informed proposals (csmc/independent.py:57-75 gradient=True, :173-190, :252-268; both AUXSSM_GRAD_* weightings: the reference's own SV protocol
// exposes --gradient at D = 30) and time-varying transitions (the step's F_t, b_t, chol Q_t re-staged into LDS before the step's first barrier) since round 4.
// (The first version of this file -- one wave per chain, one lane per particle walking its dx x dx products alone: 30.5 ms per sweep of the SV protocol against
// 2.6 now -- is in the history)
//
// In-kernel draws (AUXSSM_NOISE_THREEFRY) use the NATURAL flat indices of the explicit arrays -- eps_prop[c][t][n][k] = normal ((c T + t) N + n) dx + k of
// stream 2, u_res[c][s][n] = uniform (c (T-1) + s) N + n of stream 3, u_bwd[c][t] = uniform c T + t of stream 4 -- not the two-steps-per-block packing
// of the register kernels (csmc/_device.py::key_noise(wide=True) builds the equivalent arrays).
#include <type_traits>
#include <utility>

#include "csmc_dev.h"

namespace ax {

constexpr int CSW_MAXD = 32;

template <typename R> struct FkW {
    int proposal, potential, D;
    const R *m0, *LP0, *iLP0, *F, *b, *LQ, *iLQ;  // device arrays, matrices row-major with leading dimension D
    R c_init, c_trans, c_obs, inv_sig_y;
    int gradient;                          // AUXSSM_GRAD_*
    const R *Ft, *bt, *LQt, *ctt, *idt;    // time-varying transitions (csmc_dev.h::FkDev: row t = transition t -> t + 1), or null
};
// the transition t -> t + 1 in global memory (gradient kernel; the sweep kernels read it from LDS)
template <typename R> struct TransW {
    const R *F, *b, *LQ;
};
template <typename R> __device__ __forceinline__ TransW<R> trans_w(const FkW<R>& m, long long t) {
    const long long D = m.D;
    if (m.Ft) return TransW<R>{m.Ft + t * D * D, m.bt + t * D, m.LQt + t * D * D};
    return TransW<R>{m.F, m.b, m.LQ};
}

// e_i = exp(lw_i - max lw) over ONE wave (csmc_dev.h::block_expmax)
template <typename R> __device__ __forceinline__ R wave_expmax(R lw, R* m_out) {
    R m = wave_max_dpp(lw);
    if (!(m - m == 0)) m = 0;
    if (m_out) *m_out = m;
    return det_exp(lw - m);
}
// the descent search of the sweep contract on a wave's cumulative weights held in LDS (unpadded: at most 64 entries)
template <typename R> __device__ __forceinline__ int search_w(const R* c, int N, R r) {
    int s0 = 1;
    while (s0 * 2 < N) s0 *= 2;
    int pos = 0;
    for (int s = s0; s > 0; s >>= 1) {
        const int q = pos + s - 1;
        pos += (q < N && c[q < N ? q : N - 1] < r) ? s : 0;
    }
    return pos < N - 1 ? pos : N - 1;
}

template <typename R> __global__ void k_cw_potbound(int T, FkW<R> m, const R* __restrict__ y, R* __restrict__ gb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int D = m.D;
    R b = 0;
    if (m.potential == 1) b = m.c_obs;
    else if (m.potential == 3) {
        int nobs = 0;
        for (int k = 0; k < D; ++k) nobs += (y[(long long)t * D + k] - y[(long long)t * D + k] == 0) ? 1 : 0;
        b = (R)nobs * m.c_obs;
    } else if (m.potential == 2) {
        for (int k = 0; k < D; ++k) {
            const R yk = y[(long long)t * D + k], y2 = yk * yk;
            R v = (R)0;
            if (y2 - y2 == 0) v = y2 > (R)0 ? fma_((R)-0.5, (R)1 + det_log(y2), m.c_obs) : (R)INFINITY;
            b += v > (R)0 ? v : (R)0;
        }
    }
    gb[t] = b;
}

// additive constants and reciprocal diagonals of the time-varying transition densities (csmc_dev.h::k_csmc_ctrans with a runtime dimension, same operations)
template <typename R> __global__ void k_cw_ctrans(int n, int D, const R* __restrict__ LQt, R* __restrict__ ct, R* __restrict__ idt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    R c = 0;
    for (int k = 0; k < D; ++k) {
        const R l = LQt[((long long)t * D + k) * D + k];
        c -= det_log(l);
        idt[(long long)t * D + k] = (R)1 / l;
    }
    ct[t] = c - (R)D * (R)0.91893853320467274178;
}
// w <- (L L^T)^-1 r (csmc_dev.h::cho_solve_fixed, runtime dimension, leading dimension D)
template <typename R> __device__ __forceinline__ void cho_solve_w(int D, const R* L, const R* r, R* w) {
    R z[CSW_MAXD];
    for (int k = 0; k < D; ++k) {
        R acc = r[k];
        for (int j = 0; j < k; ++j) acc = fma_(-L[k * D + j], z[j], acc);
        z[k] = acc / L[k * D + k];
    }
    for (int k = D - 1; k >= 0; --k) {
        R acc = z[k];
        for (int j = k + 1; j < D; ++j) acc = fma_(-L[j * D + k], w[j], acc);
        w[k] = acc / L[k * D + k];
    }
}
// the gradient of the model's joint log-density at u (csmc_dev.h::k_csmc_grad, same operations in the same order; one thread per (chain, time step):
// C T threads of O(dx^2) work, once per sweep -- 0.5 M multiply-adds at the SV protocol's size)
template <typename R> __global__ void k_cw_grad(CsmcArgs a, FkW<R> m) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)a.C * a.T) return;
    const int D = m.D;
    const long long t = g % a.T;
    const R* u = (const R*)a.u + g * D;
    R gr[CSW_MAXD], r[CSW_MAXD], w[CSW_MAXD];
    const R* yv = (const R*)a.y;
    for (int k = 0; k < D; ++k) {
        const R y = yv ? yv[t * D + k] : (R)0;
        R v = 0;
        if (m.potential == 1 || (m.potential == 3 && y - y == 0)) v = ((y - u[k]) * m.inv_sig_y) * m.inv_sig_y;
        else if (m.potential == 2) {
            const R e = det_exp(-u[k]);
            v = (R)0.5 * fma_(y * y, e, (R)-1);
            v = (v == v) ? v : (R)0;
        }
        gr[k] = v;
    }
    if (t == 0) {
        for (int k = 0; k < D; ++k) r[k] = u[k] - m.m0[k];
        cho_solve_w<R>(D, m.LP0, r, w);
    } else {
        const TransW<R> tr = trans_w<R>(m, t - 1);
        for (int k = 0; k < D; ++k) {
            R acc = tr.b[k];
            for (int j = 0; j < D; ++j) acc = fma_(tr.F[k * D + j], u[j - D], acc);
            r[k] = u[k] - acc;
        }
        cho_solve_w<R>(D, tr.LQ, r, w);
    }
    for (int k = 0; k < D; ++k) gr[k] = gr[k] - w[k];
    if (t + 1 < a.T) {
        const TransW<R> tr = trans_w<R>(m, t);
        for (int k = 0; k < D; ++k) {
            R acc = tr.b[k];
            for (int j = 0; j < D; ++j) acc = fma_(tr.F[k * D + j], u[j], acc);
            r[k] = u[D + k] - acc;
        }
        cho_solve_w<R>(D, tr.LQ, r, w);
        for (int k = 0; k < D; ++k) {
            R acc = 0;
            for (int j = 0; j < D; ++j) acc = fma_(tr.F[j * D + k], w[j], acc);
            gr[k] = gr[k] + acc;
        }
    }
    for (int k = 0; k < D; ++k) ((R*)a.grad)[g * D + k] = gr[k];
}

// =================================================================================================================================================
// Both passes: NW2 waves per chain, a particle's dx components ACROSS the 32 lanes of a half-wave (two particles per wave, particle
// i = 2 NW2 s + 2 wave + half in pass s of ceil(N / (2 NW2))).  Everything the contract orders is kept in its order:
//   * a mean component is one dot product, accumulated over j = 0 .. dx - 1 by the lane that owns the component;
//   * the forward substitution runs COLUMN by column -- z_j = acc_j / L_jj is final once columns < j have been applied, it is broadcast inside the
//     half-wave by v_readlane and every lane k > j applies acc_k = fma(-L_kj, z_j, acc_k): each acc_k receives the same updates in the same order
//     as the row-oriented loop of the contract (csmc_dev.h::gauss_chol_logpdf), so z, q = sum z_k^2 (accumulated in k order by every lane alike) and the densities are bit-identical;
//   * the potential's sum over components is accumulated in component order from readlane broadcasts of the per-component terms;
//   * weights, cumulative sums, searches and the single draw of the backward pass are done by wave 0 with one lane per particle, exactly as before.
// Two workgroup barriers per time step in either pass.  In-kernel draws keep the natural flat indices (and use both normals of a Threefry block).
#ifndef CW2_ABL
#define CW2_ABL 0  // diagnostic builds: 1 no search, 2 no draws, 4 no substitution, 8 no potential sums, 16 no mean products (tools/cw2_ablate.sh)
#endif

template <typename R> struct Cw2Lds {
    int D, S;
    R *F, *LQ, *b, *iL, *c, *lwv, *xa, *xb, *eps, *blk;
    int* idx;
    __device__ Cw2Lds(char* smem, int D_) : D(D_), S(CSW_MAXD + 1) {  // rows padded with zeros to 32 columns (+ 1: odd stride): every component loop runs 32 steps, unrolled
        F = (R*)smem;           // [D][S]
        LQ = F + D * S;         // [D][S]
        b = LQ + D * S;
        iL = b + D;
        c = iL + D;             // [64]
        lwv = c + 64;           // [64]
        xa = lwv + 64;          // [64][S]
        xb = xa + 64 * S;       // [64][S]
        eps = xb + 64 * S;      // [64][S]
        blk = eps + 64 * S;     // [8][12] the 4 x 4 diagonal blocks of chol Q and the reciprocal diagonal, in the order gauss_half_blk reads them
        idx = (int*)(blk + 96);  // [64]
    }
    static size_t bytes(int D) { return ((size_t)2 * D * (CSW_MAXD + 1) + 2 * D + 128 + (size_t)3 * 64 * (CSW_MAXD + 1) + 96) * sizeof(R) + 64 * sizeof(int) + 64; }
};
template <typename R> __device__ __forceinline__ void cw2_stage(const FkW<R>& m, Cw2Lds<R>& L, int tid, int nt) {
    const int D = m.D, S = L.S;
    for (int i = tid; i < D * S; i += nt) {
        const int r = i / S, q = i - r * S;
        L.F[i] = q < D ? m.F[r * D + q] : (R)0;
        L.LQ[i] = q < D ? m.LQ[r * D + q] : (R)0;
    }
    for (int i = tid; i < D; i += nt) L.b[i] = m.b[i], L.iL[i] = m.iLQ[i];
    for (int i = tid; i < 64 * S; i += nt) L.xa[i] = 0, L.xb[i] = 0, L.eps[i] = 0;
    if (tid < 96) {  // block jb = 4 (tid / 12): [L10 L20 L21 L30 L31 L32 | i0 i1 i2 i3 | 0 0], zeros beyond D
        const int bq = tid / 12, e = tid - 12 * bq, jb = 4 * bq;
        const int rr[6] = {1, 2, 2, 3, 3, 3}, cc[6] = {0, 0, 1, 0, 1, 2};
        R v = 0;
        if (e < 6) {
            const int r = jb + rr[e], q = jb + cc[e];
            v = r < D ? m.LQ[r * D + q] : (R)0;
        } else if (e < 10) {
            v = jb + e - 6 < D ? m.iLQ[jb + e - 6] : (R)0;
        }
        L.blk[tid] = v;
    }
    __syncthreads();
}
// time-varying transitions: the rows of transition tt -> tt + 1 replace the staged model (same layout; called by every thread between the two barriers that
// separate the particle sections of consecutive steps, so no section reads a half-written model)
template <typename R> __device__ __forceinline__ void cw2_stage_t(const FkW<R>& m, Cw2Lds<R>& L, long long tt, int tid, int nt) {
    const int D = m.D, S = L.S;
    const R* F = m.Ft + tt * D * D;
    const R* LQ = m.LQt + tt * D * D;
    const R* b = m.bt + tt * D;
    const R* iL = m.idt + tt * D;
    for (int i = tid; i < D * S; i += nt) {
        const int r = i / S, q = i - r * S;
        L.F[i] = q < D ? F[r * D + q] : (R)0;
        L.LQ[i] = q < D ? LQ[r * D + q] : (R)0;
    }
    for (int i = tid; i < D; i += nt) L.b[i] = b[i], L.iL[i] = iL[i];
    if (tid < 96) {
        const int bq = tid / 12, e = tid - 12 * bq, jb = 4 * bq;
        const int rr[6] = {1, 2, 2, 3, 3, 3}, cc[6] = {0, 0, 1, 0, 1, 2};
        R v = 0;
        if (e < 6) {
            const int r = jb + rr[e], q = jb + cc[e];
            v = r < D ? LQ[r * D + q] : (R)0;
        } else if (e < 10) {
            v = jb + e - 6 < D ? iL[jb + e - 6] : (R)0;
        }
        L.blk[tid] = v;
    }
}
// value of lane J of MY half-wave: ds_swizzle in bit mode (lane' = (lane & and) | or inside each group of 32 lanes, and = 0, or = J) -- one LDS-crossbar
// instruction, no memory, no scalar round trip (two v_readlane + two v_mov + a select before: the component loops are bound by the CU's instruction issue,
// thirteen waves of one chain walk them together)
template <typename R, int J> __device__ __forceinline__ R half_bcast(R v) {
    if constexpr (sizeof(R) == 4) {
        return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), J << 5));
    } else {
        const int lo_ = __builtin_amdgcn_ds_swizzle(__double2loint(v), J << 5), hi_ = __builtin_amdgcn_ds_swizzle(__double2hiint(v), J << 5);
        return __hiloint2double(hi_, lo_);
    }
}
template <int J, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (J < N) {
        f(std::integral_constant<int, J>{});
        static_for<J + 1, N>(f);
    }
}
// sum_k ((x_k - pm_k)^2 - (x_k - u_k)^2) / (2 s^2) of the particle whose component k this lane holds (csmc_dev.h::grad_correction: the two multiply-adds per
// component, in component order, from broadcasts of the lanes' differences; components beyond D add fma(0, 0, acc) = acc)
template <typename R> __device__ __forceinline__ R grad_corr_half(int D, int k, R xk, R uk, R pmk, R s) {
    const R d1 = k < D ? xk - uk : (R)0, d2 = k < D ? xk - pmk : (R)0;
    R acc = 0;
    static_for<0, CSW_MAXD>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const R b2 = half_bcast<R, j>(d2), b1 = half_bcast<R, j>(d1);
        acc = fma_(b2, b2, acc);
        acc = fma_(-b1, b1, acc);
    });
    return acc * ((R)0.5 / (s * s));
}
// log N(x; mean, L L^T) of the particle whose component k this lane holds (x - mean in `acc`), column-oriented substitution; L: lane k's row pointer with
// element stride 1 (L[j] = L_kj), iLk = 1 / L_kk.  Every lane of the half-wave returns the same value.
template <typename R> __device__ __forceinline__ R gauss_half(int D, int k, bool hi, R acc, const R* Lrow, R iLk, R cst) {
    R q = 0;
    if (CW2_ABL & 4) return acc * cst;
    acc = k < D ? acc : (R)0;  // (components beyond D: z = 0, fma(0, 0, q) = q -- the 32 steps below are the D steps of the contract)
    R l[CSW_MAXD];
#pragma unroll
    for (int j = 0; j < CSW_MAXD; ++j) l[j] = j < D ? Lrow[j] : (R)0;
    static_for<0, CSW_MAXD>([&](auto jc) {  // (straight-line: the row of L requested up front, only the broadcast / multiply-add chain is serial)
        constexpr int j = decltype(jc)::value;
        const R zj = half_bcast<R, j>(acc * iLk);
        q = fma_(zj, zj, q);
        acc = (k > j && k < D) ? fma_(-l[j], zj, acc) : acc;
    });
    return fma_((R)-0.5, q, cst);
}
// The same density with the substitution in BLOCKS OF FOUR columns: one round of four broadcasts hands every lane the accumulators of lanes jb .. jb + 3 (final with
// respect to the columns before jb); every lane then solves the 4 x 4 triangular block for z_jb .. z_jb+3 itself -- the multiply-adds lane jb + a would apply to its own
// accumulator, in the same order, so the same bits -- and applies the four columns to its own accumulator in order.  Eight broadcast latencies per density instead of
// thirty-two (profiles/r03_d_cw2_ablation.txt: the dependent broadcast chain was half of the sweep).  blk: Cw2Lds::blk (uniform reads).
template <typename R> __device__ __forceinline__ R gauss_half_blk(int D, int k, R acc, const R* Lrow, const R* blk, R cst) {
    if (CW2_ABL & 4) return acc * cst;
    R q = 0;
    acc = k < D ? acc : (R)0;
    R l[CSW_MAXD];
#pragma unroll
    for (int j = 0; j < CSW_MAXD; ++j) l[j] = Lrow[j];  // (rows are zero-padded to 32 columns)
    static_for<0, CSW_MAXD / 4>([&](auto bc) {
        constexpr int jb = 4 * decltype(bc)::value;
        const R* e = blk + 12 * decltype(bc)::value;
        R a0 = half_bcast<R, jb>(acc), a1 = half_bcast<R, jb + 1>(acc), a2 = half_bcast<R, jb + 2>(acc), a3 = half_bcast<R, jb + 3>(acc);
        const R z0 = a0 * e[6];
        a1 = fma_(-e[0], z0, a1);
        const R z1 = a1 * e[7];
        a2 = fma_(-e[1], z0, a2);
        a2 = fma_(-e[2], z1, a2);
        const R z2 = a2 * e[8];
        a3 = fma_(-e[3], z0, a3);
        a3 = fma_(-e[4], z1, a3);
        a3 = fma_(-e[5], z2, a3);
        const R z3 = a3 * e[9];
        q = fma_(z0, z0, q);
        q = fma_(z1, z1, q);
        q = fma_(z2, z2, q);
        q = fma_(z3, z3, q);
        acc = (k > jb && k < D) ? fma_(-l[jb], z0, acc) : acc;
        acc = (k > jb + 1 && k < D) ? fma_(-l[jb + 1], z1, acc) : acc;
        acc = (k > jb + 2 && k < D) ? fma_(-l[jb + 2], z2, acc) : acc;
        acc = (k > jb + 3 && k < D) ? fma_(-l[jb + 3], z3, acc) : acc;
    });
    return fma_((R)-0.5, q, cst);
}
// g_t(x) of that particle: per-component terms in the lanes, summed in component order (csmc_dev.h::potential with a runtime dimension, same operations)
template <typename R> __device__ __forceinline__ R potential_half(const FkW<R>& m, int k, bool hi, R xk, R yk) {
    const int D = m.D;
    if (m.potential == 0) return (R)0;
    if (CW2_ABL & 8) return xk * yk;
    if (m.potential == 1 || m.potential == 3) {
        const bool obs = m.potential == 1 || (yk - yk == 0);
        const R z = (k < D && obs) ? (yk - xk) * m.inv_sig_y : (R)0;
        R q = 0;
        static_for<0, CSW_MAXD>([&](auto jc) {
            const R zj = half_bcast<R, decltype(jc)::value>(z);
            q = fma_(zj, zj, q);  // (a missing component, or one beyond D, contributes fma(0, 0, q) = q: the reference skips it)
        });
        if (m.potential == 1) return fma_((R)-0.5, q, m.c_obs);
        const unsigned long long bal = __ballot(k < D && obs);
        const int nobs = __popc((unsigned int)(hi ? bal >> 32 : bal & 0xffffffffull));
        return fma_((R)-0.5, q, (R)nobs * m.c_obs);
    }
    const R e = det_exp(-xk);
    const R sv = fma_(yk * yk, e, xk);
    R v = fma_((R)-0.5, sv, m.c_obs);
    v = (k < D && v == v) ? v : (R)0;
    R acc = 0;
    static_for<0, CSW_MAXD>([&](auto jc) { acc += half_bcast<R, decltype(jc)::value>(v); });
    return acc;
}
// this step's proposal noise into the LDS rows eps[n][k]: natural flat index ((ch T + t) N + n) D + k of stream 2, both normals of every Threefry block used
template <typename R> __device__ __forceinline__ void cw2_draw(const CsmcArgs& a, Cw2Lds<R>& L, int ch, int t, int r, int nr) {
    const int N = a.N, D = L.D, S = L.S, ND = N * D;
    if (CW2_ABL & 2) {
        for (int e = r; e < N * S; e += nr) L.eps[e] = (R)0.25;
        return;
    }
    const long long base = (((long long)ch * a.T + t) * N) * D;
    if (a.noise_mode == 0) {
        for (int e = r; e < ND; e += nr) {
            const int n = e / D, k = e - n * D;
            L.eps[n * S + k] = ((const R*)a.eps_prop)[base + e];
        }
        return;
    }
    const long long b0 = base >> 1, b1 = (base + ND - 1) >> 1;
    for (long long blk = b0 + r; blk <= b1; blk += nr) {
        R z0, z1;
        stream_normal2<R>(a.key0, a.key1, STREAM_EPS_PROP, (unsigned long long)blk, z0, z1);
        const long long e0 = 2 * blk - base, e1 = e0 + 1;
        if (e0 >= 0 && e0 < ND) {
            const int n = (int)e0 / D, k = (int)e0 - n * D;
            L.eps[n * S + k] = z0;
        }
        if (e1 >= 0 && e1 < ND) {
            const int n = (int)e1 / D, k = (int)e1 - n * D;
            L.eps[n * S + k] = z1;
        }
    }
}

template <typename R, int NW2> __global__ void __launch_bounds__(64 * NW2) k_cw2_fwd(CsmcArgs a, FkW<R> m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT_ = 64 * NW2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, N = a.N, T = a.T, D = m.D;
    const bool hi = lane >= 32;
    const int k = lane & 31;
    Cw2Lds<R> L(smem, D);
    cw2_stage<R>(m, L, tid, NT_);
    const int S = L.S, nslot = (N + 2 * NW2 - 1) / (2 * NW2);
    const int ch = a.c0 + blockIdx.x;
    const R* xstar = (const R*)a.x + (long long)ch * T * D;
    const R* uaux = (const R*)a.u + (long long)ch * T * D;
    const bool grad = m.gradient != 0 && m.proposal == 1, tv = m.Ft != nullptr;
    const R* gaux = grad ? (const R*)a.grad + (long long)ch * T * D : uaux;
    const R* yv = (const R*)a.y;
    R* xs = (R*)a.xs + (long long)ch * T * N * D;
    R* lws = (R*)a.lws + (long long)ch * T * N;
    int32_t* As = a.As ? a.As + (long long)ch * (T - 1) * N : nullptr;
    R* fmax = a.fmax ? (R*)a.fmax + (long long)ch * T : nullptr;
    const R* gbp = (const R*)a.gb;
    const bool bmode = gbp != nullptr && !(grad && m.gradient == 2);  // (the exact-gradient correction is unbounded in x: csmc.hip)
    const R ninf = -INFINITY;
    R bk = k < D ? L.b[k] : (R)0;
    const R* Frow = L.F + (k < D ? k : 0) * S;
    const R* Lrow = L.LQ + (k < D ? k : 0) * S;

    // ---- t = 0 (csmc.py:74-80)
    cw2_draw<R>(a, L, ch, 0, tid, NT_);
    __syncthreads();
    for (int s = 0; s < nslot; ++s) {
        const int i = s * 2 * NW2 + 2 * wv + (hi ? 1 : 0);
        const bool pl = i < N;  // (uniform per half-wave)
        const int ir = pl ? i : 0;
        R xk = 0, acc0 = 0, pmk = 0;
        if (k < D) {
            if (m.proposal == 0) {
                R acc = m.m0[k];
                for (int j = 0; j <= k; ++j) acc = fma_(m.LP0[k * D + j], L.eps[ir * S + j], acc);
                xk = acc;
            } else {  // AuxiliaryM0: N(u_0 [+ delta_0 / 2 grad_0], delta_0 / 2 I)  (independent.py:143-158)
                const R s0 = ((const R*)a.shd)[0];
                pmk = grad ? fma_(s0 * s0, gaux[k], uaux[k]) : uaux[k];
                xk = fma_(s0, L.eps[ir * S + k], pmk);
            }
            if (i == 0) xk = xstar[k];
            acc0 = xk - m.m0[k];
        }
        const R yk = (yv && k < D) ? yv[k] : (R)0;
        R g = potential_half<R>(m, k, hi, xk, yk);
        if (m.proposal == 1) g = g + gauss_half<R>(D, k, hi, acc0, m.LP0 + (long long)(k < D ? k : 0) * D, k < D ? m.iLP0[k] : (R)0, m.c_init);  // AuxiliaryG0
        if (grad) g = g + grad_corr_half<R>(D, k, xk, k < D ? uaux[k] : (R)0, pmk, ((const R*)a.shd)[0]);  // GradientAuxiliaryG0 (:173-190)
        if (pl && k < D) {
            L.xa[i * S + k] = xk;
            xs[(long long)i * D + k] = xk;
        }
        if (pl && k == 0) {
            L.lwv[i] = g;
            lws[i] = g;
        }
    }
    __syncthreads();
    R* xprev = L.xa;
    R* xcur = L.xb;
    for (int t = 1; t < T; ++t) {
        // the step's rows from global memory, requested before the resampling section and its barrier
        const R yk = (yv && k < D) ? yv[(long long)t * D + k] : (R)0;
        const R st = m.proposal != 0 ? ((const R*)a.shd)[t] : (R)0;
        const R uk = (m.proposal != 0 && k < D) ? uaux[(long long)t * D + k] : (R)0;
        const R xsk = k < D ? xstar[(long long)t * D + k] : (R)0;
        const R gk = (grad && k < D) ? gaux[(long long)t * D + k] : (R)0;
        const R ctr = tv ? m.ctt[t - 1] : m.c_trans;  // the transition t - 1 -> t (time-varying: row t - 1)
        if (tv) cw2_stage_t<R>(m, L, t - 1, tid, NT_);  // (the previous step's particle section is behind its barrier)
        if (wv == 0) {
            // weights of step t - 1 and the conditional multinomial resampling (resamplings.py:14-37), one lane per particle
            const bool live = lane < N;
            const R lw = live ? L.lwv[lane] : ninf;
            const int tp = t - 1;
            R Mb = (bmode ? gbp[tp] : (R)0) + (m.proposal == 1 ? (tv && tp >= 1 ? m.ctt[tp - 1] : m.c_trans) : (R)0);
            const bool used_bound = bmode && tp >= 1 && tp < T - 1 && (Mb - Mb == 0);
            R mstep, w;
            if (used_bound) {
                w = det_exp(lw - Mb);
                mstep = Mb;
            } else {
                w = wave_expmax<R>(lw, &mstep);
            }
            R cv = wave_scan_dpp(w);
            R tot = readlane_(cv, 63);
            if (used_bound && !(tot > (R)0)) {  // every weight underflowed under its bound: the exact maximum after all
                w = wave_expmax<R>(lw, &mstep);
                cv = wave_scan_dpp(w);
                tot = readlane_(cv, 63);
            }
            if (fmax && lane == 0) fmax[tp] = mstep;
            L.c[lane] = cv;
            __builtin_amdgcn_wave_barrier();
            const R un = (live && lane > 0) ? noise_uniform<R>(a, a.u_res, STREAM_U_RES, ((long long)ch * (T - 1) + (t - 1)) * N + lane) : (R)0;
            int idx = 0;
            if (CW2_ABL & 1) idx = (lane * 7) % N;
            else if (live && lane > 0) idx = search_w<R>(L.c, N, tot * ((R)1 - un));
            L.idx[lane] = idx;
            if (live && As) As[(long long)(t - 1) * N + lane] = idx;
            cw2_draw<R>(a, L, ch, t, lane, NT_);  // (its share of the draws: the other waves start with theirs)
        } else {
            cw2_draw<R>(a, L, ch, t, tid, NT_);
        }
        __syncthreads();
        if (tv) bk = k < D ? L.b[k] : (R)0;
        for (int s = 0; s < nslot; ++s) {
            const int i = s * 2 * NW2 + 2 * wv + (hi ? 1 : 0);
            const bool pl = i < N;
            const int ir = pl ? i : 0;
            const R* xp = xprev + L.idx[ir] * S;
            // the parent's transition mean, component k (csmc.py:91-92)
            R mu = bk;
            if (!(CW2_ABL & 16)) {
#pragma unroll
                for (int j = 0; j < CSW_MAXD; ++j) mu = fma_(Frow[j], xp[j], mu);  // (columns beyond D are zeros on both sides: fma(0, 0, mu) = mu)
            }
            R xk = 0, pmk = 0;
            if (k < D) {
                if (m.proposal == 0) {
                    R acc = mu;
#pragma unroll
                    for (int j = 0; j < CSW_MAXD; ++j) acc = j <= k ? fma_(Lrow[j], L.eps[ir * S + j], acc) : acc;
                    xk = acc;
                } else {  // AuxiliaryMtDynamics: N(u_t [+ delta_t / 2 grad_t], delta_t / 2 I) (independent.py:192-198)
                    pmk = grad ? fma_(st * st, gk, uk) : uk;
                    xk = fma_(st, L.eps[ir * S + k], pmk);
                }
                if (i == 0) xk = xsk;
            }
            // weights (csmc.py:95-96)
            R g = potential_half<R>(m, k, hi, xk, yk);
            if (m.proposal == 1) g = gauss_half_blk<R>(D, k, xk - mu, Lrow, L.blk, ctr) + g;  // AuxiliaryGt = Mt.logpdf + Gt (independent.py:238-248)
            // GradientAuxiliaryGt (:252-268): summed over the particles in the reference, i.e. a constant of the step (AUXSSM_GRAD_REFERENCE: nothing to add);
            // AUXSSM_GRAD_EXACT applies it per particle
            if (grad && m.gradient == 2) g = g + grad_corr_half<R>(D, k, xk, uk, pmk, st);
            if (pl && k < D) {
                xcur[i * S + k] = xk;
                xs[((long long)t * N + i) * D + k] = xk;
            }
            if (pl && k == 0) {
                L.lwv[i] = g;
                lws[(long long)t * N + i] = g;
            }
        }
        __syncthreads();
        R* tmp = xprev;
        xprev = xcur;
        xcur = tmp;
    }
    if (wv == 0) {  // the weights of the last step: exact maximum
        const bool live = lane < N;
        const R lw = live ? L.lwv[lane] : ninf;
        R mstep;
        const R w = wave_expmax<R>(lw, &mstep);
        if (fmax && lane == 0) fmax[T - 1] = mstep;
        if (live) ((R*)a.wT)[(long long)ch * N + lane] = w;
    }
}

template <typename R, int NW2> __global__ void __launch_bounds__(64 * NW2) k_cw2_bwd(CsmcArgs a, FkW<R> m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT_ = 64 * NW2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, N = a.N, T = a.T, D = m.D;
    const bool hi = lane >= 32;
    const int k = lane & 31;
    Cw2Lds<R> L(smem, D);
    cw2_stage<R>(m, L, tid, NT_);
    const int S = L.S, nslot = (N + 2 * NW2 - 1) / (2 * NW2);
    const int ch = a.c0 + blockIdx.x;
    const R* xs = (const R*)a.xs + (long long)ch * T * N * D;
    const R* lws = (const R*)a.lws + (long long)ch * T * N;
    const int32_t* As = a.As ? a.As + (long long)ch * (T - 1) * N : nullptr;
    R* xout = (R*)a.x + (long long)ch * T * D;
    int32_t* anc = a.anc + (long long)ch * T;
    const R* fmax = (const R*)a.fmax + (long long)ch * T;
    const R ninf = -INFINITY;
    const bool tv = m.Ft != nullptr;
    R bk = k < D ? L.b[k] : (R)0;
    const R* Frow = L.F + (k < D ? k : 0) * S;
    const R* Lrow = L.LQ + (k < D ? k : 0) * S;
    // B_T ~ choice(w_T) by wave 0
    if (wv == 0) {
        const bool live = lane < N;
        const R w = live ? ((const R*)a.wT)[(long long)ch * N + lane] : (R)0;
        const R cv = wave_scan_dpp(w);
        const R tot = readlane_(cv, 63);
        const R un = ((const R*)a.u_bwd)[(long long)ch * T + (T - 1)];
        int B = __popcll(__ballot(live && cv < tot * ((R)1 - un)));
        B = B < N - 1 ? B : N - 1;
        if (lane == 0) L.idx[0] = B, anc[T - 1] = B;
    }
    __syncthreads();
    int B = L.idx[0];
    R xn = k < D ? xs[((long long)(T - 1) * N + B) * D + k] : (R)0;  // x_{t+1}, component k (every half-wave holds a copy)
    if (wv == 0 && !hi && k < D) xout[(long long)(T - 1) * D + k] = xn;
    if (!a.backward) {
        if (tid == 0) {
            for (int t = T - 1; t >= 1; --t) {
                B = As[(long long)(t - 1) * N + B];
                for (int q = 0; q < D; ++q) xout[(long long)(t - 1) * D + q] = xs[((long long)(t - 1) * N + B) * D + q];
                anc[t - 1] = B;
            }
        }
        return;
    }
    constexpr int NSL = 64 / (2 * NW2);  // passes at N = 64
    R xi_nx[NSL], lw_nx[NSL];  // the rows of step t, requested one step ahead (they do not depend on the draws)
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
        const int i = s * 2 * NW2 + 2 * wv + (hi ? 1 : 0), ir = i < N ? i : 0;
        xi_nx[s] = (T >= 2 && k < D) ? xs[((long long)(T - 2) * N + ir) * D + k] : (R)0;
        lw_nx[s] = T >= 2 ? lws[(long long)(T - 2) * N + ir] : (R)0;
    }
    for (int t = T - 2; t >= 0; --t) {
        __syncthreads();  // (the draw of the step before has been read by everybody)
        const R ctr = tv ? m.ctt[t] : m.c_trans;  // the transition t -> t + 1 (time-varying: row t)
        if (tv) {
            cw2_stage_t<R>(m, L, t, tid, NT_);
            __syncthreads();
            bk = k < D ? L.b[k] : (R)0;
        }
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            if (s >= nslot) break;
            const int i = s * 2 * NW2 + 2 * wv + (hi ? 1 : 0);
            const bool pl = i < N;
            const int ir = pl ? i : 0;
            const R xik = xi_nx[s], lwik = lw_nx[s];
            if (t > 0) {
                xi_nx[s] = k < D ? xs[((long long)(t - 1) * N + ir) * D + k] : (R)0;
                lw_nx[s] = lws[(long long)(t - 1) * N + ir];
            }
            if (k < D) L.xa[ir * S + k] = xik;  // (a half-wave reads back only its own row: ordered inside the wave)
            __builtin_amdgcn_wave_barrier();
            const R* xi = L.xa + ir * S;
            R mu = bk;
            if (!(CW2_ABL & 16)) {
#pragma unroll
                for (int j = 0; j < CSW_MAXD; ++j) mu = fma_(Frow[j], xi[j], mu);
            }
            const R lwt = gauss_half_blk<R>(D, k, xn - mu, Lrow, L.blk, ctr) + lwik;  // Pt.logpdf(x_{t+1}, xs_t) + log_ws_t (csmc.py:136)
            if (pl && k == 0) L.lwv[i] = lwt;
        }
        __syncthreads();
        if (wv == 0) {
            const bool live = lane < N;
            const R lw = live ? L.lwv[lane] : ninf;
            R Mb = fmax[t] + ctr;
            if (!(Mb - Mb == 0)) Mb = 0;
            R w = det_exp(lw - Mb);
            R cv = wave_scan_dpp(w);
            R tot = readlane_(cv, 63);
            if (!(tot > (R)0)) {
                w = wave_expmax<R>(lw, nullptr);
                cv = wave_scan_dpp(w);
                tot = readlane_(cv, 63);
            }
            const R un = ((const R*)a.u_bwd)[(long long)ch * T + t];
            int Bn = __popcll(__ballot(live && cv < tot * ((R)1 - un)));
            Bn = Bn < N - 1 ? Bn : N - 1;
            if (lane == 0) L.idx[0] = Bn, anc[t] = Bn;
        }
        __syncthreads();
        B = L.idx[0];
        xn = k < D ? L.xa[B * S + k] : (R)0;
        if (wv == 0 && !hi && k < D) xout[(long long)t * D + k] = xn;
    }
}

// host: the model as one device block [m0 | LP0 | iLP0 | F | b | LQ | iLQ], constants as csmc_dev.h::fill_model computes them
template <typename R> static int run_cw(auxssm_ctx* h, const auxssm_fk_model* fk, CsmcArgs& a, R* host_block, void* ctt) {
    const int D = fk->dx;
    FkW<R> m;
    memset(&m, 0, sizeof(m));
    m.proposal = fk->proposal; m.potential = fk->potential; m.D = D;
    m.gradient = fk->gradient;
    R* p = host_block;
    R* hm0 = p; p += D;
    R* hLP0 = p; p += D * D;
    R* hiLP0 = p; p += D;
    R* hF = p; p += D * D;
    R* hb = p; p += D;
    R* hLQ = p; p += D * D;
    R* hiLQ = p; p += D;
    for (int k = 0; k < D; ++k) hm0[k] = (R)fk->m0[k], hb[k] = (R)fk->b[k];
    for (int i = 0; i < D * D; ++i) hLP0[i] = (R)fk->chol_P0[i], hF[i] = (R)fk->F[i], hLQ[i] = (R)fk->chol_Q[i];
    R ci = 0, ct = 0;
    for (int k = 0; k < D; ++k) {
        ci -= det_log(hLP0[k * D + k]);
        ct -= det_log(hLQ[k * D + k]);
        hiLP0[k] = (R)1 / hLP0[k * D + k];
        hiLQ[k] = (R)1 / hLQ[k * D + k];
    }
    const R half_log_2pi = (R)0.91893853320467274178;
    m.c_init = ci - (R)D * half_log_2pi;
    m.c_trans = ct - (R)D * half_log_2pi;
    if (fk->potential == 1) {
        m.inv_sig_y = (R)1 / (R)fk->sig_y;
        m.c_obs = -(R)D * det_log((R)fk->sig_y) - (R)D * half_log_2pi;
    } else if (fk->potential == 3) {
        m.inv_sig_y = (R)1 / (R)fk->sig_y;
        m.c_obs = -det_log((R)fk->sig_y) - half_log_2pi;
    } else {
        m.inv_sig_y = 0;
        m.c_obs = -half_log_2pi;
    }
    const size_t nb = (size_t)(p - host_block) * sizeof(R);
    {   // the handle's copy of the block: a new upload only when the content differs from the last one (a model that changes between sweeps pays one
        // stream synchronisation -- earlier sweeps may still be reading the old block -- a fixed model none: include/auxssm.h, auxssm_csmc_sweep)
        const int hdr[2] = {(int)sizeof(R), D};
        const bool same = h->cw_dev && h->cw_host.size() == sizeof(hdr) + nb && memcmp(h->cw_host.data(), hdr, sizeof(hdr)) == 0 &&
                          memcmp(h->cw_host.data() + sizeof(hdr), host_block, nb) == 0;
        if (!same) {
            AX_HIP(hipStreamSynchronize(h->stream));
            if (h->cw_dev_bytes < nb) {
                if (h->cw_dev) (void)hipFree(h->cw_dev);
                h->cw_dev = nullptr, h->cw_dev_bytes = 0;
                h->cw_host.clear();
                AX_HIP(hipMalloc(&h->cw_dev, nb));
                h->cw_dev_bytes = nb;
            }
            AX_HIP(hipMemcpy(h->cw_dev, host_block, nb, hipMemcpyHostToDevice));
            h->cw_host.resize(sizeof(hdr) + nb);
            memcpy(h->cw_host.data(), hdr, sizeof(hdr));
            memcpy(h->cw_host.data() + sizeof(hdr), host_block, nb);
        }
    }
    const R* d = (const R*)h->cw_dev;
    m.m0 = d; d += D;
    m.LP0 = d; d += D * D;
    m.iLP0 = d; d += D;
    m.F = d; d += D * D;
    m.b = d; d += D;
    m.LQ = d; d += D * D;
    m.iLQ = d;
    if (fk->F_t && a.T > 1) {  // time-varying transitions: device rows + their constants and reciprocal diagonals (csmc.hip::run_csmc does the same for dx <= 4)
        m.Ft = (const R*)fk->F_t;
        m.bt = (const R*)fk->b_t;
        m.LQt = (const R*)fk->chol_Q_t;
        m.ctt = (const R*)ctt;
        m.idt = (const R*)ctt + (a.T - 1);
        hipLaunchKernelGGL((k_cw_ctrans<R>), dim3((a.T - 1 + 255) / 256), dim3(256), 0, h->stream, a.T - 1, D, m.LQt, (R*)ctt, (R*)ctt + (a.T - 1));
    }
    if (a.gb) hipLaunchKernelGGL((k_cw_potbound<R>), dim3((a.T + 255) / 256), dim3(256), 0, h->stream, a.T, m, (const R*)a.y, (R*)a.gb);
    if (fk->proposal == 1) {
        const long long total = (long long)a.C * a.T * D;
        hipLaunchKernelGGL((k_csmc_aux<R>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, a, D);
        if (fk->gradient) {
            const long long tot = (long long)a.C * a.T;
            hipLaunchKernelGGL((k_cw_grad<R>), dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, h->stream, a, m);
        }
    }
    const size_t lds = Cw2Lds<R>::bytes(D);
    if (lds > 48 * 1024) {
        AX_HIP(hipFuncSetAttribute((const void*)k_cw2_fwd<R, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AX_HIP(hipFuncSetAttribute((const void*)k_cw2_bwd<R, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AX_HIP(hipFuncSetAttribute((const void*)k_cw2_fwd<R, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AX_HIP(hipFuncSetAttribute((const void*)k_cw2_bwd<R, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int cb = a.cb > 0 ? a.cb : a.C;
    // sixteen waves per chain (every particle of N <= 32 in its own half-wave at once: the shortest step) while the chains leave CUs to spare, eight (no idle
    // waves at N = 25, two passes) once there are more chains than CUs
    const bool wide16 = sizeof(R) == 4 && (getenv("AUXSSM_CW_WAVES") ? atoi(getenv("AUXSSM_CW_WAVES")) == 16 : a.C <= h->num_cu);  // (fp64: the unrolled loops need more than the 128 registers of a 1024-lane workgroup)
    for (int c0 = 0; c0 < a.C; c0 += cb) {
        CsmcArgs ab = a;
        ab.c0 = c0;
        ab.C = a.C - c0 < cb ? a.C - c0 : cb;
        ab.xs = (char*)a.xs - (size_t)c0 * a.xs_rec;
        ab.lws = (char*)a.lws - (size_t)c0 * a.lws_rec;
        if (a.As) ab.As = (int32_t*)((char*)a.As - (size_t)c0 * a.As_rec);
        {
            ProfScope ps(h, AUXSSM_K_CSMC_FWD);
            if (wide16) hipLaunchKernelGGL((k_cw2_fwd<R, 16>), dim3(ab.C), dim3(1024), lds, h->stream, ab, m);
            else hipLaunchKernelGGL((k_cw2_fwd<R, 8>), dim3(ab.C), dim3(512), lds, h->stream, ab, m);
        }
        {
            ProfScope ps(h, AUXSSM_K_CSMC_BWD);
            if (wide16) hipLaunchKernelGGL((k_cw2_bwd<R, 16>), dim3(ab.C), dim3(1024), lds, h->stream, ab, m);
            else hipLaunchKernelGGL((k_cw2_bwd<R, 8>), dim3(ab.C), dim3(512), lds, h->stream, ab, m);
        }
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// called by auxssm_csmc_sweep (csmc.hip) for dx > CS_MAXD
int run_csmc_wide(auxssm_ctx* h, int dtype, const auxssm_fk_model* fk, CsmcArgs& a, void* ctt) {
    const int D = fk->dx;
    std::vector<double> host((size_t)3 * D * D + 4 * D + 8);
    if (dtype == AUXSSM_F32) return run_cw<float>(h, fk, a, (float*)host.data(), ctt);
    return run_cw<double>(h, fk, a, host.data(), ctt);
}

}  // namespace ax

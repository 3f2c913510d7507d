// api.hip -- the C ABI of libauxssm.so (include/auxssm.h): handle, workspace, dispatch, the fused
// auxiliary-Kalman sweep and the Threefry fill kernels.
#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include "ctx.h"
#include "rng.h"

namespace ax {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int ws_reserve(auxssm_ctx* h, size_t bytes) {
    h->ws_off = 0;
    if (bytes <= h->ws_bytes) return AUXSSM_OK;
    if (h->ws) {
        AX_HIP(hipStreamSynchronize(h->stream));
        AX_HIP(hipFree(h->ws));
        h->ws = nullptr;
        h->ws_bytes = 0;
    }
    const size_t want = bytes + bytes / 8 + (1u << 20);
    hipError_t e = hipMalloc((void**)&h->ws, want);
    if (e != hipSuccess) {
        set_error("workspace hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return AUXSSM_ERR_NOMEM;
    }
    h->ws_bytes = want;
    return AUXSSM_OK;
}
void* ws_take(auxssm_ctx* h, size_t bytes) {
    const size_t off = (h->ws_off + 255) & ~(size_t)255;
    if (off + bytes > h->ws_bytes) {
        set_error("internal: workspace overrun (%zu + %zu > %zu)", off, bytes, h->ws_bytes);
        return nullptr;
    }
    h->ws_off = off + bytes;
    return h->ws + off;
}

// ---- model stage on the side stream (ctx.h: auxssm_ctx::SideStage) --------------------------------------------------------------------------
int side_open(auxssm_ctx* h, size_t need) {
    auxssm_ctx::SideStage& s = h->side;
    if (s.inside) {
        set_error("internal: side stage opened inside a side scope");
        return AUXSSM_ERR_ARG;
    }
    constexpr int NS = auxssm_ctx::SideStage::NS;
    if (!s.streams[0]) {
        // lowest priority: the stage has whole sweeps of slack, the chain passes it overlaps do not
        int lo = 0, hi = 0;
        static const int prio = [] { const char* e = getenv("AUXSSM_SIDE_PRIO"); return e ? atoi(e) : 1; }();  // 0 default priority, 1 lowest, 2 highest
        const bool prio_on = prio != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
        bool ok = true;
        for (int p = 0; ok && p < NS; ++p) {
            if (!prio_on || hipStreamCreateWithPriority(&s.streams[p], hipStreamNonBlocking, prio == 2 ? hi : lo) != hipSuccess)
                ok = hipStreamCreateWithFlags(&s.streams[p], hipStreamNonBlocking) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&s.done[p], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s.sweep_end[p], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&s.begun[p], hipEventDisableTiming) == hipSuccess;
        }
        ok = ok && hipEventCreateWithFlags(&s.fence, hipEventDisableTiming) == hipSuccess;
        if (!ok) {  // no further streams on this device / runtime: the sweeps stay on the one stream (s.open stays false)
            (void)hipGetLastError();
            for (int p = 0; p < NS; ++p) {
                if (s.streams[p]) (void)hipStreamDestroy(s.streams[p]);
                s.streams[p] = nullptr;
            }
            h->overlap_model_stage = 0;
            return AUXSSM_OK;
        }
    }
    if (s.open) AX_HIP(hipStreamSynchronize(h->stream));  // a sweep that failed half way never marked its end: no reader may be left behind
    s.open = false;
    const int p = (s.parity + 1) % NS;
    if (need > s.bytes[p]) {  // (first sweeps of a shape only) nobody may still use the old slab
        AX_HIP(hipStreamSynchronize(h->stream));
        for (int q = 0; q < NS; ++q) AX_HIP(hipStreamSynchronize(s.streams[q]));
        if (s.ws[p]) AX_HIP(hipFree(s.ws[p]));
        s.ws[p] = nullptr;
        s.bytes[p] = 0;
        const size_t want = need + need / 8 + (1u << 20);
        hipError_t e = hipMalloc((void**)&s.ws[p], want);
        if (e != hipSuccess) {  // no room for another slab: this sweep stays on the one stream
            (void)hipGetLastError();
            s.ws[p] = nullptr;
            return AUXSSM_OK;
        }
        s.bytes[p] = want;
        s.end_valid[p] = false;
        s.memo_use[p] = -1;  // (a new slab holds nobody's tables)
    }
    ++s.uses[p];
    if (s.end_valid[p]) AX_HIP(hipStreamWaitEvent(s.streams[p], s.sweep_end[p], 0));
    // something other than a staged sweep went through the handle since -- or the caller holds the raw stream and may have queued work on it the
    // library never saw (auxssm_stream): that work comes first
    if (h->api_calls != s.last_call + 1 || h->stream_exposed) {
        AX_HIP(hipEventRecord(s.fence, h->stream));
        AX_HIP(hipStreamWaitEvent(s.streams[p], s.fence, 0));
    }
    if (s.begun_valid) AX_HIP(hipStreamWaitEvent(s.streams[p], s.begun[s.parity], 0));  // (s.parity: the previous stage's slab)
    AX_HIP(hipEventRecord(s.begun[p], s.streams[p]));
    s.begun_valid = true;
    s.stream = s.streams[p];
    s.last_call = h->api_calls;
    s.parity = p;
    s.off = 0;
    s.last_tab = nullptr;
    s.open = true;
    return AUXSSM_OK;
}
int side_close(auxssm_ctx* h) {
    auxssm_ctx::SideStage& s = h->side;
    if (!s.open || s.inside) return AUXSSM_OK;
    AX_HIP(hipEventRecord(s.done[s.parity], s.stream));
    AX_HIP(hipStreamWaitEvent(h->stream, s.done[s.parity], 0));
    return AUXSSM_OK;
}
void side_sweep_end(auxssm_ctx* h) {
    auxssm_ctx::SideStage& s = h->side;
    if (!s.open) return;
    (void)hipEventRecord(s.sweep_end[s.parity], h->stream);
    s.end_valid[s.parity] = true;
    s.open = false;
}

// `waves` (chain tile, chunk) lanes per SIMD: a chain's state is a handful of registers and the passes stream their inputs, so more, shorter chunks
// hide more latency until the aggregate pass grows.  Measured at C2 x 256 chains, three runs each, after the noise-drawing reduce pass went from 89 to
// 60 registers: filter scan 0.909 / 0.872 / 0.816 ms at E = 64 / 32 / 24, sampler scan 0.762 / 0.727 / 0.742 -- hence 11 and 8 (kernels.hip.h::AffWaves)
AffPlan plan_aff(const auxssm_ctx* h, int S, int N, int parallel, int waves) {
    AffPlan p;
    if (!parallel || N <= 2) {
        p.E = N > 0 ? N : 1;
        p.nchunk = 1;
        return p;
    }
    const long long stiles = (S + 64 /* TB_CM */ - 1) / 64 /* TB_CM */;
    long long want = (long long)h->num_cu * 4 * waves / stiles;  // chunks
    if (want < 1) want = 1;
    long long E = (N + want - 1) / want;
    if (E < 16) E = 16;
    if (E > 1024) E = 1024;
    if (const char* ev = getenv("AUXSSM_AFF_E")) {  // tuning/debug override
        const long long v = atoll(ev);
        if (v >= 1 && v <= 65536) E = v;
    }
    p.E = (int)E;
    p.nchunk = (int)((N + E - 1) / E);
    return p;
}

ScanPlan plan_scan(const auxssm_ctx* h, int S, int n, int parallel, int waves) {
    ScanPlan p;
    if (!parallel || n <= 2) {
        p.E = n > 0 ? n : 1;
        p.nchunk = 1;
        return p;
    }
    // one wave per SIMD: the fp64 d=4 combine needs ~400 unified registers, so that is the residency anyway, and fewer,
    // longer chunks halve the aggregate-scan work (measured on C2 x 64 chains: E = 64 beats 16/32/48/96)
    const long long target = (long long)h->num_cu * 4 * 64 * (waves < 1 ? 1 : waves > SCAN_WAVES_MAX ? SCAN_WAVES_MAX : waves);
    long long E = ((long long)S * n + target - 1) / target;
    // few sequences (the chip is not full at any E): the lane-serial walk over a chunk dominates, the aggregate scan is cheap -- measured
    // optimum E = 12 at n = 65535 (1 and 8 sequences, fp64 d = 4) and E = 8 at n = 16383 (8 sequences, fp32 d = 3): E ~ sqrt(n / 450)
    long long emin = (long long)(sqrt((double)n / 450.0) + 0.5);
    if (emin < 4) emin = 4;
    if (emin > 32) emin = 32;
    if (E < 32) E = (4 * E + 2) / 3;  // a chip that is only just full of one-chunk lanes: slightly longer chunks win (24 sequences: E = 32 beats 24 by 9 %)
    if (E < emin) E = emin;
    if (E > 512) E = 512;
    if (const char* ev = getenv("AUXSSM_SCAN_E")) {  // tuning/debug override
        const long long v = atoll(ev);
        if (v >= 1 && v <= 4096) E = v;
    }
    p.E = (int)E;
    p.nchunk = (int)((n + E - 1) / E);
    return p;
}

// instantiation units
#define AX_DECL_UNIT(NAME) \
    const KalmanEntry* kalman_unit_##NAME(int P); \
    const SampleEntry* sample_unit_##NAME(); \
    const SweepLogpdfEntry* sweep_logpdf_unit_##NAME(int PO);
AX_DECL_UNIT(f32_d1) AX_DECL_UNIT(f32_d2) AX_DECL_UNIT(f32_d3) AX_DECL_UNIT(f32_d4)
AX_DECL_UNIT(f64_d1) AX_DECL_UNIT(f64_d2) AX_DECL_UNIT(f64_d3) AX_DECL_UNIT(f64_d4)

const KalmanEntry* kalman_entry(int dtype, int D, int P) {
    if (dtype == AUXSSM_F32) {
        switch (D) {
            case 1: return kalman_unit_f32_d1(P);
            case 2: return kalman_unit_f32_d2(P);
            case 3: return kalman_unit_f32_d3(P);
            case 4: return kalman_unit_f32_d4(P);
        }
    } else if (dtype == AUXSSM_F64) {
        switch (D) {
            case 1: return kalman_unit_f64_d1(P);
            case 2: return kalman_unit_f64_d2(P);
            case 3: return kalman_unit_f64_d3(P);
            case 4: return kalman_unit_f64_d4(P);
        }
    }
    return nullptr;
}
const SampleEntry* sample_entry(int dtype, int D) {
    if (dtype == AUXSSM_F32) {
        switch (D) {
            case 1: return sample_unit_f32_d1();
            case 2: return sample_unit_f32_d2();
            case 3: return sample_unit_f32_d3();
            case 4: return sample_unit_f32_d4();
        }
    } else if (dtype == AUXSSM_F64) {
        switch (D) {
            case 1: return sample_unit_f64_d1();
            case 2: return sample_unit_f64_d2();
            case 3: return sample_unit_f64_d3();
            case 4: return sample_unit_f64_d4();
        }
    }
    return nullptr;
}

const SweepLogpdfEntry* sweep_logpdf_entry(int dtype, int D, int PO) {
    if (dtype == AUXSSM_F32) {
        switch (D) {
            case 1: return sweep_logpdf_unit_f32_d1(PO);
            case 2: return sweep_logpdf_unit_f32_d2(PO);
            case 3: return sweep_logpdf_unit_f32_d3(PO);
            case 4: return sweep_logpdf_unit_f32_d4(PO);
        }
    } else if (dtype == AUXSSM_F64) {
        switch (D) {
            case 1: return sweep_logpdf_unit_f64_d1(PO);
            case 2: return sweep_logpdf_unit_f64_d2(PO);
            case 3: return sweep_logpdf_unit_f64_d3(PO);
            case 4: return sweep_logpdf_unit_f64_d4(PO);
        }
    }
    return nullptr;
}

static inline Arr cv(const auxssm_arr& a) { return Arr{a.ptr, (long long)a.sc, (long long)a.st, (long long)a.sb, 1}; }
// AUXSSM_AUX_FLY=0 (debug / comparison runs): materialise the concatenated observations and the scan elements instead
static bool aux_fly_enabled() {
    static const bool on = [] { const char* e = getenv("AUXSSM_AUX_FLY"); return e ? atoi(e) != 0 : true; }();
    return on;
}

static int check_dims(const auxssm_dims* d, bool need_dy) {
    if (!d) {
        set_error("dims is NULL");
        return AUXSSM_ERR_ARG;
    }
    if (d->C < 1 || d->T < 1 || d->B < 1 || d->dx < 1 || (need_dy && d->dy < 1)) {
        set_error("bad dims C=%d T=%d B=%d dx=%d dy=%d (all must be >= 1)", d->C, d->T, d->B, d->dx, d->dy);
        return AUXSSM_ERR_ARG;
    }
    if ((long long)d->C * d->B > (1ll << 24)) {
        set_error("C*B = %lld sequences is beyond the supported 2^24", (long long)d->C * d->B);
        return AUXSSM_ERR_ARG;
    }
    return AUXSSM_OK;
}
static int check_dtype(int dtype) {
    if (dtype != AUXSSM_F32 && dtype != AUXSSM_F64) {
        set_error("dtype must be AUXSSM_F32 (0) or AUXSSM_F64 (1), got %d", dtype);
        return AUXSSM_ERR_ARG;
    }
    return AUXSSM_OK;
}
static int check_lgssm(const auxssm_lgssm* g, int T) {
    if (!g) {
        set_error("lgssm is NULL");
        return AUXSSM_ERR_ARG;
    }
    if (!g->m0.ptr || !g->P0.ptr || !g->Hs.ptr || !g->Rs.ptr || !g->cs.ptr) {
        set_error("lgssm has a NULL m0/P0/Hs/Rs/cs pointer");
        return AUXSSM_ERR_ARG;
    }
    if (T > 1 && (!g->Fs.ptr || !g->Qs.ptr || !g->bs.ptr)) {
        set_error("lgssm has a NULL Fs/Qs/bs pointer with T > 1");
        return AUXSSM_ERR_ARG;
    }
    return AUXSSM_OK;
}
// sizes the register-resident per-lane kernels are instantiated for; everything else runs the wide-state path (wide.hip)
static bool is_wide(int D, int P) { return D > MAX_D || P > MAX_P; }
static const KalmanEntry* need_kalman(int dtype, int D, int P) {
    if (is_wide(D, P)) {
        std::string why;
        if (wide_fits(dtype, D, P, &why)) return wide_kalman_entry(dtype);
        set_error("%s", why.c_str());
        return nullptr;
    }
    const KalmanEntry* e = kalman_entry(dtype, D, P);
    if (!e) set_error("(dx=%d, dy=%d) is not instantiated in this build", D, P);
    return e;
}

static void fill_filter_args(FilterArgs& a, const auxssm_dims* d, const auxssm_lgssm* g, const auxssm_arr* ys, void* ms, void* Ps) {
    a.d = KDims{d->C, d->T, d->B};
    a.m0 = cv(g->m0); a.P0 = cv(g->P0); a.Fs = cv(g->Fs); a.Qs = cv(g->Qs); a.bs = cv(g->bs);
    a.Hs = cv(g->Hs); a.Rs = cv(g->Rs); a.cs = cv(g->cs); a.ys = cv(*ys);
    a.ms = dense_arr(ms, a.d, d->dx); a.Ps = dense_arr(Ps, a.d, (long long)d->dx * d->dx);
    a.elem = nullptr; a.ell0 = nullptr; a.ellz = nullptr;
    a.lay = ScanLayout{1, 1, 1, 1, 0, d->C * d->B};
    a.pblk = 0;
    a.dx = d->dx; a.dy = d->dy;
}
static void fill_logpdf_args(LogpdfArgs& a, const auxssm_dims* d, const auxssm_lgssm* g, const Arr& ys, const Arr& xs, int pol) {
    a.d = KDims{d->C, d->T, d->B};
    a.m0 = cv(g->m0); a.P0 = cv(g->P0); a.Fs = cv(g->Fs); a.Qs = cv(g->Qs); a.bs = cv(g->bs);
    a.Hs = cv(g->Hs); a.Rs = cv(g->Rs); a.cs = cv(g->cs); a.ys = ys; a.xs = xs; a.nan_policy = pol;
    a.dx = d->dx; a.dy = d->dy;
}

template <typename R>
__global__ void k_rng_sweep(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t c0, uint32_t c1, long long n, long long nu, unsigned g1,
                            R* eps_aux, R* eps_samp, R* u_acc);  // (defined below)
// the noise fills of a keyed sweep: the first n entries of eps_aux and eps_samp and the nu uniforms, on the handle's stream
template <typename R> static void launch_rng_sweep(auxssm_ctx* h, const uint32_t* keys, long long n, long long nu, void* eps_aux, void* eps_samp, void* u_acc) {
    const unsigned g1 = (unsigned)(((n + 1) / 2 + 255) / 256), g2 = (unsigned)(((nu + 1) / 2 + 255) / 256);
    ProfScope ps(h, AUXSSM_K_RNG);
    hipLaunchKernelGGL((k_rng_sweep<R>), dim3(2 * g1 + g2), dim3(256), 0, h->stream, keys[0], keys[1], keys[2], keys[3], keys[4], keys[5], n, nu, g1,
                       (R*)eps_aux, (R*)eps_samp, (R*)u_acc);
}

// ---- sweep helper kernels (pure data movement / reductions; runtime sizes) --------------------------------

// device-resident step size: blk = {delta, sqrt(delta / 2)} from the caller's device scalar (auxssm_kalman_sweep_dd)
template <typename R> __global__ void k_delta_block(const R* __restrict__ delta, double* __restrict__ blk) {
    const double d = (double)delta[0];
    blk[0] = d;
    blk[1] = sqrt(0.5 * d);
}

// ---- model-stage memo (ctx.h::SideStage): the stage's input arrays against the snapshot its slab was built from -------------------------------------------------
// up to 9 chain-shared arrays (chain stride 0, batch 1), array q = nt[q] time records of rec[q] reals; the snapshot stores them back to back
struct MemoDesc {
    Arr a[9];
    int nt[9], rec[9];
    long long off[10];  // prefix sums of nt * rec (reals)
    int n;
};
template <typename R> struct MemoBits;
template <> struct MemoBits<float> { using U = uint32_t; };
template <> struct MemoBits<double> { using U = unsigned long long; };
template <typename R> __device__ __forceinline__ const R* memo_src(const MemoDesc& d, long long g) {
    int q = 0;
#pragma unroll
    for (int k = 1; k < 9; ++k) q += (k < d.n && g >= d.off[k]) ? 1 : 0;
    const long long e = g - d.off[q];
    const long long t = e / d.rec[q];
    const int k = (int)(e - t * d.rec[q]);
    return at<R>(d.a[q], 0, t, 0) + (long long)k * d.a[q].se;
}
// rebuild |= (any input element's BIT PATTERN differs from the snapshot): NaNs (missing observations) compare equal to themselves
template <typename R> __global__ void __launch_bounds__(256) k_memo_check(MemoDesc d, const R* __restrict__ snap, int* __restrict__ rebuild) {
    using U = typename MemoBits<R>::U;
    const long long tot = d.off[d.n];
    bool diff = false;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < tot; g += (long long)gridDim.x * 256)
        diff = diff || *reinterpret_cast<const U*>(memo_src<R>(d, g)) != reinterpret_cast<const U*>(snap)[g];
    if (__builtin_amdgcn_ballot_w64(diff) != 0 && (threadIdx.x & 63) == 0) atomicOr(rebuild, 1);
}
template <typename R> __global__ void __launch_bounds__(256) k_memo_snap(MemoDesc d, R* __restrict__ snap, const int* __restrict__ rebuild) {
    if (*rebuild == 0) return;
    const long long tot = d.off[d.n];
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < tot; g += (long long)gridDim.x * 256) snap[g] = *memo_src<R>(d, g);
}
__global__ void k_memo_set(int* rebuild, int v) { *rebuild = v; }

// concatenated observation model of AUXSSM_KMODEL_LG_CONCAT, chain-shared part: H = [I; Hobs], R = blkdiag(d/2 I, Robs), c = [0; cobs]
template <typename R>
__global__ void k_concat_model(int T, int D, int PO, Arr Hobs, Arr Robs, Arr cobs, R half_delta, const double* dptr, R* Hc, R* Rc, R* cc, const int* memo = nullptr) {
    if (memo_skip_p(memo)) return;
    if (dptr) half_delta = (R)(0.5 * dptr[0]);  // device-resident step size (auxssm_kalman_sweep_dd)
    const int P = D + PO;
    const int per_t = P * D + P * P + P;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)T * per_t) return;
    const long long t = g / per_t;
    int r = (int)(g % per_t);
    if (r < P * D) {
        const int k = r / D, j = r % D;
        Hc[t * P * D + r] = k < D ? (k == j ? (R)1 : (R)0) : at<R>(Hobs, 0, t, 0)[(k - D) * D + j];
    } else if ((r -= P * D) < P * P) {
        const int k = r / P, l = r % P;
        R v = 0;
        if (k < D && l < D) v = (k == l) ? half_delta : (R)0;
        else if (k >= D && l >= D) v = at<R>(Robs, 0, t, 0)[(k - D) * PO + (l - D)];
        Rc[t * P * P + r] = v;
    } else {
        r -= P * P;
        cc[t * P + r] = r < D ? (R)0 : at<R>(cobs, 0, t, 0)[r - D];
    }
}
// u = x + sqrt(d/2) eps ; ys_c[c,t,:] = [u ; yobs_t].  x, eps dense (C,T,D); u, ysc through strided views (dense or chain-minor).
// cfast != 0: consecutive lanes = consecutive chains (coalesced writes of the chain-minor buffers).
// x, eps, u, ysc through strided views.  cfast != 0 (chain-minor layout everywhere): consecutive lanes = consecutive chains;
// otherwise lanes run over (t, k) of one chain (dense layout everywhere).  Either way every access of a wave is contiguous.
template <typename R>
__global__ void k_concat_obs(int C, int T, int D, int PO, Arr x, Arr eps, R shd, const double* dptr, Arr yobs, Arr u, Arr ysc, int cfast) {
    if (dptr) shd = (R)dptr[1];
    const int P = D + PO;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)C * T * P) return;
    int c, k;
    long long t;
    if (cfast) {
        c = (int)(g % C);
        const long long r = g / C;
        k = (int)(r % P);
        t = r / P;
    } else {
        k = (int)(g % P);
        const long long ct = g / P;
        c = (int)(ct / T);
        t = ct % T;
    }
    R* yo = const_cast<R*>(at<R>(ysc, c, t, 0)) + (long long)k * ysc.se;
    if (k < D) {
        const R v = at<R>(x, c, t, 0)[(long long)k * x.se] + shd * at<R>(eps, c, t, 0)[(long long)k * eps.se];
        const_cast<R*>(at<R>(u, c, t, 0))[(long long)k * u.se] = v;
        *yo = v;
    } else {
        *yo = at<R>(yobs, c, t, 0)[k - D];
    }
}
// _get_alpha + bernoulli (generic.py:70-73, 98-106).  The five totals arrive in Acc (smallmat.h) and the ratio is formed in Acc: the fp32 sweep
// rounds log alpha once, at the end, instead of differencing six rounded ~1e5-sized totals (fp64: the same arithmetic as before).
template <typename R>
__global__ void k_accept(int C, const Acc* jp_prop, const Acc* jp_rev, const R* ell_prop, const R* ell_rev, const Acc* lt_prop,
                         const Acc* lt_rev, const Acc* corr, const R* u_acc, int32_t* accepted, R* logs) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const Acc lp_prop = jp_prop[c] - (Acc)ell_prop[c];
    const Acc lp_rev = jp_rev[c] - (Acc)ell_rev[c];
    Acc la = lt_prop[c] - lt_rev[c];
    la += lp_rev - lp_prop;
    la -= corr[c];
    const Acc alpha = exp_(la != la ? la : min_(la, (Acc)0));  // jnp.minimum(0, nan) = nan (generic.py:105): a NaN ratio rejects
    accepted[c] = ((Acc)u_acc[c] < alpha) ? 1 : 0;  // NaN alpha -> reject, as jax.random.bernoulli(key, nan)
    if (logs) {
        logs[c * 5 + 0] = (R)la;
        logs[c * 5 + 1] = (R)lp_prop;
        logs[c * 5 + 2] = (R)lp_rev;
        logs[c * 5 + 3] = (R)lt_prop[c];
        logs[c * 5 + 4] = (R)lt_rev[c];
    }
}
// x <- xp for accepted chains, both through strided views; cfast as in k_concat_obs.  With running moments attached
// (auxssm_stats_attach) the same pass folds the sweep into them, the `stats = tree_map(lambda u, v: (i * u + v) / (i + 1), ...)` line of
// the reference's loop body (examples/stochastic_volatility/experiment.py:81-83, :113) -- no contraction, so the fold is the NumPy one
// bit for bit.
template <typename R> AX_HD R fold_mean(R i, R u, R v) {
#pragma clang fp contract(off)
    const R a = i * u;
    const R b = a + v;
    return b / (i + (R)1);
}
template <typename R>
__global__ void k_select(int C, int T, int D, const int32_t* accepted, Arr xp, Arr x, int cfast, R* sq_jump, R* mean, R* sq_mean, long long iter) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)C * T * D) return;
    int c, k;
    long long t;
    if (cfast) {
        c = (int)(g % C);
        const long long r = g / C;
        k = (int)(r % D);
        t = r / D;
    } else {
        k = (int)(g % D);
        const long long ct = g / D;
        c = (int)(ct / T);
        t = ct % T;
    }
    R* px = const_cast<R*>(at<R>(x, c, t, 0)) + (long long)k * x.se;
    const int acc = accepted[c];
    if (!mean) {
        if (acc) *px = at<R>(xp, c, t, 0)[(long long)k * xp.se];
        return;
    }
    const R xo = *px;
    const R xn = acc ? at<R>(xp, c, t, 0)[(long long)k * xp.se] : xo;
    if (acc) *px = xn;
    const long long off = px - (const R*)x.ptr;
    const R i = (R)iter, dj = xn - xo;
    sq_jump[off] = fold_mean<R>(i, sq_jump[off], dj * dj);
    mean[off] = fold_mean<R>(i, mean[off], xn);
    sq_mean[off] = fold_mean<R>(i, sq_mean[off], xn * xn);
}
// The same pass for the two packed layouts, without a division per element (k_select spends ~110 integer instructions per element on 64-bit div / mod
// and is VALU-bound at 4.2 TB/s; this one is a plain masked copy): x and xp are flat arrays with the same strides, `inner` contiguous elements per `outer`
// index -- chain-minor (T, D, C): outer = (t, k), inner = c; dense (C, T, D): outer = c, inner = (t, k).  blockIdx.x = outer, blockIdx.y = block of inner.
template <typename R>
__global__ void __launch_bounds__(256) k_select_rows(int inner, int chain_is_inner, const int32_t* __restrict__ accepted, const R* __restrict__ xp, R* __restrict__ x,
                                                     R* __restrict__ sq_jump, R* __restrict__ mean, R* __restrict__ sq_mean, long long iter) {
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= inner) return;
    const long long off = (long long)blockIdx.x * inner + i;
    const int acc = accepted[chain_is_inner ? i : (int)blockIdx.x];
    if (!mean) {
        if (acc) x[off] = xp[off];
        return;
    }
    const R xo = x[off];
    const R xn = acc ? xp[off] : xo;
    if (acc) x[off] = xn;
    const R it = (R)iter, dj = xn - xo;
    sq_jump[off] = fold_mean<R>(it, sq_jump[off], dj * dj);
    mean[off] = fold_mean<R>(it, mean[off], xn);
    sq_mean[off] = fold_mean<R>(it, sq_mean[off], xn * xn);
}
// the accept/select step of every Kalman sweep.  Running moments attached to the handle are bound to one resident state
// (auxssm_stats_attach: pointer, element count, dtype): a sweep over anything else is refused instead of folding out of bounds.
template <typename R> static int launch_select(auxssm_ctx* h, int C, int T, int D, const int32_t* accepted, Arr xp, Arr x, int cfast) {
    const long long total = (long long)C * T * D;
    if (h->st_mean && (h->st_x != x.ptr || h->st_n != total || h->st_dtype != (sizeof(R) == 4 ? AUXSSM_F32 : AUXSSM_F64))) {
        set_error("running moments are attached to another state (x=%p, n=%lld, dtype=%d): detach them (auxssm_stats_attach with NULLs) "
                  "before sweeping a different state on this handle", h->st_x, h->st_n, h->st_dtype);
        return AUXSSM_ERR_ARG;
    }
    ProfScope ps(h, AUXSSM_K_SELECT);
    // packed buffers of one layout (what the sweeps pass): rows of `inner` contiguous elements
    const bool cm_packed = cfast && x.sc == 1 && xp.sc == 1 && x.se == C && xp.se == C && x.st == (long long)C * D && xp.st == (long long)C * D;
    const bool dense_packed = !cfast && x.se == 1 && xp.se == 1 && x.st == D && xp.st == D && x.sc == (long long)T * D && xp.sc == (long long)T * D;
    const long long inner = cm_packed ? C : (long long)T * D, outer = cm_packed ? (long long)T * D : C;
    if ((cm_packed || dense_packed) && inner >= 64 && outer <= 0x7fffffffLL && (inner + 255) / 256 <= 65535)
        hipLaunchKernelGGL((k_select_rows<R>), dim3((unsigned)outer, (unsigned)((inner + 255) / 256)), dim3(256), 0, h->stream, (int)inner, cm_packed ? 1 : 0,
                           accepted, (const R*)xp.ptr, (R*)x.ptr, (R*)h->st_sq_jump, (R*)h->st_mean, (R*)h->st_sq_mean, h->st_iter);
    else
        hipLaunchKernelGGL((k_select<R>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, C, T, D, accepted, xp, x, cfast,
                           (R*)h->st_sq_jump, (R*)h->st_mean, (R*)h->st_sq_mean, h->st_iter);
    if (h->st_mean) ++h->st_iter;
    return AUXSSM_OK;
}

// the observation pattern of the concatenated model, chain-independent: [0 (D) ; yobs_t] (the auxiliary block is always observed).  Handed to the chain-shared
// wide filter as FilterArgs::mask_ys it needs no read-back of the chains' patterns (no host synchronisation inside a sweep)
template <typename R> __global__ void k_concat_carrier(int T, int D, int P, Arr yobs, R* __restrict__ out) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long long)T * P) return;
    const long long t = g / P;
    const int k = (int)(g % P);
    out[g] = k < D ? (R)0 : at<R>(yobs, 0, t, 0)[k - D];
}

template <typename R>
static int sweep_lg_concat(auxssm_ctx* h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* model, const auxssm_arr* yobs,
                           double delta, const double* dptr, const uint32_t* keys, int parallel, int nan_policy, int layout, void* x, const void* eps_aux, const void* eps_samp,
                           const void* u_acc, int32_t* accepted, void* logs) {
    const int C = dims->C, T = dims->T, D = dims->dx, PO = dims->dy, P = D + PO;
    // one path for the whole sweep: the register kernels when every piece is instantiated, else the wide-state path
    const bool wide = is_wide(D, P) || PO > 4;
    const KalmanEntry* ke = wide ? nullptr : need_kalman(dtype, D, P);
    const SampleEntry* se = wide ? wide_sample_entry(dtype) : sample_entry(dtype, D);
    const SweepLogpdfEntry* sl = wide ? wide_sweep_logpdf_entry(dtype) : sweep_logpdf_entry(dtype, D, PO);
    if (wide) {
        std::string why;
        if (!wide_fits(dtype, D, P, &why)) {
            set_error("%s", why.c_str());
            return AUXSSM_ERR_UNSUPPORTED;
        }
        if (layout != AUXSSM_LAYOUT_DENSE) {
            set_error("(dx=%d, dy=%d) runs the wide-state path, which takes the dense (C, T, dx) layout only", D, P);
            return AUXSSM_ERR_UNSUPPORTED;
        }
        ke = wide_kalman_entry(dtype);
    }
    if (!ke || !se || !sl) return AUXSSM_ERR_UNSUPPORTED;
    const KDims kd{C, T, 1};
    // layout 1: x and the noise are chain-minor (T, dx, C) and so is every internal per-chain buffer: lanes <-> chains
    const int cm = layout == AUXSSM_LAYOUT_CHAIN_MINOR ? 1 : 0;
    const size_t sR = sizeof(R);
    const size_t CT = (size_t)C * T;
    size_t need = 0;
    auto add = [&](size_t b) { need += b + 256; };
    add(CT * P * sR);                                  // ys_c
    add((size_t)T * (P * D + P * P + P) * sR);         // Hc, Rc, cc (three takes)
    add(512);
    add(CT * D * sR);                                  // u
    add(CT * D * sR);                                  // ms
    add(CT * D * D * sR);                              // Ps
    add(CT * D * sR);                                  // x_prop
    add((size_t)C * sR + (size_t)5 * C * sizeof(Acc) + 2048);  // ell, the five totals
    if (wide) add((size_t)T * P * sR);                 // the observation-pattern carrier of the chain-shared wide filter
    add(wide ? wide_filter_ws(h, dtype, kd, parallel, D, P) : ke->filter_ws(h, kd, parallel));
    add(wide ? wide_sample_ws(h, dtype, kd, parallel, D) : se->sample_ws(h, kd, parallel));
    add(wide ? wide_logpdf_ws(dtype, kd) : sl->ws(h, kd));
    int rc = ws_reserve(h, need);
    if (rc) return rc;
    R* ysc = (R*)ws_take(h, CT * P * sR);
    R* Hc = (R*)ws_take(h, (size_t)T * P * D * sR);
    R* Rc = (R*)ws_take(h, (size_t)T * P * P * sR);
    R* cc = (R*)ws_take(h, (size_t)T * P * sR);
    R* u = (R*)ws_take(h, CT * D * sR);
    R* ms = (R*)ws_take(h, CT * D * sR);
    R* Ps = (R*)ws_take(h, CT * D * D * sR);
    R* xp = (R*)ws_take(h, CT * D * sR);
    R* ell = (R*)ws_take(h, C * sR);
    Acc* sums = (Acc*)ws_take(h, (size_t)5 * C * sizeof(Acc));
    if (!ysc || !Hc || !Rc || !cc || !u || !ms || !Ps || !xp || !ell || !sums) return AUXSSM_ERR_NOMEM;
    const size_t mark = h->ws_off;
    // chain-shared parameters: the filtered covariances do not depend on the chain and are stored once, (T, D, D) dense with chain stride 0
    const bool shared_mode = !wide && chain_shared_mode(h, cm, C, T, model->Fs.sc == 0 && model->Qs.sc == 0 && model->bs.sc == 0 && model->P0.sc == 0);
    const bool aux_fly = cm && T > 1 && !wide && aux_fly_enabled();  // u and the concatenated observations of t >= 1 are formed inside the filter
    // a time-invariant real observation model (time stride 0 on Hs, Rs, cs: the broadcast views of a constant model) gives a time-invariant concatenated one: ONE
    // record with time stride 0 instead of T of them (C2 at one chain: 54 MB written and read back per sweep, 27 of its 240 us)
    const bool tinv = model->Hs.st == 0 && model->Rs.st == 0 && model->cs.st == 0;
    const int Tm = tinv ? 1 : T;
    // Chain-shared sweep with a host step size: the MODEL STAGE (concatenated observation model here, matrix filter + gain table in
    // run_filter_shared) reads neither a chain nor anything the previous sweep wrote, so it goes to the side stream with its own double-buffered
    // slab (ctx.h::SideStage) and overlaps the chain passes of the sweep before; its products -- Hc, Rc, cc, the shared covariances, the gain
    // rows -- live in that slab.  AUXSSM_OPT_OVERLAP_MODEL_STAGE = 0 (environment AUXSSM_OVERLAP_TAB=0): everything on the one stream, as before.
    bool overlap = h->overlap_model_stage != 0 && shared_mode && aux_fly && parallel && !dptr;
    struct SweepEnd {  // whatever way the sweep returns: everything it enqueued on `stream` precedes the mark the next stage of this parity waits for
        auxssm_ctx* h;
        ~SweepEnd() { side_sweep_end(h); }
    } sweep_end{h};
    if (overlap) {
        const size_t gain = (size_t)D * D + D + 2 * (size_t)D * P + P + (size_t)P * (P + 1) / 2 + 4, felem = 3 * (size_t)D * D + 2 * D + 8;
        const size_t need_side = (size_t)(T + 64) * sR * ((size_t)P * D + (size_t)P * P + P + (size_t)D * D + P + D + gain + 2 * felem +
                                                          8 * (size_t)D * D + 8 * D + 2 * (size_t)PO * PO + 4 * PO + 32) + (4u << 20);  // + sampler / log-density tables, chunk products
        if ((rc = side_open(h, need_side))) return rc;
        overlap = h->side.open;  // (false when the device gave no second stream)
    }
    if (overlap) {
        SideScope sc(h);
        Hc = (R*)ws_take(h, (size_t)T * P * D * sR);
        Rc = (R*)ws_take(h, (size_t)T * P * P * sR);
        cc = (R*)ws_take(h, (size_t)T * P * sR);
        Ps = (R*)ws_take(h, (size_t)T * D * D * sR);
        if (!Hc || !Rc || !cc || !Ps) return AUXSSM_ERR_NOMEM;
    }
    const Arr yscA = cm ? cm_arr(ysc, kd, P) : dense_arr(ysc, kd, P);
    const Arr uA = cm ? cm_arr(u, kd, D) : dense_arr(u, kd, D);
    const Arr msA = cm ? cm_arr(ms, kd, D) : dense_arr(ms, kd, D);
    // general chain-minor sweep: the filtered covariances are an internal buffer the sampler reads twice -- kept symmetric-packed (10 instead of 16
    // reals at d = 4)
    static const bool ps_pack_on = [] { const char* e = getenv("AUXSSM_PS_PACK"); return e ? atoi(e) != 0 : true; }();
    static const bool samp_fly_on = [] { const char* e = getenv("AUXSSM_SAMPLE_FLY"); return e ? atoi(e) != 0 : true; }();
    const bool ps_pack = ps_pack_on && samp_fly_on && cm && !shared_mode && !wide;
    // wide states, several chains on one model: ONE copy of the filtered covariances (chain stride 0) -- the chain-shared wide filter then skips its broadcast to
    // the chains' slots and the sampler builds its gain / factor tables once (wide.hip::run_sample_shared); the pattern carrier below replaces the filter's read-back
    const bool wide_shared = wide && C >= 2 && h->share_model && model->Fs.sc == 0 && model->Qs.sc == 0 && model->bs.sc == 0 && model->P0.sc == 0 &&
                             model->Hs.sc == 0 && model->Rs.sc == 0 && model->cs.sc == 0 && yobs->sc == 0;
    const Arr PsA = (shared_mode || wide_shared) ? Arr{Ps, 0, (long long)D * D, 0, 1}
                    : cm                         ? cm_arr(Ps, kd, ps_pack ? (long long)symsize(D) : (long long)D * D)
                                                 : dense_arr(Ps, kd, (long long)D * D);
    const Arr xpA = cm ? cm_arr(xp, kd, D) : dense_arr(xp, kd, D);
    const Arr xA = cm ? cm_arr(x, kd, D) : dense_arr(x, kd, D);
    const Arr epsauxA = cm ? cm_arr(eps_aux, kd, D) : dense_arr(eps_aux, kd, D);
    const Arr epsA = cm ? cm_arr(eps_samp, kd, D) : dense_arr(eps_samp, kd, D);

    // observations_factory / dynamics_factory of the LG_CONCAT device model.  With chain-shared parameters in the chain-minor layout
    // the filter builds u and the concatenated observation on the fly for t >= 1 (FilterArgs::aux_*): only row t = 0 is materialised.
    // In the chain-minor layout the filter builds u and the concatenated observation on the fly for t >= 1 (FilterArgs::aux_*), in
    // both of its modes (chain-shared parameters: gain-form recursion; otherwise: elements built inside the scan passes).
    // Keyed sweep (auxssm_kalman_sweep_keyed): the noise is a function of the keys.  Where the chain-shared affine scans run, their reduce
    // passes -- the first readers of eps_aux (t >= 1) and eps_samp -- GENERATE it and store it for the later readers (the fill kernel then
    // only draws row t = 0 and the acceptance uniforms); everywhere else the fill kernel draws all of it first.  Same values either way.
    bool gen = false;
    if (keys) {
        static const bool gen_on = !getenv("AUXSSM_NO_GEN");
        gen = gen_on && aux_fly && shared_mode && parallel && (C % 2 == 0) && plan_aff(h, C, T - 1, 1).nchunk > 1 && plan_aff(h, C, T, 1).nchunk > 1;
        launch_rng_sweep<R>(h, keys, gen ? (long long)D * C : (long long)C * T * D, C, const_cast<void*>(eps_aux), const_cast<void*>(eps_samp), const_cast<void*>(u_acc));
    }
    {
        ProfScope ps(h, AUXSSM_K_FACTORY);
        const long long n1 = (long long)Tm * (P * D + P * P + P);
        {
            SideScope sc(h);  // (model stage)
            hipLaunchKernelGGL((k_concat_model<R>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, h->stream, Tm, D, PO,
                               cv(model->Hs), cv(model->Rs), cv(model->cs), (R)(0.5 * delta), dptr, Hc, Rc, cc);
        }
        const int Tc = aux_fly ? 1 : T;
        const long long n2 = (long long)C * Tc * P;
        hipLaunchKernelGGL((k_concat_obs<R>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, h->stream, C, Tc, D, PO,
                           xA, epsauxA, (R)sqrt(0.5 * delta), dptr, cv(*yobs), uA, yscA, cm);
    }
    auxssm_lgssm gc = *model;
    gc.Hs = auxssm_arr{Hc, 0, tinv ? 0 : (int64_t)P * D, 0};
    gc.Rs = auxssm_arr{Rc, 0, tinv ? 0 : (int64_t)P * P, 0};
    gc.cs = auxssm_arr{cc, 0, tinv ? 0 : (int64_t)P, 0};
    const auxssm_arr ysc_dummy{ysc, (int64_t)T * P, (int64_t)P, 0};
    auxssm_dims dc = *dims;
    dc.dy = P;
    dc.B = 1;

    // proposal LGSSM: filter + pathwise sample (generic.py:80-86).  The factories of this model do not depend on the
    // linearisation point, so the reverse LGSSM (generic.py:67) is the same one and its filter pass is not repeated.
    FilterArgs fa;
    fill_filter_args(fa, &dc, &gc, &ysc_dummy, ms, Ps);
    fa.ys = yscA;
    fa.ms = msA;
    fa.Ps = PsA;
    fa.lay.cm = cm;
    fa.pblk = D;  // R = blkdiag(delta/2 I_d, Robs) by construction
    fa.ps_packed = ps_pack ? 1 : 0;
    if (aux_fly) {
        fa.aux_on = 1;
        fa.aux_x = xA;
        fa.aux_eps = epsauxA;
        fa.aux_u = uA;
        fa.aux_yobs = cv(*yobs);
        fa.aux_shd = sqrt(0.5 * delta);
        fa.dptr = dptr;
        if (gen) fa.aux_gen = 1, fa.gen_k0 = keys[0], fa.gen_k1 = keys[1];
    }
    if (wide_shared) {
        R* carrier = (R*)ws_take(h, (size_t)T * P * sR);
        if (!carrier) return AUXSSM_ERR_NOMEM;
        const long long tot = (long long)T * P;
        hipLaunchKernelGGL((k_concat_carrier<R>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, T, D, P, cv(*yobs), carrier);
        fa.mask_ys = Arr{carrier, 0, (long long)P, 0, 1};
    }
    rc = ke->filter(h, fa, parallel, ell);
    if (rc) return rc;
    h->ws_off = mark;
    SampleArgs sa;
    sa.d = kd;
    sa.dx = D;
    sa.Fs = cv(model->Fs); sa.Qs = cv(model->Qs); sa.bs = cv(model->bs);
    sa.ms = msA; sa.Ps = PsA; sa.eps = epsA; sa.xs = xpA; sa.elem = nullptr;
    sa.lay = ScanLayout{1, 1, 1, 1, cm, C};
    // the filtered covariances of this model do not depend on the chain when its parameters do not
    sa.ps_shared = shared_mode ? 1 : 0;
    sa.ps_packed = ps_pack ? 1 : 0;
    if (gen) sa.eps_gen = 1, sa.gen_k0 = keys[2], sa.gen_k1 = keys[3];
    rc = se->sample(h, sa, parallel);
    if (rc) return rc;
    h->ws_off = mark;

    // all proposal / target log-densities and the MH correction in one pass (generic.py:88-89, :103-105)
    {
        SweepLogpdfArgs la;
        la.d = kd;
        la.dx = D; la.po = PO;
        la.m0 = cv(model->m0); la.P0 = cv(model->P0); la.Fs = cv(model->Fs); la.Qs = cv(model->Qs); la.bs = cv(model->bs);
        la.Hs = cv(model->Hs); la.Rs = cv(model->Rs); la.cs = cv(model->cs); la.ys = cv(*yobs);
        la.x = xA; la.xp = xpA; la.u = uA; la.delta = delta; la.nan_policy = nan_policy;
        la.u_fly = aux_fly ? 1 : 0; la.eps_aux = epsauxA; la.shd = sqrt(0.5 * delta); la.dptr = dptr;
        rc = sl->run(h, la, sums);
        if (rc) return rc;
        h->ws_off = mark;
    }
    const Acc* jp_prop = sums; const Acc* jp_rev = sums + C; const Acc* lt_prop = sums + 2 * C; const Acc* lt_rev = sums + 3 * C; const Acc* corr = sums + 4 * C;
    hipLaunchKernelGGL((k_accept<R>), dim3((C + 127) / 128), dim3(128), 0, h->stream, C, jp_prop, jp_rev, (const R*)ell, (const R*)ell, lt_prop, lt_rev,
                       corr, (const R*)u_acc, accepted, (R*)logs);
    if ((rc = launch_select<R>(h, C, T, D, (const int32_t*)accepted, xpA, xA, cm))) return rc;
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// ---- the chain-shared LG_CONCAT sweep in three streaming passes (fused_shared.h) -----------------------------------------------------------
// row t = 0 of u and of the concatenated observation [u_0 ; yobs_0] from the chain's own buffer (lazy state: x of chain c lives in xa or xb);
// eps0 (D, C) = row 0 of eps_aux.  One lane per (component, chain).
template <typename R>
__global__ void k_fs_concat0(int C, int D, int PO, const R* __restrict__ xa, const R* __restrict__ xb, const int32_t* __restrict__ sel, const R* __restrict__ eps0, R shd,
                             const double* dptr, Arr yobs, R* __restrict__ u, R* __restrict__ ysc0) {
    if (dptr) shd = (R)dptr[1];
    const int P = D + PO;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= C * P) return;
    const int c = g % C, k = g / C;
    if (k < D) {
        const R* x = (sel && sel[c]) ? xb : xa;
        const R v = x[(long long)k * C + c] + shd * eps0[(long long)k * C + c];
        u[(long long)k * C + c] = v;
        ysc0[(long long)k * C + c] = v;
    } else {
        ysc0[(long long)k * C + c] = at<R>(yobs, 0, 0, 0)[k - D];
    }
}
// chain c's state gathered into xa: xa[:, :, c] <- xb[:, :, c] where sel[c]; rows of C contiguous chains (blockIdx.x = (t, k), blockIdx.y = block of chains)
template <typename R> __global__ void __launch_bounds__(256) k_fs_resolve(int C, const int32_t* __restrict__ sel, const R* __restrict__ xb, R* __restrict__ xa) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const long long off = (long long)blockIdx.x * C + c;
    if (sel[c]) xa[off] = xb[off];
}
// running moments of a LAZY state (auxssm_stats_attach; the fold launch_select does for a plain state): after the accept step chain c's new trajectory
// lives in buffer sel[c], its previous one in the other buffer if the proposal was accepted (else they coincide: no jump).  Rows of C chains.
template <typename R>
__global__ void __launch_bounds__(256) k_fs_stats(int C, const int32_t* __restrict__ accepted, const int32_t* __restrict__ sel, const R* __restrict__ xa, const R* __restrict__ xb,
                                                   R* __restrict__ sq_jump, R* __restrict__ mean, R* __restrict__ sq_mean, long long iter) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const long long off = (long long)blockIdx.x * C + c;
    const int s = sel[c], acc = accepted[c];
    const R xn = (s ? xb : xa)[off];
    const R xo = acc ? (s ? xa : xb)[off] : xn;
    const R it = (R)iter, dj = xn - xo;
    sq_jump[off] = fold_mean<R>(it, sq_jump[off], dj * dj);
    mean[off] = fold_mean<R>(it, mean[off], xn);
    sq_mean[off] = fold_mean<R>(it, sq_mean[off], xn * xn);
}
// whether (and why not) a sweep can run the fused passes
static const char* fused_refusal(const auxssm_ctx* h, const auxssm_dims* dims, const auxssm_lgssm* model, int parallel, int layout) {
    const int C = dims->C, T = dims->T, D = dims->dx, PO = dims->dy;
    static const bool off = [] { const char* e = getenv("AUXSSM_FUSED"); return e && atoi(e) == 0; }();
    if (off) return "AUXSSM_FUSED=0";
    if (layout != AUXSSM_LAYOUT_CHAIN_MINOR) return "the fused sweep takes the chain-minor layout";
    if (D > MAX_D || PO < 1 || PO > 4) return "the fused sweep is instantiated for dx <= 4, 1 <= dy <= 4";
    if (!parallel) return "the fused sweep is the parallel-in-time one";
    if (!h->share_model) return "AUXSSM_OPT_SHARE_MODEL is off";
    if (!(model->Fs.sc == 0 && model->Qs.sc == 0 && model->bs.sc == 0 && model->P0.sc == 0 && model->m0.sc == 0))
        return "the fused sweep needs chain-shared model parameters (chain stride 0)";
    if (C < 2 || (C % 2) != 0) return "the fused sweep pairs chains for its in-kernel draws: the chain count must be even";
    if (T < 64) return "the fused sweep needs T >= 64";
    return nullptr;
}
template <typename R>
static int sweep_lg_concat_fused(auxssm_ctx* h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* model, const auxssm_arr* yobs, double delta, const double* dptr,
                                 const uint32_t* keys, int nan_policy, void* x, void* x_alt, int32_t* sel, void* u_acc, int32_t* accepted, void* logs) {
    const int C = dims->C, T = dims->T, D = dims->dx, PO = dims->dy, P = D + PO;
    const SweepLogpdfEntry* sl = sweep_logpdf_entry(dtype, D, PO);
    if (!sl || !sl->fused) {
        set_error("(dx=%d, dy=%d) has no fused sweep in this build", D, PO);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    const KDims kd{C, T, 1};
    const size_t sR = sizeof(R), CT = (size_t)C * T;
    size_t need = 0;
    auto add = [&](size_t b) { need += b + 256; };
    add(CT * D * sR);                                   // u
    add(CT * D * sR);                                   // inc
    add((size_t)T * (P * D + P * P + P + D * D) * sR);  // Hc, Rc, cc, Ps (four takes; in the side slab when the stage overlaps)
    add(1024);
    add((size_t)C * (P + 3 * D) * sR + 1024);           // ysc0, m0p, eps0a, eps0s
    add(sl->fused_ws(h, kd));
    int rc = ws_reserve(h, need);
    if (rc) return rc;
    R* u = (R*)ws_take(h, CT * D * sR);
    R* inc = (R*)ws_take(h, CT * D * sR);
    R* ysc0 = (R*)ws_take(h, (size_t)C * P * sR);
    R* m0p = (R*)ws_take(h, (size_t)C * D * sR);
    R* eps0a = (R*)ws_take(h, (size_t)C * D * sR);
    R* eps0s = (R*)ws_take(h, (size_t)C * D * sR);
    if (!u || !inc || !ysc0 || !m0p || !eps0a || !eps0s) return AUXSSM_ERR_NOMEM;
    bool overlap = h->overlap_model_stage != 0 && !dptr;
    struct SweepEnd {
        auxssm_ctx* h;
        ~SweepEnd() { side_sweep_end(h); }
    } sweep_end{h};
    if (overlap) {
        const size_t need_side = sl->fused_ws(h, kd) + (size_t)(T + 64) * sR * ((size_t)P * D + (size_t)P * P + P + (size_t)D * D) + (4u << 20) +
                                 (size_t)(T + 2) * sR * (3 * (size_t)D * D + 2 * D + (size_t)PO * D + (size_t)PO * PO + 2 * PO);  // (+ the memo's snapshot of the inputs)
        if ((rc = side_open(h, need_side))) return rc;
        overlap = h->side.open;
    }
    R *Hc, *Rc, *cc, *Ps;
    const bool tinv = model->Hs.st == 0 && model->Rs.st == 0 && model->cs.st == 0;  // time-invariant observation model: one concatenated record (sweep_lg_concat)
    const int Tm = tinv ? 1 : T;
    const int* memo = nullptr;
    struct MemoGuard {  // a sweep that fails after claiming its slab's tables leaves nobody's tables behind
        auxssm_ctx* h;
        int p = -1;
        bool ok = false;
        ~MemoGuard() { if (p >= 0 && !ok) h->side.memo_use[p] = -1; }
    } memo_guard{h};
    if (overlap) {
        // MODEL-STAGE MEMO (ctx.h::SideStage): the head of the slab holds the `rebuild` word and a snapshot of the stage's inputs
        static const bool memo_on = [] { const char* e = getenv("AUXSSM_STAGE_MEMO"); return !(e && atoi(e) == 0); }();
        SideScope sc(h);
        auxssm_ctx::SideStage& sd = h->side;
        const int p = sd.parity;
        MemoDesc md{};
        const auxssm_arr* srcs[9] = {&model->m0, &model->P0, &model->Fs, &model->Qs, &model->bs, &model->Hs, &model->Rs, &model->cs, yobs};
        const int recs[9] = {D, D * D, D * D, D * D, D, PO * D, PO * PO, PO, PO};
        const int nts[9] = {1, 1, T - 1, T - 1, T - 1, T, T, T, T};
        md.n = 9;
        md.off[0] = 0;
        bool plain = memo_on;
        for (int q = 0; q < 9; ++q) {
            md.a[q] = cv(*srcs[q]);
            md.rec[q] = recs[q];
            md.nt[q] = md.a[q].st == 0 ? 1 : nts[q];
            md.off[q + 1] = md.off[q] + (long long)md.nt[q] * md.rec[q];
            plain = plain && md.a[q].sc == 0;  // (chain-shared arrays only: what the fused sweep accepts anyway)
        }
        int* rebuild = (int*)ws_take(h, 256);
        R* snap = (R*)ws_take(h, (size_t)md.off[9] * sR + 256);
        if (!rebuild || !snap) return AUXSSM_ERR_NOMEM;
        if (plain) {
            // host key: everything the tables depend on besides the arrays' contents
            std::vector<unsigned char> key;
            auto put = [&](const void* v, size_t nb) { key.insert(key.end(), (const unsigned char*)v, (const unsigned char*)v + nb); };
            const int hdr[8] = {(int)sR, C, T, D, PO, nan_policy, 0, 0};  // (the chunk length is a function of C, T and the process's environment)
            put(hdr, sizeof(hdr));
            put(&delta, sizeof(delta));
            for (int q = 0; q < 9; ++q) put(srcs[q], sizeof(auxssm_arr));
            const bool chain_ok = sd.memo_use[p] == sd.uses[p] - 1 && sd.memo_key[p] == key;
            const unsigned gmemo = (unsigned)std::min<long long>(1024, (md.off[9] + 255) / 256);
            if (chain_ok) {  // same key, and this slab's previous stage was a memoised one: compare the inputs with the snapshot on the device
                hipLaunchKernelGGL(k_memo_set, dim3(1), dim3(1), 0, h->stream, rebuild, 0);
                hipLaunchKernelGGL((k_memo_check<R>), dim3(gmemo), dim3(256), 0, h->stream, md, (const R*)snap, rebuild);
            } else {
                hipLaunchKernelGGL(k_memo_set, dim3(1), dim3(1), 0, h->stream, rebuild, 1);
            }
            hipLaunchKernelGGL((k_memo_snap<R>), dim3(gmemo), dim3(256), 0, h->stream, md, snap, (const int*)rebuild);
            sd.memo_key[p] = std::move(key);
            sd.memo_use[p] = sd.uses[p];
            memo_guard.p = p;
            memo = rebuild;
        }
    }
    {
        SideScope sc(h);  // (model stage: the side slab when a stage is open, else the main one)
        Hc = (R*)ws_take(h, (size_t)T * P * D * sR);
        Rc = (R*)ws_take(h, (size_t)T * P * P * sR);
        cc = (R*)ws_take(h, (size_t)T * P * sR);
        Ps = (R*)ws_take(h, (size_t)T * D * D * sR);
        if (!Hc || !Rc || !cc || !Ps) return AUXSSM_ERR_NOMEM;
        ProfScope ps(h, AUXSSM_K_FACTORY);
        const long long n1 = (long long)Tm * (P * D + P * P + P);
        hipLaunchKernelGGL((k_concat_model<R>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, h->stream, Tm, D, PO, cv(model->Hs), cv(model->Rs), cv(model->cs),
                           (R)(0.5 * delta), dptr, Hc, Rc, cc, memo);
    }
    // row 0 of the two normal draws and the acceptance uniforms (the rest is drawn inside passes A and C), then u_0 and [u_0 ; yobs_0]
    launch_rng_sweep<R>(h, keys, (long long)D * C, C, eps0a, eps0s, u_acc);
    hipLaunchKernelGGL((k_fs_concat0<R>), dim3((unsigned)((C * P + 255) / 256)), dim3(256), 0, h->stream, C, D, PO, (const R*)x, (const R*)x_alt, (const int32_t*)sel,
                       (const R*)eps0a, (R)sqrt(0.5 * delta), dptr, cv(*yobs), u, ysc0);
    FusedHost f{};
    auxssm_lgssm gc = *model;
    gc.Hs = auxssm_arr{Hc, 0, tinv ? 0 : (int64_t)P * D, 0};
    gc.Rs = auxssm_arr{Rc, 0, tinv ? 0 : (int64_t)P * P, 0};
    gc.cs = auxssm_arr{cc, 0, tinv ? 0 : (int64_t)P, 0};
    const auxssm_arr ysc_dummy{ysc0, (int64_t)P, (int64_t)P, 0};
    auxssm_dims dc = *dims;
    dc.dy = P;
    dc.B = 1;
    fill_filter_args(f.fa, &dc, &gc, &ysc_dummy, m0p, Ps);
    const KDims k1{C, 1, 1};
    f.fa.ys = cm_arr(ysc0, k1, P);
    f.fa.ms = cm_arr(m0p, k1, D);
    f.fa.Ps = Arr{Ps, 0, (long long)D * D, 0, 1};
    f.fa.lay.cm = 1;
    f.fa.pblk = D;
    f.fa.aux_on = 1;
    f.fa.aux_yobs = cv(*yobs);
    f.fa.aux_shd = sqrt(0.5 * delta);
    f.fa.dptr = dptr;
    f.sa.d = kd;
    f.sa.dx = D;
    f.sa.Fs = cv(model->Fs); f.sa.Qs = cv(model->Qs); f.sa.bs = cv(model->bs);
    f.sa.Ps = f.fa.Ps;
    f.sa.ps_shared = 1;
    f.sa.elem = nullptr;
    f.sa.lay = ScanLayout{1, 1, 1, 1, 1, C};
    SweepLogpdfArgs& la = f.la;
    la.d = kd;
    la.dx = D; la.po = PO;
    la.m0 = cv(model->m0); la.P0 = cv(model->P0); la.Fs = cv(model->Fs); la.Qs = cv(model->Qs); la.bs = cv(model->bs);
    la.Hs = cv(model->Hs); la.Rs = cv(model->Rs); la.cs = cv(model->cs); la.ys = cv(*yobs);
    la.x = cm_arr(x, kd, D); la.xp = cm_arr(x_alt, kd, D); la.u = cm_arr(u, kd, D);
    la.delta = delta; la.nan_policy = nan_policy; la.u_fly = 0; la.shd = sqrt(0.5 * delta); la.dptr = dptr;
    f.memo = memo;
    f.xa = x; f.xb = x_alt; f.sel = sel; f.u = u; f.inc = inc; f.keys = keys; f.eps0s = eps0s; f.u_acc = u_acc; f.accepted = accepted; f.logs = logs;
    if (h->st_mean && (h->st_x != x || h->st_n != (long long)C * T * D || h->st_dtype != (sizeof(R) == 4 ? AUXSSM_F32 : AUXSSM_F64))) {
        set_error("running moments are attached to another state (x=%p, n=%lld, dtype=%d): detach them (auxssm_stats_attach with NULLs) "
                  "before sweeping a different state on this handle", h->st_x, h->st_n, h->st_dtype);
        return AUXSSM_ERR_ARG;
    }
    if ((rc = sl->fused(h, f))) return rc;
    if (!sel) {  // plain state: x' sits in x_alt, the usual select moves the accepted chains (and folds attached moments)
        if ((rc = launch_select<R>(h, C, T, D, (const int32_t*)accepted, cm_arr(x_alt, kd, D), cm_arr(x, kd, D), 1))) return rc;
    } else if (h->st_mean) {  // lazy state: the moments' fold as its own pass over the pair of buffers
        ProfScope ps(h, AUXSSM_K_SELECT);
        const long long rows = (long long)T * D;
        hipLaunchKernelGGL((k_fs_stats<R>), dim3((unsigned)rows, (unsigned)((C + 255) / 256)), dim3(256), 0, h->stream, C, (const int32_t*)accepted, (const int32_t*)sel,
                           (const R*)x, (const R*)x_alt, (R*)h->st_sq_jump, (R*)h->st_mean, (R*)h->st_sq_mean, h->st_iter);
        ++h->st_iter;
    }
    AX_HIP(hipGetLastError());
    memo_guard.ok = true;
    return AUXSSM_OK;
}

// ---- stochastic-volatility device factories (examples/stochastic_volatility/auxiliary_kalman.py:22-48, model.py:56-82) ----------
// potential log g_t(x) = sum_k log N(y_k; 0, exp(x_k)); grad_k = (y_k^2 e^{-x_k} - 1) / 2, hess_kk = -y_k^2 e^{-x_k} / 2.
//   first order  (:28-35): ys = u + delta/2 grad(x_lin),                      H = I, R = delta/2 I,  c = 0
//   second order (:37-46): Om = (-hess + 2/delta I)^-1 (diagonal), ys = Om (2u/delta + grad - hess x_lin), H = I, R = Om, c = 0
template <typename R> AX_HD R sv_nan_to_num(R v) { return nan_to_num<R>(v); }
// pass 1 (eps != null): u = x + sqrt(delta/2) eps and the observations linearised at x; pass 2: linearised at xlin, u given
template <typename R>
__global__ void k_sv_obs(long long total, int C, int T, int D, int order, int cfast, const R* __restrict__ xlin, const R* __restrict__ eps, R shd,
                         R delta, const double* dptr, Arr yobs, R* __restrict__ u, R* __restrict__ ys, R* __restrict__ Rs) {
    if (dptr) delta = (R)dptr[0], shd = (R)dptr[1];
    // flat index g walks the (C, T, D) arrays in storage order, dense (c, t, k) or chain-minor (t, k, c)
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    int k, c;
    long long t;
    if (cfast) {
        c = (int)(g % C);
        const long long r = g / C;
        k = (int)(r % D);
        t = r / D;
    } else {
        k = (int)(g % D);
        const long long ct = g / D;
        t = ct % T;
        c = (int)(ct / T);
    }
    const R x = xlin[g];
    R uu;
    if (eps) {
        uu = x + shd * eps[g];
        u[g] = uu;
    } else {
        uu = u[g];
    }
    const R y = at<R>(yobs, 0, t, 0)[k];
    const R w = y * y * exp_(-x);
    const R grad = nan_to_num<R>((R)0.5 * (w - (R)1));
    if (order == 1) {
        ys[g] = uu + (R)0.5 * delta * grad;
    } else {
        const R hess = (R)-0.5 * w;
        const R om = (R)1 / (-hess + (R)2 / delta);
        ys[g] = om * ((R)2 * uu / delta + grad - hess * x);
        // row k of the diagonal D x D record of (c, t): dense [c][t][k][j], chain-minor [t][k][j][c]
        const long long base = cfast ? ((t * D + k) * D) * (long long)C + c : (((long long)c * T + t) * D + k) * D;
        const long long js = cfast ? C : 1;
        for (int j = 0; j < D; ++j) Rs[base + j * js] = j == k ? om : (R)0;
    }
}
template <typename R> __global__ void k_scaled_eye(int D, R v, R* eye, const double* half_of = nullptr) {
    if (half_of) v = (R)(0.5 * half_of[0]);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D * D) eye[i] = (i / D == i % D) ? v : (R)0;
}
template <typename R> __global__ void k_fill(long long n, R v, R* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}
// per chain (one workgroup): pot(xp), pot(x), la1 = sum_t log N(ys1_t; xp_t, R1_t), la2 = sum_t log N(ys2_t; x_t, R2_t) (R diagonal;
// a step whose term is NaN is dropped, as the reference's nansum over time steps does), corr (generic.py:103-105).  out [5][C].
template <typename R>
__global__ void __launch_bounds__(256) k_sv_terms(int C, int T, int D, R delta, const double* dptr, const R* __restrict__ x, const R* __restrict__ xp, const R* __restrict__ u,
                                                  Arr yobs, const R* __restrict__ ys1, const R* __restrict__ ys2, const R* __restrict__ R1,
                                                  const R* __restrict__ R2, R* __restrict__ out) {
    if (dptr) delta = (R)dptr[0];
    __shared__ R sh[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    R acc[5] = {0, 0, 0, 0, 0};
    for (long long t = tid; t < T; t += 256) {
        R pp = 0, px = 0, l1 = 0, l2 = 0, cr = 0;
        for (int k = 0; k < D; ++k) {
            const long long g = ((long long)c * T + t) * D + k;
            const R y = at<R>(yobs, 0, t, 0)[k];
            const R a = xp[g], b = x[g], uu = u[g];
            pp += nan_to_num<R>((R)(-0.5 * LOG_2PI) - (R)0.5 * a - (R)0.5 * y * y * exp_(-a));
            px += nan_to_num<R>((R)(-0.5 * LOG_2PI) - (R)0.5 * b - (R)0.5 * y * y * exp_(-b));
            const R r1 = R1 ? R1[g * D + k] : (R)0.5 * delta, r2 = R2 ? R2[g * D + k] : (R)0.5 * delta;
            const R s1 = sqrt_(r1), s2 = sqrt_(r2);
            const R z1 = (ys1[g] - a) / s1, z2 = (ys2[g] - b) / s2;
            l1 += (R)-0.5 * z1 * z1 - log_(s1) - (R)(0.5 * LOG_2PI);
            l2 += (R)-0.5 * z2 * z2 - log_(s2) - (R)(0.5 * LOG_2PI);
            const R e1 = a - uu, e2 = b - uu;
            cr += (e1 * e1 - e2 * e2) / delta;
        }
        acc[0] += pp;
        acc[1] += px;
        acc[2] += isnan_(l1) ? (R)0 : l1;
        acc[3] += isnan_(l2) ? (R)0 : l2;
        acc[4] += cr;
    }
    for (int q = 0; q < 5; ++q) {
        sh[tid] = acc[q];
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) sh[tid] += sh[tid + off];
            __syncthreads();
        }
        if (tid == 0) out[(long long)q * C + c] = sh[0];
        __syncthreads();
    }
}
// out1[c] = (R) in[c], out2[c] = (R) in[C + c]
template <typename R> __global__ void k_acc_to_real(int C, const Acc* __restrict__ in, R* __restrict__ out1, R* __restrict__ out2) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < C) out1[g] = (R)in[g];
    else if (g < 2 * C) out2[g - C] = (R)in[g];
}
// lt = joint - la + pot (target = prior + potential; joint = auxiliary log-likelihood + prior), then _get_alpha + bernoulli
template <typename R>
__global__ void k_sv_accept(int C, const R* j1, const R* j2, const R* ell1, const R* ell2, const R* terms, const R* u_acc, int32_t* accepted, R* logs) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const R pot_p = terms[c], pot_x = terms[C + c], la1 = terms[2 * C + c], la2 = terms[3 * C + c], corr = terms[4 * C + c];
    const R lp_prop = j1[c] - ell1[c], lp_rev = j2[c] - ell2[c];
    const R lt_prop = (j1[c] - la1) + pot_p, lt_rev = (j2[c] - la2) + pot_x;
    R la = lt_prop - lt_rev;
    la += lp_rev - lp_prop;
    la -= corr;
    const R alpha = exp_(la != la ? la : min_(la, (R)0));  // jnp.minimum(0, nan) = nan (generic.py:105): a NaN ratio rejects
    accepted[c] = (u_acc[c] < alpha) ? 1 : 0;
    if (logs) {
        logs[c * 5 + 0] = la;
        logs[c * 5 + 1] = lp_prop;
        logs[c * 5 + 2] = lp_rev;
        logs[c * 5 + 3] = lt_prop;
        logs[c * 5 + 4] = lt_rev;
    }
}

// kernel(key, state, delta) of kalman/generic.py:53-76 with the SV factories: both linearisation points (x for the proposal,
// x_prop for the reverse move) get their own observation set and filter pass, as in the reference.
template <typename R>
static int sweep_sv(auxssm_ctx* h, int dtype, int order, const auxssm_dims* dims, const auxssm_lgssm* model, const auxssm_arr* yobs,
                    double delta, const double* dptr, int parallel, int nan_policy, int layout, void* x, const void* eps_aux, const void* eps_samp,
                    const void* u_acc, int32_t* accepted, void* logs) {
    const int C = dims->C, T = dims->T, D = dims->dx;
    const bool wide = is_wide(D, D);
    const KalmanEntry* ke = need_kalman(dtype, D, D);
    const SampleEntry* se = wide ? wide_sample_entry(dtype) : sample_entry(dtype, D);
    if (!ke || !se) return AUXSSM_ERR_UNSUPPORTED;
    // layout 1 (register kernels only): state, noise and every internal per-chain buffer chain-minor, lanes <-> chains
    const int cm = layout == AUXSSM_LAYOUT_CHAIN_MINOR ? 1 : 0;
    if (cm && wide) {
        set_error("dx=%d runs the wide-state path, which takes the dense (C, T, dx) layout only", D);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    const KDims kd{C, T, 1};
    const size_t sR = sizeof(R), CT = (size_t)C * T;
    const bool second = order == 2;
    // Per-chain observation model in the chain-minor layout (second order, or first order without chain-shared dynamics): no observation arrays -- the
    // scan passes and the log-density pass re-form the pseudo-observations from (x, u, y) (FilterArgs::sv_order; AUXSSM_SV_FLY=0: the array path)
    static const bool sv_fly_on = [] { const char* e = getenv("AUXSSM_SV_FLY"); return e ? atoi(e) != 0 : true; }();
    const bool fly = sv_fly_on && cm && !wide && T > 1 &&
                     !(!second && chain_shared_mode(h, cm, C, T, model->Fs.sc == 0 && model->Qs.sc == 0 && model->bs.sc == 0 && model->P0.sc == 0));
    size_t need = 0;
    auto add = [&](size_t b) { need += b + 256; };
    for (int q = 0; q < (fly ? 2 : 4); ++q) add(CT * D * sR);   // u, x_prop (, ys1, ys2)
    if (second && !fly) add(2 * CT * D * D * sR);              // Rs1, Rs2
    add(CT * D * sR);                                        // ms
    add(CT * D * D * sR);                                    // Ps
    add((size_t)(D * D + D * D + D) * sR + (size_t)16 * C * sR + (size_t)5 * C * sizeof(Acc) + 4096);
    const bool wide_carrier = wide && !second && C >= 2 && h->share_model;  // (below: the observation pattern said to the chain-shared wide filter)
    if (wide_carrier) add((size_t)T * D * sR);
    if (wide_carrier) add(wide_gain_tab_bytes(dtype, T, D, D));  // the gain rows: built by the proposal filter, reused by the reverse filter
    add(wide ? wide_filter_ws(h, dtype, kd, parallel, D, D) : ke->filter_ws(h, kd, parallel));
    add(wide ? wide_sample_ws(h, dtype, kd, parallel, D) : se->sample_ws(h, kd, parallel));
    add(wide ? wide_logpdf_ws(dtype, kd) : std::max(ke->logpdf_ws(h, kd), se->sv_logpdf_ws(h, kd)));
    int rc = ws_reserve(h, need);
    if (rc) return rc;
    R* u = (R*)ws_take(h, CT * D * sR);
    R* ys1 = fly ? u : (R*)ws_take(h, CT * D * sR);  // (fly: never read; non-null for the checks below)
    R* ys2 = fly ? u : (R*)ws_take(h, CT * D * sR);
    R* xp = (R*)ws_take(h, CT * D * sR);
    R* Rs1 = second && !fly ? (R*)ws_take(h, CT * D * D * sR) : nullptr;
    R* Rs2 = second && !fly ? (R*)ws_take(h, CT * D * D * sR) : nullptr;
    R* ms = (R*)ws_take(h, CT * D * sR);
    R* Ps = (R*)ws_take(h, CT * D * D * sR);
    R* eye = (R*)ws_take(h, (size_t)D * D * sR);
    R* Rc = (R*)ws_take(h, (size_t)D * D * sR);
    R* zero = (R*)ws_take(h, (size_t)D * sR);
    R* sc = (R*)ws_take(h, (size_t)16 * C * sR);
    Acc* sums = (Acc*)ws_take(h, (size_t)5 * C * sizeof(Acc));
    R* wide_mask = wide_carrier ? (R*)ws_take(h, (size_t)T * D * sR) : nullptr;
    void* wide_gtab = wide_carrier ? ws_take(h, wide_gain_tab_bytes(dtype, T, D, D) - 256) : nullptr;
    if (!u || !ys1 || !ys2 || !xp || !ms || !Ps || !eye || !Rc || !zero || !sc || !sums || (wide_carrier && !wide_mask)) return AUXSSM_ERR_NOMEM;
    R* ell1 = sc; R* ell2 = sc + C; R* j1 = sc + 2 * C; R* j2 = sc + 3 * C; R* terms = sc + 4 * C;
    const size_t mark = h->ws_off;
    const long long tot = (long long)CT * D;
    const unsigned gb = (unsigned)((tot + 255) / 256);
    auto arr = [&](const void* p, long long rec) { return cm ? cm_arr(p, kd, rec) : dense_arr(p, kd, rec); };
    const Arr xA = arr(x, D), xpA = arr(xp, D), uA = arr(u, D), y1A = arr(ys1, D), y2A = arr(ys2, D);
    const Arr R1A = Rs1 ? arr(Rs1, (long long)D * D) : Arr{nullptr, 0, 0, 0, 1};
    const Arr R2A = Rs2 ? arr(Rs2, (long long)D * D) : Arr{nullptr, 0, 0, 0, 1};

    // First-order factory with chain-shared dynamics: the filtered covariances and the gain rows depend on the model and the step size only (the
    // pseudo-observations are finite whatever the data: every component is observed), the same for the proposal and the reverse filter.  That MODEL
    // STAGE runs once per sweep, on the side stream beside the previous sweep (ctx.h::SideStage, as in sweep_lg_concat); the reverse filter reuses
    // its gain rows.
    const int ps_shared = (!second && !wide && chain_shared_mode(h, cm, C, T, model->Fs.sc == 0 && model->Qs.sc == 0 && model->bs.sc == 0 && model->P0.sc == 0)) ? 1 : 0;
    bool overlap = h->overlap_model_stage != 0 && ps_shared && cm && parallel && !dptr;
    struct SweepEnd {
        auxssm_ctx* h;
        ~SweepEnd() { side_sweep_end(h); }
    } sweep_end{h};
    R* mask_carrier = nullptr;
    if (overlap) {
        const size_t P_ = D, gain = (size_t)D * D + D + 2 * (size_t)D * P_ + P_ + P_ * (P_ + 1) / 2 + 4, felem = 3 * (size_t)D * D + 2 * D + 8;
        const size_t need_side = (size_t)(T + 64) * sR * (3 * (size_t)D * D + 4 * D + gain + 2 * felem + 8 * (size_t)D * D + 8 * D + 32) + (4u << 20);
        if ((rc = side_open(h, need_side))) return rc;
        overlap = h->side.open;
    }
    if (overlap) {
        SideScope sc_(h);
        eye = (R*)ws_take(h, (size_t)D * D * sR);
        Rc = (R*)ws_take(h, (size_t)D * D * sR);
        zero = (R*)ws_take(h, (size_t)D * sR);
        Ps = (R*)ws_take(h, (size_t)T * D * D * sR);
        mask_carrier = (R*)ws_take(h, (size_t)T * D * sR);
        if (!eye || !Rc || !zero || !Ps || !mask_carrier) return AUXSSM_ERR_NOMEM;
        hipLaunchKernelGGL((k_fill<R>), dim3((unsigned)(((long long)T * D + 255) / 256)), dim3(256), 0, h->stream, (long long)T * D, (R)0, mask_carrier);
    }
    {
        SideScope sc_(h);
        hipLaunchKernelGGL((k_scaled_eye<R>), dim3((D * D + 255) / 256), dim3(256), 0, h->stream, D, (R)1, eye);
        hipLaunchKernelGGL((k_fill<R>), dim3(1), dim3(256), 0, h->stream, (long long)D, (R)0, zero);
    }
    // wide states, first order: the pseudo-observations u + delta/2 grad are finite by construction, so every chain observes every component -- said to the chain-shared
    // wide filter with a carrier of zeros, which then needs no read-back of the observation patterns (no host synchronisation inside a sweep)
    if (wide_carrier) {
        mask_carrier = wide_mask;
        hipLaunchKernelGGL((k_fill<R>), dim3((unsigned)(((long long)T * D + 255) / 256)), dim3(256), 0, h->stream, (long long)T * D, (R)0, mask_carrier);
    }
    AX_HIP(hipGetLastError());
    // observation LGSSMs of the two linearisation points (H = I, c = 0; R = delta/2 I or the per-step diagonal Omega)
    auxssm_lgssm g1 = *model;
    g1.Hs = auxssm_arr{eye, 0, 0, 0};
    g1.cs = auxssm_arr{zero, 0, 0, 0};
    g1.Rs = second ? auxssm_arr{Rs1, (int64_t)T * D * D, (int64_t)D * D, 0} : auxssm_arr{Rc, 0, 0, 0};
    auxssm_lgssm g2 = g1;
    if (second) g2.Rs = auxssm_arr{Rs2, (int64_t)T * D * D, (int64_t)D * D, 0};
    if (!second) {
        SideScope sc_(h);
        hipLaunchKernelGGL((k_scaled_eye<R>), dim3((D * D + 255) / 256), dim3(256), 0, h->stream, D, (R)(0.5 * delta), Rc, dptr);
    }
    auxssm_dims dc = *dims;
    dc.dy = D;
    dc.B = 1;
    const auxssm_arr y1d{ys1, (int64_t)T * D, (int64_t)D, 0}, y2d{ys2, (int64_t)T * D, (int64_t)D, 0};
    // the filtered covariances do not depend on the chain when neither the dynamics nor R do (first order); wide states: one copy too -- the chain-shared wide filter
    // then skips its broadcast to the chains' slots and the sampler builds its gain / factor tables once per time step (wide.hip::run_sample_shared)
    const bool wide_ps_once = wide_carrier && model->Fs.sc == 0 && model->Qs.sc == 0 && model->bs.sc == 0 && model->P0.sc == 0 && model->m0.sc == 0;
    const Arr PsA = (ps_shared || wide_ps_once) ? Arr{Ps, 0, (long long)D * D, 0, 1} : arr(Ps, (long long)D * D);

    // proposal: observations linearised at x, filter, pathwise sample (generic.py:80-86)
    if (!fly) {
        ProfScope ps(h, AUXSSM_K_FACTORY);
        hipLaunchKernelGGL((k_sv_obs<R>), dim3(gb), dim3(256), 0, h->stream, tot, C, T, D, order, cm, (const R*)x, (const R*)eps_aux,
                           (R)sqrt(0.5 * delta), (R)delta, dptr, cv(*yobs), u, ys1, Rs1);
    }
    FilterArgs fa;
    fill_filter_args(fa, &dc, &g1, &y1d, ms, Ps);
    fa.ys = y1A;
    fa.ms = arr(ms, D);
    fa.Ps = PsA;
    fa.lay.cm = cm;
    if (second) fa.Rs = R1A;
    if (fly) {
        fa.sv_order = order; fa.sv_delta = delta; fa.aux_shd = sqrt(0.5 * delta); fa.dptr = dptr;
        fa.aux_x = xA; fa.aux_eps = arr(eps_aux, D); fa.aux_u = uA; fa.aux_yobs = cv(*yobs);
    }
    if (overlap || wide_carrier) fa.mask_ys = Arr{mask_carrier, 0, (long long)D, 0, 1};
    if (wide_ps_once && wide_gtab) fa.pc = wide_gtab;  // (the chain-shared wide filter leaves its gain rows here)
    rc = ke->filter(h, fa, parallel, ell1);
    if (rc) return rc;
    h->ws_off = mark;
    SampleArgs sa;
    sa.d = kd;
    sa.dx = D;
    sa.Fs = cv(model->Fs); sa.Qs = cv(model->Qs); sa.bs = cv(model->bs);
    sa.ms = arr(ms, D); sa.Ps = PsA;
    sa.eps = arr(eps_samp, D); sa.xs = xpA; sa.elem = nullptr;
    sa.lay = ScanLayout{1, 1, 1, 1, cm, C};
    sa.ps_shared = cm ? ps_shared : 0;
    rc = se->sample(h, sa, parallel);
    if (rc) return rc;
    h->ws_off = mark;
    // reverse move: observations linearised at x_prop, filter for its marginal likelihood (generic.py:67)
    if (!fly) {
        ProfScope ps(h, AUXSSM_K_FACTORY);
        hipLaunchKernelGGL((k_sv_obs<R>), dim3(gb), dim3(256), 0, h->stream, tot, C, T, D, order, cm, (const R*)xp, (const R*)nullptr, (R)0,
                           (R)delta, dptr, cv(*yobs), u, ys2, Rs2);
    }
    fill_filter_args(fa, &dc, &g2, &y2d, ms, Ps);
    fa.ys = y2A;
    fa.ms = arr(ms, D);
    fa.Ps = PsA;
    fa.lay.cm = cm;
    if (second) fa.Rs = R2A;
    if (fly) {  // (only ell2 is used: generic.py:67)
        fa.sv_order = order; fa.sv_delta = delta; fa.aux_shd = sqrt(0.5 * delta); fa.dptr = dptr;
        fa.aux_x = xpA; fa.aux_eps = Arr{nullptr, 0, 0, 0, 1}; fa.aux_u = uA; fa.aux_yobs = cv(*yobs);
        fa.no_moments = 1;
    }
    if (overlap && h->side.last_tab) {  // the proposal filter's gain rows (same model, step size and mask)
        fa.mask_ys = Arr{mask_carrier, 0, (long long)D, 0, 1};
        fa.tab = h->side.last_tab;
        fa.tab_ready = 1;
    }
    if (wide_carrier) fa.mask_ys = Arr{mask_carrier, 0, (long long)D, 0, 1};
    if (wide_ps_once && wide_gtab && parallel && T >= 5) {  // same model, same step size, same pattern: the proposal filter's covariances and gain rows (if it took the shared form:
        fa.pc = wide_gtab;                                   // it does whenever this one would -- same sizes, same strides)
        fa.tab_ready = 1;
    }
    rc = ke->filter(h, fa, parallel, ell2);
    if (rc) return rc;
    h->ws_off = mark;
    if (!wide) {
        // every log-density of the MH ratio in one pass over the chains (generic.py:88-89, :98-106)
        SvLogpdfArgs la;
        la.d = kd;
        la.m0 = cv(model->m0); la.P0 = cv(model->P0); la.Fs = cv(model->Fs); la.Qs = cv(model->Qs); la.bs = cv(model->bs);
        la.yobs = cv(*yobs);
        la.x = xA; la.xp = xpA; la.u = uA; la.ys1 = y1A; la.ys2 = y2A; la.R1 = R1A; la.R2 = R2A;
        la.delta = delta;
        la.dptr = dptr;
        la.fly_order = fly ? order : 0;
        rc = se->sv_logpdf(h, la, sums);  // [5][C] = jp_prop, jp_rev, lt_prop, lt_rev, corr
        if (rc) return rc;
        h->ws_off = mark;
        hipLaunchKernelGGL((k_accept<R>), dim3((C + 127) / 128), dim3(128), 0, h->stream, C, (const Acc*)sums, (const Acc*)(sums + C), (const R*)ell1,
                           (const R*)ell2, (const Acc*)(sums + 2 * C), (const Acc*)(sums + 3 * C), (const Acc*)(sums + 4 * C), (const R*)u_acc, accepted,
                           (R*)logs);
    } else {
        // wide-state path: joint log-densities of both auxiliary models (posterior_logpdf + ell, base.py:72-96), then the SV terms
        bool both = false;
        const SweepLogpdfEntry* wsl = wide_sweep_logpdf_entry(dtype);
        if (wide_ps_once && nan_policy == AUXSSM_NAN_REFERENCE && wsl && wsl->wide_shared) {
            // first order on one model: both joints from ONE launch pair -- Q_t^-1, R^-1 and their determinants once per time step, the chains as columns, ys1 scored
            // against x' and ys2 against x (wide_shared.h::wk_lp_cols); sums [2] / [3] = observation + transition terms of x' / x
            SweepLogpdfArgs sl_;
            sl_.d = kd;
            sl_.dx = D; sl_.po = D;
            sl_.m0 = cv(model->m0); sl_.P0 = cv(model->P0); sl_.Fs = cv(model->Fs); sl_.Qs = cv(model->Qs); sl_.bs = cv(model->bs);
            sl_.Hs = cv(g1.Hs); sl_.Rs = cv(g1.Rs); sl_.cs = cv(g1.cs);
            sl_.ys = cv(y1d); sl_.ys_x = cv(y2d);
            sl_.x = dense_arr(x, kd, D); sl_.xp = dense_arr(xp, kd, D); sl_.u = dense_arr(u, kd, D);
            sl_.delta = delta; sl_.dptr = dptr; sl_.nan_policy = nan_policy; sl_.u_fly = 0; sl_.shd = sqrt(0.5 * delta);
            rc = wsl->wide_shared(h, sl_, sums);
            if (rc == AUXSSM_OK) {
                hipLaunchKernelGGL((k_acc_to_real<R>), dim3((2 * C + 127) / 128), dim3(128), 0, h->stream, C, (const Acc*)(sums + 2 * C), j1, j2);
                both = true;
            } else if (rc != 1) {
                return rc;
            }
            h->ws_off = mark;
        }
        if (!both) {
            LogpdfArgs la;
            fill_logpdf_args(la, &dc, &g1, cv(y1d), dense_arr(xp, kd, D), nan_policy);
            rc = ke->logpdf(h, la, j1);
            if (rc) return rc;
            h->ws_off = mark;
            fill_logpdf_args(la, &dc, &g2, cv(y2d), dense_arr(x, kd, D), nan_policy);
            rc = ke->logpdf(h, la, j2);
            if (rc) return rc;
            h->ws_off = mark;
        }
        hipLaunchKernelGGL((k_sv_terms<R>), dim3(C), dim3(256), 0, h->stream, C, T, D, (R)delta, dptr, (const R*)x, (const R*)xp, (const R*)u,
                           cv(*yobs), (const R*)ys1, (const R*)ys2, (const R*)Rs1, (const R*)Rs2, terms);
        hipLaunchKernelGGL((k_sv_accept<R>), dim3((C + 127) / 128), dim3(128), 0, h->stream, C, (const R*)j1, (const R*)j2, (const R*)ell1,
                           (const R*)ell2, (const R*)terms, (const R*)u_acc, accepted, (R*)logs);
    }
    if ((rc = launch_select<R>(h, C, T, D, (const int32_t*)accepted, xpA, xA, cm))) return rc;
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// ---- Lorenz-63 device factories (examples/lorenz/auxiliary_kalman.py:14-52, model.py:10-25, linearisation.py:11-44) ------------
// dynamics_factory(x) = first-order extended linearisation of mean(x) = x + dt (phi_0(x) + theta * phi(x)) at every x_t, with the
// analytic Jacobian in place of jacfwd:  F_t = I + dt J(x_t),  b_t = mean(x_t) - F_t x_t,  Q = model Qs;
// observations_factory = the auxiliary observations concatenated with the real ones (as LG_CONCAT);
// log_likelihood_fn(x) = log N(x_0; m0, P0) + sum_t log N(x_{t+1}; mean(x_t), Q) + nansum_t log N(y_t; H_t x_t + c_t, R_t).
template <typename R>
__global__ void k_lorenz_dyn(int C, int T, int cfast, const R* __restrict__ par, long long psc, const R* __restrict__ xlin, R* __restrict__ Fs,
                             R* __restrict__ bs) {
    // one thread per (chain, transition); dense: x (C, T, 3), Fs (C, n, 9), bs (C, n, 3); chain-minor: x (T, 3, C), Fs (n, 9, C), bs (n, 3, C)
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = T - 1;
    if (g >= (long long)C * n) return;
    const long long c = cfast ? g % C : g / n, i = cfast ? g / C : g % n;
    par += c * psc;  // per-chain theta (a Gibbs sampler over (x, theta) keeps one theta per chain)
    const R th[3] = {par[0], par[1], par[2]};
    const R dt = par[3];
    const long long xs_ = cfast ? C : 1;
    const R* xq = xlin + (cfast ? i * 3 * (long long)C + c : (c * T + i) * 3);
    const R x[3] = {xq[0], xq[xs_], xq[2 * xs_]};
    R F[9], mu[3];
    lorenz_lin_F<R>(th, dt, x, F);
    lorenz_mean<R>(th, dt, x, mu);
    R* Fo = Fs + (cfast ? i * 9 * (long long)C + c : g * 9);
    R* bo = bs + (cfast ? i * 3 * (long long)C + c : g * 3);
    for (int k = 0; k < 9; ++k) Fo[k * xs_] = F[k];
    for (int r = 0; r < 3; ++r) bo[r * xs_] = mu[r] - (F[r * 3] * x[0] + F[r * 3 + 1] * x[1] + F[r * 3 + 2] * x[2]);
}

template <typename R>
static int sweep_lorenz(auxssm_ctx* h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* model, const auxssm_arr* yobs, double delta, const double* dptr,
                        int parallel, int nan_policy, int layout, void* x, const void* eps_aux, const void* eps_samp, const void* u_acc,
                        int32_t* accepted, void* logs) {
    const int C = dims->C, T = dims->T, D = 3, PO = dims->dy, P = D + PO;
    const KalmanEntry* ke = need_kalman(dtype, D, P);
    const SampleEntry* se = sample_entry(dtype, D);
    const SweepLogpdfEntry* sl = sweep_logpdf_entry(dtype, D, PO);
    if (!ke || !se || !sl || !sl->lorenz) return AUXSSM_ERR_UNSUPPORTED;
    const int cm = layout == AUXSSM_LAYOUT_CHAIN_MINOR ? 1 : 0;  // state, noise and every per-chain buffer (T, ., C): lanes <-> chains
    const KDims kd{C, T, 1};
    const size_t sR = sizeof(R), CT = (size_t)C * T, n = (size_t)(T > 1 ? T - 1 : 1);
    size_t need = 0;
    auto add = [&](size_t b) { need += b + 256; };
    add(CT * P * sR);
    add((size_t)T * (P * D + P * P + P) * sR + 1024);
    for (int q = 0; q < 4; ++q) add(CT * D * sR);             // u, ms, xp, (spare)
    add(CT * D * D * sR);                                     // Ps
    add(2 * (size_t)C * n * (9 + 3) * sR + 1024);             // Fs1, bs1, Fs2, bs2
    add((size_t)16 * C * sR + (size_t)5 * C * sizeof(Acc) + 2048);
    add(ke->filter_ws(h, kd, parallel));
    add(se->sample_ws(h, kd, parallel));
    add(sl->ws(h, kd));
    int rc = ws_reserve(h, need);
    if (rc) return rc;
    R* ysc = (R*)ws_take(h, CT * P * sR);
    R* Hc = (R*)ws_take(h, (size_t)T * P * D * sR);
    R* Rc = (R*)ws_take(h, (size_t)T * P * P * sR);
    R* cc = (R*)ws_take(h, (size_t)T * P * sR);
    R* u = (R*)ws_take(h, CT * D * sR);
    R* ms = (R*)ws_take(h, CT * D * sR);
    R* xp = (R*)ws_take(h, CT * D * sR);
    R* Ps = (R*)ws_take(h, CT * D * D * sR);
    R* Fs1 = (R*)ws_take(h, (size_t)C * n * 9 * sR);
    R* bs1 = (R*)ws_take(h, (size_t)C * n * 3 * sR);
    R* Fs2 = (R*)ws_take(h, (size_t)C * n * 9 * sR);
    R* bs2 = (R*)ws_take(h, (size_t)C * n * 3 * sR);
    R* sc = (R*)ws_take(h, (size_t)16 * C * sR);
    Acc* sums = (Acc*)ws_take(h, (size_t)5 * C * sizeof(Acc));
    if (!ysc || !Hc || !Rc || !cc || !u || !ms || !xp || !Ps || !Fs1 || !bs1 || !Fs2 || !bs2 || !sc || !sums) return AUXSSM_ERR_NOMEM;
    R* ell1 = sc; R* ell2 = sc + C;
    const size_t mark = h->ws_off;
    const R* par = (const R*)model->Fs.ptr;  // [theta1, theta2, theta3, dt], chain stride model->Fs.sc (0 = one theta for all chains)
    const long long psc = model->Fs.sc;
    auto arr = [&](const void* p, long long rec) { return cm ? cm_arr(p, kd, rec) : dense_arr(p, kd, rec); };
    const Arr xA = arr(x, D), xpA = arr(xp, D), uA = arr(u, D), yscA = arr(ysc, P);
    const bool aux_fly = cm && T > 1 && aux_fly_enabled();  // u and the concatenated observations of t >= 1 are formed inside the filter
    // per-chain transition arrays have n = T - 1 rows: same strides as a T-row array of that record size
    const Arr F1A = arr(Fs1, 9), b1A = arr(bs1, 3), F2A = arr(Fs2, 9), b2A = arr(bs2, 3);
    {
        ProfScope ps(h, AUXSSM_K_FACTORY);
        const long long n1 = (long long)T * (P * D + P * P + P);
        hipLaunchKernelGGL((k_concat_model<R>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, h->stream, T, D, PO, cv(model->Hs),
                           cv(model->Rs), cv(model->cs), (R)(0.5 * delta), dptr, Hc, Rc, cc);
        const int Tc = aux_fly ? 1 : T;
        const long long n2 = (long long)C * Tc * P;
        hipLaunchKernelGGL((k_concat_obs<R>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, h->stream, C, Tc, D, PO, xA,
                           arr(eps_aux, D), (R)sqrt(0.5 * delta), dptr, cv(*yobs), uA, yscA, cm);
    }
    const unsigned gd = (unsigned)(((long long)C * (T - 1) + 255) / 256);
    auxssm_lgssm g1 = *model;
    g1.Hs = auxssm_arr{Hc, 0, (int64_t)P * D, 0};
    g1.Rs = auxssm_arr{Rc, 0, (int64_t)P * P, 0};
    g1.cs = auxssm_arr{cc, 0, (int64_t)P, 0};
    g1.Fs = auxssm_arr{Fs1, (int64_t)n * 9, 9, 0};
    g1.bs = auxssm_arr{bs1, (int64_t)n * 3, 3, 0};
    auxssm_lgssm g2 = g1;
    g2.Fs = auxssm_arr{Fs2, (int64_t)n * 9, 9, 0};
    g2.bs = auxssm_arr{bs2, (int64_t)n * 3, 3, 0};
    auxssm_dims dc = *dims;
    dc.dy = P;
    dc.B = 1;
    const auxssm_arr yd{ysc, (int64_t)T * P, (int64_t)P, 0};
    auto set_views = [&](FilterArgs& fa, const Arr& FA, const Arr& bA) {
        fa.ys = yscA;
        fa.ms = arr(ms, D);
        fa.Ps = arr(Ps, (long long)D * D);
        fa.lay.cm = cm;
        fa.pblk = D;
        if (cm) {
            fa.Fs = FA;
            fa.bs = bA;
        }
        if (aux_fly) {
            fa.aux_on = 1;
            fa.aux_x = xA;
            fa.aux_eps = arr(eps_aux, D);
            fa.aux_u = uA;
            fa.aux_yobs = cv(*yobs);
            fa.aux_shd = sqrt(0.5 * delta);
        fa.dptr = dptr;
        }
    };

    // proposal: dynamics linearised at x (generic.py:80-86)
    if (T > 1) {
        ProfScope ps(h, AUXSSM_K_FACTORY);
        hipLaunchKernelGGL((k_lorenz_dyn<R>), dim3(gd), dim3(256), 0, h->stream, C, T, cm, par, psc, (const R*)x, Fs1, bs1);
    }
    FilterArgs fa;
    fill_filter_args(fa, &dc, &g1, &yd, ms, Ps);
    set_views(fa, F1A, b1A);
    rc = ke->filter(h, fa, parallel, ell1);
    if (rc) return rc;
    h->ws_off = mark;
    SampleArgs sa;
    sa.d = kd;
    sa.dx = D;
    sa.Fs = cm ? F1A : cv(g1.Fs); sa.Qs = cv(model->Qs); sa.bs = cm ? b1A : cv(g1.bs);
    sa.ms = arr(ms, D); sa.Ps = arr(Ps, (long long)D * D);
    sa.eps = arr(eps_samp, D); sa.xs = xpA; sa.elem = nullptr;
    sa.lay = ScanLayout{1, 1, 1, 1, cm, C};
    rc = se->sample(h, sa, parallel);
    if (rc) return rc;
    h->ws_off = mark;
    // reverse move: dynamics linearised at x_prop (generic.py:67)
    if (T > 1) {
        ProfScope ps(h, AUXSSM_K_FACTORY);
        hipLaunchKernelGGL((k_lorenz_dyn<R>), dim3(gd), dim3(256), 0, h->stream, C, T, cm, par, psc, (const R*)xp, Fs2, bs2);
    }
    fill_filter_args(fa, &dc, &g2, &yd, ms, Ps);
    set_views(fa, F2A, b2A);
    rc = ke->filter(h, fa, parallel, ell2);
    if (rc) return rc;
    h->ws_off = mark;
    // every log-density of the MH ratio in one pass (generic.py:88-89, :98-106); the linearised transitions are rebuilt from x / x_prop
    {
        SweepLogpdfArgs la;
        la.d = kd;
        la.dx = D; la.po = PO;
        la.m0 = cv(model->m0); la.P0 = cv(model->P0); la.Qs = cv(model->Qs);
        la.Fs = Arr{nullptr, 0, 0, 0, 1}; la.bs = Arr{nullptr, 0, 0, 0, 1};
        la.Hs = cv(model->Hs); la.Rs = cv(model->Rs); la.cs = cv(model->cs); la.ys = cv(*yobs);
        la.x = xA; la.xp = xpA; la.u = uA; la.delta = delta; la.nan_policy = nan_policy;
        la.u_fly = aux_fly ? 1 : 0; la.eps_aux = arr(eps_aux, D); la.shd = sqrt(0.5 * delta); la.dptr = dptr;
        la.lor_par = par; la.lor_psc = psc;
        rc = sl->lorenz(h, la, sums);  // [5][C]: jp_prop, jp_rev, lt_prop, lt_rev, corr
        if (rc) return rc;
        h->ws_off = mark;
    }
    hipLaunchKernelGGL((k_accept<R>), dim3((C + 127) / 128), dim3(128), 0, h->stream, C, (const Acc*)sums, (const Acc*)(sums + C), (const R*)ell1,
                       (const R*)ell2, (const Acc*)(sums + 2 * C), (const Acc*)(sums + 3 * C), (const Acc*)(sums + 4 * C), (const R*)u_acc, accepted,
                       (R*)logs);
    if ((rc = launch_select<R>(h, C, T, D, (const int32_t*)accepted, xpA, xA, cm))) return rc;
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

// ---- RNG fill -----------------------------------------------------------------------------------------------------
// one Threefry block -> out[2 i], out[2 i + 1] (both fills)
template <typename R> __global__ void k_rng_uniform(uint32_t k0, uint32_t k1, uint32_t stream, long long n, R* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i >= n) return;
    R u0, u1;
    stream_uniform2<R>(k0, k1, stream, (unsigned long long)i, u0, u1);
    out[2 * i] = u0;
    if (2 * i + 1 < n) out[2 * i + 1] = u1;
}
template <typename R> __global__ void k_rng_normal(uint32_t k0, uint32_t k1, uint32_t stream, long long n, R* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i >= n) return;
    uint32_t x0, x1;
    stream_counter(stream, (unsigned long long)i, x0, x1);
    threefry2x32(k0, k1, x0, x1);
    R z0, z1;
    bits_to_normal2<R>(x0, x1, z0, z1);
    out[2 * i] = z0;
    if (2 * i + 1 < n) out[2 * i + 1] = z1;
}

// the three noise fills of one Kalman sweep in one launch: the same values as auxssm_rng_normal (keys a, b; stream 0; n each) and
// auxssm_rng_uniform (key c; stream 0; nu) -- workgroups [0, g1) fill eps_aux, [g1, 2 g1) eps_samp, the rest u_acc
template <typename R>
__global__ void k_rng_sweep(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t c0, uint32_t c1, long long n, long long nu, unsigned g1,
                            R* eps_aux, R* eps_samp, R* u_acc) {
    const unsigned blk = blockIdx.x;
    if (blk < 2 * g1) {
        const bool second = blk >= g1;
        const long long i = (long long)(second ? blk - g1 : blk) * blockDim.x + threadIdx.x;
        if (2 * i >= n) return;
        uint32_t x0, x1;
        stream_counter(0, (unsigned long long)i, x0, x1);
        threefry2x32(second ? b0 : a0, second ? b1 : a1, x0, x1);
        R z0, z1;
        bits_to_normal2<R>(x0, x1, z0, z1);
        R* out = second ? eps_samp : eps_aux;
        out[2 * i] = z0;
        if (2 * i + 1 < n) out[2 * i + 1] = z1;
    } else {
        const long long i = (long long)(blk - 2 * g1) * blockDim.x + threadIdx.x;
        if (2 * i >= nu) return;
        R u0, u1;
        stream_uniform2<R>(c0, c1, 0, (unsigned long long)i, u0, u1);
        u_acc[2 * i] = u0;
        if (2 * i + 1 < nu) u_acc[2 * i + 1] = u1;
    }
}

}  // namespace ax

using namespace ax;

extern "C" {

int auxssm_version(void) { return AUXSSM_VERSION; }
const char* auxssm_last_error(void) { return g_err.c_str(); }

int auxssm_device_count(int* count) {
    if (!count) return AUXSSM_ERR_ARG;
    AX_HIP(hipGetDeviceCount(count));
    return AUXSSM_OK;
}

int auxssm_create(int device, auxssm_handle* out) {
    if (!out) {
        set_error("out is NULL");
        return AUXSSM_ERR_ARG;
    }
    int n = 0;
    AX_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        set_error("device %d out of range (have %d)", device, n);
        return AUXSSM_ERR_ARG;
    }
    AX_HIP(hipSetDevice(device));
    auxssm_ctx* h = new auxssm_ctx();
    h->device = device;
    hipDeviceProp_t prop;
    AX_HIP(hipGetDeviceProperties(&prop, device));
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    AX_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    if (const char* e = getenv("AUXSSM_SHARED")) h->share_model = atoi(e) != 0;  // default of AUXSSM_OPT_SHARE_MODEL
    if (const char* e = getenv("AUXSSM_OVERLAP_TAB")) h->overlap_model_stage = atoi(e) != 0;  // default of AUXSSM_OPT_OVERLAP_MODEL_STAGE
    *out = h;
    return AUXSSM_OK;
}

int auxssm_destroy(auxssm_handle h) {
    if (!h) return AUXSSM_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    auxssm_prof_disable(h);
    if (h->ws) (void)hipFree(h->ws);
    if (h->dblock) (void)hipFree(h->dblock);
    if (h->cw_dev) (void)hipFree(h->cw_dev);
    if (h->side.streams[0]) {
        for (int p = 0; p < auxssm_ctx::SideStage::NS; ++p) {
            (void)hipStreamSynchronize(h->side.streams[p]);
            if (h->side.ws[p]) (void)hipFree(h->side.ws[p]);
            (void)hipEventDestroy(h->side.done[p]);
            (void)hipEventDestroy(h->side.sweep_end[p]);
            (void)hipEventDestroy(h->side.begun[p]);
            (void)hipStreamDestroy(h->side.streams[p]);
        }
        (void)hipEventDestroy(h->side.fence);
    }
    if (h->fork_stream) {
        (void)hipEventDestroy(h->fork_ev);
        (void)hipEventDestroy(h->join_ev);
        (void)hipStreamDestroy(h->fork_stream);
    }
    (void)hipStreamDestroy(h->stream);
    delete h;
    return AUXSSM_OK;
}

#define AX_NEED_H(h)                         \
    do {                                     \
        if (!(h)) {                          \
            set_error("handle is NULL");     \
            return AUXSSM_ERR_ARG;           \
        }                                    \
        AX_HIP(hipSetDevice((h)->device));   \
        ++(h)->api_calls;                    \
    } while (0)
// entry points that neither enqueue work nor change device data (ctx.h: SideStage::last_call)
#define AX_NEED_H_RO(h)                      \
    do {                                     \
        AX_NEED_H(h);                        \
        --(h)->api_calls;                    \
    } while (0)

int auxssm_sync(auxssm_handle h) {
    AX_NEED_H_RO(h);
    AX_HIP(hipStreamSynchronize(h->stream));
    return AUXSSM_OK;
}
int auxssm_stream(auxssm_handle h, void** stream) {
    AX_NEED_H(h);
    if (!stream) return AUXSSM_ERR_ARG;
    *stream = (void*)h->stream;
    h->stream_exposed = true;
    return AUXSSM_OK;
}
int auxssm_malloc(auxssm_handle h, size_t bytes, void** dptr) {
    AX_NEED_H(h);
    if (!dptr) return AUXSSM_ERR_ARG;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return AUXSSM_ERR_NOMEM;
    }
    return AUXSSM_OK;
}
int auxssm_free(auxssm_handle h, void* dptr) {
    AX_NEED_H(h);
    if (dptr) {
        AX_HIP(hipStreamSynchronize(h->stream));
        AX_HIP(hipFree(dptr));
    }
    return AUXSSM_OK;
}
int auxssm_memcpy_h2d(auxssm_handle h, void* dst, const void* src, size_t bytes) {
    AX_NEED_H(h);
    AX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    AX_HIP(hipStreamSynchronize(h->stream));
    return AUXSSM_OK;
}
int auxssm_memcpy_d2h(auxssm_handle h, void* dst, const void* src, size_t bytes) {
    AX_NEED_H_RO(h);
    AX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    AX_HIP(hipStreamSynchronize(h->stream));
    return AUXSSM_OK;
}
int auxssm_memcpy_d2d(auxssm_handle h, void* dst, const void* src, size_t bytes) {
    AX_NEED_H(h);
    AX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->stream));
    return AUXSSM_OK;
}
int auxssm_memset(auxssm_handle h, void* dst, int value, size_t bytes) {
    AX_NEED_H(h);
    AX_HIP(hipMemsetAsync(dst, value, bytes, h->stream));
    return AUXSSM_OK;
}

int auxssm_set_option(auxssm_handle h, int option, int value) {
    AX_NEED_H(h);
    if (option == AUXSSM_OPT_SHARE_MODEL) {
        h->share_model = value != 0;
        return AUXSSM_OK;
    }
    if (option == AUXSSM_OPT_OVERLAP_MODEL_STAGE) {
        h->overlap_model_stage = value != 0;
        return AUXSSM_OK;
    }
    set_error("unknown option %d", option);
    return AUXSSM_ERR_ARG;
}

int auxssm_get_option(auxssm_handle h, int option, int* value) {
    AX_NEED_H_RO(h);
    if (!value) return AUXSSM_ERR_ARG;
    if (option == AUXSSM_OPT_SHARE_MODEL) *value = h->share_model;
    else if (option == AUXSSM_OPT_OVERLAP_MODEL_STAGE) *value = h->overlap_model_stage;
    else {
        set_error("unknown option %d", option);
        return AUXSSM_ERR_ARG;
    }
    return AUXSSM_OK;
}

int auxssm_prof_disable(auxssm_handle h) {
    if (!h) return AUXSSM_ERR_ARG;
    Prof& p = h->prof;
    for (auto e : p.start) (void)hipEventDestroy(e);
    for (auto e : p.stop) (void)hipEventDestroy(e);
    p.start.clear();
    p.stop.clear();
    p.ids.clear();
    p.kernel_id = 0;
    p.max_launches = 0;
    p.used = 0;
    return AUXSSM_OK;
}
int auxssm_prof_enable(auxssm_handle h, int kernel_id, int max_launches) {
    AX_NEED_H_RO(h);
    if (max_launches < 1 || max_launches > (1 << 16)) {
        set_error("max_launches must be in [1, 65536]");
        return AUXSSM_ERR_ARG;
    }
    auxssm_prof_disable(h);
    Prof& p = h->prof;
    p.start.resize(max_launches);
    p.stop.resize(max_launches);
    p.ids.assign(max_launches, 0);
    for (int i = 0; i < max_launches; ++i) {
        AX_HIP(hipEventCreate(&p.start[i]));
        AX_HIP(hipEventCreate(&p.stop[i]));
    }
    p.kernel_id = kernel_id;
    p.max_launches = max_launches;
    p.used = 0;
    return AUXSSM_OK;
}
int auxssm_prof_read(auxssm_handle h, int* launches, double* total_ms) {
    AX_NEED_H_RO(h);
    AX_HIP(hipStreamSynchronize(h->stream));
    Prof& p = h->prof;
    double tot = 0;
    for (int i = 0; i < p.used; ++i) {
        float ms = 0;
        AX_HIP(hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
        tot += ms;
    }
    if (launches) *launches = p.used;
    if (total_ms) *total_ms = tot;
    p.used = 0;
    return AUXSSM_OK;
}

int auxssm_prof_read_groups(auxssm_handle h, int n_ids, int* launches, double* total_ms) {
    AX_NEED_H_RO(h);
    if (n_ids < 1 || !launches || !total_ms) {
        set_error("n_ids must be >= 1 and launches / total_ms non-NULL");
        return AUXSSM_ERR_ARG;
    }
    AX_HIP(hipStreamSynchronize(h->stream));
    Prof& p = h->prof;
    for (int k = 0; k < n_ids; ++k) launches[k] = 0, total_ms[k] = 0;
    for (int i = 0; i < p.used; ++i) {
        float ms = 0;
        AX_HIP(hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
        const int id = p.ids[i];
        if (id >= 0 && id < n_ids) ++launches[id], total_ms[id] += ms;
    }
    p.used = 0;
    return AUXSSM_OK;
}

int auxssm_kalman_filter(auxssm_handle h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* lgssm,
                         const auxssm_arr* ys, int parallel, void* ms, void* Ps, void* ell) {
    AX_NEED_H(h);
    int rc;
    if ((rc = check_dtype(dtype)) || (rc = check_dims(dims, true)) || (rc = check_lgssm(lgssm, dims->T))) return rc;
    if (!ys || !ys->ptr || !ms || !Ps || !ell) {
        set_error("ys/ms/Ps/ell must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    const KalmanEntry* e = need_kalman(dtype, dims->dx, dims->dy);
    if (!e) return AUXSSM_ERR_UNSUPPORTED;
    const KDims kd{dims->C, dims->T, dims->B};
    const size_t wsb = is_wide(dims->dx, dims->dy) ? wide_filter_ws(h, dtype, kd, parallel, dims->dx, dims->dy) : e->filter_ws(h, kd, parallel);
    if ((rc = ws_reserve(h, wsb + 4096))) return rc;
    FilterArgs a;
    fill_filter_args(a, dims, lgssm, ys, ms, Ps);
    // a wide batch axis (the reference's spatial example: B = 64 scalar LGSSMs side by side, examples/spatial/model.py:103-112): lanes <-> (c, b) sequences, which
    // are contiguous along b in every (C, T, B, .) array -- no element buffer, the sequential recursion inside each time chunk (kernels.hip.h::run_filter)
    static const int lanes_b = [] { const char* e = getenv("AUXSSM_FILTER_BATCH_LANES"); return e ? atoi(e) : 32; }();
    if (!is_wide(dims->dx, dims->dy) && lanes_b > 0 && dims->B >= lanes_b && (long long)dims->C * dims->B >= 256) a.lay.cm = 1;
    return e->filter(h, a, parallel, ell);
}

int auxssm_kalman_sample(auxssm_handle h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* lgssm,
                         const void* ms, const void* Ps, const void* eps, int parallel, void* xs) {
    AX_NEED_H(h);
    int rc;
    if ((rc = check_dtype(dtype)) || (rc = check_dims(dims, false))) return rc;
    if (!lgssm || !ms || !Ps || !eps || !xs || (dims->T > 1 && (!lgssm->Fs.ptr || !lgssm->Qs.ptr || !lgssm->bs.ptr))) {
        set_error("lgssm(Fs,Qs,bs)/ms/Ps/eps/xs must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    const bool wide = dims->dx > MAX_D;
    std::string why;
    if (wide && !wide_fits(dtype, dims->dx, 0, &why)) {
        set_error("%s", why.c_str());
        return AUXSSM_ERR_UNSUPPORTED;
    }
    const SampleEntry* e = wide ? wide_sample_entry(dtype) : sample_entry(dtype, dims->dx);
    if (!e) {
        set_error("dx=%d is not instantiated in this build", dims->dx);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    const KDims kd{dims->C, dims->T, dims->B};
    const size_t wsb = wide ? wide_sample_ws(h, dtype, kd, parallel, dims->dx) : e->sample_ws(h, kd, parallel);
    if ((rc = ws_reserve(h, wsb + 4096))) return rc;
    SampleArgs a;
    a.d = kd;
    a.dx = dims->dx;
    a.Fs = cv(lgssm->Fs); a.Qs = cv(lgssm->Qs); a.bs = cv(lgssm->bs);
    a.ms = dense_arr(ms, kd, dims->dx); a.Ps = dense_arr(Ps, kd, (long long)dims->dx * dims->dx);
    a.eps = dense_arr(eps, kd, dims->dx); a.xs = dense_arr(xs, kd, dims->dx); a.elem = nullptr;
    a.lay = ScanLayout{1, 1, 1, 1, 0, kd.C * kd.B};
    return e->sample(h, a, parallel);
}

int auxssm_kalman_dnc_sample(auxssm_handle h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* lgssm, const void* ms, const void* Ps, const void* eps, void* xs) {
    AX_NEED_H(h);
    int rc;
    if ((rc = check_dtype(dtype)) || (rc = check_dims(dims, false))) return rc;
    if (!lgssm || !ms || !Ps || !eps || !xs || (dims->T > 1 && (!lgssm->Fs.ptr || !lgssm->Qs.ptr || !lgssm->bs.ptr))) {
        set_error("lgssm(Fs,Qs,bs)/ms/Ps/eps/xs must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    if (dims->B != 1) {  // dnc_sampling.py:42-43
        set_error("Batched sampling is not supported for this function. Use auxssm_kalman_sample instead.");
        return AUXSSM_ERR_ARG;
    }
    if (dims->dx > MAX_D) {
        set_error("the divide-and-conquer sampler is instantiated for dx <= %d (dx = %d)", MAX_D, dims->dx);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    const KDims kd{dims->C, dims->T, 1};
    return run_dnc(h, dtype, dims->C, dims->T, dims->dx, cv(lgssm->Fs), cv(lgssm->Qs), cv(lgssm->bs), dense_arr(ms, kd, dims->dx),
                   dense_arr(Ps, kd, (long long)dims->dx * dims->dx), eps, xs);
}

int auxssm_kalman_joint_logpdf(auxssm_handle h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* lgssm,
                               const auxssm_arr* ys, const auxssm_arr* xs, int nan_policy, void* out) {
    AX_NEED_H(h);
    int rc;
    if ((rc = check_dtype(dtype)) || (rc = check_dims(dims, true)) || (rc = check_lgssm(lgssm, dims->T))) return rc;
    if (!ys || !ys->ptr || !xs || !xs->ptr || !out) {
        set_error("ys/xs/out must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    if (nan_policy != AUXSSM_NAN_REFERENCE && nan_policy != AUXSSM_NAN_MASKED) {
        set_error("nan_policy must be 0 (reference) or 1 (masked)");
        return AUXSSM_ERR_ARG;
    }
    const KalmanEntry* e = need_kalman(dtype, dims->dx, dims->dy);
    if (!e) return AUXSSM_ERR_UNSUPPORTED;
    const KDims kd{dims->C, dims->T, dims->B};
    const size_t wsb = is_wide(dims->dx, dims->dy) ? wide_logpdf_ws(dtype, kd) : e->logpdf_ws(h, kd);
    if ((rc = ws_reserve(h, wsb + 4096))) return rc;
    LogpdfArgs a;
    fill_logpdf_args(a, dims, lgssm, cv(*ys), cv(*xs), nan_policy);
    return e->logpdf(h, a, out);
}

// delta_dev != NULL: the step size is a device scalar of `dtype` (auxssm_kalman_sweep_dd); `delta` is then only a placeholder
static int kalman_sweep_impl(auxssm_handle h, int dtype, int model_kind, const auxssm_dims* dims, const auxssm_lgssm* model,
                             const auxssm_arr* yobs, double delta, const void* delta_dev, const uint32_t* keys, int parallel, int nan_policy, int layout,
                             void* x, const void* eps_aux, const void* eps_samp, const void* u_acc, int32_t* accepted, void* logs) {
    AX_NEED_H(h);
    int rc;
    if ((rc = check_dtype(dtype)) || (rc = check_dims(dims, true))) return rc;
    const bool sv = model_kind == AUXSSM_KMODEL_SV_FIRST || model_kind == AUXSSM_KMODEL_SV_SECOND;
    const bool lorenz = model_kind == AUXSSM_KMODEL_LORENZ63_EXT;
    if (model_kind != AUXSSM_KMODEL_LG_CONCAT && !sv && !lorenz) {
        set_error("unknown model_kind %d", model_kind);
        return AUXSSM_ERR_ARG;
    }
    if (sv) {
        if (!model || !model->m0.ptr || !model->P0.ptr || (dims->T > 1 && (!model->Fs.ptr || !model->Qs.ptr || !model->bs.ptr))) {
            set_error("model needs m0, P0 (and Fs, Qs, bs with T > 1)");
            return AUXSSM_ERR_ARG;
        }
        if (dims->dy != dims->dx) {
            set_error("stochastic-volatility models observe every state component: dy (%d) must equal dx (%d)", dims->dy, dims->dx);
            return AUXSSM_ERR_ARG;
        }
    } else if (lorenz) {
        if (!model || !model->m0.ptr || !model->P0.ptr || !model->Fs.ptr || !model->Qs.ptr || !model->Hs.ptr || !model->Rs.ptr || !model->cs.ptr) {
            set_error("Lorenz model needs m0, P0, Fs (= [theta1, theta2, theta3, dt] on the device), Qs, Hs, Rs, cs");
            return AUXSSM_ERR_ARG;
        }
        if (dims->dx != 3 || dims->dy < 1 || dims->dy > 3) {
            set_error("Lorenz-63 has dx = 3 and 1..3 observed combinations (got dx=%d, dy=%d)", dims->dx, dims->dy);
            return AUXSSM_ERR_ARG;
        }
    } else if ((rc = check_lgssm(model, dims->T))) {
        return rc;
    }
    if (dims->B != 1) {
        set_error("auxssm_kalman_sweep needs B == 1");
        return AUXSSM_ERR_ARG;
    }
    if (!(delta > 0)) {
        set_error("delta must be > 0");
        return AUXSSM_ERR_ARG;
    }
    if (layout != AUXSSM_LAYOUT_DENSE && layout != AUXSSM_LAYOUT_CHAIN_MINOR) {
        set_error("layout must be AUXSSM_LAYOUT_DENSE (0) or AUXSSM_LAYOUT_CHAIN_MINOR (1)");
        return AUXSSM_ERR_ARG;
    }
    if (!yobs || !yobs->ptr || !x || !eps_aux || !eps_samp || !u_acc || !accepted) {
        set_error("yobs/x/eps_aux/eps_samp/u_acc/accepted must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    if (nan_policy != AUXSSM_NAN_REFERENCE && nan_policy != AUXSSM_NAN_MASKED) {
        set_error("nan_policy must be 0 (reference) or 1 (masked)");
        return AUXSSM_ERR_ARG;
    }
    // The data and the REAL observation model are what the chains have in common (the reference's factories close over them,
    // examples/lorenz/auxiliary_kalman.py:26-35): the concatenated model is built once per time step, so per-chain copies are refused
    // rather than silently read at chain 0.
    if (yobs->sc != 0 || (!sv && (model->Hs.sc != 0 || model->Rs.sc != 0 || model->cs.sc != 0))) {
        set_error("yobs and the observation model (Hs, Rs, cs) are shared by the chains of a sweep: their chain strides must be 0");
        return AUXSSM_ERR_ARG;
    }
    const double* dptr = nullptr;
    if (delta_dev) {  // {delta, sqrt(delta / 2)} for the kernels of this sweep, on the stream: nothing returns to the host
        if (!h->dblock) AX_HIP(hipMalloc((void**)&h->dblock, 2 * sizeof(double)));
        if (dtype == AUXSSM_F32) hipLaunchKernelGGL((k_delta_block<float>), dim3(1), dim3(1), 0, h->stream, (const float*)delta_dev, h->dblock);
        else hipLaunchKernelGGL((k_delta_block<double>), dim3(1), dim3(1), 0, h->stream, (const double*)delta_dev, h->dblock);
        dptr = h->dblock;
    }
    if (keys && (lorenz || sv)) {  // these sweeps read the noise from the arrays: draw all of it first (auxssm_kalman_draw)
        const long long nn = (long long)dims->C * dims->T * dims->dx;
        if (dtype == AUXSSM_F32) launch_rng_sweep<float>(h, keys, nn, dims->C, const_cast<void*>(eps_aux), const_cast<void*>(eps_samp), const_cast<void*>(u_acc));
        else launch_rng_sweep<double>(h, keys, nn, dims->C, const_cast<void*>(eps_aux), const_cast<void*>(eps_samp), const_cast<void*>(u_acc));
    }
    if (lorenz) {
        rc = dtype == AUXSSM_F32 ? sweep_lorenz<float>(h, dtype, dims, model, yobs, delta, dptr, parallel, nan_policy, layout, x, eps_aux, eps_samp, u_acc, accepted, logs)
                                 : sweep_lorenz<double>(h, dtype, dims, model, yobs, delta, dptr, parallel, nan_policy, layout, x, eps_aux, eps_samp, u_acc, accepted, logs);
    } else if (sv) {
        const int order = model_kind == AUXSSM_KMODEL_SV_FIRST ? 1 : 2;
        rc = dtype == AUXSSM_F32 ? sweep_sv<float>(h, dtype, order, dims, model, yobs, delta, dptr, parallel, nan_policy, layout, x, eps_aux, eps_samp, u_acc, accepted, logs)
                                 : sweep_sv<double>(h, dtype, order, dims, model, yobs, delta, dptr, parallel, nan_policy, layout, x, eps_aux, eps_samp, u_acc, accepted, logs);
    } else {
        rc = dtype == AUXSSM_F32 ? sweep_lg_concat<float>(h, dtype, dims, model, yobs, delta, dptr, keys, parallel, nan_policy, layout, x, eps_aux, eps_samp, u_acc, accepted, logs)
                                 : sweep_lg_concat<double>(h, dtype, dims, model, yobs, delta, dptr, keys, parallel, nan_policy, layout, x, eps_aux, eps_samp, u_acc, accepted, logs);
    }
    return rc;
}

int auxssm_kalman_sweep(auxssm_handle h, int dtype, int model_kind, const auxssm_dims* dims, const auxssm_lgssm* model,
                        const auxssm_arr* yobs, double delta, int parallel, int nan_policy, int layout, void* x, const void* eps_aux,
                        const void* eps_samp, const void* u_acc, int32_t* accepted, void* logs) {
    return kalman_sweep_impl(h, dtype, model_kind, dims, model, yobs, delta, nullptr, nullptr, parallel, nan_policy, layout, x, eps_aux, eps_samp, u_acc,
                             accepted, logs);
}
int auxssm_kalman_sweep_dd(auxssm_handle h, int dtype, int model_kind, const auxssm_dims* dims, const auxssm_lgssm* model,
                           const auxssm_arr* yobs, const void* delta_dev, int parallel, int nan_policy, int layout, void* x, const void* eps_aux,
                           const void* eps_samp, const void* u_acc, int32_t* accepted, void* logs) {
    if (!delta_dev) {
        set_error("delta_dev must be a device pointer to one scalar of `dtype`");
        return AUXSSM_ERR_ARG;
    }
    return kalman_sweep_impl(h, dtype, model_kind, dims, model, yobs, 1.0, delta_dev, nullptr, parallel, nan_policy, layout, x, eps_aux, eps_samp, u_acc,
                             accepted, logs);
}
int auxssm_kalman_sweep_keyed(auxssm_handle h, int dtype, int model_kind, const auxssm_dims* dims, const auxssm_lgssm* model,
                              const auxssm_arr* yobs, double delta, const void* delta_dev, const uint32_t* keys, int parallel, int nan_policy, int layout,
                              void* x, void* eps_aux, void* eps_samp, void* u_acc, int32_t* accepted, void* logs) {
    if (!keys) {
        set_error("keys must point at six uint32: {aux0, aux1, samp0, samp1, acc0, acc1}");
        return AUXSSM_ERR_ARG;
    }
    return kalman_sweep_impl(h, dtype, model_kind, dims, model, yobs, delta_dev ? 1.0 : delta, delta_dev, keys, parallel, nan_policy, layout, x, eps_aux,
                             eps_samp, u_acc, accepted, logs);
}

int auxssm_kalman_sweep_fused(auxssm_handle h, int dtype, int model_kind, const auxssm_dims* dims, const auxssm_lgssm* model, const auxssm_arr* yobs, double delta,
                              const void* delta_dev, const uint32_t* keys, int parallel, int nan_policy, int layout, void* x, void* x_alt, int32_t* sel, void* u_acc,
                              int32_t* accepted, void* logs) {
    AX_NEED_H(h);
    int rc;
    if ((rc = check_dtype(dtype)) || (rc = check_dims(dims, true))) return rc;
    if (model_kind != AUXSSM_KMODEL_LG_CONCAT) {
        set_error("auxssm_kalman_sweep_fused runs AUXSSM_KMODEL_LG_CONCAT (model_kind %d: use auxssm_kalman_sweep_keyed)", model_kind);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    if ((rc = check_lgssm(model, dims->T))) return rc;
    if (dims->B != 1 || !(delta_dev || delta > 0) || !keys || !yobs || !yobs->ptr || !x || !x_alt || !u_acc || !accepted ||
        (nan_policy != AUXSSM_NAN_REFERENCE && nan_policy != AUXSSM_NAN_MASKED)) {
        set_error("auxssm_kalman_sweep_fused: B must be 1, delta > 0 (or delta_dev), keys / yobs / x / x_alt / u_acc / accepted non-NULL, nan_policy 0 or 1");
        return AUXSSM_ERR_ARG;
    }
    if (yobs->sc != 0 || model->Hs.sc != 0 || model->Rs.sc != 0 || model->cs.sc != 0) {
        set_error("yobs and the observation model (Hs, Rs, cs) are shared by the chains of a sweep: their chain strides must be 0");
        return AUXSSM_ERR_ARG;
    }
    if (const char* why = fused_refusal(h, dims, model, parallel, layout)) {  // (nothing has been enqueued: the caller runs auxssm_kalman_sweep_keyed instead)
        set_error("auxssm_kalman_sweep_fused: %s", why);
        return AUXSSM_ERR_UNSUPPORTED;
    }
    const double* dptr = nullptr;
    if (delta_dev) {
        if (!h->dblock) AX_HIP(hipMalloc((void**)&h->dblock, 2 * sizeof(double)));
        if (dtype == AUXSSM_F32) hipLaunchKernelGGL((k_delta_block<float>), dim3(1), dim3(1), 0, h->stream, (const float*)delta_dev, h->dblock);
        else hipLaunchKernelGGL((k_delta_block<double>), dim3(1), dim3(1), 0, h->stream, (const double*)delta_dev, h->dblock);
        dptr = h->dblock;
    }
    const double dl = delta_dev ? 1.0 : delta;
    return dtype == AUXSSM_F32 ? sweep_lg_concat_fused<float>(h, dtype, dims, model, yobs, dl, dptr, keys, nan_policy, x, x_alt, sel, u_acc, accepted, logs)
                               : sweep_lg_concat_fused<double>(h, dtype, dims, model, yobs, dl, dptr, keys, nan_policy, x, x_alt, sel, u_acc, accepted, logs);
}
int auxssm_kalman_state_resolve(auxssm_handle h, int dtype, const auxssm_dims* dims, void* x, const void* x_alt, int32_t* sel) {
    AX_NEED_H(h);
    int rc;
    if ((rc = check_dtype(dtype)) || (rc = check_dims(dims, false))) return rc;
    if (!x || !x_alt || !sel) {
        set_error("x, x_alt and sel must be non-NULL");
        return AUXSSM_ERR_ARG;
    }
    const long long rows = (long long)dims->T * dims->dx;
    const int C = dims->C;
    if (rows > 0x7fffffffLL || (C + 255) / 256 > 65535) {
        set_error("state too large for one resolve launch");
        return AUXSSM_ERR_ARG;
    }
    if (dtype == AUXSSM_F32) hipLaunchKernelGGL((k_fs_resolve<float>), dim3((unsigned)rows, (unsigned)((C + 255) / 256)), dim3(256), 0, h->stream, C, (const int32_t*)sel, (const float*)x_alt, (float*)x);
    else hipLaunchKernelGGL((k_fs_resolve<double>), dim3((unsigned)rows, (unsigned)((C + 255) / 256)), dim3(256), 0, h->stream, C, (const int32_t*)sel, (const double*)x_alt, (double*)x);
    AX_HIP(hipMemsetAsync(sel, 0, (size_t)C * sizeof(int32_t), h->stream));
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}

static int rng_fill(auxssm_handle h, int dtype, bool normal, uint32_t k0, uint32_t k1, uint32_t stream, int64_t n, void* out) {
    AX_NEED_H(h);
    int rc;
    if ((rc = check_dtype(dtype))) return rc;
    if (n < 0 || !out) {
        set_error("n must be >= 0 and out non-NULL");
        return AUXSSM_ERR_ARG;
    }
    if (n == 0) return AUXSSM_OK;
    const long long work = (n + 1) / 2;
    const unsigned grid = (unsigned)((work + 255) / 256);
    ProfScope ps(h, AUXSSM_K_RNG);
    if (dtype == AUXSSM_F32) {
        if (normal) hipLaunchKernelGGL((k_rng_normal<float>), dim3(grid), dim3(256), 0, h->stream, k0, k1, stream, (long long)n, (float*)out);
        else hipLaunchKernelGGL((k_rng_uniform<float>), dim3(grid), dim3(256), 0, h->stream, k0, k1, stream, (long long)n, (float*)out);
    } else {
        if (normal) hipLaunchKernelGGL((k_rng_normal<double>), dim3(grid), dim3(256), 0, h->stream, k0, k1, stream, (long long)n, (double*)out);
        else hipLaunchKernelGGL((k_rng_uniform<double>), dim3(grid), dim3(256), 0, h->stream, k0, k1, stream, (long long)n, (double*)out);
    }
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}
int auxssm_kalman_draw(auxssm_handle h, int dtype, const uint32_t* keys, int64_t n, int64_t nu, void* eps_aux, void* eps_samp, void* u_acc) {
    AX_NEED_H(h);
    int rc;
    if ((rc = check_dtype(dtype))) return rc;
    if (!keys || n < 1 || nu < 1 || !eps_aux || !eps_samp || !u_acc) {
        set_error("keys/eps_aux/eps_samp/u_acc must be non-NULL and n, nu >= 1");
        return AUXSSM_ERR_ARG;
    }
    const unsigned g1 = (unsigned)(((n + 1) / 2 + 255) / 256), g2 = (unsigned)(((nu + 1) / 2 + 255) / 256);
    ProfScope ps(h, AUXSSM_K_RNG);
    if (dtype == AUXSSM_F32)
        hipLaunchKernelGGL((k_rng_sweep<float>), dim3(2 * g1 + g2), dim3(256), 0, h->stream, keys[0], keys[1], keys[2], keys[3], keys[4], keys[5],
                           (long long)n, (long long)nu, g1, (float*)eps_aux, (float*)eps_samp, (float*)u_acc);
    else
        hipLaunchKernelGGL((k_rng_sweep<double>), dim3(2 * g1 + g2), dim3(256), 0, h->stream, keys[0], keys[1], keys[2], keys[3], keys[4], keys[5],
                           (long long)n, (long long)nu, g1, (double*)eps_aux, (double*)eps_samp, (double*)u_acc);
    AX_HIP(hipGetLastError());
    return AUXSSM_OK;
}
int auxssm_rng_normal(auxssm_handle h, int dtype, uint32_t key0, uint32_t key1, uint32_t stream, int64_t n, void* out) {
    return rng_fill(h, dtype, true, key0, key1, stream, n, out);
}
int auxssm_rng_uniform(auxssm_handle h, int dtype, uint32_t key0, uint32_t key1, uint32_t stream, int64_t n, void* out) {
    return rng_fill(h, dtype, false, key0, key1, stream, n, out);
}

}  // extern "C"

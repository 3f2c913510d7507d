// ctx.h -- handle, workspace and type-erased launch records shared by api.hip and the instantiation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/auxssm.h"
#include "kalman_bodies.h"

namespace ax {

void set_error(const char* fmt, ...);

#define AX_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            ax::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return AUXSSM_ERR_HIP;                                                           \
        }                                                                                    \
    } while (0)

struct Prof {
    int kernel_id = 0;  // AUXSSM_K_ALL (-1): every kernel group is bracketed, each slot remembers its group
    int max_launches = 0;
    int used = 0;
    std::vector<hipEvent_t> start, stop;
    std::vector<int> ids;
};

}  // namespace ax

struct auxssm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // workspace: one slab, bump-allocated per API call, grown on demand
    char* ws = nullptr;
    size_t ws_bytes = 0;
    size_t ws_off = 0;
    ax::Prof prof;
    int num_cu = 256;
    int share_model = 1;  // AUXSSM_OPT_SHARE_MODEL: hoist whatever depends only on chain-shared model parameters out of the chain loop
    // auxssm_stats_attach: running moments folded in by the accept/select step of every Kalman sweep (layout of x); iter = sweeps folded so far
    void* st_sq_jump = nullptr;
    void* st_mean = nullptr;
    void* st_sq_mean = nullptr;
    long long st_iter = 0;
    // ... bound to ONE resident state: its base pointer, element count and dtype; a sweep on any other state is refused while attached
    const void* st_x = nullptr;
    long long st_n = 0;
    int st_dtype = -1;
    unsigned long long api_calls = 0;  // entry points that may enqueue work or change device data (not: sync, device-to-host copies, profiler reads)
    int overlap_model_stage = 0;       // AUXSSM_OPT_OVERLAP_MODEL_STAGE (off unless the caller opts in: include/auxssm.h)
    bool stream_exposed = false;       // auxssm_stream() has handed out `stream`: work the library cannot see may be queued on it, so a model stage
                                       // always waits for the tail of `stream` from then on (side_open)
    // a helper stream for memory-bound side work of ONE call that needs none of the call's later results (the covariance broadcast of the chain-shared wide
    // filter): forked from `stream` by an event, joined back before the call returns -- never visible to the caller (lazily created)
    hipStream_t fork_stream = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    // the wide-state cSMC's model block [m0 | chol P0 | 1/diag | F | b | chol Q | 1/diag] lives on the handle: uploaded (one blocking copy behind the stream's tail)
    // only when its CONTENT changes, so a sampling loop with a fixed model never synchronises with the host
    void* cw_dev = nullptr;
    size_t cw_dev_bytes = 0;
    std::vector<unsigned char> cw_host;  // {dtype, D, the block as uploaded}
    double* dblock = nullptr;  // {delta, sqrt(delta / 2)} of a sweep whose step size is device-resident (auxssm_kalman_sweep_dd); lazily allocated
    // The MODEL STAGE of a chain-shared sweep -- concatenated observation model, matrix filter on one sequence, gain table: ~0.4 ms of short dependent
    // launches that read the model and the step size only, never a chain -- runs on a second stream with its own double-buffered slab, so that the
    // stage of sweep k + 1 overlaps the chain passes of sweep k (api.hip: side_*).  Joined into `stream` before its first consumer.
    // Round 3: a RING of NS slabs, each with its own stream -- the stage of sweep k + 2 may start as soon as sweep k - 1 has finished with its slab, so a
    // stage has two sweeps of slack instead of one (beside full-chip passes its dozen short dependent launches take 1.3 ms, as long as the passes of
    // the fused sweep themselves) and consecutive stages overlap each other.
    struct SideStage {
        static constexpr int NS = 3;
        hipStream_t streams[NS] = {nullptr, nullptr, nullptr};
        hipStream_t stream = nullptr;                  // streams[parity] of the open stage (null until the first stage)
        hipEvent_t done[NS] = {nullptr, nullptr, nullptr};       // stage of this slab finished (recorded on its stream)
        hipEvent_t sweep_end[NS] = {nullptr, nullptr, nullptr};  // last sweep that read this slab finished enqueueing (recorded on `stream` of the handle)
        hipEvent_t begun[NS] = {nullptr, nullptr, nullptr};      // this slab's stage got past its waits (recorded on its stream): the NEXT stage starts after it,
        bool begun_valid = false;                                 // so a fence one stage waited for (new data behind a foreign call) orders every later stage too
        bool end_valid[NS] = {false, false, false};
        char* ws[NS] = {nullptr, nullptr, nullptr};
        size_t bytes[NS] = {0, 0, 0};
        int parity = 0;                                // slab of the open (or last) stage
        bool open = false, inside = false;
        // A stage runs ahead of whatever `stream` still holds.  That is safe behind another sweep (a sweep writes none of a stage's inputs), not
        // behind anything else the caller may have enqueued through the handle (an upload of new parameters, a memset, another kind of call): every
        // entry point counts itself in auxssm_ctx::api_calls, and a stage that is not opened by the call right after the last staged sweep first
        // waits for the tail of `stream` (one sweep without overlap).
        // MODEL-STAGE MEMO (round 4).  The stage of the fused chain-shared sweep reads the model arrays, the data and the step size only.  In the sampling phase of a
        // run (fixed step size, fixed model) it rebuilt the same tables every sweep -- ~12 short dependent launches that take 0.7 ms alone and 1.3 ms beside full-chip
        // passes: as long as the sweep itself.  Each slab now remembers WHAT it was built from: a host key (shapes, pointers, strides, step size, options; memo_key) and a
        // device snapshot of the input arrays' bytes at the head of the slab.  A later stage on the same slab with the same key launches one comparison kernel
        // (inputs vs snapshot, bit patterns) that sets the slab's `rebuild` word, and every stage kernel returns at once when it is 0 (kalman_bodies.h::memo_skip).
        // Exact: the tables are reused only when their inputs are byte for byte the same; nothing is trusted to the caller; no host synchronisation.
        // `uses` counts the stages a slab has hosted; memo_use is the count at which its tables were last built or validated: a stage of another kind (SV, keyed
        // sweep) in between breaks the chain and forces a rebuild.
        std::vector<unsigned char> memo_key[NS];
        long long uses[NS] = {0, 0, 0};
        long long memo_use[NS] = {-1, -1, -1};
        const void* last_tab = nullptr;  // gain rows the open stage built (run_filter_shared), for a second filter of the same sweep
        unsigned long long last_call = 0;
        hipEvent_t fence = nullptr;
        size_t off = 0;                                // bump offset inside the open slab
        hipStream_t m_stream = nullptr;                // the main context while `inside`
        char* m_ws = nullptr;
        size_t m_bytes = 0, m_off = 0;
    } side;
};

namespace ax {

// Reserve the slab for one API call (sum of the call's needs), then carve from it.
int ws_reserve(auxssm_ctx* h, size_t bytes);
void* ws_take(auxssm_ctx* h, size_t bytes);

// model stage on the side stream (see auxssm_ctx::SideStage).  side_open: next parity, slab of at least `need` bytes, the side stream waits for the
// last sweep that read it; SideScope: launches and ws_take inside the scope go to the side stream / slab (no-op unless a stage is open);
// side_close: `stream` waits for the stage; side_sweep_end: marks the end of the sweep that consumed it.
// dnc.hip: the reference's divide-and-conquer pathwise sampler (dx <= 4, B = 1)
int run_dnc(auxssm_ctx* h, int dtype, int C, int T, int D, const Arr& Fs, const Arr& Qs, const Arr& bs, const Arr& ms, const Arr& Ps, const void* eps, void* xs);
int side_open(auxssm_ctx* h, size_t need);
int side_close(auxssm_ctx* h);
void side_sweep_end(auxssm_ctx* h);
struct SideScope {
    auxssm_ctx* h;
    bool on;
    explicit SideScope(auxssm_ctx* h_) : h(h_), on(h_->side.open && !h_->side.inside) {
        if (!on) return;
        auxssm_ctx::SideStage& s = h->side;
        s.m_stream = h->stream; s.m_ws = h->ws; s.m_bytes = h->ws_bytes; s.m_off = h->ws_off;
        h->stream = s.stream; h->ws = s.ws[s.parity]; h->ws_bytes = s.bytes[s.parity]; h->ws_off = s.off;
        s.inside = true;
    }
    ~SideScope() {
        if (!on) return;
        auxssm_ctx::SideStage& s = h->side;
        s.off = h->ws_off;
        h->stream = s.m_stream; h->ws = s.m_ws; h->ws_bytes = s.m_bytes; h->ws_off = s.m_off;
        s.inside = false;
    }
};

struct ProfScope {
    auxssm_ctx* h;
    int slot;
    ProfScope(auxssm_ctx* h_, int kernel_id) : h(h_), slot(-1) {
        Prof& p = h->prof;
        if ((p.kernel_id == kernel_id || p.kernel_id == AUXSSM_K_ALL) && p.used < p.max_launches) {
            slot = p.used++;
            p.ids[slot] = kernel_id;
            (void)hipEventRecord(p.start[slot], h->stream);
        }
    }
    ~ProfScope() {
        if (slot >= 0) (void)hipEventRecord(h->prof.stop[slot], h->stream);
    }
};

// The chain-minor sweep treats a model as chain-shared (the filter's gain-form path of affine_shared.h, the sampler's gain tables, the
// log-density factor tables) under exactly these conditions; callers use it to lay the (then chain-independent) covariances out once.
inline bool chain_shared_mode(const auxssm_ctx* h, int cm, int C, int T, bool params_chain_stride_0) {
    return h->share_model != 0 && cm != 0 && C > 1 && T > 1 && params_chain_stride_0;
}

// ---- type-erased launch entry points, one set per (dtype, D) instantiation unit ---------------------
struct ScanPlan {
    int E;       // elements per thread chunk
    int nchunk;  // chunks per sequence
};
// waves: how many waves per SIMD the scan passes of the caller's operator can keep resident (1: the fp64 d = 4 operators; the chain-minor d = 1, 2 operators
// hold 4 / 2 and are issue-bound, so shorter chunks -- more lanes -- pay until the chip holds that many; kernels.hip.h::scan_waves)
constexpr int SCAN_WAVES_MAX = 4;
ScanPlan plan_scan(const auxssm_ctx* h, int S, int n, int parallel, int waves = 1);
struct AffPlan {  // chunking of the chain-shared affine scans (kernels.hip.h: run_affine)
    int E, nchunk;
};
AffPlan plan_aff(const auxssm_ctx* h, int S, int N, int parallel, int waves = 8);

typedef int (*filter_fn)(auxssm_ctx*, const FilterArgs&, int parallel, void* ell_out /*[C]*/);
typedef int (*sample_fn)(auxssm_ctx*, const SampleArgs&, int parallel);
typedef int (*logpdf_fn)(auxssm_ctx*, const LogpdfArgs&, void* out /*[C]*/);
typedef size_t (*filter_ws_fn)(const auxssm_ctx*, const KDims&, int parallel);
typedef size_t (*sample_ws_fn)(const auxssm_ctx*, const KDims&, int parallel);
typedef size_t (*logpdf_ws_fn)(const auxssm_ctx*, const KDims&);

struct KalmanEntry {
    filter_fn filter;
    filter_ws_fn filter_ws;
    logpdf_fn logpdf;
    logpdf_ws_fn logpdf_ws;
};
typedef int (*sweep_logpdf_fn)(auxssm_ctx*, const SweepLogpdfArgs&, void* out /*[5][C]*/);
typedef size_t (*sweep_logpdf_ws_fn)(const auxssm_ctx*, const KDims&);
// host-side description of one fused chain-shared sweep (fused_shared.h::run_fused_shared; built by api.hip)
struct FusedHost {
    FilterArgs fa;        // the concatenated model; ys = row 0 of [u_0 ; yobs_0] (chain-minor, one row), ms = the (D, C) slot of the t = 0 mean, Ps shared
    SampleArgs sa;        // Fs, Qs, bs, Ps (shared)
    SweepLogpdfArgs la;   // the real observation model, yobs, u (row 0 is read), delta / shd / dptr / nan_policy
    const void* xa;
    void* xb;
    int32_t* sel;         // null: xa is the state, xb receives the proposals
    void* u;
    void* inc;
    const uint32_t* keys;  // {aux0, aux1, samp0, samp1, acc0, acc1}
    const void* eps0s;     // row 0 of eps_samp
    const void* u_acc;
    int32_t* accepted;
    void* logs;
    const int* memo = nullptr;  // model-stage memo of this sweep's stage (ctx.h::SideStage), or null
};
typedef int (*fused_fn)(auxssm_ctx*, FusedHost&);
typedef size_t (*fused_ws_fn)(const auxssm_ctx*, const KDims&);
struct SweepLogpdfEntry {
    sweep_logpdf_fn run;
    sweep_logpdf_ws_fn ws;
    sweep_logpdf_fn lorenz = nullptr;  // the Lorenz-63 sweep's fused pass (dx = 3 units only); same workspace as `run`
    fused_fn fused = nullptr;          // the chain-shared LG_CONCAT sweep in three streaming passes (fused_shared.h); null in the wide-state entry
    fused_ws_fn fused_ws = nullptr;
    sweep_logpdf_fn wide_shared = nullptr;  // wide-state entry only: the chains-as-columns form alone (returns 1 when it does not apply, nothing enqueued)
};
typedef int (*sv_logpdf_fn)(auxssm_ctx*, const SvLogpdfArgs&, void* out /*[5][C]*/);
typedef size_t (*sv_logpdf_ws_fn)(const auxssm_ctx*, const KDims&);
struct SampleEntry {
    sample_fn sample;
    sample_ws_fn sample_ws;
    sv_logpdf_fn sv_logpdf = nullptr;  // the SV sweep's fused log-density pass (register kernels only; null in the wide-state entry)
    sv_logpdf_ws_fn sv_logpdf_ws = nullptr;
};

constexpr int MAX_D = 4;
constexpr int MAX_P = 8;

// defined by the instantiation units (inst_*.hip); nullptr entries = not built
const KalmanEntry* kalman_entry(int dtype, int D, int P);
const SampleEntry* sample_entry(int dtype, int D);
const SweepLogpdfEntry* sweep_logpdf_entry(int dtype, int D, int PO);

// wide.hip: one workgroup per time step / scan element, matrices in LDS -- every (dx, dy) the register kernels do not cover.
// wide_fits() says whether the LDS plan of the largest kernel fits this device for (dtype, dx, dy); why = message if not.
const KalmanEntry* wide_kalman_entry(int dtype);
const SampleEntry* wide_sample_entry(int dtype);
const SweepLogpdfEntry* wide_sweep_logpdf_entry(int dtype);
bool wide_fits(int dtype, int dx, int dy, std::string* why);
// the wide entries' *_ws members return 0 (the table signatures carry no sizes): api.hip asks through these instead
size_t wide_filter_ws(const auxssm_ctx* h, int dtype, const KDims& kd, int parallel, int d, int p);
size_t wide_sample_ws(const auxssm_ctx* h, int dtype, const KDims& kd, int parallel, int d);
size_t wide_logpdf_ws(int dtype, const KDims& kd);
size_t wide_gain_tab_bytes(int dtype, int T, int d, int p);  // the chain-shared wide filter's per-step gain rows (FilterArgs::pc)

}  // namespace ax

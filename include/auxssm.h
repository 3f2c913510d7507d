/*
 * auxssm.h -- C ABI of libauxssm.so: MI355X (gfx950) native auxiliary-Kalman / conditional-SMC hot path.
 *
 * This is the drop-in boundary for AdrienCorenflos/aux-ssm-samplers.  The reference has no FFI of its
 * own (it is pure Python on JAX); every entry point below replaces the body of one reference function
 * and is what a ctypes binding of that function binds.  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.  All functions return 0 on success and a
 *     negative auxssm_status otherwise; auxssm_last_error() returns the message for the calling thread.
 *   - Data pointers of compute entry points are DEVICE pointers (hipMalloc'd, or from auxssm_malloc).
 *     The library never frees caller memory; scratch lives in the handle's workspace, grown on demand.
 *   - One handle = one HIP device + one stream.  Calls on one handle are serialised by the caller and are
 *     asynchronous w.r.t. the host unless stated; auxssm_sync() waits for the handle's stream.
 *   - Shapes.  C = independent chains, T = time steps, B = batched independent LGSSMs (reference
 *     _primitives/kalman/base.py:40-49), dx = state dim, dy = observation dim.  Every array is described by
 *     an auxssm_arr {ptr, chain stride, time stride, batch stride} in ELEMENTS; the trailing matrix/vector
 *     axes are dense row-major.  A stride of 0 broadcasts (e.g. model parameters shared by all chains, or
 *     time-invariant F).  The natural dense layout is (C, T, B, ...).
 *   - dtype: AUXSSM_F32 or AUXSSM_F64 for every real array of a call; ancestors are int32.
 */
#ifndef AUXSSM_H
#define AUXSSM_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUXSSM_VERSION 108

typedef struct auxssm_ctx* auxssm_handle;

typedef enum { AUXSSM_F32 = 0, AUXSSM_F64 = 1 } auxssm_dtype;

typedef enum {
    AUXSSM_OK = 0,
    AUXSSM_ERR_ARG = -1,      /* bad argument / shape (Python glue raises ValueError) */
    AUXSSM_ERR_UNSUPPORTED = -2, /* (dx, dy) not instantiated in this build */
    AUXSSM_ERR_HIP = -3,      /* HIP runtime error */
    AUXSSM_ERR_NOMEM = -4
} auxssm_status;

/* nansum policy of log_likelihood (reference base.py:137-166): REFERENCE drops a whole time step whose
 * residual has any non-finite component (what jnp.nansum over per-step logpdfs does); MASKED scores the
 * finite components of a partially observed step. */
typedef enum { AUXSSM_NAN_REFERENCE = 0, AUXSSM_NAN_MASKED = 1 } auxssm_nan_policy;

typedef struct {
    const void* ptr;
    int64_t sc; /* chain stride  (elements) */
    int64_t st; /* time stride   (elements) */
    int64_t sb; /* batch stride  (elements) */
} auxssm_arr;

/* LGSSM parameter bundle == reference `LGSSM` NamedTuple (_primitives/kalman/base.py:12-69).
 * m0 (dx) P0 (dx,dx) [no time axis: st ignored]; Fs,Qs (T-1,dx,dx) bs (T-1,dx); Hs (T,dy,dx) Rs (T,dy,dy) cs (T,dy). */
typedef struct {
    auxssm_arr m0, P0, Fs, Qs, bs, Hs, Rs, cs;
} auxssm_lgssm;

typedef struct {
    int32_t C, T, B, dx, dy;
} auxssm_dims;

/* ---- lifecycle ------------------------------------------------------------------------------- */
int auxssm_version(void);
const char* auxssm_last_error(void);
int auxssm_device_count(int* count);
int auxssm_create(int device, auxssm_handle* out);
int auxssm_destroy(auxssm_handle h);
int auxssm_sync(auxssm_handle h);
/* Options of a handle.
 * AUXSSM_OPT_SHARE_MODEL (default 1; environment AUXSSM_SHARED=0 changes the default): in the chain-minor sweep, when the
 *   model's parameter arrays do not depend on the chain (chain stride 0: the factories of a linear-Gaussian model ignore the
 *   linearisation point), everything that depends on the parameters only -- element matrices, gains, Cholesky factors of Q_t /
 *   R_t, the filtered covariances -- is computed once per time step and sweep instead of once per chain (what jax.vmap leaves
 *   unbatched in the reference).  0 forces the general per-chain path.  Results agree to rounding.
 * AUXSSM_OPT_OVERLAP_MODEL_STAGE (default 0 = every call ordered on the one stream; environment AUXSSM_OVERLAP_TAB=1 / 0 changes the default;
 *   the Python host layer, whose every device write goes through an auxssm_* call, switches it on): a chain-shared sweep with a host step
 *   size runs its MODEL STAGE -- the concatenated observation model, the matrix filter on one sequence, the gain / sampler / log-density
 *   tables: everything that reads the model and the step size only -- on a second stream with a double-buffered workspace, so that the stage
 *   of one sweep overlaps the chain passes of the sweep before (same results bit for bit).  The stage runs ahead of earlier SWEEPS only: when
 *   any other call went through the handle since the last such sweep, or once auxssm_stream() has handed the raw stream to the caller (who
 *   may queue work on it the library cannot see), the stage first waits for the tail of the stream.  Still outside the library's sight with
 *   the option on: model arrays written by ANOTHER stream or handle without host synchronisation -- call auxssm_sync first, or leave it 0. */
typedef enum { AUXSSM_OPT_SHARE_MODEL = 1, AUXSSM_OPT_OVERLAP_MODEL_STAGE = 2 } auxssm_option;
int auxssm_set_option(auxssm_handle h, int option, int value);
int auxssm_get_option(auxssm_handle h, int option, int* value);

/* the hipStream_t the handle launches on (as an opaque pointer) */
int auxssm_stream(auxssm_handle h, void** stream);

/* ---- device memory owned by the caller --------------------------------------------------------- */
int auxssm_malloc(auxssm_handle h, size_t bytes, void** dptr);
int auxssm_free(auxssm_handle h, void* dptr);
int auxssm_memcpy_h2d(auxssm_handle h, void* dst, const void* src, size_t bytes); /* synchronous */
int auxssm_memcpy_d2h(auxssm_handle h, void* dst, const void* src, size_t bytes); /* synchronous */
int auxssm_memcpy_d2d(auxssm_handle h, void* dst, const void* src, size_t bytes); /* async on stream */
int auxssm_memset(auxssm_handle h, void* dst, int value, size_t bytes);           /* async on stream */

/* ---- in-library kernel timing with HIP events on the handle's stream ----------------------------
 * kernel_id: one of AUXSSM_K_*.  While enabled, every launch of that kernel group is bracketed by a pair of
 * events from a pool of `max_launches`; read() synchronises and returns the count and the summed ms.
 * AUXSSM_K_ALL brackets every group; read_groups() then returns launches[id] / total_ms[id] for id < n_ids
 * (the event pairs serialise nothing: groups run back to back on one stream anyway). */
typedef enum {
    AUXSSM_K_ALL = -1,
    AUXSSM_K_NONE = 0,
    AUXSSM_K_FILTER_INIT = 1,
    AUXSSM_K_FILTER_SCAN = 2, /* the three launches of the filter's associative scan, timed as one unit */
    AUXSSM_K_FILTER_ELL = 3,  /* chain-shared wide filter: the mean / log-likelihood scan of all sequences (else unused: the scan elements carry the log-likelihood) */
    AUXSSM_K_SAMPLE_INIT = 4,
    AUXSSM_K_SAMPLE_SCAN = 5,
    AUXSSM_K_LOGPDF = 6,
    AUXSSM_K_CSMC_FWD = 7,
    AUXSSM_K_CSMC_BWD = 8,
    AUXSSM_K_PIT_STITCH = 9, /* the log2(T) stitching launches of the parallel-in-time cSMC sweep, timed as one unit */
    AUXSSM_K_RNG = 10,       /* Threefry fills (auxssm_rng_*, auxssm_kalman_draw) */
    AUXSSM_K_SELECT = 11,    /* accept / select step (+ running moments) of the Kalman sweep */
    AUXSSM_K_FACTORY = 12,   /* device factories of a sweep: concatenated / linearised model, observation tables */
    AUXSSM_K_FILTER_TAB = 13,/* chain-shared model: the one-sequence matrix filter and the gain tables derived from it */
    AUXSSM_K_COUNT = 14
} auxssm_kernel_id;
int auxssm_prof_enable(auxssm_handle h, int kernel_id, int max_launches);
int auxssm_prof_read(auxssm_handle h, int* launches, double* total_ms);
int auxssm_prof_read_groups(auxssm_handle h, int n_ids, int* launches /* [n_ids] */, double* total_ms /* [n_ids] */);
int auxssm_prof_disable(auxssm_handle h);

/* ---- Kalman primitives --------------------------------------------------------------------------
 * auxssm_kalman_filter  == filtering(ys, lgssm, parallel)        (_primitives/kalman/filtering.py:18-46)
 *   ys (C,T,B,dy) -> ms (C,T,B,dx), Ps (C,T,B,dx,dx) dense outputs, ell (C) [summed over B, :43-45].
 *   parallel != 0: associative scan over T (filtering.py:49-63); 0: the same kernels run the scan with one
 *   chunk per sequence, i.e. the sequential recursion (filtering.py:66-79).  NaN observations = missing.
 *   Wide states (dx > 4 or dy > 8), parallel != 0, C * B >= 2 sequences whose parameter arrays (P0, Fs, Qs, bs, Hs, Rs, cs) all have chain
 *   and batch stride 0, AUXSSM_OPT_SHARE_MODEL on: the covariance recursion runs ONCE (sequence 0) and every sequence keeps the affine mean /
 *   log-likelihood recursion of the gain form (csrc/wide_shared.h).  That form needs all sequences to miss the same observations; the call
 *   checks it on the device and reads the 4-byte answer back (one stream synchronisation per call), falling back to the per-sequence scan
 *   when they differ.  Same results to rounding.
 */
int auxssm_kalman_filter(auxssm_handle h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* lgssm,
                         const auxssm_arr* ys, int parallel, void* ms, void* Ps, void* ell);

/* auxssm_kalman_sample  == sampling(key, ms, Ps, lgssm, parallel) (_primitives/kalman/sampling.py:11-40)
 *   with the N(0,I) draws given explicitly: eps (C,T,B,dx) dense == jax.random.normal(key, ms.shape) (:128).
 *   ms, Ps dense as produced by auxssm_kalman_filter.  -> xs (C,T,B,dx) dense. */
int auxssm_kalman_sample(auxssm_handle h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* lgssm,
                         const void* ms, const void* Ps, const void* eps, int parallel, void* xs);

/* auxssm_kalman_dnc_sample == dnc_sampling.sampling(key, ms, Ps, lgssm) (_primitives/kalman/dnc_sampling.py:17-77: the divide-and-conquer pathwise sampler --
 *   leaves _init_elems :128-137, pairwise combination _combination_operator_impl :104-118 with the odd interval carried up :140-169, the two ends :46-68, the
 *   mid-points level by level :70-86) with the N(0, I) draws given explicitly: eps (C,T,dx) dense, time index t is sampled with eps[:, t].  The reference calls it a
 *   proof of concept (:38-41); same distribution as auxssm_kalman_sample.  dx <= 4, B must be 1 (:42-43: AUXSSM_ERR_ARG).  -> xs (C,T,dx) dense.
 *   One stream synchronisation per call (the tree's index plan is uploaded). */
int auxssm_kalman_dnc_sample(auxssm_handle h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* lgssm, const void* ms, const void* Ps, const void* eps, void* xs);

/* auxssm_kalman_joint_logpdf == log_likelihood(ys, xs, lgssm) + prior_logpdf(xs, lgssm)
 *   (_primitives/kalman/base.py:99-166); posterior_logpdf (:72-96) is this minus ell.
 *   xs (C,T,B,dx) described by an auxssm_arr; out (C). */
int auxssm_kalman_joint_logpdf(auxssm_handle h, int dtype, const auxssm_dims* dims, const auxssm_lgssm* lgssm,
                               const auxssm_arr* ys, const auxssm_arr* xs, int nan_policy, void* out);

/* ---- one auxiliary-Kalman MH sweep, fully on device ------------------------------------------------
 * == kernel(key, state, delta) of aux_samplers.kalman.get_kernel (kalman/generic.py:53-76) for models whose
 * factories are the built-in device factory below, for C chains at once.
 *
 * Device factory AUXSSM_KMODEL_LG_CONCAT (linear-Gaussian model, auxiliary observations concatenated with
 * the real ones, the pattern of examples/lorenz/auxiliary_kalman.py:26-35):
 *   dynamics_factory(x)          -> (m0, P0, Fs, Qs, bs)  = `model` dynamics, independent of x
 *   observations_factory(x,u,d)  -> ys = [u_t ; y_t], Hs = [I ; Hobs_t], Rs = blkdiag(d/2 I, Robs_t), cs = [0 ; cobs_t]
 *   log_likelihood_fn(x)         -> prior_logpdf(x) + sum_t log N(y_t; Hobs_t x_t + cobs_t, Robs_t)
 * `model` holds the dynamics and the REAL observation model (dy = dims->dy = p_obs); yobs (T, p_obs).
 *
 * Noise is explicit (the parity contract): eps_aux, eps_samp (C,T,dx) ~ N(0,I), u_acc (C) ~ U[0,1).
 * x (C,T,dx) is updated in place; accepted (C) int32; logs (C,5) = log_alpha, lp_prop, lp_rev, lt_prop, lt_rev
 * (may be NULL).  B must be 1.
 *
 * layout: AUXSSM_LAYOUT_DENSE -- x, eps_aux, eps_samp are (C,T,dx) row-major and lanes run over time;
 *         AUXSSM_LAYOUT_CHAIN_MINOR -- they are (T,dx,C) row-major (chain index fastest).  Then lanes run over CHAINS: every
 *         per-chain buffer inside the sweep is [t][component][chain], so each wave access is one contiguous run and the
 *         chain-shared model parameters are wave-uniform loads; there is no transposition anywhere in the sweep.  This is the
 *         layout to use with >= 32 chains (bench.py); results are identical up to the combination tree of the scan.  LG_CONCAT and the
 *         SV and LORENZ63_EXT factories (dx <= 4) take it; the wide-state sizes (dx > 4) are dense only.
 * Chain-minor SV sweeps whose observation model differs per chain (second order; first order without chain-shared dynamics) materialise neither the
 * pseudo-observations nor scan elements: the passes fold every step from (x, u, y) (csrc/kalman_bodies.h::FilterOpFlySV).  Wide-state sweeps (dx > 4)
 * of C >= 2 chains on one model (chain stride 0 on every model array) keep ONE copy of the filtered covariances and build the sampler's gains / factors
 * and the log-densities' inverses once per time step, the chains as columns (csrc/wide_shared.h); no sweep synchronises with the host.
 */
typedef enum {
    AUXSSM_KMODEL_LG_CONCAT = 1,
    /* Stochastic volatility y_{t,k} ~ N(0, exp(x_{t,k})) on linear-Gaussian dynamics (`model`: m0, P0, Fs, Qs, bs; Hs/Rs/cs unused),
     * examples/stochastic_volatility/auxiliary_kalman.py:22-48: observations_factory(x, u, d) is the first-order
     * (ys = u + d/2 grad log g(x), R = d/2 I) or second-order (R = (-hess + 2/d I)^-1, ys = R (2u/d + grad - hess x)) auxiliary
     * observation set with H = I, c = 0, rebuilt at both linearisation points (x and x_prop) each sweep;
     * log_likelihood_fn(x) = prior_logpdf(x) + sum log g.  yobs (T, dx), dims->dy = dx; dense or (dx <= 4) chain-minor layout. */
    AUXSSM_KMODEL_SV_FIRST = 2,
    AUXSSM_KMODEL_SV_SECOND = 3,
    /* Stochastic Lorenz-63 (examples/lorenz/auxiliary_kalman.py:14-52, model.py:10-25): dynamics_factory(x) is the first-order
     * extended linearisation (linearisation.py:11-44, analytic Jacobian) of mean(x) = x + dt (phi_0(x) + theta * phi(x)) at every x_t,
     * rebuilt at both linearisation points each sweep; observations_factory concatenates the auxiliary and the real observations as
     * LG_CONCAT does; log_likelihood_fn(x) = log N(x_0; m0, P0) + sum log N(x_{t+1}; mean(x_t), Q) + nansum_t log N(y_t; H_t x_t + c_t, R_t).
     * `model`: m0, P0, Qs as usual; Fs.ptr -> DEVICE array [theta1, theta2, theta3, dt] of `dtype`, chain stride Fs.sc (0: one theta for all chains; 4: one row per chain) (bs unused); Hs, Rs, cs = the
     * real observation model (rows of unobserved steps may be NaN); yobs (T, dy) with NaN = missing; dx = 3, dy <= 3; dense or chain-minor layout. */
    AUXSSM_KMODEL_LORENZ63_EXT = 4
} auxssm_kalman_model;
typedef enum { AUXSSM_LAYOUT_DENSE = 0, AUXSSM_LAYOUT_CHAIN_MINOR = 1 } auxssm_layout;
int auxssm_kalman_sweep(auxssm_handle h, int dtype, int model_kind, const auxssm_dims* dims,
                        const auxssm_lgssm* model, const auxssm_arr* yobs, double delta, int parallel,
                        int nan_policy, int layout, void* x, const void* eps_aux, const void* eps_samp, const void* u_acc,
                        int32_t* accepted, void* logs);
/* The same sweep with a DEVICE-RESIDENT step size: delta_dev points at one scalar of `dtype` in device memory (e.g. the array
 * auxssm_delta_adapt updates, m = 1), read by the sweep's kernels on the handle's stream -- the adaptation loop of
 * the experiment drivers under examples/ (common.py:4-32 delta_adaptation between sweeps) then runs without any host round trip.  Results are bit-identical
 * to auxssm_kalman_sweep called with the same value.  delta_dev must hold a value > 0 (not checked: that would need a read-back). */
int auxssm_kalman_sweep_dd(auxssm_handle h, int dtype, int model_kind, const auxssm_dims* dims,
                           const auxssm_lgssm* model, const auxssm_arr* yobs, const void* delta_dev, int parallel,
                           int nan_policy, int layout, void* x, const void* eps_aux, const void* eps_samp, const void* u_acc,
                           int32_t* accepted, void* logs);
/* The sweep as the reference's kernel(key, state, delta) takes it: with the KEYS of its three draws instead of the drawn arrays
 * (keys = {aux0, aux1, samp0, samp1, acc0, acc1}, the three children of the sweep's key, kalman/generic.py:58).  Equivalent to
 * auxssm_kalman_draw(keys, ...) into (eps_aux, eps_samp, u_acc) followed by auxssm_kalman_sweep[_dd] on them -- the three buffers (caller-owned
 * scratch of x's size / C entries) hold exactly those values afterwards -- but the library may generate the noise inside its first
 * consumer instead of in a separate pass (chain-minor LG_CONCAT sweeps with chain-shared parameters do).  delta_dev != NULL: device-resident
 * step size as auxssm_kalman_sweep_dd (delta ignored). */
int auxssm_kalman_sweep_keyed(auxssm_handle h, int dtype, int model_kind, const auxssm_dims* dims,
                              const auxssm_lgssm* model, const auxssm_arr* yobs, double delta, const void* delta_dev,
                              const uint32_t* keys, int parallel, int nan_policy, int layout, void* x, void* eps_aux, void* eps_samp,
                              void* u_acc, int32_t* accepted, void* logs);

/* The keyed LG_CONCAT sweep for chain-shared model parameters in TWO streaming passes over the chains (csrc/fused_shared.h; three up to round 3): the draws, the
 * filter, the pathwise sampler and every log-density of the MH ratio (kalman/generic.py:53-106) are folded into a forward pass over x that emits the auxiliary
 * variables and the sampler's chunk-local increments, and a backward pass that emits x' -- 6 instead of 15 reads / writes of a (C, T, dx) array per sweep, no
 * eps_aux / eps_samp buffers at all.  The chain-independent model stage of the sweep is memoised on the device: rebuilt only when the model arrays, the data or the
 * step size differ byte for byte from what its tables were built from (no host synchronisation).  Same keys -> same draws as auxssm_kalman_sweep_keyed; x' and the five log terms agree with it to rounding
 * (the per-chain totals are summed in a different order).
 *   x, x_alt : two (T, dx, C) chain-minor buffers.
 *   sel == NULL : x is the state (in / out), x_alt scratch for the proposals (accepted chains are copied back by the usual select pass).
 *   sel != NULL : LAZY state -- chain c's current trajectory is x_alt[:, :, c] if sel[c] != 0 else x[:, :, c]; the sweep reads it from there,
 *                 writes the proposal to the other buffer and acceptance flips sel[c]: no select pass.  auxssm_kalman_state_resolve gathers the
 *                 state into x (and zeroes sel) before anything else reads x.
 * Returns AUXSSM_ERR_UNSUPPORTED -- before enqueueing anything -- when the sweep cannot run fused (other model kinds, dense layout, per-chain
 * parameters, AUXSSM_OPT_SHARE_MODEL off, odd chain count, T < 64, parallel == 0, dx > 4 or dy outside 1..4): the caller then runs
 * auxssm_kalman_sweep_keyed.  Running moments (auxssm_stats_attach) are NOT a refusal: attached to `x` they are folded by the sweep (one extra pass over
 * the buffer pair; a rejected chain folds a zero jump and its unchanged state), attached to any other state the call returns AUXSSM_ERR_ARG. */
int auxssm_kalman_sweep_fused(auxssm_handle h, int dtype, int model_kind, const auxssm_dims* dims, const auxssm_lgssm* model,
                              const auxssm_arr* yobs, double delta, const void* delta_dev, const uint32_t* keys, int parallel, int nan_policy,
                              int layout, void* x, void* x_alt, int32_t* sel, void* u_acc, int32_t* accepted, void* logs);
int auxssm_kalman_state_resolve(auxssm_handle h, int dtype, const auxssm_dims* dims, void* x, const void* x_alt, int32_t* sel);

/* ---- conditional SMC (particle Gibbs) sweep ------------------------------------------------------------
 * == kernel(key, state) of aux_samplers._primitives.csmc.get_kernel (csmc.py:16-66: forward pass _csmc :69-107 with
 * conditional multinomial resampling resamplings.py:14-37, then _backward_scanning_pass :110-124 or
 * _backward_sampling_pass :127-149), and -- with proposal = AUX_INDEPENDENT -- kernel(key, state, delta) of
 * aux_samplers.csmc.get_independent_kernel's classical branch (csmc/generic.py:56-72 + csmc/independent.py:57-75,
 * 143-169, 192-198, 238-248), for C chains at once.
 *
 * A Python Mt/Gt object cannot run inside a kernel; the Feynman-Kac model is a closed family instead:
 *   transition  x_t | x_{t-1} ~ N(F x_{t-1} + b, Q)   (time-invariant), initial N(m0, P0)
 *   potential   FLAT: 0 | GAUSS_OBS: log N(y_t; x_t, sig_y^2 I) | SV: sum_k log N(y_{t,k}; 0, exp(x_{t,k}))
 *   proposal    BOOTSTRAP_LG:    M0 = initial, Mt = transition, G0/Gt = potential          (test_csmc/common.py fixtures)
 *               AUX_INDEPENDENT: u = x + sqrt(delta_t/2) eps_aux; M0/Mt = N(u_t, delta_t/2 I);
 *                                G0 = log initial + potential, Gt = log transition + potential, Pt = transition
 * m0, chol_P0 (lower), F, b, chol_Q (lower) are small HOST arrays of doubles; y (T, dx) is a DEVICE array of `dtype`
 * shared by all chains; sqrt_half_delta (T) is a DEVICE array of `dtype` (AUX_INDEPENDENT only).
 *
 * Noise: EXPLICIT device arrays (the parity contract): eps_aux (C,T,dx) [AUX only], eps_prop (C,T,N,dx) ~ N(0,1) (row t
 * feeds M0 / Mt.sample at time t), u_res (C,T-1,N) ~ U[0,1) (row t-1 resamples into time t), u_bwd (C,T) (entry t draws
 * B_t; only entry T-1 is used when backward == 0); or THREEFRY: the same quantities generated in-kernel from
 * (key0, key1) with streams 1..4 of the auxssm_rng_normal/uniform counter space: eps_aux and u_bwd over the same flat
 * indices; eps_prop and u_res with two consecutive time steps sharing one Threefry block (index map in csrc/csmc.hip,
 * k_csmc_fwd), so a THREEFRY run equals an EXPLICIT run on the correspondingly permuted fills.
 *
 * x (C,T,dx): reference trajectories in, new trajectories out.  ancestors (C,T) int32 out (updated = ancestors != 0,
 * csmc.py:59).  xs_out (C,T,N,dx), log_ws_out (C,T,N), As_out (C,T-1,N) int32: optional full particle history
 * (NULL -> kept in the handle's workspace).  N <= 1024.  Arithmetic uses the fixed reduction orders and the
 * bit-reproducible exp/log documented in csrc/csmc.hip, so ancestors are bit-exact against oracle/csmc_ref.c. */
typedef enum { AUXSSM_PROP_BOOTSTRAP_LG = 0, AUXSSM_PROP_AUX_INDEPENDENT = 1 } auxssm_fk_proposal;
typedef enum {
    AUXSSM_POT_FLAT = 0,
    AUXSSM_POT_GAUSS_OBS = 1,        /* y_t ~ N(x_t, sig_y^2 I) */
    AUXSSM_POT_SV = 2,               /* y_{t,k} ~ N(0, exp(x_{t,k})) */
    AUXSSM_POT_GAUSS_OBS_MASKED = 3  /* y_{t,k} ~ N(x_{t,k}, sig_y^2) for the finite y_{t,k} only: missing components / steps skipped
                                        (examples/lorenz/model.py:43-56: x2, x3 observed every 80th step) */
} auxssm_fk_potential;
/* transition x_{t+1} | x_t ~ N(mean(x_t), Q): LINEAR mean = F x + b; LORENZ63_EM mean = x + dt (phi_0(x) + theta * phi(x)), the
 * Euler-Maruyama step of examples/lorenz/model.py:10-25 (dx = 3; theta = F[0..2], dt = b[0], Q = chol_Q chol_Q^T = dt sigma_x^2 I) */
typedef enum { AUXSSM_TRANS_LINEAR = 0, AUXSSM_TRANS_LORENZ63_EM = 1 } auxssm_fk_transition;
typedef enum { AUXSSM_NOISE_EXPLICIT = 0, AUXSSM_NOISE_THREEFRY = 1 } auxssm_noise_mode;
typedef struct {
    int32_t proposal, potential, dx, transition;
    const double* m0;      /* host (dx) */
    const double* chol_P0; /* host (dx,dx) lower */
    const double* F;       /* host (dx,dx) */
    const double* b;       /* host (dx) */
    const double* chol_Q;  /* host (dx,dx) lower */
    const void* y;         /* device (T,dx), may be NULL for FLAT */
    double sig_y;
    /* Time-varying LINEAR transitions (the reference scans Mt.params over time, _primitives/csmc/csmc.py:103, csmc/base.py:56-71):
     * DEVICE arrays of `dtype`, row t = the transition x_t -> x_{t+1}: F_t (T-1,dx,dx), b_t (T-1,dx), chol_Q_t (T-1,dx,dx) lower.
     * All three NULL: the time-invariant host F / b / chol_Q above; otherwise all three non-NULL (auxssm_csmc_sweep only). */
    const void* F_t;
    const void* b_t;
    const void* chol_Q_t;
    /* Gradient-informed independent proposals (csmc/independent.py:57-75 with gradient=True, :121-134, :173-190, :252-268), AUX_INDEPENDENT
     * only: x_t^i ~ N(u_t + delta_t/2 grad_t, delta_t/2 I) with grad = the gradient at u of the model's joint log-density
     * log M0(u_0) + G0(u_0) + sum_t [log Mt(u_{t+1} | u_t) + Gt(u_{t+1})], evaluated in closed form for the model family.  The weights get the
     * importance correction  sum_k [log N(x_k; u_k, s) - log N(x_k; u_k + delta/2 grad_k, s)]:
     *   AUXSSM_GRAD_REFERENCE  at t = 0 only -- the reference's GradientAuxiliaryGt sums its correction over ALL particles (jnp.sum without
     *                          an axis, :265-266), a constant that cancels in every normalisation, so for t >= 1 it is no correction at all;
     *   AUXSSM_GRAD_EXACT      at every step (the weights the construction intends). */
    int32_t gradient;
    int32_t reserved;
} auxssm_fk_model;
typedef enum { AUXSSM_GRAD_NONE = 0, AUXSSM_GRAD_REFERENCE = 1, AUXSSM_GRAD_EXACT = 2 } auxssm_fk_gradient;
typedef struct {
    int32_t mode;
    uint32_t key0, key1;
    int32_t reserved;
    const void* eps_aux;
    const void* eps_prop;
    const void* u_res;
    const void* u_bwd;
} auxssm_csmc_noise;
/* xs_out / log_ws_out / As_out NULL: the particle systems live in the handle's workspace, T N (dx + 1) reals (+ 4 T N bytes of traced ancestors)
 * per chain.  When those of all C chains exceed what the device has free, the forward + backward pair runs over batches of chains (whole rounds of
 * the chip: one workgroup per chain and CU) -- same results bit for bit, every array and random stream being indexed by the global chain;
 * AUXSSM_ERR_NOMEM only when a single chain does not fit. */
int auxssm_csmc_sweep(auxssm_handle h, int dtype, const auxssm_fk_model* model, int32_t C, int32_t T, int32_t N,
                      int32_t backward, const void* sqrt_half_delta, void* x, const auxssm_csmc_noise* noise,
                      int32_t* ancestors, void* xs_out, void* log_ws_out, int32_t* As_out);

/* ---- parallel-in-time conditional SMC (conditional dSMC) ------------------------------------------------------------
 * == kernel(key, state, delta) of aux_samplers.csmc.get_independent_kernel(..., parallel=True), classical branch
 * (csmc/independent.py:78-118 on _primitives/csmc/pit/csmc.py:69-114, operator.py, dc_map.py), for C chains at once: all T x N
 * proposals x_t^n ~ N(u_t, delta_t/2 I) are drawn at once (slot 0 = the reference trajectory), then a binary tree over time stitches
 * neighbouring blocks: N x N weights G_t(x_t^j, x_{t-1}^i) w_{t-1}^i w_t^j between the left block's last and the right block's first step,
 * N conditional multinomial draws of (i, j) pairs (pair 0 pinned to (0, 0)); the root draws ONE pair and the selected trajectory is
 * read off the tree.  log2(T) launches instead of T sequential steps; work (T - 1) N^2.
 * model: proposal must be AUXSSM_PROP_AUX_INDEPENDENT (the tree needs proposals that are independent across time); any transition /
 * potential of the family.  sqrt_half_delta (T) device.  Noise: EXPLICIT eps_aux (C,T,dx), eps_prop (C,T,N,dx), u_res (C,T,N) [row t
 * feeds the stitch at the boundary (t-1 | t); entry (t, 0) is used by the root only; row 0 never] or THREEFRY (streams 1, 2, 3 at the
 * same flat indices).  x (C,T,dx) in/out; ancestors (C,T) int32 = the leaf particle index selected at each step (updated =
 * ancestors != 0).  T >= 2, 2 <= N <= 1024.  Reduction orders and exp/log are fixed (csrc/pit.hip): bit-exact vs oracle/csmc_ref.c. */
int auxssm_csmc_pit_sweep(auxssm_handle h, int dtype, const auxssm_fk_model* model, int32_t C, int32_t T, int32_t N,
                          const void* sqrt_half_delta, void* x, const auxssm_csmc_noise* noise, int32_t* ancestors);

/* auxssm_normalize_resample == normalize(log_weights) (_primitives/math/utils.py:23-39) and/or
 * multinomial(key, weights) (_primitives/csmc/resamplings.py:14-37) with the U[0,1) draws given explicitly, for `rows`
 * independent weight vectors of length N <= 1024 (row-major).  Give log_weights (-> normalised weights in weights_out, may be
 * NULL) or already normalised weights; if `indices` is non-NULL the conditional multinomial ancestors (index 0 pinned to 0)
 * are written from `uniforms` (rows, N).  Same reduction orders and exp/log as the cSMC kernels (bit-exact vs oracle). */
int auxssm_normalize_resample(auxssm_handle h, int dtype, int32_t rows, int32_t N, const void* log_weights, const void* weights,
                              const void* uniforms, void* weights_out, int32_t* indices);

/* ---- the MCMC loop around the sweeps: running statistics, step-size adaptation, Lorenz theta step --------------------
 * == the body of `loop` in examples/stochastic_volatility/experiment.py:88-128 and examples/lorenz/experiment.py:120-169
 * (stats_fn :81-83; moving averages :110-113; delta_adaptation aux_samplers/common.py:4-32), kept on the device so that a run of
 * sweeps needs no host round trip.  `iter` is the reference's loop counter i (sweeps already folded in).
 *
 * auxssm_stats_attach: from now on the accept/select step of every auxssm_kalman_sweep on this handle also folds the sweep into the
 *   running means  sq_jump <- (i sq_jump + (x' - x)^2) / (i + 1),  mean <- (i mean + x') / (i + 1),  sq_mean <- (i sq_mean + x'^2) / (i + 1)
 *   (x the state before the sweep, x' after), arrays of x's shape, layout and dtype, and advances i by one; `iter` sets i for the next
 *   sweep.  The moments are bound to ONE resident state: `x` is its device pointer (the `x` argument of the sweeps to fold), `n` = C*T*dx
 *   its element count, `dtype` its type; while attached, an auxssm_kalman_sweep on this handle over any other state (pointer, size or
 *   dtype) returns AUXSSM_ERR_ARG instead of folding.  All three moment pointers NULL detaches (dtype, n, x ignored).  The fold costs no
 *   extra pass over x.
 * auxssm_stats_update: the same fold as a standalone pass over n = C*T*dx elements (cSMC sweeps: keep a copy of x before the sweep).
 * auxssm_accept_update: flags (C, m) int32, nonzero = updated (`accepted` of a Kalman sweep, m = 1; `ancestors` of a cSMC sweep,
 *   m = T, csmc.py:59):  avg <- (i avg + f) / (i + 1),  window <- beta f + (1 - beta) window,  both (C, m) of `dtype`.
 * auxssm_delta_adapt: delta_j <- clip(delta_j exp(rate (mean_c window[c, j] - target)), min_delta, max_delta), j < m: the reference's rule
 *   on the chain-averaged windowed acceptance, because the chains of one sweep call share delta (C = 1: exactly the reference).  delta (m)
 *   and, if non-NULL, sqrt_half_delta (m) = sqrt(delta / 2) (the cSMC sweep's input) are DEVICE arrays of `dtype`.
 * auxssm_lorenz_theta_update: theta | x of the stochastic Lorenz-63 model (examples/lorenz/model.py:59-79: three independent conjugate
 *   linear regressions of dx - dt phi_0(x) on dt phi(x), prior N(0, sigma_theta^2)), then theta = mean + chol * eps
 *   (experiment.py:112-113).  x (C, T, 3) dense or (T, 3, C) chain-minor (`layout`, as the sweep's); eps (C, 3) ~ N(0, 1); par (C, 4) rows [theta1, theta2, theta3, dt]: dt is read, theta
 *   overwritten -- the array model->Fs points at for AUXSSM_KMODEL_LORENZ63_EXT (chain stride 4).  mean_chol (C, 6) out, may be NULL. */
int auxssm_stats_attach(auxssm_handle h, int dtype, int64_t n, const void* x, void* sq_jump, void* mean, void* sq_mean, int64_t iter);
int auxssm_stats_update(auxssm_handle h, int dtype, int64_t n, int64_t iter, const void* x_prev, const void* x_next, void* sq_jump,
                        void* mean, void* sq_mean);
int auxssm_accept_update(auxssm_handle h, int dtype, int32_t C, int32_t m, int64_t iter, double beta, const int32_t* flags, void* avg,
                         void* window);
int auxssm_delta_adapt(auxssm_handle h, int dtype, int32_t C, int32_t m, const void* window, double target, double rate, double min_delta,
                       double max_delta, void* delta, void* sqrt_half_delta);
int auxssm_lorenz_theta_update(auxssm_handle h, int dtype, int32_t C, int32_t T, int layout, const void* x, double sigma_theta,
                               double sigma_x, const void* eps, void* par, void* mean_chol);

/* auxssm_systematic_resample == systematic(key, weights, N) (_primitives/csmc/resamplings.py:40-86): conditional systematic resampling
 * (Chopin & Singh, Algorithm 4) of `rows` independent weight vectors of length M <= 1024 into N <= 1024 indices each, index 0 kept at
 * position 0; uvw (rows, 3) ~ U[0,1)^3 are the three uniforms of :61.  No reference kernel calls it (csmc.py uses multinomial); it
 * completes resamplings.py.  Same cumsum order as the multinomial path (bit-exact vs oracle/csmc_ref.c::csmc_ref_systematic). */
int auxssm_systematic_resample(auxssm_handle h, int dtype, int32_t rows, int32_t M, int32_t N, const void* weights, const void* uvw,
                               int32_t* indices);

/* auxssm_mvn_logpdf == mvn.logpdf(x, m, chol) (_primitives/math/mvn/base.py:15-58, with tril_log_det :108-128) for n independent triplets of
 * dimension dim <= 64: x, m (dim), chol (dim, dim) lower-triangular row-major, triplet g at x + g sx, m + g sm, chol + g sl (ELEMENT strides;
 * 0 broadcasts).  Reference semantics, IEEE-literal: non-finite entries of chol act as +inf in the forward substitution, the dimension
 * counts the finite diagonal entries only, non-finite diagonal entries drop out of the log-determinant.  out (n).
 * Inside the filter / log-density kernels the same function is csrc/kalman_math.h::gauss_logpdf; this is its standalone export
 * (aux_samplers.mvn.logpdf, reference aux_samplers/__init__.py:3). */
int auxssm_mvn_logpdf(auxssm_handle h, int dtype, int64_t n, int32_t dim, const void* x, int64_t sx, const void* m, int64_t sm, const void* chol,
                      int64_t sl, void* out);

/* auxssm_rng_jax: jax.random's own draws -- `jax.vmap(lambda k: jax.random.uniform(k, (n,), dtype, minval, maxval))(keys)` (kind 0) or `... jax.random.normal(k, (n,), dtype)`
 * (kind 1; minval / maxval ignored) for the threefry2x32 implementation in its non-partitionable layout (JAX's default up to 0.4.x; jax/_src/prng.py: threefry_2x32,
 * threefry_random_bits; jax/_src/random.py: _uniform, _normal_real).  keys: DEVICE (nkeys, 2) uint32 (jax.random.split's rows: aux_ssm_samplers_amd.random.jax_split);
 * out[c * key_stride + i * elem_stride], strides in elements.  The reference's call sites: kalman/generic.py:58-61 (split(key, 3), normal(k, x.shape)), :73 (bernoulli),
 * _primitives/kalman/sampling.py:128, csmc/generic.py:64-67, _primitives/csmc/csmc.py:71-85, :129-138 (split per time step; choice = one uniform per index).
 * Pinned by the values JAX's documentation prints for PRNGKey(0) / PRNGKey(42) (tests/test_rng.py); float64 normals agree with XLA's erfinv to rounding. */
int auxssm_rng_jax(auxssm_handle h, int dtype, int kind, int64_t nkeys, int64_t n, const uint32_t* keys, double minval, double maxval, void* out, int64_t key_stride,
                   int64_t elem_stride);

/* auxssm_mvn_optimal_covariance == mvn.get_optimal_covariance(chol_P, chol_Sig) (_primitives/math/mvn/base.py:78-105): the Cholesky factor of the dominating
 * covariance of Section 3 of the paper -- Y = chol_P^-1 chol_Sig, (w, V) = eigh(Y^T Y), w <- min(w, 1), L = chol_Sig V diag(w^-1/2), out = chol(L L^T) -- for
 * dim <= 64 (lower-triangular row-major inputs and output, dim x dim).  vector != 0: the reference's scalar / diagonal branch (:94-95), out = max(chol_P, chol_Sig)
 * elementwise over dim entries.  The symmetric eigen-decomposition is a cyclic Jacobi iteration; the result does not depend on the eigenvectors' order or signs. */
int auxssm_mvn_optimal_covariance(auxssm_handle h, int dtype, int32_t dim, int vector, const void* chol_P, const void* chol_Sig, void* out);

/* auxssm_ess == effective_sample_size(input_array, var) (examples/rare_event/ess.py:28-160: BlackJAX's estimator -- Geyer's initial positive, monotone sequence of
 * paired autocorrelations, autocovariances averaged over the chains -- with the option of dividing by the TRUE variance): a (M chains, N draws, K series) dense,
 * var (K) or NULL, out (K).  Autocovariances by direct sums in double (the reference's FFT to rounding).  N >= 4, K <= 65535. */
int auxssm_ess(auxssm_handle h, int dtype, int64_t M, int64_t N, int64_t K, const void* a, const void* var, void* out);

/* auxssm_linearise == extended / cubature / gauss_hermite(mean, cov, params, x_star, P_star[, order]) (_primitives/linearisation.py:11-44, :78-104, :47-75 on
 * _generic_sigma_points :107-127) for n linearisation points at once, for the conditional means the device knows in closed form (cov(x) = Qc):
 *   AUXSSM_FN_AFFINE    mean(x) = A x + a, A (dim_out, dim), a (dim_out)    (the reference's own test of the three methods, test_linearisation.py:13-48: 4 -> 2)
 *   AUXSSM_FN_LORENZ63  mean(x) = x + dt (phi_0(x) + theta * phi(x)), A = (theta_0, theta_1, theta_2, dt), a unused, dim = dim_out = 3 (examples/lorenz/model.py:10-25)
 * x_star (n, dim) dense; P_star (dim, dim) with point stride sP in elements (0 = one matrix for every point; unused by EXTENDED, may be NULL);
 * GAUSS_HERMITE: `nodes`, `weights` = HOST doubles, the `order`-point rule for N(0, 1) (order <= 8), tensorised on the device (order^dim points);
 * CUBATURE: the 2 dim points +- sqrt(dim) e_i.  EXTENDED uses the analytic Jacobian where the reference differentiates with jacfwd.
 * Out: F (n, dim_out, dim), Q (n, dim_out, dim_out), b (n, dim_out) -- what a dynamics_factory hands to get_kernel.  dim, dim_out <= 4; Qc (dim_out, dim_out). */
typedef enum { AUXSSM_LIN_EXTENDED = 0, AUXSSM_LIN_CUBATURE = 1, AUXSSM_LIN_GAUSS_HERMITE = 2 } auxssm_lin_method;
typedef enum { AUXSSM_FN_AFFINE = 0, AUXSSM_FN_LORENZ63 = 1 } auxssm_lin_fn;
int auxssm_linearise(auxssm_handle h, int dtype, int method, int order, int fn_kind, int64_t n, int32_t dim, int32_t dim_out, const void* A, const void* a,
                     const void* Qc, const void* nodes, const void* weights, const void* x_star, const void* P_star, int64_t sP, void* F, void* Q, void* b);

/* ---- device RNG: Threefry-2x32-20 counter stream -> N(0,1) / U[0,1) fill -----------------------------
 * out[i], i < n, is a pure function of (key0, key1, stream, i) -- number (i & 1) of Threefry block i >> 1: see
 * oracle/rng_np.py for the restatement. */
int auxssm_rng_normal(auxssm_handle h, int dtype, uint32_t key0, uint32_t key1, uint32_t stream, int64_t n, void* out);
/* the noise of one auxssm_kalman_sweep in ONE launch: keys = {aux0, aux1, samp0, samp1, acc0, acc1} (the three children of the sweep's key,
 * kalman/generic.py:58); eps_aux, eps_samp (n) <- auxssm_rng_normal(key, stream 0), u_acc (nu) <- auxssm_rng_uniform(key, stream 0): the same values. */
int auxssm_kalman_draw(auxssm_handle h, int dtype, const uint32_t* keys, int64_t n, int64_t nu, void* eps_aux, void* eps_samp, void* u_acc);
int auxssm_rng_uniform(auxssm_handle h, int dtype, uint32_t key0, uint32_t key1, uint32_t stream, int64_t n, void* out);

#ifdef __cplusplus
}
#endif
#endif /* AUXSSM_H */

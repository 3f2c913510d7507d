"""ctypes binding of libauxssm.so (include/auxssm.h) and the thin device-memory layer on top of it.

No PyTorch / JAX anywhere on this path: Python host code -> ctypes -> hand-written HIP kernels.
The library is REQUIRED: there is no CPU fallback.  If it is missing or no GPU is present the product
raises; only the test oracle (oracle/) computes on the CPU, and the product never imports it.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AUXSSM_LIB", os.path.join(_HERE, "libauxssm.so"))  # override: diagnostic builds (tools/csmc_ablate.sh)

F32, F64 = 0, 1
NAN_REFERENCE, NAN_MASKED = 0, 1
KMODEL_LG_CONCAT, KMODEL_SV_FIRST, KMODEL_SV_SECOND, KMODEL_LORENZ63_EXT = 1, 2, 3, 4
LAYOUT_DENSE, LAYOUT_CHAIN_MINOR = 0, 1
OPT_SHARE_MODEL = 1
OPT_OVERLAP_MODEL_STAGE = 2
ERR_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM = -1, -2, -3, -4
(K_NONE, K_FILTER_INIT, K_FILTER_SCAN, K_FILTER_ELL, K_SAMPLE_INIT, K_SAMPLE_SCAN, K_LOGPDF, K_CSMC_FWD,
 K_CSMC_BWD, K_PIT_STITCH, K_RNG, K_SELECT, K_FACTORY, K_FILTER_TAB, K_COUNT) = range(15)
K_ALL = -1
K_NAMES = ("none", "filter_init", "filter_scan", "filter_ell", "sample_init", "sample_scan", "logpdf", "csmc_fwd", "csmc_bwd",
           "pit_stitch", "rng", "select", "factory", "filter_tab")


class AuxSSMError(RuntimeError):
    pass


class Arr(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("sc", C.c_int64), ("st", C.c_int64), ("sb", C.c_int64)]


class Lgssm(C.Structure):
    _fields_ = [(n, Arr) for n in ("m0", "P0", "Fs", "Qs", "bs", "Hs", "Rs", "cs")]


class FkModel(C.Structure):
    _fields_ = [("proposal", C.c_int32), ("potential", C.c_int32), ("dx", C.c_int32), ("transition", C.c_int32),
                ("m0", C.c_void_p), ("chol_P0", C.c_void_p), ("F", C.c_void_p), ("b", C.c_void_p), ("chol_Q", C.c_void_p),
                ("y", C.c_void_p), ("sig_y", C.c_double), ("F_t", C.c_void_p), ("b_t", C.c_void_p), ("chol_Q_t", C.c_void_p),
                ("gradient", C.c_int32), ("reserved", C.c_int32)]


class CsmcNoise(C.Structure):
    _fields_ = [("mode", C.c_int32), ("key0", C.c_uint32), ("key1", C.c_uint32), ("reserved", C.c_int32),
                ("eps_aux", C.c_void_p), ("eps_prop", C.c_void_p), ("u_res", C.c_void_p), ("u_bwd", C.c_void_p)]


PROP_BOOTSTRAP_LG, PROP_AUX_INDEPENDENT = 0, 1
POT_FLAT, POT_GAUSS_OBS, POT_SV, POT_GAUSS_OBS_MASKED = 0, 1, 2, 3
TRANS_LINEAR, TRANS_LORENZ63_EM = 0, 1
NOISE_EXPLICIT, NOISE_THREEFRY = 0, 1
GRAD_NONE, GRAD_REFERENCE, GRAD_EXACT = 0, 1, 2


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("C", "T", "B", "dx", "dy")]


_lib = None


def load():
    """Load libauxssm.so (once).  Fails loudly: the HIP extension is the product, not an accelerator option."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AuxSSMError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(or `make -C aux_ssm_samplers_amd/csrc -j8`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_double
    P = C.POINTER
    sig = {
        "auxssm_version": ([], C.c_int),
        "auxssm_last_error": ([], C.c_char_p),
        "auxssm_device_count": ([P(C.c_int)], C.c_int),
        "auxssm_create": ([i32, P(vp)], C.c_int),
        "auxssm_destroy": ([vp], C.c_int),
        "auxssm_sync": ([vp], C.c_int),
        "auxssm_stream": ([vp, P(vp)], C.c_int),
        "auxssm_set_option": ([vp, C.c_int, C.c_int], C.c_int),
        "auxssm_get_option": ([vp, C.c_int, P(C.c_int)], C.c_int),
        "auxssm_malloc": ([vp, C.c_size_t, P(vp)], C.c_int),
        "auxssm_free": ([vp, vp], C.c_int),
        "auxssm_memcpy_h2d": ([vp, vp, vp, C.c_size_t], C.c_int),
        "auxssm_memcpy_d2h": ([vp, vp, vp, C.c_size_t], C.c_int),
        "auxssm_memcpy_d2d": ([vp, vp, vp, C.c_size_t], C.c_int),
        "auxssm_memset": ([vp, vp, i32, C.c_size_t], C.c_int),
        "auxssm_prof_enable": ([vp, i32, i32], C.c_int),
        "auxssm_prof_read": ([vp, P(C.c_int), P(dbl)], C.c_int),
        "auxssm_prof_read_groups": ([vp, i32, P(C.c_int), P(dbl)], C.c_int),
        "auxssm_prof_disable": ([vp], C.c_int),
        "auxssm_kalman_filter": ([vp, i32, P(Dims), P(Lgssm), P(Arr), i32, vp, vp, vp], C.c_int),
        "auxssm_kalman_sample": ([vp, i32, P(Dims), P(Lgssm), vp, vp, vp, i32, vp], C.c_int),
        "auxssm_kalman_dnc_sample": ([vp, i32, P(Dims), P(Lgssm), vp, vp, vp, vp], C.c_int),
        "auxssm_kalman_joint_logpdf": ([vp, i32, P(Dims), P(Lgssm), P(Arr), P(Arr), i32, vp], C.c_int),
        "auxssm_kalman_sweep": ([vp, i32, i32, P(Dims), P(Lgssm), P(Arr), dbl, i32, i32, i32, vp, vp, vp, vp, vp, vp], C.c_int),
        "auxssm_kalman_sweep_dd": ([vp, i32, i32, P(Dims), P(Lgssm), P(Arr), vp, i32, i32, i32, vp, vp, vp, vp, vp, vp], C.c_int),
        "auxssm_kalman_sweep_keyed": ([vp, i32, i32, P(Dims), P(Lgssm), P(Arr), dbl, vp, P(C.c_uint32), i32, i32, i32, vp, vp, vp, vp, vp, vp], C.c_int),
        "auxssm_kalman_sweep_fused": ([vp, i32, i32, P(Dims), P(Lgssm), P(Arr), dbl, vp, P(C.c_uint32), i32, i32, i32, vp, vp, vp, vp, vp, vp], C.c_int),
        "auxssm_kalman_state_resolve": ([vp, i32, P(Dims), vp, vp, vp], C.c_int),
        "auxssm_csmc_sweep": ([vp, i32, P(FkModel), C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, P(CsmcNoise), vp, vp, vp, vp], C.c_int),
        "auxssm_csmc_pit_sweep": ([vp, i32, P(FkModel), C.c_int32, C.c_int32, C.c_int32, vp, vp, P(CsmcNoise), vp], C.c_int),
        "auxssm_normalize_resample": ([vp, i32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp], C.c_int),
        "auxssm_systematic_resample": ([vp, i32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp], C.c_int),
        "auxssm_stats_attach": ([vp, i32, i64, vp, vp, vp, vp, i64], C.c_int),
        "auxssm_stats_update": ([vp, i32, i64, i64, vp, vp, vp, vp, vp], C.c_int),
        "auxssm_accept_update": ([vp, i32, C.c_int32, C.c_int32, i64, dbl, vp, vp, vp], C.c_int),
        "auxssm_delta_adapt": ([vp, i32, C.c_int32, C.c_int32, vp, dbl, dbl, dbl, dbl, vp, vp], C.c_int),
        "auxssm_lorenz_theta_update": ([vp, i32, C.c_int32, C.c_int32, i32, vp, dbl, dbl, vp, vp, vp], C.c_int),
        "auxssm_mvn_logpdf": ([vp, i32, i64, C.c_int32, vp, i64, vp, i64, vp, i64, vp], C.c_int),
        "auxssm_mvn_optimal_covariance": ([vp, i32, C.c_int32, i32, vp, vp, vp], C.c_int),
        "auxssm_linearise": ([vp, i32, i32, i32, i32, i64, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp], C.c_int),
        "auxssm_ess": ([vp, i32, i64, i64, i64, vp, vp, vp], C.c_int),
        "auxssm_kalman_draw": ([vp, i32, P(u32), i64, i64, vp, vp, vp], C.c_int),
        "auxssm_rng_jax": ([vp, i32, i32, i64, i64, vp, dbl, dbl, vp, i64, i64], C.c_int),
        "auxssm_rng_normal": ([vp, i32, u32, u32, u32, i64, vp], C.c_int),
        "auxssm_rng_uniform": ([vp, i32, u32, u32, u32, i64, vp], C.c_int),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = the .so does not export what include/auxssm.h declares
        fn.argtypes = argtypes
        fn.restype = restype
    lib._auxssm_signatures = sig
    _lib = lib
    return lib


def exported_symbols():
    """Names include/auxssm.h declares (used by the CPU test that checks the .so exports all of them)."""
    return sorted(load()._auxssm_signatures)


def check(rc):
    if rc != 0:
        msg = load().auxssm_last_error().decode(errors="replace")
        if rc in (-1, -2):
            raise ValueError(f"auxssm: {msg}")
        if rc == -4:
            raise MemoryError(f"auxssm: {msg}")
        raise AuxSSMError(f"auxssm (status {rc}): {msg}")


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise ValueError(f"dtype must be float32 or float64, got {dt}")


class Handle:
    """One HIP device + one stream (auxssm_create)."""

    def __init__(self, device=0):
        lib = load()
        h = C.c_void_p()
        check(lib.auxssm_create(int(device), C.byref(h)))
        self.lib, self.h, self.device = lib, h, int(device)
        # the C ABI ships the model-stage overlap OFF (include/auxssm.h); this layer routes every device write through an auxssm_* entry
        # point (which the library fences), so it opts in -- unless the environment pins the default
        if "AUXSSM_OVERLAP_TAB" not in os.environ:
            check(lib.auxssm_set_option(h, OPT_OVERLAP_MODEL_STAGE, 1))

    def close(self):
        if self.h:
            self.lib.auxssm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.auxssm_sync(self.h))

    def stream(self):
        """auxssm_stream: the raw hipStream_t (as an int).  Handing it out makes every later model stage wait for the tail of the stream (the caller
        may queue work on it the library cannot see)."""
        s = C.c_void_p()
        check(self.lib.auxssm_stream(self.h, C.byref(s)))
        return s.value

    def get_option(self, option):
        v = C.c_int()
        check(self.lib.auxssm_get_option(self.h, int(option), C.byref(v)))
        return v.value

    def set_option(self, option, value):
        """auxssm_set_option: e.g. (OPT_SHARE_MODEL, 0) forces the general per-chain path of the chain-minor sweep, (OPT_OVERLAP_MODEL_STAGE, 0)
        keeps the chain-shared sweep's model stage on the one stream."""
        check(self.lib.auxssm_set_option(self.h, int(option), int(value)))

    # ---- memory ----
    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def to_device(self, a, dtype=None):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = DeviceArray(self, a.shape, a.dtype)
        if a.nbytes:
            check(self.lib.auxssm_memcpy_h2d(self.h, d.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return d

    def zeros(self, shape, dtype):
        d = DeviceArray(self, shape, dtype)
        if d.nbytes:
            check(self.lib.auxssm_memset(self.h, d.ptr, 0, d.nbytes))
        return d

    # ---- profiling (HIP events on the handle's stream, around one kernel kind) ----
    def prof_enable(self, kernel_id, max_launches):
        check(self.lib.auxssm_prof_enable(self.h, kernel_id, max_launches))

    def prof_read(self):
        n, ms = C.c_int(), C.c_double()
        check(self.lib.auxssm_prof_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def prof_read_groups(self):
        """after prof_enable(K_ALL, n): {group name: (launches, total ms)} for the groups that ran"""
        n = (C.c_int * K_COUNT)()
        ms = (C.c_double * K_COUNT)()
        check(self.lib.auxssm_prof_read_groups(self.h, K_COUNT, n, ms))
        return {K_NAMES[k]: (n[k], ms[k]) for k in range(K_COUNT) if n[k]}

    def prof_disable(self):
        check(self.lib.auxssm_prof_disable(self.h))

    # ---- the MCMC loop around the sweeps (include/auxssm.h: running statistics, adaptation, Lorenz theta step) ----
    def stats_attach(self, stats, it, x=None):
        """stats: (sq_jump, mean, sq_mean) DeviceArrays shaped like the resident state `x` (a DeviceArray), or None to detach"""
        if stats is None:
            check(self.lib.auxssm_stats_attach(self.h, 0, 0, None, None, None, None, int(it)))
            return
        if x is None:
            raise ValueError("stats_attach needs the resident state the moments belong to")
        check(self.lib.auxssm_stats_attach(self.h, dtype_code(x.dtype), x.size, x.ptr, stats[0].ptr, stats[1].ptr, stats[2].ptr, int(it)))

    def stats_update(self, it, x_prev, x_next, stats):
        check(self.lib.auxssm_stats_update(self.h, dtype_code(x_next.dtype), x_next.size, int(it), x_prev.ptr, x_next.ptr, stats[0].ptr,
                                           stats[1].ptr, stats[2].ptr))

    def accept_update(self, it, beta, flags, avg, window):
        """flags (C, m) int32 (nonzero = updated); avg, window (C, m)"""
        Cn = flags.shape[0]
        check(self.lib.auxssm_accept_update(self.h, dtype_code(avg.dtype), Cn, flags.size // max(Cn, 1), int(it), float(beta), flags.ptr,
                                            avg.ptr, window.ptr))

    def delta_adapt(self, window, target, rate, delta, sqrt_half_delta=None, min_delta=1e-20, max_delta=1e20):
        Cn = window.shape[0]
        check(self.lib.auxssm_delta_adapt(self.h, dtype_code(delta.dtype), Cn, delta.size, window.ptr, float(target), float(rate),
                                          float(min_delta), float(max_delta), delta.ptr,
                                          sqrt_half_delta.ptr if sqrt_half_delta is not None else None))

    def lorenz_theta_update(self, x, sigma_theta, sigma_x, eps, par, mean_chol=None, layout=LAYOUT_DENSE):
        """x (C, T, 3) [dense] or (T, 3, C) [chain-minor], eps (C, 3), par (C, 4) DeviceArrays"""
        if layout == LAYOUT_CHAIN_MINOR:
            T, _, Cn = x.shape
        else:
            Cn, T, _ = x.shape
        check(self.lib.auxssm_lorenz_theta_update(self.h, dtype_code(x.dtype), Cn, T, int(layout), x.ptr, float(sigma_theta), float(sigma_x),
                                                  eps.ptr, par.ptr, mean_chol.ptr if mean_chol is not None else None))

    # ---- RNG fill ----
    def rng_normal(self, key, stream, shape, dtype):
        out = self.empty(shape, dtype)
        check(self.lib.auxssm_rng_normal(self.h, dtype_code(dtype), key[0], key[1], stream, out.size, out.ptr))
        return out

    def rng_normal_into(self, key, stream, out):
        check(self.lib.auxssm_rng_normal(self.h, dtype_code(out.dtype), int(key[0]), int(key[1]), stream, out.size, out.ptr))

    def kalman_draw(self, k_aux, k_samp, k_acc, eps_aux, eps_samp, u_acc):
        """the three noise fills of one Kalman sweep in one launch (same values as rng_normal_into x2 + rng_uniform_into, stream 0)"""
        keys = (C.c_uint32 * 6)(int(k_aux[0]), int(k_aux[1]), int(k_samp[0]), int(k_samp[1]), int(k_acc[0]), int(k_acc[1]))
        check(self.lib.auxssm_kalman_draw(self.h, dtype_code(eps_aux.dtype), keys, eps_aux.size, u_acc.size, eps_aux.ptr, eps_samp.ptr, u_acc.ptr))

    def rng_uniform_into(self, key, stream, out):
        check(self.lib.auxssm_rng_uniform(self.h, dtype_code(out.dtype), int(key[0]), int(key[1]), stream, out.size, out.ptr))

    def rng_uniform(self, key, stream, shape, dtype):
        out = self.empty(shape, dtype)
        check(self.lib.auxssm_rng_uniform(self.h, dtype_code(dtype), key[0], key[1], stream, out.size, out.ptr))
        return out


class DeviceArray:
    """A dense, C-contiguous array in HBM owned by Python (auxssm_malloc / auxssm_free)."""

    def __init__(self, handle, shape, dtype):
        self.handle = handle
        self.shape = tuple(int(s) for s in np.atleast_1d(shape)) if not isinstance(shape, tuple) else tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape, dtype=np.int64)) if len(self.shape) else 1
        self.nbytes = self.size * self.dtype.itemsize
        p = C.c_void_p()
        check(handle.lib.auxssm_malloc(handle.h, self.nbytes, C.byref(p)))
        self.ptr = p

    def __del__(self):
        try:
            if self.ptr and self.handle.h:
                self.handle.lib.auxssm_free(self.handle.h, self.ptr)
            self.ptr = None
        except Exception:
            pass

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            check(self.handle.lib.auxssm_memcpy_d2h(self.handle.h, out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def copy_from_host(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.shape != self.shape:
            raise ValueError(f"shape mismatch {a.shape} vs {self.shape}")
        if a.nbytes:
            check(self.handle.lib.auxssm_memcpy_h2d(self.handle.h, self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def copy_from(self, other):
        if other.nbytes != self.nbytes:
            raise ValueError("size mismatch")
        check(self.handle.lib.auxssm_memcpy_d2d(self.handle.h, self.ptr, other.ptr, self.nbytes))

    def arr(self, sc, st, sb=0):
        return Arr(self.ptr.value, int(sc), int(st), int(sb))


_default_handles = {}


def default_handle(device=None):
    """Per-process handle for `device` (default: $LOCAL_RANK, else 0) -- one process per GPU."""
    if device is None:
        device = int(os.environ.get("AUXSSM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    h = _default_handles.get(device)
    if h is None:
        h = _default_handles[device] = Handle(device)
    return h

"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/csmc_ref.c (the plain-C restatement of the
reference's conditional SMC sweep; see that file's header for the reference file:line map and the float contract).
Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcsmc_ref.so")

BOOTSTRAP_LG, AUX_INDEPENDENT = 0, 1
POT_FLAT, POT_GAUSS_OBS, POT_SV, POT_GAUSS_OBS_MASKED = 0, 1, 2, 3
TRANS_LINEAR, TRANS_LORENZ63_EM = 0, 1


class _Model(C.Structure):
    _fields_ = [("proposal", C.c_int), ("potential", C.c_int), ("D", C.c_int), ("backward", C.c_int),
                ("m0", C.c_void_p), ("LP0", C.c_void_p), ("F", C.c_void_p), ("b", C.c_void_p), ("LQ", C.c_void_p),
                ("sig_y", C.c_double), ("transition", C.c_int), ("F_t", C.c_void_p), ("b_t", C.c_void_p), ("LQ_t", C.c_void_p),
                ("gradient", C.c_int)]

GRAD_NONE, GRAD_REFERENCE, GRAD_EXACT = 0, 1, 2


def _model(model, D, backward):
    """(ctypes struct, arrays to keep alive).  Optional keys: F_t, b_t, chol_Q_t (time-varying transitions, T-1 rows), gradient."""
    keep = [np.ascontiguousarray(np.asarray(model[k], np.float64)) for k in ("m0", "chol_P0", "F", "b", "chol_Q")]
    tv = [None, None, None]
    if model.get("F_t") is not None:
        tv = [np.ascontiguousarray(np.asarray(model[k], np.float64)) for k in ("F_t", "b_t", "chol_Q_t")]
    m = _Model(int(model["proposal"]), int(model["potential"]), D, int(bool(backward)), _p(keep[0]), _p(keep[1]), _p(keep[2]),
               _p(keep[3]), _p(keep[4]), float(model.get("sig_y", 1.0)), int(model.get("transition", 0)), _p(tv[0]), _p(tv[1]), _p(tv[2]),
               int(model.get("gradient", 0)))
    return m, keep + tv


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "csmc_ref.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = C.CDLL(_SO)
        _lib.csmc_ref_expf.restype = C.c_float
        _lib.csmc_ref_expf.argtypes = [C.c_float]
        _lib.csmc_ref_logf.restype = C.c_float
        _lib.csmc_ref_logf.argtypes = [C.c_float]
        _lib.csmc_ref_exp.restype = C.c_double
        _lib.csmc_ref_exp.argtypes = [C.c_double]
        _lib.csmc_ref_log.restype = C.c_double
        _lib.csmc_ref_log.argtypes = [C.c_double]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def sweep(model, x, N, backward, *, y=None, sqrt_half_delta=None, eps_aux=None, eps_prop, u_res, u_bwd, dtype=np.float32):
    """One cSMC sweep of one chain.  model: dict(proposal, potential, m0, chol_P0, F, b, chol_Q, sig_y[, transition]).
    Returns dict(x, ancestors, xs, log_ws, As)."""
    dtype = np.dtype(dtype)
    x = np.array(x, dtype, order="C")
    T, D = x.shape
    m, keep = _model(model, D, backward)
    cv = lambda a: None if a is None else np.ascontiguousarray(a, dtype)
    y, shd, eps_aux, eps_prop, u_res, u_bwd = map(cv, (y, sqrt_half_delta, eps_aux, eps_prop, u_res, u_bwd))
    assert eps_prop.shape == (T, N, D) and u_bwd.shape == (T,) and (T == 1 or u_res.shape == (T - 1, N))
    anc = np.zeros(T, np.int32)
    xs = np.zeros((T, N, D), dtype)
    lws = np.zeros((T, N), dtype)
    As = np.zeros((max(T - 1, 1), N), np.int32)
    fn = lib().csmc_ref_sweep_f32 if dtype == np.float32 else lib().csmc_ref_sweep_f64
    rc = fn(C.byref(m), T, N, _p(x), _p(y), _p(shd), _p(eps_aux), _p(eps_prop), _p(u_res), _p(u_bwd), _p(anc), _p(xs), _p(lws), _p(As))
    assert rc == 0
    return dict(x=x, ancestors=anc, xs=xs, log_ws=lws, As=As[:T - 1])


def pit_sweep(model, x, N, *, y=None, sqrt_half_delta, eps_aux, eps_prop, u_res, dtype=np.float32):
    """One parallel-in-time cSMC sweep (conditional dSMC) of one chain, evaluated with full block gathers as the reference's operator
    does (csmc_ref.c::csmc_ref_pit_sweep).  u_res (T, N): row t feeds the stitch at the boundary (t-1 | t).
    Returns dict(x, ancestors, xs)."""
    dtype = np.dtype(dtype)
    x = np.array(x, dtype, order="C")
    T, D = x.shape
    m, keep = _model(model, D, 0)
    cv = lambda a: None if a is None else np.ascontiguousarray(a, dtype)
    y, shd, eps_aux, eps_prop, u_res = map(cv, (y, sqrt_half_delta, eps_aux, eps_prop, u_res))
    assert T >= 2 and eps_prop.shape == (T, N, D) and u_res.shape == (T, N) and eps_aux.shape == (T, D) and shd.shape == (T,)
    anc = np.zeros(T, np.int32)
    xs = np.zeros((T, N, D), dtype)
    fn = lib().csmc_ref_pit_sweep_f32 if dtype == np.float32 else lib().csmc_ref_pit_sweep_f64
    rc = fn(C.byref(m), T, N, _p(x), _p(y), _p(shd), _p(eps_aux), _p(eps_prop), _p(u_res), _p(anc), _p(xs))
    assert rc == 0
    return dict(x=x, ancestors=anc, xs=xs)


def grad_logpi(model, u, y=None, dtype=np.float64):
    """gradient at u (T, D) of the model's joint log-density (csmc_ref.c::grad_logpi; independent.py:121-134)"""
    dtype = np.dtype(dtype)
    u = np.ascontiguousarray(u, dtype)
    T, D = u.shape
    m, keep = _model(model, D, 0)
    y = None if y is None else np.ascontiguousarray(y, dtype)
    g = np.zeros((T, D), dtype)
    fn = lib().csmc_ref_grad_f32 if dtype == np.float32 else lib().csmc_ref_grad_f64
    fn(C.byref(m), T, _p(u), _p(y), _p(g))
    return g


def multinomial(w, un, dtype=np.float32):
    w = np.ascontiguousarray(w, dtype)
    un = np.ascontiguousarray(un, dtype)
    idx = np.zeros(len(w), np.int32)
    fn = lib().csmc_ref_multinomial_f32 if np.dtype(dtype) == np.float32 else lib().csmc_ref_multinomial_f64
    fn(_p(w), len(w), _p(un), _p(idx))
    return idx


def systematic(w, uvw, N=None, dtype=np.float32):
    """conditional systematic resampling of one weight vector given (U, V, W) (csmc_ref.c::csmc_ref_systematic)"""
    w = np.ascontiguousarray(w, dtype)
    uvw = np.ascontiguousarray(uvw, dtype)
    N = len(w) if N is None else int(N)
    idx = np.zeros(N, np.int32)
    fn = lib().csmc_ref_systematic_f32 if np.dtype(dtype) == np.float32 else lib().csmc_ref_systematic_f64
    fn(_p(w), len(w), N, _p(uvw), _p(idx))
    return idx


def normalize(lw, dtype=np.float32):
    lw = np.ascontiguousarray(lw, dtype)
    w = np.zeros_like(lw)
    fn = lib().csmc_ref_normalize_f32 if np.dtype(dtype) == np.float32 else lib().csmc_ref_normalize_f64
    fn(_p(lw), len(lw), _p(w))
    return w

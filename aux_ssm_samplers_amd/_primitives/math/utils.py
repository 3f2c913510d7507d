"""reference: aux_samplers/_primitives/math/utils.py."""
import numpy as np

from ... import _lib


def normalize(log_weights, handle=None):
    """normalize(log_weights) = exp(log_weights - logsumexp(log_weights))   (math/utils.py:23-39); (M,) or (rows, M), M <= 1024."""
    handle = handle or _lib.default_handle()
    lw = np.asarray(log_weights)
    single = lw.ndim == 1
    l2 = lw[None] if single else lw
    rows, M = l2.shape
    dtype = np.dtype(np.float32) if l2.dtype == np.float32 else np.dtype(np.float64)
    ld = handle.to_device(l2, dtype)
    out = handle.empty((rows, M), dtype)
    _lib.check(handle.lib.auxssm_normalize_resample(handle.h, _lib.dtype_code(dtype), rows, M, ld.ptr, None, None, out.ptr, None))
    w = out.to_host()
    return w[0] if single else w


LOG_HALF = float(np.log(0.5))


def log1mexp(x):
    """log(1 - exp(x)) for x < 0, stable on both sides of log(1/2) (math/utils.py:18-20).  Host helper (never on the sampler path)."""
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(x < LOG_HALF, np.log1p(-np.exp(x)), np.log(-np.expm1(x)))


def logsubexp(x1, x2):
    """log|exp(x1) - exp(x2)| (math/utils.py:11-15).  Host helper (never on the sampler path)."""
    x1, x2 = np.asarray(x1, dtype=np.float64), np.asarray(x2, dtype=np.float64)
    amax = np.maximum(x1, x2)
    return amax + log1mexp(-np.abs(x1 - x2))

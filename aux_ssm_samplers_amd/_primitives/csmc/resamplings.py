"""Conditional resampling (reference: aux_samplers/_primitives/csmc/resamplings.py).

Conditional *multinomial* resampling is the one on the hot path (csmc.py:54); conditional `systematic` resampling (never called by a
reference kernel) is provided for completeness of the module."""
import numpy as np

from ... import _lib, random as _random


def multinomial(key, weights, N=None, u=None, handle=None):
    """multinomial(key, weights, N=None) -> indices   (resamplings.py:14-37): jax.random.choice(key, M, p=weights, (N,))
    i.e. searchsorted(cumsum(w), c[-1] (1 - U)), then index 0 forced to 0.  `u`: explicit U[0,1) draws of shape weights.shape.
    weights: (M,) or (rows, M); N must equal M (the only form the kernels use)."""
    handle = handle or _lib.default_handle()
    w = np.asarray(weights)
    single = w.ndim == 1
    w2 = w[None] if single else w
    rows, M = w2.shape
    if N is not None and N != M:
        raise NotImplementedError("N != len(weights) is not used by any kernel of the reference and is not provided")
    dtype = np.dtype(np.float32) if w2.dtype == np.float32 else np.dtype(np.float64)
    wd = handle.to_device(w2, dtype)
    ud = handle.rng_uniform(_random.as_key(key), 3, (rows, M), dtype) if u is None else handle.to_device(np.reshape(u, (rows, M)), dtype)
    idx = handle.zeros((rows, M), np.int32)
    _lib.check(handle.lib.auxssm_normalize_resample(handle.h, _lib.dtype_code(dtype), rows, M, None, wd.ptr, ud.ptr, None, idx.ptr))
    out = idx.to_host()
    return out[0] if single else out


def systematic(key, weights, N=None, uvw=None, handle=None):
    """systematic(key, weights, N=None) -> indices   (resamplings.py:40-86): conditional systematic resampling (Chopin & Singh, Algorithm 4),
    index 0 left unchanged.  weights (M,) or (rows, M), normalised; N draws (default M), M, N <= 1024.  `uvw`: the three U[0,1) draws per
    row given explicitly ((3,) or (rows, 3)); default: from `key`."""
    handle = handle or _lib.default_handle()
    w = np.asarray(weights)
    single = w.ndim == 1
    w2 = w[None] if single else w
    rows, M = w2.shape
    N = M if N is None else int(N)
    dtype = np.dtype(np.float32) if w2.dtype == np.float32 else np.dtype(np.float64)
    wd = handle.to_device(w2, dtype)
    ud = handle.rng_uniform(_random.as_key(key), 5, (rows, 3), dtype) if uvw is None else handle.to_device(np.reshape(uvw, (rows, 3)), dtype)
    idx = handle.zeros((rows, N), np.int32)
    _lib.check(handle.lib.auxssm_systematic_resample(handle.h, _lib.dtype_code(dtype), rows, M, N, wd.ptr, ud.ptr, idx.ptr))
    out = idx.to_host()
    return out[0] if single else out

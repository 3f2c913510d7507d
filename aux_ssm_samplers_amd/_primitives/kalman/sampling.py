"""Pathwise sampling from the smoothing distribution of an LGSSM (reference: _primitives/kalman/sampling.py)."""
import ctypes as C

import numpy as np

from ... import _layout, _lib, random as _random
from .base import DeviceLGSSM, _common_dtype


def sampling(key, ms, Ps, lgssm, parallel, eps=None, handle=None):
    """sampling(key, ms, Ps, lgssm, parallel) -> xs   (sampling.py:11-40).

    The N(0, I) draws of sampling.py:128 come from `key` (device Threefry fill) unless `eps` (shape ms.shape)
    is given explicitly -- the form the parity tests use."""
    handle = handle or _lib.default_handle()
    ms = np.asarray(ms)
    batched = ms.ndim == 3
    T = ms.shape[0]
    B = ms.shape[1] if batched else 1
    dx = ms.shape[-1]
    dtype = _common_dtype(ms, Ps, *lgssm[:5])
    lg = list(lgssm[:5]) + [None, None, None]
    dl = DeviceLGSSM(handle, lg, 1, T, B, dx, 1, batched, dtype)
    msd = handle.to_device(np.asarray(ms, dtype).reshape(1, T, B, dx))
    Psd = handle.to_device(np.asarray(Ps, dtype).reshape(1, T, B, dx, dx))
    if eps is None:
        epd = handle.rng_normal(_random.as_key(key), 0, (1, T, B, dx), dtype)
    else:
        epd = handle.to_device(np.asarray(eps, dtype).reshape(1, T, B, dx))
    xs = handle.empty((1, T, B, dx), dtype)
    dims = _lib.Dims(1, T, B, dx, 1)
    _lib.check(handle.lib.auxssm_kalman_sample(handle.h, _lib.dtype_code(dtype), C.byref(dims), C.byref(dl.c),
                                                msd.ptr, Psd.ptr, epd.ptr, int(bool(parallel)), xs.ptr))
    out = xs.to_host()[0]
    return out if batched else out[:, 0]

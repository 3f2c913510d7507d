"""Kalman filtering on the GPU (reference: aux_samplers/_primitives/kalman/filtering.py)."""
import ctypes as C

import numpy as np

from ... import _layout, _lib
from .base import DeviceLGSSM, _common_dtype, _upload_arr


def filtering(ys, lgssm, parallel, handle=None):
    """filtering(ys, lgssm, parallel) -> (ms, Ps, ell)   (filtering.py:18-46).

    ys (T, dy) or batched (T, B, dy); NaN entries are missing observations.  parallel=True runs the
    associative scan over T (filtering.py:49-63); False runs the same HIP kernels as one chunk per sequence,
    i.e. the sequential recursion (filtering.py:66-79).  ell is summed over the batch axis (:43-45)."""
    handle = handle or _lib.default_handle()
    C_, T, B, dx, dy, batched = _layout.infer_dims(ys, lgssm, False)
    dtype = _common_dtype(ys, *lgssm)
    dl = DeviceLGSSM(handle, lgssm, C_, T, B, dx, dy, batched, dtype)
    ybuf, yarr = _upload_arr(handle, ys, (dy,), C_, T, B, batched, False, dtype, "ys")
    ms = handle.empty((C_, T, B, dx), dtype)
    Ps = handle.empty((C_, T, B, dx, dx), dtype)
    ell = handle.empty((C_,), dtype)
    dims = _lib.Dims(C_, T, B, dx, dy)
    _lib.check(handle.lib.auxssm_kalman_filter(handle.h, _lib.dtype_code(dtype), C.byref(dims), C.byref(dl.c),
                                                C.byref(yarr), int(bool(parallel)), ms.ptr, Ps.ptr, ell.ptr))
    ms_h, Ps_h = ms.to_host()[0], Ps.to_host()[0]
    if not batched:
        ms_h, Ps_h = ms_h[:, 0], Ps_h[:, 0]
    return ms_h, Ps_h, ell.to_host()[0]

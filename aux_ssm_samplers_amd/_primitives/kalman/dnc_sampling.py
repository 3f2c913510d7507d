"""reference: aux_samplers/_primitives/kalman/dnc_sampling.py -- the divide-and-conquer pathwise sampler of an LGSSM, which the reference itself
declares "a proof-of-concept (and not efficient) feature kept mostly for pedagogical reasons" (:38-41).  Since round 4 this is the algorithm itself on the device
(`csrc/dnc.hip`, `auxssm_kalman_dnc_sample`: leaves :128-137, pairwise combination :104-118 with the odd interval carried up :140-169, the two ends :46-68, the
mid-points level by level :70-86) -- same signature, same warning, same error for batched input.  Noise: time index t is sampled with eps[t] (explicit `eps`, or drawn
from `key` by the device fill); the reference splits its key per tree level instead, and the parity contract is on explicit noise (oracle: `oracle/kalman_np.py::dnc_sampling`)."""
import ctypes as C
import warnings

import numpy as np

from ... import _lib, random as _random
from .base import DeviceLGSSM, _common_dtype


def sampling(key, ms, Ps, lgssm, eps=None, handle=None):
    warnings.warn("`dnc_sampling.sampling` is a proof-of-concept (and not efficient) feature kept mostly for pedagogical reasons."
                  "Use `sampling.sampling` with the argument `parallel=True` instead.", UserWarning)
    if np.ndim(ms) > 2:
        raise ValueError("Batched sampling is not supported for this function. Use `sampling.sampling` instead.")
    handle = handle or _lib.default_handle()
    ms = np.asarray(ms)
    T, dx = ms.shape
    if dx > 4:
        raise ValueError("the divide-and-conquer sampler is built for dx <= 4; use `sampling.sampling` (parallel=True)")
    dtype = _common_dtype(ms, Ps, *lgssm[:5])
    dl = DeviceLGSSM(handle, list(lgssm[:5]) + [None, None, None], 1, T, 1, dx, 1, False, dtype)
    msd = handle.to_device(np.asarray(ms, dtype).reshape(1, T, dx))
    Psd = handle.to_device(np.asarray(Ps, dtype).reshape(1, T, dx, dx))
    epd = handle.rng_normal(_random.as_key(key), 0, (1, T, dx), dtype) if eps is None else handle.to_device(np.asarray(eps, dtype).reshape(1, T, dx))
    xs = handle.empty((1, T, dx), dtype)
    dims = _lib.Dims(1, T, 1, dx, 1)
    _lib.check(handle.lib.auxssm_kalman_dnc_sample(handle.h, _lib.dtype_code(dtype), C.byref(dims), C.byref(dl.c), msd.ptr, Psd.ptr, epd.ptr, xs.ptr))
    return xs.to_host()[0]

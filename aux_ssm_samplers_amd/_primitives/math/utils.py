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

"""reference: aux_samplers/_primitives/math/mvn/base.py.

`logpdf` runs on the device (auxssm_mvn_logpdf; inside the filter / log-density kernels the same function is
csrc/kalman_math.h::gauss_logpdf).  `rvs` draws its normals with the device Threefry generator.  `get_optimal_covariance` runs on the
device too (auxssm_mvn_optimal_covariance).  `tril_log_det` of a host array is a set-up-time one-liner (the device
form lives inside k_mvn_logpdf and the log-density kernels)."""
import numpy as np

from .... import _lib


def logpdf(x, m, chol, handle=None):
    """mvn.logpdf(x, m, chol), signature (n),(n),(n,n)->() with NumPy broadcasting over leading axes (mvn/base.py:15-58):
    log N(x; m, chol chol^T); non-finite entries of `chol` are "numerically ignored" exactly as the reference does."""
    handle = handle or _lib.default_handle()
    x, m, chol = np.asarray(x), np.asarray(m), np.asarray(chol)
    dtype = np.dtype(np.float32) if all(a.dtype == np.float32 for a in (x, m, chol)) else np.dtype(np.float64)
    if chol.ndim < 2 or chol.shape[-1] != chol.shape[-2]:
        raise ValueError(f"chol must be (..., n, n), got {chol.shape}")
    n = chol.shape[-1]
    if x.shape[-1:] != (n,) or m.shape[-1:] != (n,):
        raise ValueError(f"x {x.shape} / m {m.shape} do not match chol {chol.shape}")
    batch = np.broadcast_shapes(x.shape[:-1], m.shape[:-1], chol.shape[:-2])
    nb = int(np.prod(batch, dtype=np.int64)) if batch else 1

    def dev(a, core):
        # a broadcast operand is uploaded once and read with stride 0
        if a.shape[:a.ndim - len(core)] == batch and nb > 1:
            return handle.to_device(np.ascontiguousarray(a, dtype).reshape((nb,) + core)), int(np.prod(core))
        if a.ndim == len(core) or int(np.prod(a.shape[:a.ndim - len(core)])) == 1:
            return handle.to_device(np.ascontiguousarray(a, dtype).reshape(core)), 0
        return handle.to_device(np.ascontiguousarray(np.broadcast_to(a, batch + core), dtype).reshape((nb,) + core)), int(np.prod(core))

    xd, sx = dev(x, (n,))
    md, sm = dev(m, (n,))
    Ld, sl = dev(chol, (n, n))
    out = handle.empty((nb,), dtype)
    _lib.check(handle.lib.auxssm_mvn_logpdf(handle.h, _lib.dtype_code(dtype), nb, n, xd.ptr, sx, md.ptr, sm, Ld.ptr, sl, out.ptr))
    res = out.to_host().reshape(batch)
    return res if batch else res[()]


def rvs(key, m, chol, handle=None):
    """m + chol @ eps, eps ~ N(0, I) of m's shape drawn on the device from `key` (mvn/base.py:61-75)."""
    from .... import random as R
    m, chol = np.asarray(m), np.asarray(chol)
    eps = R.normal(key, m.shape, m.dtype if m.dtype == np.float32 else np.float64, handle=handle)
    return m + np.einsum("...ij,...j->...i", chol, eps)


def tril_log_det(chol):
    """log |det| of a lower-triangular matrix (or of its diagonal given as a vector), non-finite diagonal entries ignored
    (mvn/base.py:108-128)."""
    chol = np.asarray(chol)
    d = np.diag(chol) if chol.ndim == 2 else chol
    d = np.where(np.isfinite(d), d, 1.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.nansum(np.log(np.abs(d)))


def get_optimal_covariance(chol_P, chol_Sig, handle=None):
    """Cholesky factor of the dominating covariance of Section 3 of the paper (mvn/base.py:78-105): with Y = chol_P^-1 chol_Sig,
    eigen-decompose Y^T Y = V diag(w) V^T, clip w at 1 and return chol(L L^T), L = chol_Sig V diag(min(w, 1)^-1/2) -- on the device
    (auxssm_mvn_optimal_covariance: one workgroup, cyclic Jacobi eigen-decomposition in LDS), dim <= 64.  Scalars / vectors take the
    reference's elementwise-maximum branch (:94-95)."""
    handle = handle or _lib.default_handle()
    chol_P, chol_Sig = np.asarray(chol_P), np.asarray(chol_Sig)
    dtype = np.dtype(np.float32) if chol_P.dtype == np.float32 and chol_Sig.dtype == np.float32 else np.dtype(np.float64)
    vector = (chol_P.ndim < 2 and chol_Sig.ndim < 2) or chol_P.shape[0] == 1
    if vector:
        shape = np.broadcast_shapes(chol_P.shape, chol_Sig.shape)
        a, b = (np.ascontiguousarray(np.broadcast_to(v, shape), dtype).reshape(-1) for v in (chol_P, chol_Sig))
        n = a.size
    else:
        if chol_P.ndim != 2 or chol_P.shape[0] != chol_P.shape[1] or chol_Sig.shape != chol_P.shape:
            raise ValueError(f"chol_P {chol_P.shape} and chol_Sig {chol_Sig.shape} must be the same square shape")
        shape, n = chol_P.shape, chol_P.shape[0]
        a, b = np.ascontiguousarray(chol_P, dtype), np.ascontiguousarray(chol_Sig, dtype)
    ad, bd, out = handle.to_device(a), handle.to_device(b), handle.empty(a.shape, dtype)
    _lib.check(handle.lib.auxssm_mvn_optimal_covariance(handle.h, _lib.dtype_code(dtype), n, int(vector), ad.ptr, bd.ptr, out.ptr))
    res = out.to_host().reshape(shape)
    return res if shape else res[()]

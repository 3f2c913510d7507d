"""LGSSM bundle and trajectory log-densities (reference: aux_samplers/_primitives/kalman/base.py)."""
import ctypes as C
from typing import Any, NamedTuple

import numpy as np

from ... import _layout, _lib


class LGSSM(NamedTuple):
    """Same fields, order and shapes as the reference NamedTuple (base.py:12-69):
    m0 ([B,] dx), P0 ([B,] dx, dx), Fs/Qs (T-1, [B,] dx, dx), bs (T-1, [B,] dx),
    Hs (T, [B,] dy, dx), Rs (T, [B,] dy, dy), cs (T, [B,] dy).  NumPy arrays; broadcast views
    (np.broadcast_to) are honoured and stored once on the device."""
    m0: Any
    P0: Any
    Fs: Any
    Qs: Any
    bs: Any
    Hs: Any
    Rs: Any
    cs: Any


def _common_dtype(*arrays):
    dt = np.result_type(*[np.asarray(a).dtype for a in arrays if a is not None])
    return np.dtype(np.float32) if dt == np.float32 else np.dtype(np.float64)


class DeviceLGSSM:
    """An LGSSM uploaded once and described for the C ABI (keeps the device buffers alive)."""

    def __init__(self, handle, lgssm, C_, T, B, dx, dy, batched, dtype, chain_axis=False):
        self.handle, self.dims = handle, (C_, T, B, dx, dy)
        self.dtype = np.dtype(dtype)
        desc = _layout.describe_lgssm(lgssm, C_, T, B, dx, dy, batched, self.dtype, chain_axis)
        self.bufs = {}
        self.c = _lib.Lgssm()
        for name in _layout.LGSSM_FIELDS:
            if name in desc:
                d = desc[name]
                buf = handle.to_device(d.buf)
                self.bufs[name] = buf
                setattr(self.c, name, buf.arr(d.sc, d.st, d.sb))


def _upload_arr(handle, a, core, C_, T, B, batched, chains, dtype, name):
    d = _layout.describe(a, core, chains=C_ if chains else None, time_len=T, batch=B if batched else None, dtype=dtype, name=name)
    buf = handle.to_device(d.buf)
    return buf, buf.arr(d.sc, d.st, d.sb)


def joint_logpdf(ys, xs, lgssm, nan_policy="reference", handle=None):
    """log_likelihood(ys, xs, lgssm) + prior_logpdf(xs, lgssm)  (base.py:99-166), one fused HIP pass."""
    handle = handle or _lib.default_handle()
    C_, T, B, dx, dy, batched = _layout.infer_dims(ys, lgssm, False)
    dtype = _common_dtype(ys, xs, *lgssm)
    dl = DeviceLGSSM(handle, lgssm, C_, T, B, dx, dy, batched, dtype)
    ybuf, yarr = _upload_arr(handle, ys, (dy,), C_, T, B, batched, False, dtype, "ys")
    xbuf, xarr = _upload_arr(handle, xs, (dx,), C_, T, B, batched, False, dtype, "xs")
    out = handle.empty((C_,), dtype)
    dims = _lib.Dims(C_, T, B, dx, dy)
    pol = {"reference": _lib.NAN_REFERENCE, "masked": _lib.NAN_MASKED}[nan_policy]
    _lib.check(handle.lib.auxssm_kalman_joint_logpdf(handle.h, _lib.dtype_code(dtype), C.byref(dims), C.byref(dl.c),
                                                      C.byref(yarr), C.byref(xarr), pol, out.ptr))
    return out.to_host()[0]


def posterior_logpdf(ys, xs, ell, lgssm, nan_policy="reference", handle=None):
    """log p(x_{0:T} | y_{0:T}) = log_likelihood - ell + prior_logpdf  (base.py:72-96)."""
    return joint_logpdf(ys, xs, lgssm, nan_policy, handle) - ell


def log_likelihood(ys, xs, lgssm, nan_policy="reference", handle=None):
    """base.py:137-166.  Computed as joint_logpdf minus the prior part (both on the device)."""
    return joint_logpdf(ys, xs, lgssm, nan_policy, handle) - prior_logpdf(xs, lgssm, handle)


def prior_logpdf(xs, lgssm, handle=None):
    """base.py:99-134: joint_logpdf with every observation missing under the masked policy (obs term == 0)."""
    m0, P0, Fs, Qs, bs, Hs, Rs, cs = lgssm
    T = np.shape(xs)[0]
    dtype = _common_dtype(xs, m0, P0, Fs, Qs, bs)
    lead = np.shape(xs)[:-1]
    dx = np.shape(xs)[-1]
    ys = np.full(lead + (1,), np.nan, dtype)
    bt = np.broadcast_to
    H1 = bt(np.zeros((1, dx), dtype), lead + (1, dx))
    R1 = bt(np.ones((1, 1), dtype), lead + (1, 1))
    c1 = bt(np.zeros((1,), dtype), lead + (1,))
    return joint_logpdf(ys, xs, LGSSM(m0, P0, Fs, Qs, bs, H1, R1, c1), "masked", handle)

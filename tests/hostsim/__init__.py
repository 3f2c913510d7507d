"""TEST INFRASTRUCTURE ONLY: ctypes front-end of tests/hostsim/libhostsim.so, which runs the product's
per-lane kernel bodies (csrc/kalman_bodies.h) on the CPU in plain loops.  Lets the CPU test-suite check the
math + indexing of the HIP path against the oracle.  Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

from aux_ssm_samplers_amd import _layout
from aux_ssm_samplers_amd._lib import Arr

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhostsim.so")
_SRC = os.path.join(_HERE, "hostsim.cpp")
_CSRC = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "aux_ssm_samplers_amd", "csrc")


def build(force=False):
    deps = [_SRC] + [os.path.join(_CSRC, f) for f in ("smallmat.h", "kalman_math.h", "kalman_bodies.h")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in deps):
        return _SO
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", _SRC, "-o", _SO])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _arr(d):
    return Arr(d.buf.ctypes.data, d.sc, d.st, d.sb)


def _dt(dtype):
    return 0 if np.dtype(dtype) == np.float32 else 1


def _garr(desc):
    keep = [desc[k] for k in _layout.LGSSM_FIELDS if k in desc]
    g = (Arr * 8)()
    for i, k in enumerate(_layout.LGSSM_FIELDS):
        if k in desc:
            g[i] = _arr(desc[k])
    return g, keep


def filtering(ys, lgssm, E=0, dtype=np.float64, chains=False, chain_axis=False, pblk=0):
    C_, T, B, dx, dy, batched = _layout.infer_dims(ys, lgssm, chains)
    desc = _layout.describe_lgssm(lgssm, C_, T, B, dx, dy, batched, dtype, chain_axis)
    yd = _layout.describe(ys, (dy,), chains=C_ if chains else None, time_len=T, batch=B if batched else None, dtype=dtype, name="ys")
    g, keep = _garr(desc)
    ms = np.empty((C_, T, B, dx), dtype)
    Ps = np.empty((C_, T, B, dx, dx), dtype)
    ell = np.empty((C_,), dtype)
    ya = _arr(yd)
    rc = lib().hs_filter(_dt(dtype), dx, dy, C_, T, B, g, C.byref(ya), int(E), ms.ctypes.data_as(C.c_void_p),
                         Ps.ctypes.data_as(C.c_void_p), ell.ctypes.data_as(C.c_void_p), int(pblk))
    assert rc == 0, rc
    return _squeeze(ms, chains, batched), _squeeze(Ps, chains, batched), (ell if chains else ell[0])


def _squeeze(a, chains, batched):
    if not batched:
        a = a[:, :, 0]
    if not chains:
        a = a[0]
    return a


def _dense(a, C_, T, B, core, dtype, chains, batched):
    a = np.asarray(a, dtype)
    return np.ascontiguousarray(a.reshape((C_, T, B) + core))


def sampling(eps, ms, Ps, lgssm, E=0, dtype=np.float64, chains=False, chain_axis=False):
    ms = np.asarray(ms)
    nlead = ms.ndim - 1 - (1 if chains else 0)
    batched = nlead == 2
    off = 1 if chains else 0
    C_ = ms.shape[0] if chains else 1
    T = ms.shape[off]
    B = ms.shape[off + 1] if batched else 1
    dx = ms.shape[-1]
    lg = list(lgssm[:5]) + [None, None, None]
    desc = _layout.describe_lgssm(lg, C_, T, B, dx, 1, batched, dtype, chain_axis)
    g, keep = _garr(desc)
    msd = _dense(ms, C_, T, B, (dx,), dtype, chains, batched)
    Psd = _dense(Ps, C_, T, B, (dx, dx), dtype, chains, batched)
    epd = _dense(eps, C_, T, B, (dx,), dtype, chains, batched)
    xs = np.empty((C_, T, B, dx), dtype)
    rc = lib().hs_sample(_dt(dtype), dx, C_, T, B, g, msd.ctypes.data_as(C.c_void_p), Psd.ctypes.data_as(C.c_void_p),
                         epd.ctypes.data_as(C.c_void_p), int(E), xs.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return _squeeze(xs, chains, batched)


def joint_logpdf(ys, xs, lgssm, nan_policy=0, dtype=np.float64, chains=False, chain_axis=False):
    C_, T, B, dx, dy, batched = _layout.infer_dims(ys, lgssm, chains)
    desc = _layout.describe_lgssm(lgssm, C_, T, B, dx, dy, batched, dtype, chain_axis)
    bt = B if batched else None
    yd = _layout.describe(ys, (dy,), chains=C_ if chains else None, time_len=T, batch=bt, dtype=dtype, name="ys")
    xd = _layout.describe(xs, (dx,), chains=C_ if chains else None, time_len=T, batch=bt, dtype=dtype, name="xs")
    g, keep = _garr(desc)
    out = np.empty((C_,), dtype)
    ya, xa = _arr(yd), _arr(xd)
    rc = lib().hs_logpdf(_dt(dtype), dx, dy, C_, T, B, g, C.byref(ya), C.byref(xa), int(nan_policy), out.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return out if chains else out[0]


def sv_logpdf(lg5, yobs, x, xp, u, ys1, ys2, R1, R2, delta, chain_minor=False, dtype=np.float64):
    """The SV sweep's fused log-density pass (csrc/kalman_bodies.h::body_sv_logpdf) on the host: x, xp, u, ys1, ys2 (C, T, D), R1 / R2
    (C, T, D, D) or None; lg5 = (m0, P0, Fs, Qs, bs) chain-shared.  Returns (5, C): jp_prop, jp_rev, lt_prop, lt_rev, corr."""
    x = np.asarray(x)
    C_, T, D = x.shape
    lg = list(lg5) + [None, None, None]
    desc = _layout.describe_lgssm(lg, 1, T, 1, D, 1, False, dtype, False)
    g, keep = _garr(desc)
    cv = lambda a: None if a is None else np.ascontiguousarray(a, dtype)
    arrs = [cv(a) for a in (yobs, x, xp, u, ys1, ys2, R1, R2)]
    out = np.empty((5, C_), dtype)
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib().hs_sv_logpdf(_dt(dtype), D, C_, T, g, *[ptr(a) for a in arrs], C.c_double(float(delta)), int(bool(chain_minor)),
                            out.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return out


def lorenz_logpdf(lg, yobs, x, xp, u, par, delta, nan_policy=0, chain_minor=False, dtype=np.float64):
    """The Lorenz sweep's fused log-density pass (csrc/kalman_bodies.h::body_lorenz_logpdf) on the host: x, xp, u (C, T, 3); par (C, 4) rows
    [theta, dt] or (4,); lg = (m0, P0, Fs*, Qs, bs*, Hs, Rs, cs) with the REAL observation model (Fs / bs unused).  Returns (5, C)."""
    x = np.asarray(x)
    C_, T, _ = x.shape
    po = np.shape(yobs)[-1]
    desc = _layout.describe_lgssm(list(lg), 1, T, 1, 3, po, False, dtype, False)
    g, keep = _garr(desc)
    cv = lambda a: np.ascontiguousarray(a, dtype)
    yo, xx, xxp, uu, pp = cv(yobs), cv(x), cv(xp), cv(u), cv(par)
    psc = 4 if pp.ndim == 2 and pp.shape[0] > 1 else 0
    out = np.empty((5, C_), dtype)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().hs_lorenz_logpdf(_dt(dtype), po, C_, T, g, ptr(yo), ptr(xx), ptr(xxp), ptr(uu), ptr(pp), psc, C.c_double(float(delta)), int(nan_policy),
                                int(bool(chain_minor)), ptr(out))
    assert rc == 0, rc
    return out


def fold_aux(dtype, F, Q, bd, Lobs, gobs, qobs, u, inv_hd, ldR, dim, b0, C0):
    """log-likelihood increment of one information-form step with the auxiliary block folded in around the origin / kept apart (hostsim.cpp::fold_aux_T):
    returns (zinc_folded, zinc_apart, max |db'|, max |dC'|) in `dtype` arithmetic"""
    d = len(bd)
    a = [np.ascontiguousarray(v, np.float64) for v in (F, Q, bd, Lobs, gobs, u, b0, C0)]
    P = C.POINTER(C.c_double)
    out = np.zeros(4)
    rc = lib().hs_fold_aux(C.c_int(1 if np.dtype(dtype) == np.float32 else 0), C.c_int(d), a[0].ctypes.data_as(P), a[1].ctypes.data_as(P), a[2].ctypes.data_as(P),
                           a[3].ctypes.data_as(P), a[4].ctypes.data_as(P), C.c_double(qobs), a[5].ctypes.data_as(P), C.c_double(inv_hd), C.c_double(ldR),
                           C.c_double(dim), a[6].ctypes.data_as(P), a[7].ctypes.data_as(P), out.ctypes.data_as(P))
    assert rc == 0, rc
    return tuple(out)


def fold_check(F, Q, bd, Lam, g0, q0, ldR, dim, acc):
    """max |prefix (+) element(step) - fold(prefix, step)| over all fields (kalman_math.h::filter_fold_step / filter_apply_step vs
    filter_elem_from_lam + filter_combine); Lam packed upper-symmetric, acc = [A | b | C packed | eta | J packed | z]"""
    d = len(bd)
    f = lib().hs_fold_check
    f.restype = C.c_double
    a = [np.ascontiguousarray(v, np.float64) for v in (F, Q, bd, Lam, g0, acc)]
    P = C.POINTER(C.c_double)
    return f(C.c_int(d), a[0].ctypes.data_as(P), a[1].ctypes.data_as(P), a[2].ctypes.data_as(P), a[3].ctypes.data_as(P), a[4].ctypes.data_as(P),
             C.c_double(q0), C.c_double(ldR), C.c_double(dim), a[5].ctypes.data_as(P))

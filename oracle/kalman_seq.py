"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/kalman_seq.c: the reference's SEQUENTIAL auxiliary-Kalman sweep
(`parallel=False`) for the LG-concat model, chains over OpenMP threads.  Used by bench.py's cpu_baseline leg and by
tests/test_oracle_kalman_seq.py.  Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libkalman_seq.so")


class _Model(C.Structure):
    _fields_ = [("T", C.c_int), ("d", C.c_int), ("po", C.c_int)] + [(n, C.c_void_p) for n in
                ("m0", "P0", "Fs", "Qs", "bs", "Hobs", "Robs", "cobs", "yobs")] + [(n, C.c_long) for n in ("sF", "sQ", "sb", "sH", "sR", "sc")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "kalman_seq.c")
        if not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(_SO) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", _HERE, "_build/libkalman_seq.so"])
        _lib = C.CDLL(_SO)
        _lib.kseq_sweep.restype = C.c_int
        _lib.kseq_max_threads.restype = C.c_int
    return _lib


def max_threads():
    return int(lib().kseq_max_threads())


class Model:
    """m0 (d), P0 (d,d); Fs, Qs (T-1,d,d) or (d,d); bs (T-1,d) or (d); Hobs (T,po,d) or (po,d); Robs; cobs; yobs (T,po)."""

    def __init__(self, m0, P0, Fs, Qs, bs, Hobs, Robs, cobs, yobs):
        f = lambda a: np.ascontiguousarray(a, np.float64)
        self.yobs = f(yobs)
        self.T, self.po = self.yobs.shape
        self.m0, self.P0 = f(m0), f(P0)
        self.d = self.m0.shape[0]
        d, po, T = self.d, self.po, self.T
        self.keep = []

        def tv(a, core, n):
            a = np.asarray(a)
            if a.ndim == len(core):
                a = f(a)
                st = 0
            elif a.strides[0] == 0:  # broadcast view: time-invariant
                a = f(a[0])
                st = 0
            else:
                a = f(a)
                assert a.shape[0] >= n, (a.shape, n)
                st = int(np.prod(core))
            self.keep.append(a)
            return a, st

        (self.Fs, sF), (self.Qs, sQ), (self.bs, sb) = tv(Fs, (d, d), T - 1), tv(Qs, (d, d), T - 1), tv(bs, (d,), T - 1)
        (self.Hobs, sH), (self.Robs, sR), (self.cobs, sc) = tv(Hobs, (po, d), T), tv(Robs, (po, po), T), tv(cobs, (po,), T)
        p = lambda a: a.ctypes.data
        self.c = _Model(T, d, po, p(self.m0), p(self.P0), p(self.Fs), p(self.Qs), p(self.bs), p(self.Hobs), p(self.Robs), p(self.cobs),
                        p(self.yobs), sF, sQ, sb, sH, sR, sc)


def sweep(model, x, delta, eps_aux, eps_samp, u_acc, nthreads=0, want_prop=False):
    """x (C, T, d) -> dict(x, accepted, logs (C, 5) = log_alpha, lp_prop, lp_rev, lt_prop, lt_rev[, x_prop])."""
    x = np.array(x, np.float64, order="C", copy=True)
    Cn = x.shape[0]
    ea, es = np.ascontiguousarray(eps_aux, np.float64), np.ascontiguousarray(eps_samp, np.float64)
    ua = np.ascontiguousarray(np.broadcast_to(u_acc, (Cn,)), np.float64)
    assert x.shape == ea.shape == es.shape == (Cn, model.T, model.d)
    acc = np.zeros(Cn, np.int32)
    logs = np.zeros((Cn, 5))
    xp = np.zeros_like(x) if want_prop else None
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib().kseq_sweep(C.byref(model.c), Cn, C.c_double(float(delta)), vp(x), vp(ea), vp(es), vp(ua), vp(acc), vp(logs), vp(xp), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"kseq_sweep failed ({rc})")
    out = dict(x=x, accepted=acc.astype(bool), logs=logs)
    if want_prop:
        out["x_prop"] = xp
    return out

"""Shape logic between the reference's array conventions and the C ABI's strided descriptors.

Reference shapes (_primitives/kalman/base.py:29-49): time-major, optional batch axis B after time:
    Fs (T-1, [B,] dx, dx)   ys (T, [B,] dy)   m0 ([B,] dx) ...
This module adds an optional leading chain axis C and maps every array to (contiguous buffer, chain/time/batch
strides in elements).  Axes that are broadcast views (stride 0, e.g. ``np.broadcast_to(F, (T-1, d, d))``) are
collapsed and get stride 0, so time-invariant or chain-shared parameters are stored (and read from HBM) once.
"""
import numpy as np


class Described:
    __slots__ = ("buf", "sc", "st", "sb")

    def __init__(self, buf, sc, st, sb):
        self.buf, self.sc, self.st, self.sb = buf, sc, st, sb


def describe(a, core_shape, *, chains, time_len, batch, dtype, name="array"):
    """a: array with leading axes [C][T][B] then `core_shape`.
    chains: None (no chain axis) or C.  time_len: None (no time axis) or the expected length.  batch: None or B."""
    a = np.asarray(a)
    lead = []
    if chains is not None:
        lead.append(("c", chains))
    if time_len is not None:
        lead.append(("t", time_len))
    if batch is not None:
        lead.append(("b", batch))
    want = tuple(n for _, n in lead) + tuple(core_shape)
    if a.shape != want:
        raise ValueError(f"{name}: expected shape {want}, got {a.shape}")
    if a.dtype != dtype:
        a = a.astype(dtype)
    # collapse broadcast (stride-0) leading axes
    idx = []
    kept = []
    for ax, (tag, n) in enumerate(lead):
        if n > 1 and a.strides[ax] == 0:
            idx.append(0)
        else:
            idx.append(slice(None))
            kept.append(tag)
    buf = np.ascontiguousarray(a[tuple(idx)])
    strides = {"c": 0, "t": 0, "b": 0}
    rec = int(np.prod(core_shape, dtype=np.int64)) if len(core_shape) else 1
    step = rec
    for tag in reversed(kept):
        strides[tag] = step
        step *= dict(lead)[tag]
    # axes of length 1 need no stride
    return Described(buf, strides["c"], strides["t"], strides["b"])


LGSSM_FIELDS = ("m0", "P0", "Fs", "Qs", "bs", "Hs", "Rs", "cs")


def infer_dims(ys, lgssm, chains):
    """(C, T, B, dx, dy, batched) from reference-shaped inputs."""
    ys = np.asarray(ys) if not hasattr(ys, "shape") else ys
    nlead = ys.ndim - 1 - (1 if chains else 0)
    if nlead not in (1, 2):
        raise ValueError(f"ys must be ([C,] T, [B,] dy); got shape {ys.shape}")
    batched = nlead == 2
    off = 1 if chains else 0
    C = ys.shape[0] if chains else 1
    T = ys.shape[off]
    B = ys.shape[off + 1] if batched else 1
    dy = ys.shape[-1]
    dx = np.shape(lgssm[0])[-1]
    return C, T, B, dx, dy, batched


def describe_lgssm(lgssm, C, T, B, dx, dy, batched, dtype, chain_axis):
    """chain_axis: dict name -> bool (does this array carry the leading chain axis?) or a single bool."""
    m0, P0, Fs, Qs, bs, Hs, Rs, cs = lgssm
    bt = B if batched else None

    def ca(name):
        has = chain_axis.get(name, False) if isinstance(chain_axis, dict) else chain_axis
        return C if has else None

    n = T - 1
    out = {}
    out["m0"] = describe(m0, (dx,), chains=ca("m0"), time_len=None, batch=bt, dtype=dtype, name="m0")
    out["P0"] = describe(P0, (dx, dx), chains=ca("P0"), time_len=None, batch=bt, dtype=dtype, name="P0")
    out["Fs"] = describe(Fs, (dx, dx), chains=ca("Fs"), time_len=n, batch=bt, dtype=dtype, name="Fs")
    out["Qs"] = describe(Qs, (dx, dx), chains=ca("Qs"), time_len=n, batch=bt, dtype=dtype, name="Qs")
    out["bs"] = describe(bs, (dx,), chains=ca("bs"), time_len=n, batch=bt, dtype=dtype, name="bs")
    if Hs is not None:
        out["Hs"] = describe(Hs, (dy, dx), chains=ca("Hs"), time_len=T, batch=bt, dtype=dtype, name="Hs")
        out["Rs"] = describe(Rs, (dy, dy), chains=ca("Rs"), time_len=T, batch=bt, dtype=dtype, name="Rs")
        out["cs"] = describe(cs, (dy,), chains=ca("cs"), time_len=T, batch=bt, dtype=dtype, name="cs")
    return out

#!/usr/bin/env python3
"""wide-path filter against the NumPy oracle over observation counts around the blocked elimination's limit (n = 64) -- diagnostic for wide.hip::spd_solve's split."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aux_ssm_samplers_amd._primitives.kalman as P  # noqa: E402
from oracle import kalman_np as K  # noqa: E402

rng = np.random.default_rng(0)
for dtype in (np.float64, np.float32):
    for d, p in ((8, 60), (8, 64), (8, 65), (8, 68), (8, 100), (8, 128), (34, 68), (34, 64), (20, 70), (40, 66)):
        T = 12
        F = 0.9 * np.eye(d) + 0.02 * rng.standard_normal((d, d))
        Q = 0.3 * np.eye(d)
        H = rng.standard_normal((p, d)) / np.sqrt(d)
        A = rng.standard_normal((p, p)) / np.sqrt(p)
        Rm = 0.5 * np.eye(p) + 0.1 * A @ A.T
        bt = np.broadcast_to
        lg = (np.zeros(d), np.eye(d), bt(F, (T - 1, d, d)), bt(Q, (T - 1, d, d)), bt(np.zeros(d), (T - 1, d)), bt(H, (T, p, d)), bt(Rm, (T, p, p)), bt(np.zeros(p), (T, p)))
        ys = rng.standard_normal((T, p))
        ys[3] = np.nan
        ys[5, : p // 2] = np.nan
        ys[7, p - 3:] = np.nan
        try:
            ms, Ps, ell = P.filtering(ys.astype(dtype), P.LGSSM(*[np.ascontiguousarray(a, dtype) for a in lg]), True)
        except ValueError as e:
            print(np.dtype(dtype).name, d, p, "skipped:", str(e)[:70])
            continue
        oms, oPs, oell = K.filtering(ys, lg, True)
        print(np.dtype(dtype).name, d, p, "max|dm| %.2e  |dell| %.2e  ell %.3f" % (np.nanmax(np.abs(ms - oms)), abs(ell - oell), oell), "NaN!" if not np.isfinite(ms).all() else "")

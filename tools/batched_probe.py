#!/usr/bin/env python3
"""The reference's spatial example as it runs it (examples/spatial/model.py:103-112): B = 64 independent scalar LGSSMs on the batch axis, C sequences of observations
on shared parameters -- `auxssm_kalman_filter` through the C ABI on resident buffers, wall time per call.  usage: batched_probe.py [T:C ...]   (default 1024:64 8192:16
1024:1 8192:1);  AUXSSM_FILTER_BATCH_LANES=0 runs the time-minor passes for comparison."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib  # noqa: E402
from aux_ssm_samplers_amd._primitives.kalman.base import DeviceLGSSM  # noqa: E402
from aux_ssm_samplers_amd.workloads import c5_batched_model  # noqa: E402


def main():
    shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(1024, 64), (8192, 16), (1024, 1), (8192, 1)]
    h = _lib.default_handle()
    f32 = np.float32
    for T, Cn in shapes:
        ub, lgb, xb = c5_batched_model(T)
        B = ub.shape[1]
        dl = DeviceLGSSM(h, tuple(lgb), 1, T, B, 1, 1, True, f32)
        ys = (ub[None] + 0.3 * np.random.default_rng(5).standard_normal((Cn, T, B, 1))).astype(f32)
        yd = h.to_device(ys)
        yarr = yd.arr(T * B, B, 1)
        ms, Ps, ell = h.empty((Cn, T, B, 1), f32), h.empty((Cn, T, B, 1, 1), f32), h.empty((Cn,), f32)
        dims = _lib.Dims(Cn, T, B, 1, 1)
        for par in (1, 0):
            def step():
                _lib.check(h.lib.auxssm_kalman_filter(h.h, _lib.F32, C.byref(dims), C.byref(dl.c), C.byref(yarr), par, ms.ptr, Ps.ptr, ell.ptr))
            for _ in range(3):
                step()
            h.sync()
            n = 20
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            h.sync()
            el = (time.perf_counter() - t0) / n
            print(json.dumps(dict(config=f"B={B} scalar LGSSMs x {Cn} sequences, T={T}, fp32, parallel={par}", ms_per_call=round(el * 1e3, 4),
                                  scalar_filters_per_s=round(Cn * B / el, 1), steps_per_s=round(Cn * B * T / el / 1e9, 3), ell0=float(ell.to_host()[0]))), flush=True)


if __name__ == "__main__":
    main()

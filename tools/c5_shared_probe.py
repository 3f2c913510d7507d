"""Launch-group times of the chain-shared wide filter at C5's sizes (d = p = 64, T = 8192, fp32, S sequences): python tools/c5_shared_probe.py [S] [reps]
(AUXSSM_LIB=<variant .so> for the phase-ablation builds of wk_gain_tab: -DAUXSSM_GT_PHASE=k)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib  # noqa: E402
from aux_ssm_samplers_amd._primitives.kalman.base import DeviceLGSSM  # noqa: E402
from aux_ssm_samplers_amd.workloads import c5_model  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T, d = 8192, 64
u, lg64, _ = c5_model(T, d)
h = _lib.default_handle()
f32 = np.float32
dl = DeviceLGSSM(h, tuple(lg64), 1, T, 1, d, d, False, f32)
ys = (u[None] + np.concatenate([np.zeros((1, T, d)), 0.3 * np.random.default_rng(4).standard_normal((S - 1, T, d))])).astype(f32)
yd = h.to_device(ys)
yarr = yd.arr(T * d, d, 0)
ms, Ps, ell = h.empty((S, T, 1, d), f32), h.empty((S, T, 1, d, d), f32), h.empty((S,), f32)
dims = _lib.Dims(S, T, 1, d, d)


def step():
    _lib.check(h.lib.auxssm_kalman_filter(h.h, _lib.F32, C.byref(dims), C.byref(dl.c), C.byref(yarr), 1, ms.ptr, Ps.ptr, ell.ptr))


step()
h.sync()
h.prof_enable(_lib.K_ALL, 64 * reps)
for _ in range(reps):
    step()
g = h.prof_read_groups()
h.prof_disable()
t0 = time.perf_counter()
for _ in range(reps):
    step()
h.sync()
el = (time.perf_counter() - t0) / reps
print(os.environ.get("AUXSSM_LIB", "default"), f"S={S} wall {el * 1e3:.3f} ms/call  {S / el:.0f} filters/s ", {k: round(v[1] / reps, 3) for k, v in g.items()}, flush=True)

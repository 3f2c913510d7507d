#!/usr/bin/env python3
"""Which piece of the fp32 Lorenz sweep loses precision at C4's size (T = 16384, dt = 1.25e-4)?  The auxiliary LGSSM linearised at the true path through the
filtering / posterior_logpdf primitives in fp32 and fp64, parallel and sequential: ell, filtered means / covariances, the joint log-density.  GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aux_ssm_samplers_amd._primitives.kalman as P  # noqa: E402
from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
rng = np.random.default_rng(0)
u = xtrue + np.sqrt(delta / 2) * rng.standard_normal((T, 3))
dyn = model.dynamics_factory(xtrue)
obs = model.observations_factory(xtrue, u, delta)
res = {}
for dt_ in (np.float64, np.float32):
    lg = P.LGSSM(*[np.ascontiguousarray(a, dt_) for a in (dyn[0], dyn[1], dyn[2], dyn[3], dyn[4], obs[1], obs[2], obs[3])])
    ys = np.ascontiguousarray(obs[0], dt_)
    for par in (True, False):
        ms, Ps, ell = P.filtering(ys, lg, par)
        xs = P.sampling(None, ms, Ps, lg, par, eps=rng.standard_normal((T, 3)).astype(dt_) * 0 + 0.3)
        lp = P.posterior_logpdf(ys, xs, ell, lg)
        res[(dt_.__name__, par)] = (np.asarray(ms, np.float64), np.asarray(Ps, np.float64), float(ell), float(lp), np.asarray(xs, np.float64))
ref = res[("float64", False)]
for k, (ms, Ps, ell, lp, xs) in res.items():
    print(k, f"ell {ell:.4f} (d {ell - ref[2]:+.4f})  lp {lp:.4f} (d {lp - ref[3]:+.4f})  max|dm| {np.abs(ms - ref[0]).max():.3e}  "
          f"max rel dP {np.abs(Ps - ref[1]).max() / np.abs(ref[1]).max():.3e}  max|dx| {np.abs(xs - ref[4]).max():.3e}")
# where along the horizon do the fp32 covariances lose accuracy?
for par in (True, False):
    P32, P64 = res[("float32", par)][1], res[("float64", False)][1]
    rel = np.abs(P32 - P64).reshape(T, -1).max(1) / np.abs(P64).reshape(T, -1).max(1)
    pick = [0, 1, 2, 10, 79, 80, 81, 100, 1000, 8000, T - 1]
    print("parallel" if par else "sequential", "rel dP at t:", {t: float(f"{rel[t]:.2e}") for t in pick}, "steps with rel dP > 1e-2:", int((rel > 1e-2).sum()),
          "> 1e-3:", int((rel > 1e-3).sum()), "median", float(np.median(rel)))
    m32, m64 = res[("float32", par)][0], res[("float64", False)][0]
    print("   |dm| at t:", {t: float(f"{np.abs(m32[t] - m64[t]).max():.2e}") for t in pick})

#!/usr/bin/env python3
"""A whole auxiliary Kalman sweep on a WIDE linear-Gaussian model with several chains (dense layout): filter (chain-shared matrix recursion), pathwise sampler,
log-densities, accept -- launch-group times from the library's profiler and wall time per sweep.  usage: wide_sweep_probe.py [d po T C [f64]]; AUXSSM_SHARED=0 /
OPT_SHARE_MODEL off (second pass of this script) is the per-chain path."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402


def model_(T, d, po, seed=0):
    rng = np.random.default_rng(seed)
    F = 0.9 * np.eye(d) + 0.04 * (np.eye(d, k=1) + np.eye(d, k=-1))
    Q = 0.2 * np.eye(d)
    Hobs = rng.standard_normal((po, d)) / np.sqrt(d)
    Robs = 0.5 * np.eye(po)
    x = np.zeros((T, d))
    x[0] = rng.standard_normal(d)
    for t in range(1, T):
        x[t] = F @ x[t - 1] + np.sqrt(0.2) * rng.standard_normal(d)
    y = x @ Hobs.T + np.sqrt(0.5) * rng.standard_normal((T, po))
    bt = np.broadcast_to
    return LGConcatModel(np.zeros(d), np.eye(d), bt(F, (T - 1, d, d)), bt(Q, (T - 1, d, d)), bt(np.zeros(d), (T - 1, d)), bt(Hobs, (T, po, d)), bt(Robs, (T, po, po)),
                         bt(np.zeros(po), (T, po)), y), x


def main():
    a = sys.argv[1:]
    d, po, T, C = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (64, 8, 8192, 16)
    dtype = np.float64 if "f64" in a else np.float32
    model, xt = model_(T, d, po)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    h = _lib.default_handle()
    x0 = (xt[None] + 0.3 * np.random.default_rng(1).standard_normal((C, T, d))).astype(dtype)
    for share in (1, 0):
        h.set_option(_lib.OPT_SHARE_MODEL, share)
        ch = DeviceChains(h, x0)
        st = KalmanSampler(x=ch, updated=None)
        keys = R.split(R.PRNGKey(3), 8)
        kernel(keys[0], st, 0.3)
        h.sync()
        h.prof_enable(_lib.K_ALL, 512)
        kernel(keys[1], st, 0.3)
        groups = h.prof_read_groups()
        h.prof_disable()
        n = 3
        t0 = time.perf_counter()
        for k in range(n):
            kernel(keys[2 + k], st, 0.3)
        h.sync()
        el = (time.perf_counter() - t0) / n
        print(json.dumps(dict(config=f"wide LG sweep d={d} po={po} T={T} chains={C} {np.dtype(dtype).name}, share_model={share}", ms_per_sweep_call=round(el * 1e3, 3),
                              chain_sweeps_per_s=round(C / el, 1), groups_ms={k: round(v[1], 3) for k, v in groups.items()}, accept=float(ch.accepted.to_host().mean()))), flush=True)
    h.set_option(_lib.OPT_SHARE_MODEL, 1)


if __name__ == "__main__":
    main()

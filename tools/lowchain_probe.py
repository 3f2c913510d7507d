#!/usr/bin/env python3
"""Low-chain-count sweeps (latency-bound: the time-minor path): C2 at 1 / 8 chains (fp64) and C4's 8-chain shard (Lorenz, fp32).
usage: lowchain_probe.py [c2_1 c2_8 c4_8 ...] -- prints sweeps/s; run under rocprofv3 --kernel-trace + tools/timeline.py <csv> k_accept for the launch chain."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402
from aux_ssm_samplers_amd.workloads import lg_model, lorenz_kalman_setup  # noqa: E402


def run(name, kernel, ch, delta, steps=30, warmup=5):
    h = ch.handle
    st = KalmanSampler(x=ch, updated=None)
    keys = R.split(R.PRNGKey(1), steps + warmup)
    for k in range(warmup):
        kernel(keys[k], st, delta)
    h.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        kernel(keys[warmup + k], st, delta)
    h.sync()
    el = time.perf_counter() - t0
    print(json.dumps(dict(config=name, chains=ch.C, sweeps_per_s=round(ch.C * steps / el, 1), ms_per_sweep_call=round(el / steps * 1e3, 4),
                          accept=float(ch.accepted.to_host().mean()))), flush=True)


def main():
    what = sys.argv[1:] or ["c2_1", "c2_8", "c4_8"]
    h = _lib.default_handle()
    for w in what:
        if w.startswith("c2_"):
            C = int(w[3:])
            T, d = 65536, 4
            m = lg_model(T, d)
            bt = np.broadcast_to
            model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)), bt(m["Hobs"], (T, d, d)),
                                  bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), m["y"])
            init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
            ch = DeviceChains(h, m["x_true"][None] + 0.3 * np.random.default_rng(0).standard_normal((C, T, d)), model=model)
            run(f"C2 LG-SSM T={T} d={d} fp64", kernel, ch, 0.5)
        elif w.startswith("c4_"):
            C = int(w[3:])
            T = 16384
            model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
            init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
            ch = DeviceChains(h, np.repeat(xtrue[None], C, axis=0).astype(np.float32))
            run(f"C4 Lorenz-63 T={T} fp32", kernel, ch, 1e-4)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The general per-chain path (what every nonlinear model runs): C2 with the chain-shared mode off (`c2g_<chains>`), the SV second-order auxiliary Kalman sampler
(`sv2_<chains>`, `sv1_<chains>`) and the Lorenz sampler (`lz_<chains>`), a few keyed sweeps each.  Prints sweeps/s; under `rocprofv3 --kernel-trace` +
tools/timeline.py <csv> k_accept it gives the launch chain of one sweep."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel, SVModel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402
from aux_ssm_samplers_amd.workloads import lg_model, lorenz_kalman_setup, sv_setup  # noqa: E402


def run(name, kernel, ch, delta, steps=8, warmup=3):
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
    what = sys.argv[1:] or ["c2g_256", "sv2_1024"]
    h = _lib.default_handle()
    for w in what:
        kind, C = w.split("_")
        C = int(C)
        if kind == "c2g":
            h.set_option(_lib.OPT_SHARE_MODEL, 0)
            T, d = 65536, 4
            m = lg_model(T, d)
            bt = np.broadcast_to
            model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)), bt(m["Hobs"], (T, d, d)),
                                  bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), m["y"])
            init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
            ch = DeviceChains(h, m["x_true"][None] + 0.3 * np.random.default_rng(0).standard_normal((C, T, d)))
            run(f"C2 LG-SSM T={T} d={d} fp64, general per-chain path", kernel, ch, 0.5)
            h.set_option(_lib.OPT_SHARE_MODEL, 1)
        elif kind in ("sv1", "sv2"):
            T = 65536
            y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, 1, rho=0.0)
            model = SVModel(y, m0, P0, F, Q, b, order=int(kind[2]))
            init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
            ch = DeviceChains(h, np.repeat(xtrue[None], C, axis=0))
            run(f"SV d=1 T={T} aux-Kalman order {kind[2]} fp64", kernel, ch, 0.0567 if kind == "sv2" else 0.0212)
        elif kind == "lz":
            T = 16384
            model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
            init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
            ch = DeviceChains(h, np.repeat(xtrue[None], C, axis=0).astype(np.float32))
            run(f"C4 Lorenz-63 T={T} fp32", kernel, ch, 1e-4)


if __name__ == "__main__":
    main()

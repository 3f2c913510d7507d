#!/usr/bin/env python3
"""C2 (T = 65536, d = 4, fp64, 256 chains) with random.set_compat("jax"): the sweeps consume jax.random's own draws (auxssm_rng_jax fills into the resident noise buffers,
then the explicit-noise sweep) -- sweeps/s beside the default (in-kernel draws, fused sweep).  GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib, random as R
from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
from aux_ssm_samplers_amd.workloads import lg_model

T, d, C = 65536, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = lg_model(T, d)
bt = np.broadcast_to
model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)), bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)),
                      bt(m["cobs"], (T, d)), m["y"])
init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
h = _lib.default_handle()
x0 = m["x_true"][None] + 0.3 * np.random.default_rng(0).standard_normal((C, T, d))
for mode in (None, "jax"):
    R.set_compat(mode)
    ch = DeviceChains(h, x0, model=model)
    st = KalmanSampler(x=ch, updated=None)
    keys = R.jax_split(np.array([0, 1], np.uint32), 16)
    for k in keys[:4]:
        kernel(k, st, 0.5)
    h.sync()
    t0 = time.perf_counter()
    for k in keys[4:]:
        kernel(k, st, 0.5)
    h.sync()
    el = time.perf_counter() - t0
    print(json.dumps(dict(mode=mode or "own streams", chains=C, sweeps_per_s=round(C * 12 / el, 1), ms_per_sweep=round(el / 12 * 1e3, 3),
                          accept=float(ch.accepted.to_host().mean()), max_abs_log_alpha=float(np.abs(ch.logs.to_host()[:, 0]).max()))), flush=True)
R.set_compat(None)

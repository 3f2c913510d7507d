#!/usr/bin/env python3
"""Does overlapping the issue-bound pass AC of one half of the chains with the HBM-bound pass E of the other half pay?  C2 (T = 65536, d = 4, fp64): one handle with 256
chains against K handles (own streams) with 256 / K chains each, sweeps enqueued round-robin; aggregate sweeps/s.  GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib, random as R
from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
from aux_ssm_samplers_amd.workloads import lg_model

T, d, Ctot = 65536, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = lg_model(T, d)
bt = np.broadcast_to
model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)), bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)),
                      bt(m["cobs"], (T, d)), m["y"])
init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
x0 = m["x_true"][None] + 0.3 * np.random.default_rng(0).standard_normal((Ctot, T, d))
for K in (1, 2, 4):
    hs = [_lib.Handle(0) for _ in range(K)]
    Ck = Ctot // K
    chs = [DeviceChains(h, x0[i * Ck:(i + 1) * Ck], chain_minor=True) for i, h in enumerate(hs)]
    sts = [KalmanSampler(x=c, updated=None) for c in chs]
    steps, warm = 30, 6
    keys = R.split(R.PRNGKey(5), (steps + warm) * K)
    for s in range(warm):
        for i in range(K):
            kernel(keys[s * K + i], sts[i], 0.5)
    for h in hs:
        h.sync()
    t0 = time.perf_counter()
    for s in range(warm, warm + steps):
        for i in range(K):
            kernel(keys[s * K + i], sts[i], 0.5)
    for h in hs:
        h.sync()
    el = time.perf_counter() - t0
    print(json.dumps(dict(handles=K, chains_each=Ck, sweeps_per_s=round(Ctot * steps / el, 1), ms_per_round=round(el / steps * 1e3, 4))), flush=True)
    del chs, sts, hs

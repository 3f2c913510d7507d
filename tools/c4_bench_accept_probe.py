#!/usr/bin/env python3
"""bench.py's C4 Kalman leg sweep by sweep: acceptance of every sweep with the bench's own keys (diagnostic for the `accept_rate` it prints)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib, random as R
from aux_ssm_samplers_amd.parallel import chain_key
from aux_ssm_samplers_amd.kalman import get_kernel
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup
T, Cn = 16384, 64
h = _lib.default_handle()
model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
ch = DeviceChains(h, np.repeat(xtrue[None], Cn, axis=0).astype(np.float32))
st = KalmanSampler(x=ch, updated=None)
keys = R.split(chain_key(R.PRNGKey(4), 0), 24)
seq = list(range(3)) + list(range(3)) + list(range(3, 23))
acc = []
for k in seq:
    kernel(keys[k], st, 1e-4)
    acc.append(float(ch.accepted.to_host().mean()))
print("acceptance per sweep (bench order: 3 warm-up, 3 profile, 20 timed):", np.round(acc, 3))
print("log alpha of the last sweep:", np.round(ch.logs.to_host()[:, 0], 1))

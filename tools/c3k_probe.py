#!/usr/bin/env python3
"""One configuration of the SV aux-Kalman sweep for rocprofv3: python3 tools/c3k_probe.py order chains [steps]  (C3's model, T = 65536, fp64)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from aux_ssm_samplers_amd.workloads import sv_setup  # noqa: E402
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.kalman import get_kernel, SVModel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402

order, chains = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
T = 65536
y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, 1, rho=0.0)
model = SVModel(y, m0, P0, F, Q, b, order=order)
init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
h = _lib.default_handle()
ch = DeviceChains(h, np.repeat(xtrue[None], chains, axis=0))
st = KalmanSampler(x=ch, updated=None)
keys = R.split(R.PRNGKey(1), steps + 1)
for k in keys:
    kernel(k, st, 0.02 if order == 1 else 0.057)
h.sync()
print("accept", ch.accepted.to_host().mean())

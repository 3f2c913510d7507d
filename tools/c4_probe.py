#!/usr/bin/env python3
"""The Lorenz-63 aux-Kalman sweep of config C4 for rocprofv3: python3 tools/c4_probe.py chains [steps]  (T = 16384, fp32)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup  # noqa: E402
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.kalman import get_kernel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402

chains = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T = 16384
model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
h = _lib.default_handle()
ch = DeviceChains(h, np.repeat(xtrue[None], chains, axis=0).astype(np.float32))
st = KalmanSampler(x=ch, updated=None)
keys = R.split(R.PRNGKey(1), steps + 2)
kernel(keys[0], st, 1e-4)
kernel(keys[1], st, 1e-4)
h.sync()
t0 = time.perf_counter()
for k in keys[2:]:
    kernel(k, st, 1e-4)
h.sync()
el = (time.perf_counter() - t0) / steps
print(f"chains={chains}: {el * 1e3:.3f} ms per step, {chains / el:.0f} sweeps/s, accept {ch.accepted.to_host().mean():.2f}")

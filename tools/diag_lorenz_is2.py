"""Long run of the Lorenz sampler (delta = 50) with batch-mean standard errors, against the adaptive importance-sampling values (mean -0.903 / var 70.3 for x_1 at t = 0)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.kalman import get_kernel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402
from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup  # noqa: E402

T, C = 3, 4096
model, xtrue = lorenz_kalman_setup(T, every=1, dt=0.05, seed=3)
init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
h = _lib.default_handle()
for cm in (False,):
    chains = DeviceChains(h, xtrue[None] + 0.5 * np.random.default_rng(2).standard_normal((C, T, 3)), chain_minor=cm)
    state = KalmanSampler(x=chains, updated=None)
    keys = R.split(R.PRNGKey(11), 8200)
    m1, m2 = [], []
    for i, k in enumerate(keys):
        kernel(k, state, 50.0)
        if i >= 200:
            xs = chains.to_host().reshape(C, 9)
            m1.append(xs.mean(0)); m2.append((xs ** 2).mean(0))
    m1, m2 = np.array(m1), np.array(m2)
    B = 16
    b1 = m1.reshape(B, -1, 9).mean(1); b2 = m2.reshape(B, -1, 9).mean(1)
    mean = m1.mean(0); var = m2.mean(0) - mean ** 2
    bv = b2 - b1 ** 2
    print("mean", np.round(mean, 3), "\n se ", np.round(b1.std(0, ddof=1) / np.sqrt(B), 3))
    print("var ", np.round(var, 3), "\n se ", np.round(bv.std(0, ddof=1) / np.sqrt(B), 3))

"""Lorenz extended-Kalman sampler on the T = 3 model of tests/test_gpu_nonlinear_kalman.py: chain variances of the unobserved x_1 components against the step size and the
number of sweeps (python tools/diag_lorenz_is.py; GPU box).  Importance-sampling values of the test: var x_1 = 67.1 / 16.75 / 5.15 at t = 0, 1, 2."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.kalman import get_kernel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402
from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup  # noqa: E402

T, C = 3, 1024
model, xtrue = lorenz_kalman_setup(T, every=1, dt=0.05, seed=3)
init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
h = _lib.default_handle()
for delta in (0.5, 5.0, 50.0):
    chains = DeviceChains(h, xtrue[None] + 0.5 * np.random.default_rng(2).standard_normal((C, T, 3)), chain_minor=False)
    state = KalmanSampler(x=chains, updated=None)
    keys = R.split(R.PRNGKey(8), 2100)
    s1, s2, n, acc = np.zeros(9), np.zeros(9), 0, 0.0
    for i, k in enumerate(keys):
        kernel(k, state, delta)
        if i >= 100:
            xs = chains.to_host().reshape(C, 9)
            s1 += xs.mean(0); s2 += (xs ** 2).mean(0); n += 1; acc += chains.accepted.to_host().mean()
            if n in (500, 2000):
                v = s2 / n - (s1 / n) ** 2
                print(f"delta {delta} sweeps {n}: mean x1 {np.round((s1 / n)[[0, 3, 6]], 3)} var x1 {np.round(v[[0, 3, 6]], 2)} acc {acc / n:.3f}", flush=True)

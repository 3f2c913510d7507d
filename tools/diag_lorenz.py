import os, sys
sys.path.insert(0, os.getcwd())
from functools import partial
import numpy as np
from aux_ssm_samplers_amd import _lib, random as R
from aux_ssm_samplers_amd.common import delta_adaptation
from aux_ssm_samplers_amd.kalman import get_kernel, LorenzModel
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
from aux_ssm_samplers_amd.loop import loop, LorenzThetaStep
GOLD = "tests/golden"
dtype = np.float64 if len(sys.argv) > 1 else np.float32
data = np.loadtxt(os.path.join(GOLD, "lorenz_data.csv"), delimiter=",", skiprows=1)
t_end, obs_freq, dt = data[-1, 0], data[1, 0] - data[0, 0], 20 * 1e-4
n_steps = int(t_end / dt + 1e-6) + 1
every = int(obs_freq / dt + 1e-6)
ys = np.full((n_steps, 2), np.nan); ys[::every] = data[:, 1:]
Hs = np.full((n_steps, 2, 3), np.nan); Hs[::every] = np.array([[0, 1.0, 0], [0, 0, 1.0]])
Rs = np.broadcast_to(5.0 * np.eye(2), (n_steps, 2, 2))
C = 4
model = LorenzModel(ys, Hs, Rs, np.zeros((n_steps, 2)), [1.5, -1.5, 25.0], np.diag([400.0, 20.0, 20.0]), np.tile([5.0, 15.0, 6.0], (C, 1)), 3.0, dt)
init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
ts = np.linspace(0, t_end, n_steps)
x0 = np.stack([np.interp(ts, data[:, 0], data[:, 1]), np.interp(ts, data[:, 0], data[:, 1]), np.interp(ts, data[:, 0], data[:, 2])], 1)
h = _lib.default_handle()
import sys
for seed in range(1, 9):
    chains = DeviceChains(h, np.repeat(x0[None], C, 0).astype(dtype), chain_minor=False)
    model.theta0 = None
    model = LorenzModel(ys, Hs, Rs, np.zeros((n_steps, 2)), [1.5, -1.5, 25.0], np.diag([400.0, 20.0, 20.0]), np.tile([5.0, 15.0, 6.0], (C, 1)), 3.0, dt)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    step = LorenzThetaStep(model, 1e3 ** 0.5)
    rec = []
    def cb(i, s):
        if i % 25 == 0:
            rec.append((i, chains.accepted.to_host().copy(), step.theta(chains).copy(), chains.logs.to_host()[:, 0].copy()))
    burn = loop(R.PRNGKey(seed), 1e-5, KalmanSampler(x=chains, updated=True), kernel, partial(delta_adaptation, min_delta=1e-15), 1500,
                target_alpha=0.234, lr=1.0, beta=0.05, theta_step=step, callback=cb)
    th = step.theta(chains)
    print("seed", seed, "delta", burn[3], "final theta0 per chain", th[:, 0], "acc-window", np.asarray(burn[4].to_host() if hasattr(burn[4],"to_host") else burn[4]).ravel(), flush=True)
    badc = np.nonzero(np.abs(th[:, 0] - 10) > 5)[0]
    for c in badc:
        print("  chain", c, "theta0 path", [round(float(r[2][c, 0]), 1) for r in rec])
        print("  chain", c, "log alpha path", [round(float(r[3][c]), 1) for r in rec])

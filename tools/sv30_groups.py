import sys, json, numpy as np
sys.path.insert(0, '/root/repo')
from aux_ssm_samplers_amd import _lib, random as R
from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
from aux_ssm_samplers_amd.workloads import sv_setup
T, D = 250, 30
y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, D)
h = _lib.default_handle()
for order in (1, 2):
    model = SVModel(y, m0, P0, F, Q, b, order=order)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    for C_ in (16, 64):
        ch = DeviceChains(h, np.repeat(xtrue[None], C_, axis=0), chain_minor=False)
        st = KalmanSampler(x=ch, updated=None)
        kernel(R.PRNGKey(0), st, 0.01)
        h.sync()
        h.prof_enable(_lib.K_ALL, 512)
        kernel(R.PRNGKey(1), st, 0.01)
        g = h.prof_read_groups()
        h.prof_disable()
        print(order, C_, {k: round(v[1], 3) for k, v in g.items()}, round(sum(v[1] for v in g.values()), 3))

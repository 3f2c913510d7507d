import sys, json, numpy as np
sys.path.insert(0, '.')
from aux_ssm_samplers_amd import _lib, random as R
from aux_ssm_samplers_amd.kalman import get_kernel, DeviceChains, KalmanSampler, SVModel
from aux_ssm_samplers_amd.workloads import sv_setup
T = 65536
y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, 1, rho=0.0)
h = _lib.default_handle()
for chains in (64, 256):
    model = SVModel(y, m0, P0, F, Q, b, order=1)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    ch = DeviceChains(h, np.repeat(xtrue[None], chains, axis=0), chain_minor=None)
    st = KalmanSampler(x=ch, updated=True)
    for i in range(3): kernel(R.PRNGKey(i), st, 0.02)
    h.sync()
    h.prof_enable(_lib.K_ALL, 64 * 6)
    for i in range(5): kernel(R.PRNGKey(10 + i), st, 0.02)
    h.sync()
    g = h.prof_read_groups(); h.prof_disable()
    print(chains, {k: (n / 5, round(ms / 5, 4)) for k, (n, ms) in g.items()})

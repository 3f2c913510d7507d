import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from aux_ssm_samplers_amd import _lib, random as R
from aux_ssm_samplers_amd.kalman import get_kernel
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup
T, Cn = 16384, 64
h = _lib.default_handle()
model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
for dtype in (np.float32, np.float64):
    x0 = np.repeat(xtrue[None], Cn, axis=0).astype(dtype)
    a, b = DeviceChains(h, x0), DeviceChains(h, x0)
    keys = R.split(R.PRNGKey(4), 24)
    accA, accB = [], []
    for k in range(23):
        kernel(keys[k], KalmanSampler(x=a, updated=None), 1e-4)
        ea, es, ua = kernel.draw(h, keys[k], b)
        kernel.sweep(h, b, 1e-4, ea, es, ua)
        accA.append(a.accepted.to_host().mean()); accB.append(b.accepted.to_host().mean())
        la, lb = a.logs.to_host(), b.logs.to_host()
    print(dtype.__name__, "keyed acc per sweep", np.round(accA, 2))
    print(dtype.__name__, "explicit acc per sweep", np.round(accB, 2))
    print("last logs equal:", np.array_equal(la, lb), "max diff", np.abs(la - lb).max(), "log alpha keyed[:6]", np.round(la[:6, 0], 1))

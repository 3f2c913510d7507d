"""Layout chosen for resident chains (kalman.DeviceChains): a chain-shared linear-Gaussian model takes the chain-minor layout -- and with it the fused sweep -- from
4 chains on; everything else keeps the time-minor general path below 32 chains.  (The random streams are indexed by the flat position in the layout, so the
two layouts draw different noise from the same key: the fused sweep is compared with the unfused chain-minor sweep, as tests/test_gpu_fused.py does.)"""
import numpy as np
import numpy.testing as npt
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Cn", [4, 8, 16])
def test_few_chains_on_a_shared_lg_model_run_the_fused_sweep_and_agree_with_the_time_minor_path(Cn):
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman import get_kernel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    from aux_ssm_samplers_amd.workloads import lg_model
    T, d = 1024, 4
    m = lg_model(T, d)
    bt = np.broadcast_to
    model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)), bt(m["Hobs"], (T, d, d)),
                          bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), m["y"])
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    h = _lib.default_handle()
    x0 = m["x_true"][None] + 0.3 * np.random.default_rng(Cn).standard_normal((Cn, T, d))
    a = DeviceChains(h, x0, model=model)          # the hint: chain-minor, fused
    b = DeviceChains(h, x0)                       # no hint: time-minor below 32 chains
    c = DeviceChains(h, x0[:3], model=model)      # odd count: the fused sweep pairs chains, so the hint does not apply
    assert a.chain_minor and not b.chain_minor and not c.chain_minor
    k = DeviceChains(h, x0, chain_minor=True, fused=False)   # the same layout, keyed (unfused) sweeps
    keys = R.split(R.PRNGKey(5), 4)
    for key in keys:
        kernel(key, KalmanSampler(x=a, updated=None), 0.5)
        kernel(key, KalmanSampler(x=k, updated=None), 0.5)
        kernel(key, KalmanSampler(x=b, updated=None), 0.5)
    assert a.fused is True and k.fused is False and b.fused is not True
    npt.assert_array_equal(a.accepted.to_host(), k.accepted.to_host())
    npt.assert_allclose(a.to_host(), k.to_host(), rtol=1e-8, atol=1e-8)
    # exact proposals on a linear-Gaussian model: every layout accepts everything, log alpha = 0 to rounding
    for ch in (a, k, b):
        assert ch.accepted.to_host().all() and np.abs(ch.logs.to_host()[:, 0]).max() < 1e-7
    # a host-array state through the kernel takes the same decision as the hinted resident chains
    out = kernel(keys[0], init(x0), 0.5)
    ref = DeviceChains(h, x0, model=model)
    kernel(keys[0], KalmanSampler(x=ref, updated=None), 0.5)
    npt.assert_allclose(out.x, ref.to_host(), rtol=1e-12, atol=1e-12)

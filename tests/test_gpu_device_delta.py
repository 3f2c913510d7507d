"""auxssm_kalman_sweep_dd: the sweep with a DEVICE-RESIDENT step size (boundary gap: the adaptation loop of
examples/*/experiment.py with common.py:4-32 delta_adaptation read C floats back per burn-in sweep).  Same kernels, delta read from device
memory: results must be bit-identical to the host-scalar sweep for every device model and layout."""
import numpy as np
import numpy.testing as npt
import pytest

from aux_ssm_samplers_amd import _lib
from aux_ssm_samplers_amd.kalman import get_kernel, DeviceChains, KalmanSampler
from tests.helpers import lg_model

pytestmark = pytest.mark.gpu


def _models(dtype):
    from aux_ssm_samplers_amd.kalman import LGConcatModel, SVModel
    from tests.helpers import sv_setup, lorenz_kalman_setup
    T, d = 400, 2
    m = lg_model(T, d, dtype=dtype)
    full = lambda a, n: np.ascontiguousarray(np.broadcast_to(a, (n,) + a.shape))
    yield "lg", LGConcatModel(m["m0"], m["P0"], full(m["F"], T - 1), full(m["Q"], T - 1), full(m["b"], T - 1), full(m["Hobs"], T), full(m["Robs"], T),
                              full(m["cobs"], T), m["y"]), 0.4, T, d
    y, _, (m0, P0, F, Q, b) = sv_setup(64, 2)
    for order in (1, 2):
        yield f"sv{order}", SVModel(y, m0, P0, F, Q, b, order=order), 0.05, 64, 2
    yield "lorenz", lorenz_kalman_setup(160, every=8, dt=1e-3)[0], 1e-3, 160, 3


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("C,chain_minor", [(3, False), (64, True)])
def test_device_delta_sweep_is_bit_identical_to_the_host_scalar_sweep(dtype, C, chain_minor):
    h = _lib.default_handle()
    for name, model, delta, T, d in _models(dtype):
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        rng = np.random.default_rng(7)
        x0 = rng.standard_normal((C, T, d)).astype(dtype) * 0.3
        noise = dict(eps_aux=rng.standard_normal((C, T, d)).astype(dtype), eps_samp=rng.standard_normal((C, T, d)).astype(dtype),
                     u_accept=rng.random(C).astype(dtype))
        cm = chain_minor and not getattr(model, "dense_only", False)
        a = DeviceChains(h, x0, chain_minor=cm)
        b = DeviceChains(h, x0, chain_minor=cm)
        kernel(None, KalmanSampler(x=a, updated=None), delta, noise=noise)
        dd = h.to_device(np.full(1, delta, dtype), dtype)
        kernel(None, KalmanSampler(x=b, updated=None), dd, noise=noise)
        npt.assert_array_equal(a.to_host(), b.to_host(), err_msg=name)
        npt.assert_array_equal(a.logs.to_host(), b.logs.to_host(), err_msg=name)
        npt.assert_array_equal(a.accepted.to_host(), b.accepted.to_host(), err_msg=name)


def test_wrong_dtype_device_delta_is_refused():
    h = _lib.default_handle()
    name, model, delta, T, d = next(_models(np.float64))
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    ch = DeviceChains(h, np.zeros((2, T, d)), chain_minor=False)
    with pytest.raises(ValueError):
        kernel(0, KalmanSampler(x=ch, updated=None), h.to_device(np.full(1, 0.3, np.float32), np.float32))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("C,chain_minor,share", [(3, False, 1), (64, True, 1), (64, True, 0), (130, True, 1)])
def test_keyed_sweep_equals_draw_then_sweep(dtype, C, chain_minor, share):
    """auxssm_kalman_sweep_keyed: kernel(key, state, delta) with the KEYS of the three draws.  Where the chain-shared affine scans run the noise
    is generated inside their reduce passes (lane pairs sharing Threefry blocks); it must be bit for bit what auxssm_kalman_draw puts into the
    buffers, and the sweep's results must be those of draw + sweep."""
    from aux_ssm_samplers_amd import random as R
    h = _lib.default_handle()
    h.set_option(_lib.OPT_SHARE_MODEL, share)
    try:
        for name, model, delta, T, d in _models(dtype):
            if name != "lg" and C == 130:
                continue
            T2 = T if name != "lg" else T  # (short horizons: few chunks; the gen path still needs more than one chunk)
            init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
            rng = np.random.default_rng(11)
            x0 = rng.standard_normal((C, T, d)).astype(dtype) * 0.3
            cm = chain_minor and not getattr(model, "dense_only", False)
            a = DeviceChains(h, x0, chain_minor=cm)
            b = DeviceChains(h, x0, chain_minor=cm)
            key = R.PRNGKey(123)
            kernel(key, KalmanSampler(x=a, updated=None), delta)                     # keyed
            k_aux, k_samp, k_acc = R.split(key, 3)
            h.kalman_draw(k_aux, k_samp, k_acc, b.eps_aux, b.eps_samp, b.u_acc)      # the three fills ...
            ea, es, ua = b.eps_aux.to_host(), b.eps_samp.to_host(), b.u_acc.to_host()
            npt.assert_array_equal(a.eps_aux.to_host(), ea, err_msg=name)
            npt.assert_array_equal(a.eps_samp.to_host(), es, err_msg=name)
            npt.assert_array_equal(a.u_acc.to_host(), ua, err_msg=name)
            to_ctd = (lambda e: np.ascontiguousarray(e.transpose(2, 0, 1))) if cm else (lambda e: e)
            kernel(None, KalmanSampler(x=b, updated=None), delta, noise=dict(eps_aux=to_ctd(ea), eps_samp=to_ctd(es), u_accept=ua))  # ... then the sweep
            npt.assert_array_equal(a.to_host(), b.to_host(), err_msg=name)
            npt.assert_array_equal(a.logs.to_host(), b.logs.to_host(), err_msg=name)
    finally:
        h.set_option(_lib.OPT_SHARE_MODEL, 1)

"""GPU parity tests of the parallel-in-time cSMC sweep (auxssm_csmc_pit_sweep, csrc/pit.hip; reference _primitives/csmc/pit +
csmc/independent.py:78-118): the tree kernels, which keep only boundary indices, against the plain-C oracle, which gathers whole
blocks at every stitch as the reference's operator does -- on identical explicit noise, trajectories and ancestors BIT-EXACT."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import csmc as O
from tests.test_gpu_csmc import _models, _odesc, _pot

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,N,T", [(1, 2, 2), (1, 16, 3), (1, 32, 64), (1, 100, 37), (2, 64, 33), (3, 40, 20), (4, 33, 17), (1, 256, 9), (2, 1024, 5)])
@pytest.mark.parametrize("potential", [O.POT_FLAT, O.POT_GAUSS_OBS, O.POT_SV])
def test_pit_sweep_bit_exact_vs_oracle(dtype, d, N, T, potential):
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(1000 * d + N + T)
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(potential, y)
    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
    x0 = rng.standard_normal((T, d)).astype(dtype)
    delta = 0.5 + rng.random(T)
    noise = dict(eps_aux=rng.standard_normal((T, d)), eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T, N)))
    noise = {k: np.asarray(v, dtype) for k, v in noise.items()}
    x, anc = _device.pit_sweep(fk, x0, N, noise={k: v[None] for k, v in noise.items()}, delta=delta)
    ref = O.pit_sweep(_odesc(O.AUX_INDEPENDENT, potential, M0, Mt, 0.7), x0, N, y=y if potential else None,
                      sqrt_half_delta=np.sqrt(0.5 * delta), dtype=dtype, **noise)
    npt.assert_array_equal(anc, ref["ancestors"])
    npt.assert_array_equal(x, ref["x"])
    assert anc.any() or N == 2


def test_pit_lorenz_transition_bit_exact():
    from tests.test_gpu_csmc import lorenz_setup
    from aux_ssm_samplers_amd.csmc import _device
    T, N = 50, 64
    M0, Mt, G0, Gt, xt, y, sig_y = lorenz_setup(T, every=5, dt=0.01)
    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
    rng = np.random.default_rng(1)
    x0 = xt.astype(np.float32)
    noise = dict(eps_aux=rng.standard_normal((T, 3)), eps_prop=rng.standard_normal((T, N, 3)), u_res=rng.random((T, N)))
    noise = {k: np.asarray(v, np.float32) for k, v in noise.items()}
    x, anc = _device.pit_sweep(fk, x0, N, noise={k: v[None] for k, v in noise.items()}, delta=0.05)
    od = dict(proposal=O.AUX_INDEPENDENT, potential=fk.potential, m0=fk.m0, chol_P0=fk.chol_P0, F=fk.F, b=fk.b, chol_Q=fk.chol_Q,
              sig_y=fk.sig_y, transition=fk.transition)
    ref = O.pit_sweep(od, x0, N, y=fk.y, sqrt_half_delta=np.full(T, np.sqrt(0.025)), dtype=np.float32, **noise)
    npt.assert_array_equal(anc, ref["ancestors"])
    npt.assert_array_equal(x, ref["x"])
    assert anc.any()


@pytest.mark.parametrize("T", [2, 7, 64, 100])
def test_multichain_equals_single_chain_and_threefry_equals_explicit(T):
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(7)
    d, N, C = 2, 48, 4
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(O.POT_SV, y)
    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
    x0 = rng.standard_normal((C, T, d)).astype(np.float32)
    key = R.PRNGKey(5)
    h = _lib.default_handle()
    xa, anca = _device.pit_sweep(fk, x0, N, key=key, delta=0.5)
    noise = dict(eps_aux=h.rng_normal(key, 1, (C, T, d), np.float32).to_host(), eps_prop=h.rng_normal(key, 2, (C, T, N, d), np.float32).to_host(),
                 u_res=h.rng_uniform(key, 3, (C, T, N), np.float32).to_host())
    xb, ancb = _device.pit_sweep(fk, x0, N, noise=noise, delta=0.5)
    npt.assert_array_equal(xa, xb)
    npt.assert_array_equal(anca, ancb)
    for c in range(C):
        xc, ancc = _device.pit_sweep(fk, x0[c], N, noise={k: v[c:c + 1] for k, v in noise.items()}, delta=0.5)
        npt.assert_array_equal(xc, xb[c])
        npt.assert_array_equal(ancc, ancb[c])


def test_parallel_kernel_targets_the_smoothing_distribution():
    """get_independent_kernel(..., parallel=True) as a Gibbs kernel on the linear-Gaussian model of the reference's
    test_pit_csmc.py (T = 25, N = 32): many resident chains, device noise, against the exact smoother."""
    from tests.test_oracle_pit import _lg, smoother
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.csmc import (get_independent_kernel, CsmcChains, CSMCState, GaussianInit, LinearGaussianDynamics,
                                           GaussianObsPotential)
    T, N, rho, sig_y, C, M, B = 25, 32, 0.9, 0.5, 512, 120, 40
    model, xtrue, y = _lg(T, rho, sig_y, seed=5)
    mean, var = smoother(T, rho, sig_y, y[:, 0])
    M0 = GaussianInit(m0=[0.0], P0=[[1.0]])
    Mt = LinearGaussianDynamics(F=[[rho]], b=[0.0], Q=[[1 - rho ** 2]])
    init, kernel = get_independent_kernel(M0, GaussianObsPotential(sig=sig_y, y=y[0]), Mt, GaussianObsPotential(sig=sig_y, params=y[1:]), N,
                                          parallel=True)
    h = _lib.default_handle()
    chains = CsmcChains(h, np.zeros((C, T, 1), np.float64), delta=0.8)
    state = CSMCState(x=chains, updated=None)
    keys = R.split(R.PRNGKey(11), M)
    acc, acc2, upd, n = np.zeros(T), np.zeros(T), np.zeros(T), 0
    for i in range(M):
        state = kernel(keys[i], state, None)
        if i >= B:
            xs = chains.to_host()[:, :, 0]
            acc += xs.sum(0)
            acc2 += (xs ** 2).sum(0)
            upd += (chains.ancestors.to_host() != 0).sum(0)
            n += C
    m_hat = acc / n
    v_hat = acc2 / n - m_hat ** 2
    assert (upd / n).min() > 0.3
    npt.assert_allclose(m_hat, mean, atol=0.02)
    npt.assert_allclose(v_hat, var, rtol=0.05)
    # host-array states go through the same kernel
    out = kernel(R.PRNGKey(1), init(np.zeros((T, 1))), 0.8)
    assert out.x.shape == (T, 1) and out.updated.shape == (T,) and out.updated.dtype == bool


@pytest.mark.parametrize("gradient", [False, True])
def test_reference_call_shape_through_pit_get_kernel(gradient):
    """what independent.py:78-118 does by hand -- draw u, build mt / g0 / gt (/ qt), call `_primitives.csmc.pit.get_kernel(mt, g0, gt, N, qt)` -- against the
    device sweep on the same auxiliary variables and the same proposal / resampling draws.  x, the step scale and the auxiliary noise are dyadic so that the
    eps_aux the wrapper recovers from u reproduces u exactly; arbitrary Python objects still raise."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd._primitives.csmc import pit
    from aux_samplers._primitives.csmc import pit as pit_ref_path  # noqa: F401  (the reference's import path resolves to the same module)
    from aux_ssm_samplers_amd.csmc import _device
    from aux_ssm_samplers_amd.csmc.independent import AuxiliaryMtDistribution, AuxiliaryG0, AuxiliaryGt
    rng = np.random.default_rng(3)
    d, N, T = 2, 32, 40
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(O.POT_SV, y)
    x = np.round(rng.standard_normal((T, d)) * 64) / 64
    scale = 0.5
    eps_aux = np.round(rng.standard_normal((T, d)) * 64) / 64
    u = x + scale * eps_aux
    mt = AuxiliaryMtDistribution(params=(u, scale * np.ones(T), np.zeros_like(u) if gradient else None))
    qt = AuxiliaryMtDistribution(params=(u, scale * np.ones(T), None)) if gradient else None
    init, kernel = pit.get_kernel(mt, AuxiliaryG0(M0=M0, G0=G0), AuxiliaryGt(Mt=Mt, Gt=Gt), N, qt)
    key = R.PRNGKey(11)
    st = init(x)
    assert st.updated.shape == (T,) and not st.updated.any()
    out = kernel(key, st)
    h = _lib.default_handle()
    k_prop, k_res = R.split(key, 2)
    noise = dict(eps_aux=eps_aux[None], eps_prop=h.rng_normal(k_prop, 2, (1, T, N, d), np.float64).to_host(), u_res=h.rng_uniform(k_res, 3, (1, T, N), np.float64).to_host())
    fk = _device.describe_independent(M0, G0, Mt, Gt, None, _lib.GRAD_EXACT if gradient else _lib.GRAD_NONE)
    xr, ancr = _device.pit_sweep(fk, x, N, noise=noise, delta=2 * scale ** 2)
    npt.assert_array_equal(out.x, xr)
    npt.assert_array_equal(out.ancestors, ancr)
    npt.assert_array_equal(out.updated, ancr != 0)
    assert ancr.any()
    with pytest.raises(NotImplementedError):
        pit.get_kernel(object(), AuxiliaryG0(M0=M0, G0=G0), AuxiliaryGt(Mt=Mt, Gt=Gt), N)
    with pytest.raises(NotImplementedError):
        pit.get_kernel(mt, AuxiliaryG0(M0=M0, G0=G0), AuxiliaryGt(Mt=Mt, Gt=Gt), N, None if gradient else mt)


def test_argument_errors():
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(0)
    M0, Mt = _models(1, rng)
    G0, Gt = _pot(O.POT_FLAT, None)
    fkb = _device.describe_bootstrap(M0, G0, Mt, Gt, Mt)
    with pytest.raises(ValueError):  # bootstrap proposals depend on the parent: not independent across time
        _device.pit_sweep(fkb, np.zeros((4, 1)), 8, key=0, delta=0.5)
    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
    with pytest.raises(ValueError):
        _device.pit_sweep(fk, np.zeros((1, 1)), 8, key=0, delta=0.5)   # T >= 2
    with pytest.raises(ValueError):
        _device.pit_sweep(fk, np.zeros((4, 1)), 2048, key=0, delta=0.5)


def test_full_size_properties_T65536():
    """BASELINE config C3's horizon (T = 65536, SV model, fp32), N = 64: size-independent properties of the tree -- draws forced onto the
    first pair return the reference trajectory untouched; a keyed sweep is deterministic, its ancestors are valid leaf indices, every
    output value is one of that step's proposals around u_t, and a fresh key moves most of the trajectory."""
    import bench
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics, SVPotential
    T, N = 65536, 64
    phi, q, xsv, ysv = bench.sv_data(T, 0)
    M0 = GaussianInit(m0=[0.0], P0=[[q]])
    Mt = LinearGaussianDynamics(F=[[phi]], b=[0.0], Q=[[q]])
    fk = _device.describe_independent(M0, SVPotential(y=ysv[0]), Mt, SVPotential(params=ysv[1:]), Mt)
    x0 = xsv.astype(np.float32)
    h = _lib.default_handle()
    key = R.PRNGKey(123)
    noise = dict(eps_aux=h.rng_normal(key, 1, (1, T, 1), np.float32).to_host(), eps_prop=h.rng_normal(key, 2, (1, T, N, 1), np.float32).to_host(),
                 u_res=np.full((1, T, N), np.float32(1.0) - np.float32(2.0 ** -24)))
    x, anc = _device.pit_sweep(fk, x0, N, noise=noise, delta=0.5)
    npt.assert_array_equal(anc, 0)
    npt.assert_array_equal(x, x0)
    xa, anca = _device.pit_sweep(fk, x0, N, key=key, delta=0.5)
    xb, ancb = _device.pit_sweep(fk, x0, N, key=key, delta=0.5)
    npt.assert_array_equal(xa, xb)
    npt.assert_array_equal(anca, ancb)
    assert anca.min() >= 0 and anca.max() < N and np.isfinite(xa).all()
    shd = np.float32(np.sqrt(0.25))
    leaves = np.float32(x0 + shd * noise["eps_aux"][0])[:, None, :] + shd * noise["eps_prop"][0]
    leaves[:, 0] = x0
    npt.assert_allclose(xa, leaves[np.arange(T), anca], rtol=1e-6, atol=1e-6)
    assert (anca != 0).mean() > 0.2  # (a high-variance statistic: a top-level stitch that keeps slot 0 on both sides leaves a whole block on the reference path)

"""GPU parity tests of the MCMC loop around the sweeps (include/auxssm.h "the MCMC loop around the sweeps"):
device kernels vs oracle/loop_np.py.  Running means and acceptance averages: BIT-EXACT (the kernels are built without contraction);
step-size rule: rtol 1e-14 fp64 / 1e-6 fp32 (device exp); theta step: rtol 1e-10 fp64 / 2e-4 fp32 (reduction order)."""
import functools

import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K
from oracle import loop_np as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    from aux_ssm_samplers_amd import _lib
    return _lib.default_handle()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [1, 255, 4099])
def test_stats_update_bit_exact(handle, dtype, n):
    rng = np.random.default_rng(n)
    xs = rng.standard_normal((6, n)).astype(dtype) * 3
    stats_h = L.stats_fn(xs[0], xs[0])
    stats_d = tuple(handle.zeros((n,), dtype) for _ in range(3))
    for i in range(5):
        stats_h = tuple(L.fold(i, u, v) for u, v in zip(stats_h, L.stats_fn(xs[i], xs[i + 1])))
        handle.stats_update(i, handle.to_device(xs[i]), handle.to_device(xs[i + 1]), stats_d)
    for a, b in zip(stats_d, stats_h):
        assert b.dtype == dtype
        npt.assert_array_equal(a.to_host(), b)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("C,m", [(1, 1), (5, 1), (3, 130)])
def test_accept_update_and_delta_adapt(handle, dtype, C, m):
    rng = np.random.default_rng(C * 1000 + m)
    avg_h = np.ones((C, m), dtype)
    win_h = np.ones((C, m), dtype)
    avg, win = handle.to_device(avg_h), handle.to_device(win_h)
    delta_h = (0.1 + rng.random(m)).astype(dtype)
    delta, shd = handle.to_device(delta_h), handle.zeros((m,), dtype)
    beta, target, lr, n_iter = 0.07, 0.4, 0.8, 6
    for i in range(n_iter):
        anc = (rng.integers(0, 3, (C, m)) * rng.integers(0, 2, (C, m))).astype(np.int32)
        avg_h, win_h = L.accept_update(i, beta, anc != 0, avg_h, win_h)
        handle.accept_update(i, beta, handle.to_device(anc), avg, win)
        npt.assert_array_equal(avg.to_host(), avg_h)
        npt.assert_array_equal(win.to_host(), win_h)
        lr_i = (n_iter - i) * lr / n_iter
        delta_h = L.pooled_delta_adaptation(delta_h.astype(np.float64), target, win_h.astype(np.float64), lr_i, 0.2, 1.0)
        handle.delta_adapt(win, target, lr_i, delta, shd, 0.2, 1.0)
        tol = 1e-13 if dtype == np.float64 else 2e-6
        npt.assert_allclose(delta.to_host(), delta_h, rtol=tol)
        npt.assert_allclose(shd.to_host(), np.sqrt(0.5 * delta_h), rtol=tol)
        delta_h = delta.to_host()  # no drift between the two recursions
    assert delta_h.min() >= 0.2 and delta_h.max() <= 1.0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("T", [2, 300, 1025])
def test_lorenz_theta_update_vs_oracle(handle, dtype, T):
    rng = np.random.default_rng(T)
    C, dt, sx, sth = 3, 0.01, 3.0, 5.0
    theta = np.array([10.0, 28.0, 8.0 / 3.0])
    x = np.zeros((C, T, 3))
    x[:, 0] = np.array([1.5, -1.5, 25.0]) + rng.standard_normal((C, 3))
    for t in range(1, T):
        p = x[:, t - 1]
        x[:, t] = p + dt * (L.phi_0(p) + theta * L.phi(p)) + sx * np.sqrt(dt) * rng.standard_normal((C, 3))
    x = x.astype(dtype)
    eps = rng.standard_normal((C, 3)).astype(dtype)
    par = handle.to_device(np.concatenate([np.zeros((C, 3)), np.full((C, 1), dt)], 1), dtype)
    mc = handle.zeros((C, 6), dtype)
    handle.lorenz_theta_update(handle.to_device(x), sth, sx, handle.to_device(eps), par, mc)
    tol = 1e-10 if dtype == np.float64 else 2e-4
    dth = float(np.asarray(dt, dtype))
    for c in range(C):
        mean, chol = L.theta_posterior_mean_and_chol(x[c], sth, dth, sx)
        npt.assert_allclose(mc.to_host()[c, :3], mean, rtol=tol, atol=tol)
        npt.assert_allclose(mc.to_host()[c, 3:], chol, rtol=tol)
        npt.assert_allclose(par.to_host()[c, :3], mean + chol * eps[c], rtol=tol, atol=tol)
        assert par.to_host()[c, 3] == np.asarray(dt, dtype)
    if T == 1025 and dtype == np.float64:  # the regression finds the parameters that generated the path
        assert np.all(np.abs(mc.to_host()[:, :3] - theta) < 6 * mc.to_host()[:, 3:] + 0.5)


def _device_noise(handle, key, shape, C, dtype, chains):
    """the explicit arrays the keyed Kalman sweep draws (kalman/generic.py `draw`), as (C, T, dx) / (C,)"""
    from aux_ssm_samplers_amd import random as R
    k_aux, k_samp, k_acc = R.split(key, 3)
    ea = chains.stats_to_host(handle.rng_normal(k_aux, 0, chains.x.shape, dtype))
    es = chains.stats_to_host(handle.rng_normal(k_samp, 0, chains.x.shape, dtype))
    ua = handle.rng_uniform(k_acc, 0, (C,), dtype).to_host()
    return ea, es, ua


@pytest.mark.parametrize("C,chain_minor", [(1, False), (3, False), (32, True)])
@pytest.mark.parametrize("adapt", [False, True])
def test_kalman_loop_vs_oracle_loop(handle, C, chain_minor, adapt):
    """SV model (second-order factory): sweeps get rejected, the step size adapts; C chains vs C runs of the oracle's loop on the noise
    the device draws.  Moments fp64 rtol 1e-8 (the sweep's tolerance), acceptance averages exact."""
    from tests.test_gpu_nonlinear_kalman import sv_setup, oracle_target
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.common import delta_adaptation
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from aux_ssm_samplers_amd.loop import loop
    T, d, n_iter = 64, 1, 6
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d, seed=4, rho=0.0)
    if chain_minor:  # the chain-minor layout is the LG_CONCAT sweep's: use the linear-Gaussian model there
        from tests.helpers import lg_model
        from aux_ssm_samplers_amd.kalman import LGConcatModel
        m = lg_model(T, 2)
        bt = np.broadcast_to
        d = 2
        model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)),
                              bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), m["y"])
        xtrue = m["x_true"]
        lgo = (m["m0"], m["P0"], model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
        target = lambda z: K.log_likelihood(m["y"], z, lgo) + K.prior_logpdf(z, lgo)
    else:
        model = SVModel(y, m0, P0, F, Q, b, order=2)
        target = oracle_target(model)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    rng = np.random.default_rng(C)
    x0 = xtrue[None] + 0.5 * rng.standard_normal((C, T, d))
    chains = DeviceChains(handle, x0, chain_minor=chain_minor)
    key = R.PRNGKey(77)
    delta_fn = functools.partial(delta_adaptation, min_delta=1e-3, max_delta=10.0) if adapt else None
    init_delta, beta, target_alpha, lr = 1.5, 0.2, 0.5, 0.6
    n, stats, state, delta, window, avg = loop(key, init_delta, KalmanSampler(x=chains, updated=True), kernel, delta_fn, n_iter,
                                               target_alpha=target_alpha, lr=lr, beta=beta)
    assert n == n_iter
    # oracle: the same loop, chain by chain, delta pooled over chains as the device does
    keys = R.split(key, n_iter)
    xs = x0.copy()
    st = [L.stats_fn(xs[c], xs[c]) for c in range(C)]
    avg_h, win_h = np.ones((C, 1)), np.ones((C, 1))
    dl = init_delta
    n_acc = 0
    for i in range(n_iter):
        ea, es, ua = _device_noise(handle, keys[i], None, C, np.float64, chains)
        upd = np.zeros((C, 1), bool)
        for c in range(C):
            ref = K.kalman_sweep(xs[c], dl, model.dynamics_factory, model.observations_factory, target, True, eps_aux=ea[c], eps_samp=es[c],
                                 u_accept=ua[c])
            st[c] = tuple(L.fold(i, u, v) for u, v in zip(st[c], L.stats_fn(xs[c], ref["x"])))
            xs[c] = ref["x"]
            upd[c, 0] = ref["accepted"]
        n_acc += upd.sum()
        avg_h, win_h = L.accept_update(i, beta, upd, avg_h, win_h)
        if adapt:
            dl = float(L.pooled_delta_adaptation(dl, target_alpha, win_h, (n_iter - i) * lr / n_iter, 1e-3, 10.0)[0])
    if not chain_minor:
        assert 0 < n_acc < C * n_iter or C == 1  # the test exercises both branches of the accept step
    npt.assert_allclose(chains.to_host(), xs, rtol=1e-8, atol=1e-9)
    for k in range(3):
        got = chains.stats_to_host(stats[k])
        npt.assert_allclose(got, np.stack([st[c][k] for c in range(C)]), rtol=1e-8, atol=1e-9)
    npt.assert_array_equal(avg.to_host().reshape(C, 1), avg_h)
    npt.assert_array_equal(window.to_host().reshape(C, 1), win_h)
    npt.assert_allclose(delta, dl, rtol=1e-12)
    # detached afterwards: a further sweep leaves the moments alone
    before = stats[1].to_host()
    kernel(R.PRNGKey(1), state, 0.5)
    npt.assert_array_equal(stats[1].to_host(), before)


@pytest.mark.parametrize("chain_minor", [False, True])
def test_lorenz_gibbs_loop_vs_oracle(handle, chain_minor):
    """The (x, theta) Gibbs sampler of examples/lorenz/experiment.py:106-115 with one theta per chain, vs the oracle chain by chain
    (dense and chain-minor resident layouts: the theta step reads the state through the same layout as the sweep)."""
    from tests.test_gpu_nonlinear_kalman import lorenz_kalman_setup
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.kalman import get_kernel, LorenzModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from aux_ssm_samplers_amd.loop import loop, LorenzThetaStep
    T, C, n_iter, sth = 120, 2, 4, 10.0
    base, xtrue = lorenz_kalman_setup(T)
    theta0 = np.array([[10.0, 28.0, 8.0 / 3.0], [9.0, 27.0, 3.0]])

    def mk(theta):
        return LorenzModel(base.yobs, base.Hobs, base.Robs, base.cobs, base.m0, base.P0, theta, base.sigma_x, base.dt)

    model = mk(theta0)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    rng = np.random.default_rng(3)
    x0 = xtrue[None] + 0.05 * rng.standard_normal((C, T, 3))
    chains = DeviceChains(handle, x0, chain_minor=chain_minor)
    step = LorenzThetaStep(model, sth)
    key = R.PRNGKey(9)
    thetas = []
    n, stats, state, delta, window, avg = loop(key, 0.02, KalmanSampler(x=chains, updated=True), kernel, None, n_iter, beta=0.1,
                                               theta_step=step, callback=lambda i, s: thetas.append(step.theta(chains)))
    keys = R.split(key, n_iter)
    xs, th = x0.copy(), theta0.copy()
    for i in range(n_iter):
        k_sweep, k_theta = R.split(keys[i], 2)
        ea, es, ua = _device_noise(handle, k_sweep, None, C, np.float64, chains)
        et = handle.rng_normal(k_theta, 0, (C, 3), np.float64).to_host()
        for c in range(C):
            mc = mk(th[c])
            ref = K.kalman_sweep(xs[c], 0.02, mc.dynamics_factory, mc.observations_factory, mc.log_likelihood_fn, True, eps_aux=ea[c],
                                 eps_samp=es[c], u_accept=ua[c])
            xs[c] = ref["x"]
            mean, chol = L.theta_posterior_mean_and_chol(xs[c], sth, base.dt, base.sigma_x)
            th[c] = mean + chol * et[c]
        npt.assert_allclose(thetas[i], th, rtol=1e-6)
    npt.assert_allclose(chains.to_host(), xs, rtol=1e-7, atol=1e-8)
    assert np.abs(th[0] - th[1]).max() > 1e-3  # the chains really carry their own theta


@pytest.mark.parametrize("backward", [False, True, "pit"])
def test_csmc_loop_statistics_and_per_step_adaptation(handle, backward):
    """cSMC chains resident on the device: the loop's moments / per-time-step acceptance equal a host recomputation from the states
    the sweeps leave (each sweep bit-exact vs the C oracle elsewhere), and delta_t follows the rule on the chain-pooled window."""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.common import delta_adaptation
    from aux_ssm_samplers_amd.csmc import (get_independent_kernel, CsmcChains, GaussianInit, LinearGaussianDynamics, SVPotential, CSMCState)
    from aux_ssm_samplers_amd.loop import loop
    T, C, N, n_iter, dtype = 50, 3, 32, 5, np.float32
    rng = np.random.default_rng(0)
    y = rng.standard_normal((T, 1))
    M0 = GaussianInit(np.zeros(1), np.eye(1) * 5.0)
    Mt = LinearGaussianDynamics(0.9 * np.eye(1), np.zeros(1), np.eye(1))
    G0, Gt = SVPotential(y[0]), SVPotential(None, params=y[1:])
    if backward == "pit":  # the parallel-in-time kernel drives the same loop
        init, kernel = get_independent_kernel(M0, G0, Mt, Gt, N, parallel=True)
    else:
        init, kernel = get_independent_kernel(M0, G0, Mt, Gt, N, backward=backward, Pt=Mt)
    x0 = rng.standard_normal((C, T, 1)).astype(dtype)
    chains = CsmcChains(handle, x0)
    snaps = [x0.copy()]
    ancs, deltas = [], []

    def cb(i, state):
        snaps.append(chains.to_host())
        ancs.append(chains.ancestors.to_host())
        deltas.append(chains.delta.to_host())

    beta, target, lr = 0.3, 0.5, 0.7
    n, stats, state, delta, window, avg = loop(R.PRNGKey(3), 0.8, CSMCState(x=chains, updated=np.zeros(T, bool)), kernel, delta_adaptation,
                                               n_iter, target_alpha=target, lr=lr, beta=beta, callback=cb)
    st = L.stats_fn(snaps[0], snaps[0])
    avg_h = np.zeros((C, T), dtype)
    win_h = np.zeros((C, T), dtype)
    dl = np.full(T, 0.8)
    for i in range(n_iter):
        st = tuple(L.fold(i, u, v) for u, v in zip(st, L.stats_fn(snaps[i], snaps[i + 1])))
        avg_h, win_h = L.accept_update(i, beta, ancs[i] != 0, avg_h, win_h)
        dl = L.pooled_delta_adaptation(dl, target, win_h.astype(np.float64), (n_iter - i) * lr / n_iter)
        npt.assert_allclose(deltas[i], dl, rtol=3e-6)
        dl = deltas[i].astype(np.float64)
    for k in range(3):
        npt.assert_array_equal(stats[k].to_host(), st[k])
    npt.assert_array_equal(avg.to_host(), avg_h)
    npt.assert_array_equal(window.to_host(), win_h)
    npt.assert_allclose(chains.sqrt_half_delta.to_host(), np.sqrt(0.5 * dl), rtol=1e-6)
    assert np.any(np.abs(snaps[-1] - snaps[0]) > 0) and np.ptp(dl) > 0


def test_full_size_running_moments_T65536(handle):
    """BASELINE config C2's horizon (T = 65536, d = 4, fp64), 8 chains: the moments the sweeps fold in equal the plain means of the
    states the sweeps leave (the linear-Gaussian sweep always accepts: log alpha = 0), and the acceptance averages are 1."""
    import bench
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.kalman import get_kernel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from aux_ssm_samplers_amd.loop import loop
    T, d, C, n_iter = 65536, 4, 8, 4
    m, model = bench.build_model(T, d, np.float64)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    x0 = m["x_true"][None] + 0.3 * np.random.default_rng(0).standard_normal((C, T, d))
    chains = DeviceChains(handle, x0, chain_minor=True)
    snaps = [x0]
    n, stats, state, delta, window, avg = loop(R.PRNGKey(4), 0.5, KalmanSampler(x=chains, updated=True), kernel, None, n_iter,
                                               callback=lambda i, s: snaps.append(chains.to_host()))
    snaps = np.stack(snaps)
    npt.assert_allclose(chains.stats_to_host(stats[1]), snaps[1:].mean(0), rtol=1e-12, atol=1e-12)
    npt.assert_allclose(chains.stats_to_host(stats[2]), (snaps[1:] ** 2).mean(0), rtol=1e-12, atol=1e-12)
    npt.assert_allclose(chains.stats_to_host(stats[0]), ((snaps[1:] - snaps[:-1]) ** 2).mean(0), rtol=1e-12, atol=1e-12)
    npt.assert_array_equal(avg.to_host(), 1.0)
    npt.assert_array_equal(window.to_host(), 1.0)
    assert np.abs(snaps[-1] - snaps[0]).max() > 0.1

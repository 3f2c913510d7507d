"""GPU parity tests of the auxiliary-Kalman sweep on the nonlinear example models (stochastic volatility, Lorenz-63) (device factories AUXSSM_KMODEL_SV_FIRST / SV_SECOND,
reference examples/stochastic_volatility/auxiliary_kalman.py:22-48): device sweep vs the oracle's restatement of
kalman/generic.py:53-106 driven by the same closed-form NumPy factories, on identical explicit noise.  fp64 tolerances:
x_prop rtol 1e-9 / atol 1e-10, log-densities rtol 1e-9."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K

from tests.helpers import sv_setup, lorenz_kalman_setup, sv_posterior_by_quadrature  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def oracle_target(model):
    def f(z):
        lg = (model.m0, model.P0, model.Fs, model.Qs, model.bs, None, None, None)
        return K.prior_logpdf(z, lg) + model.log_potential(z)
    return f


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("d,T", [(1, 400), (2, 257), (4, 150), (6, 60), (30, 24)])
@pytest.mark.parametrize("parallel", [True, False])
def test_sv_device_sweep_vs_oracle(order, d, T, parallel):
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d)
    model = SVModel(y, m0, P0, F, Q, b, order=order)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, parallel)
    rng = np.random.Generator(np.random.PCG64(1000 + d))
    x = xtrue + 0.2 * rng.standard_normal((T, d))
    delta = 0.3
    for rep in range(2):
        noise = dict(eps_aux=rng.standard_normal((T, d)), eps_samp=rng.standard_normal((T, d)), u_accept=rng.random())
        ref = K.kalman_sweep(x, delta, model.dynamics_factory, model.observations_factory, oracle_target(model), parallel, **noise)
        out = kernel(None, init(x), delta, noise=noise)
        npt.assert_allclose(out.logs[0, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-9)
        npt.assert_allclose(out.log_alpha, ref["log_alpha"], rtol=1e-6, atol=1e-7)
        assert out.updated == ref["accepted"]
        npt.assert_allclose(out.x, ref["x"], rtol=1e-9, atol=1e-10)
        x = ref["x"]


@pytest.mark.parametrize("order", [1, 2])
def test_sv_host_factory_path_equals_device_sweep(order):
    """Hiding the model behind lambdas forces the generic host-factory path (NumPy factories + GPU primitives)."""
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    T, d = 120, 2
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d, seed=3)
    model = SVModel(y, m0, P0, F, Q, b, order=order)
    init, kdev = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    init2, khost = get_kernel(lambda z: model.dynamics_factory(z), lambda z, u, dl: model.observations_factory(z, u, dl),
                              lambda z: model.log_likelihood_fn(z), True)
    rng = np.random.default_rng(5)
    x = xtrue + 0.2 * rng.standard_normal((T, d))
    noise = dict(eps_aux=rng.standard_normal((T, d)), eps_samp=rng.standard_normal((T, d)), u_accept=0.3)
    a = kdev(None, init(x), 0.3, noise=noise)
    bb = khost(None, init2(x), 0.3, noise=noise)
    npt.assert_allclose(a.x, bb.x, rtol=1e-9, atol=1e-10)
    npt.assert_allclose(a.log_alpha, bb.log_alpha, rtol=1e-6, atol=1e-7)
    assert a.updated == bb.updated


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("d,T,C", [(1, 300, 40), (2, 129, 33), (4, 70, 64)])
@pytest.mark.parametrize("share", [1, 0])
def test_sv_chain_minor_sweep_vs_oracle(order, d, T, C, share):
    """>= 32 chains run the SV sweep chain-minor (state (T, dx, C), lanes over chains; first order additionally with the chain-shared
    tables): every chain vs the oracle's sweep on its own noise, and vs the dense-layout sweep of the same chains."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d, seed=7)
    model = SVModel(y, m0, P0, F, Q, b, order=order)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    rng = np.random.Generator(np.random.PCG64(50 + d))
    x0 = xtrue[None] + 0.2 * rng.standard_normal((C, T, d))
    noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
    h = _lib.default_handle()
    h.set_option(_lib.OPT_SHARE_MODEL, share)
    try:
        outs = {}
        for cmin in (True, False):
            chains = DeviceChains(h, x0, chain_minor=cmin)
            assert chains.chain_minor == cmin
            kernel(None, KalmanSampler(x=chains, updated=None), 0.3, noise=noise)
            outs[cmin] = (chains.to_host(), chains.accepted.to_host(), chains.logs.to_host())
    finally:
        h.set_option(_lib.OPT_SHARE_MODEL, 1)
    npt.assert_allclose(outs[True][0], outs[False][0], rtol=1e-9, atol=1e-10)
    npt.assert_array_equal(outs[True][1], outs[False][1])
    npt.assert_allclose(outs[True][2][:, 1:], outs[False][2][:, 1:], rtol=1e-9)
    for c in range(0, C, 7):
        ref = K.kalman_sweep(x0[c], 0.3, model.dynamics_factory, model.observations_factory, oracle_target(model), True,
                             eps_aux=noise["eps_aux"][c], eps_samp=noise["eps_samp"][c], u_accept=noise["u_accept"][c])
        npt.assert_allclose(outs[True][2][c, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-9)
        assert bool(outs[True][1][c]) == ref["accepted"]
        npt.assert_allclose(outs[True][0][c], ref["x"], rtol=1e-9, atol=1e-10)
    assert 0 < outs[True][1].sum() < C or d == 4  # both branches of the accept step


def test_sv_chain_moves_and_targets_posterior():
    """Many sweeps of many chains, device Threefry noise: acceptance is healthy and the chain stays in the high-probability region
    (a crude but reference-free sanity check of the MH ratio: a wrong ratio drives log pi(x) away or freezes the chain)."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    T, d, C = 200, 1, 16
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d, seed=11, rho=0.0)
    model = SVModel(y, m0, P0, F, Q, b, order=2)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    h = _lib.default_handle()
    chains = DeviceChains(h, np.repeat(xtrue[None], C, axis=0), chain_minor=False)
    state = KalmanSampler(x=chains, updated=None)
    keys = R.split(R.PRNGKey(5), 60)
    acc = []
    for k in keys:
        kernel(k, state, 0.5)
        acc.append(chains.accepted.to_host().mean())
    xs = chains.to_host()
    assert 0.05 < np.mean(acc) <= 1.0
    lp = np.array([model.log_likelihood_fn(xs[c]) for c in range(C)])
    assert np.all(np.isfinite(lp)) and np.all(lp > model.log_likelihood_fn(xtrue) - 4 * T)
    assert np.abs(xs - xtrue[None]).max() > 1e-3


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("chain_minor", [True, False])
def test_sv_sampler_targets_the_exact_posterior_by_quadrature(order, chain_minor):
    """K7 / K9 without any reference restatement in the loop: the auxiliary Kalman sampler of the SV model (first / second order
    linearisation, examples/stochastic_volatility/auxiliary_kalman.py:22-48; MH ratio of kalman/generic.py:79-106) must leave the TRUE posterior invariant.  T = 3,
    d = 1: posterior means and variances by numerical quadrature on a 151^3 grid; 1024 device chains, 40 (second order) / 300 (first order) burn-in + 300 sweeps with Threefry noise.  A wrong log
    alpha (a missing term, a sign, the wrong reverse density) biases these moments by far more than the Monte-Carlo error (tolerance: 5 standard errors at an
    integrated autocorrelation time of 10, ~0.02).  Step sizes: 1.5 for the second-order sampler, 0.3 for the first-order one -- the first-order (Langevin-like)
    proposals mix very slowly on this target at larger steps (x_2 has an almost uninformative observation and a heavy left tail where the gradient explodes: at
    delta = 1.5 / 6 the chain averages are still 0.1 / 0.5 off after 2000 sweeps of 1024 chains, and an independent dense NumPy simulation of the same algorithm
    shows the same: tools/diag_sv_quadrature.py), so a larger step would test the mixing of the reference's algorithm, not the arithmetic of its kernel."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    T, d, C, M = 3, 1, 1024, 300
    burn = 40 if order == 2 else 300  # (the small first-order steps need that long to forget the initial spread around the simulated path)
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d, seed=4, rho=0.0)
    exact = sv_posterior_by_quadrature(y[:, 0], m0[0], P0[0, 0], F[0, 0], Q[0, 0], b[0])
    model = SVModel(y, m0, P0, F, Q, b, order=order)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    h = _lib.default_handle()
    rng = np.random.default_rng(1)
    chains = DeviceChains(h, xtrue[None] + rng.standard_normal((C, T, d)), chain_minor=chain_minor)
    state = KalmanSampler(x=chains, updated=None)
    keys = R.split(R.PRNGKey(77 + order), burn + M)
    s1, s2, acc = np.zeros(T), np.zeros(T), 0.0
    for i, k in enumerate(keys):
        kernel(k, state, 1.5 if order == 2 else 0.3)
        if i >= burn:
            xs = chains.to_host()[:, :, 0]
            s1 += xs.mean(0)
            s2 += (xs ** 2).mean(0)
            acc += chains.accepted.to_host().mean()
    mean, var = s1 / M, s2 / M - (s1 / M) ** 2
    assert 0.2 < acc / M < 0.99, acc / M
    se = np.sqrt(exact[:, 1] * 10 / (C * M))
    assert np.all(np.abs(mean - exact[:, 0]) < 5 * se + 0.01), (mean, exact[:, 0], se)
    npt.assert_allclose(var, exact[:, 1], rtol=0.04)


@pytest.mark.parametrize("chain_minor", [False, True])
def test_lorenz_sampler_targets_the_posterior_by_importance_sampling(chain_minor):
    """The same question for the extended-linearisation sampler of the Lorenz-63 model (examples/lorenz/auxiliary_kalman.py:14-52): T = 3, every step observed,
    dt = 0.05 so that the Euler-Maruyama drift is visibly nonlinear over a step, step size 50 (acceptance ~0.6: the linearisation error is what the MH ratio corrects).
    Ground truth: self-normalised importance sampling of the model's OWN unnormalised log-density (a vectorised copy, checked against log_likelihood_fn up to its constant)
    under a Student-t proposal (5 degrees of freedom) scaled from the chains' output -- consistent whatever the chains did, so a biased sampler cannot hide behind a proposal
    centred on its own bias (effective sample size asserted).  The prior covariance is 20 I here, not the example's diag(400, 20, 20): with the unobserved x_1 that diffuse the
    posterior is heavy-tailed enough for importance sampling to UNDERESTIMATE its variance by 4 % at an apparent effective sample size of 2 10^5 (an independent random-walk
    Metropolis run sides with the device chains: tools/diag_lorenz_is.py, diag_lorenz_is2.py) -- a property of the check, not of the sampler; with 20 I importance sampling
    and random-walk Metropolis agree to 0.5 %.  1024 device chains, 100 + 500 sweeps; all nine posterior means within 4 combined standard errors + 2 % of a posterior
    standard deviation, variances within 5 %."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman import get_kernel, LorenzModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    T, C, burn, M = 3, 1024, 100, 500
    m_, xtrue = lorenz_kalman_setup(T, every=1, dt=0.05, seed=3)
    model = LorenzModel(m_.yobs, m_.Hobs, m_.Robs, m_.cobs, m_.m0, 20.0 * np.eye(3), m_.theta, m_.sigma_x, m_.dt)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    h = _lib.default_handle()
    rng = np.random.default_rng(2)
    chains = DeviceChains(h, xtrue[None] + 0.5 * rng.standard_normal((C, T, 3)), chain_minor=chain_minor)
    state = KalmanSampler(x=chains, updated=None)
    keys = R.split(R.PRNGKey(8), burn + M)
    s1, s2, acc = np.zeros(9), np.zeros((9, 9)), 0.0
    for i, k in enumerate(keys):
        kernel(k, state, 50.0)
        if i >= burn:
            xs = chains.to_host().reshape(C, 9)
            s1 += xs.mean(0)
            s2 += xs.T @ xs / C
            acc += chains.accepted.to_host().mean()
    mean = s1 / M
    cov = s2 / M - np.outer(mean, mean)
    assert 0.1 < acc / M < 0.9, acc / M  # (the step must be large enough for the linearisation error to show in the acceptance rate)

    y, q, p0d = np.asarray(model.yobs, np.float64), model.Q[0, 0], np.diag(model.P0)

    def logpi(X):  # (n, T, 3): log N(x_0; m0, P0) + sum_t log N(x_t; mean(x_t-1), Q) + sum_t log N(y_t; (x_t2, x_t3), 5 I) up to a constant
        lp = -0.5 * np.sum((X[:, 0] - model.m0) ** 2 / p0d, -1)
        for t in range(1, T):
            lp += -0.5 * np.sum((X[:, t] - model.mean(X[:, t - 1])) ** 2, -1) / q
        for t in range(T):
            lp += -0.5 * np.sum((y[t] - X[:, t, 1:]) ** 2, -1) / 5.0
        return lp

    probe = xtrue[None] + rng.standard_normal((6, T, 3))
    assert np.ptp(logpi(probe) - np.array([model.log_likelihood_fn(x) for x in probe])) < 1e-9
    n_is, nu = 400000, 5.0
    Lq = np.linalg.cholesky(1.5 * cov)
    z = rng.standard_normal((n_is, 9)) / np.sqrt(rng.chisquare(nu, (n_is, 1)) / nu)
    xq = mean + z @ Lq.T
    lw = logpi(xq.reshape(n_is, T, 3)) + 0.5 * (nu + 9.0) * np.log1p((z ** 2).sum(1) / nu)
    w = np.exp(lw - lw.max())
    w /= w.sum()
    ess = 1.0 / (w ** 2).sum()
    assert ess > 50000, ess
    m_is = w @ xq
    v_is = w @ (xq - m_is) ** 2
    sd = np.sqrt(v_is)
    # 4 combined standard errors (importance sampling: sd / sqrt(ESS); chains: integrated autocorrelation time taken as 30) + 2 % of a posterior sd
    tol = 4.0 * sd * np.sqrt(1.0 / ess + 30.0 / (C * M)) + 0.02 * sd
    assert np.all(np.abs(mean - m_is) < tol), (mean - m_is, tol, ess)
    npt.assert_allclose(np.diag(cov), v_is, rtol=0.05)


@pytest.mark.parametrize("T", [120, 257])
@pytest.mark.parametrize("parallel", [True, False])
@pytest.mark.parametrize("nan_policy", ["reference"])
def test_lorenz_device_sweep_vs_oracle(T, parallel, nan_policy):
    """Config C4's Kalman half (Lorenz-63, extended linearisation rebuilt at x and at x_prop, sparse observations with NaN rows)
    on the device vs the oracle's sweep driven by the same NumPy factories."""
    from aux_ssm_samplers_amd.kalman import get_kernel
    model, xtrue = lorenz_kalman_setup(T)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, parallel)
    rng = np.random.Generator(np.random.PCG64(5))
    x = xtrue + 0.05 * rng.standard_normal((T, 3))
    delta = 0.02
    for rep in range(2):
        noise = dict(eps_aux=rng.standard_normal((T, 3)), eps_samp=rng.standard_normal((T, 3)), u_accept=rng.random())
        ref = K.kalman_sweep(x, delta, model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, parallel, **noise)
        out = kernel(None, init(x), delta, noise=noise)
        npt.assert_allclose(out.logs[0, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-8)
        npt.assert_allclose(out.log_alpha, ref["log_alpha"], rtol=1e-5, atol=1e-6)
        assert out.updated == ref["accepted"]
        npt.assert_allclose(out.x, ref["x"], rtol=1e-8, atol=1e-9)
        x = ref["x"]


def test_lorenz_host_factory_path_equals_device_sweep():
    from aux_ssm_samplers_amd.kalman import get_kernel
    T = 100
    model, xtrue = lorenz_kalman_setup(T, seed=2)
    init, kdev = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    init2, khost = get_kernel(lambda z: model.dynamics_factory(z), lambda z, u, dl: model.observations_factory(z, u, dl),
                              lambda z: model.log_likelihood_fn(z), True)
    rng = np.random.default_rng(9)
    x = xtrue + 0.05 * rng.standard_normal((T, 3))
    noise = dict(eps_aux=rng.standard_normal((T, 3)), eps_samp=rng.standard_normal((T, 3)), u_accept=0.4)
    a = kdev(None, init(x), 0.02, noise=noise)
    b = khost(None, init2(x), 0.02, noise=noise)
    npt.assert_allclose(a.x, b.x, rtol=1e-8, atol=1e-9)
    npt.assert_allclose(a.log_alpha, b.log_alpha, rtol=1e-5, atol=1e-6)
    assert a.updated == b.updated


@pytest.mark.parametrize("T,C", [(120, 33), (257, 64)])
@pytest.mark.parametrize("per_chain_theta", [False, True])
def test_lorenz_chain_minor_sweep_vs_oracle(T, C, per_chain_theta):
    """>= 32 chains run the Lorenz sweep chain-minor (state (T, 3, C), lanes over chains, per-chain linearised dynamics (n, 9, C)): every 7th
    chain vs the oracle's sweep on its own noise (and its own theta), all chains vs the dense-layout sweep."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel, LorenzModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    base, xtrue = lorenz_kalman_setup(T)
    rng = np.random.Generator(np.random.PCG64(8))
    theta = base.theta + (0.5 * rng.standard_normal((C, 3)) if per_chain_theta else 0.0) * np.ones((C, 3))

    def mk(th):
        return LorenzModel(base.yobs, base.Hobs, base.Robs, base.cobs, base.m0, base.P0, th, base.sigma_x, base.dt)

    x0 = xtrue[None] + 0.05 * rng.standard_normal((C, T, 3))
    noise = dict(eps_aux=rng.standard_normal((C, T, 3)), eps_samp=rng.standard_normal((C, T, 3)), u_accept=rng.random(C))
    h = _lib.default_handle()
    outs = {}
    for cmin in (True, False):
        model = mk(theta if per_chain_theta else theta[0])
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        chains = DeviceChains(h, x0, chain_minor=cmin)
        kernel(None, KalmanSampler(x=chains, updated=None), 0.02, noise=noise)
        outs[cmin] = (chains.to_host(), chains.accepted.to_host(), chains.logs.to_host())
    npt.assert_allclose(outs[True][0], outs[False][0], rtol=1e-8, atol=1e-9)
    npt.assert_array_equal(outs[True][1], outs[False][1])
    npt.assert_allclose(outs[True][2][:, 1:], outs[False][2][:, 1:], rtol=1e-8)
    for c in range(0, C, 7):
        mc = mk(theta[c])
        ref = K.kalman_sweep(x0[c], 0.02, mc.dynamics_factory, mc.observations_factory, mc.log_likelihood_fn, True, eps_aux=noise["eps_aux"][c],
                             eps_samp=noise["eps_samp"][c], u_accept=noise["u_accept"][c])
        npt.assert_allclose(outs[True][2][c, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-8)
        assert bool(outs[True][1][c]) == ref["accepted"]
        npt.assert_allclose(outs[True][0][c], ref["x"], rtol=1e-8, atol=1e-9)
    assert 0 < outs[True][1].sum() < C


@pytest.mark.parametrize("T", [1, 2, 3])
@pytest.mark.parametrize("kind", ["sv1", "sv2", "lorenz"])
def test_tiny_horizons_chain_minor_equals_dense(T, kind):
    """T = 1 (no transition at all), 2, 3 in both layouts: the degenerate scans and the t = 0 head terms agree."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel, LorenzModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    C = 33
    rng = np.random.default_rng(T)
    if kind == "lorenz":
        base, xtrue = lorenz_kalman_setup(max(T, 9))
        model = LorenzModel(base.yobs[:T], base.Hobs[:T], base.Robs[:T], base.cobs[:T], base.m0, base.P0, base.theta, base.sigma_x, base.dt)
        xtrue, d, delta = xtrue[:T], 3, 0.02
    else:
        y, xtrue, (m0, P0, F, Q, b) = sv_setup(max(T, 4), 2, seed=1)
        model = SVModel(y[:T], m0, P0, F, Q, b, order=1 if kind == "sv1" else 2)
        xtrue, d, delta = xtrue[:T], 2, 0.3
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    x0 = xtrue[None] + 0.1 * rng.standard_normal((C, T, d))
    noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
    h = _lib.default_handle()
    outs = {}
    for cmin in (True, False):
        chains = DeviceChains(h, x0, chain_minor=cmin)
        kernel(None, KalmanSampler(x=chains, updated=None), delta, noise=noise)
        outs[cmin] = (chains.to_host(), chains.accepted.to_host(), chains.logs.to_host())
    assert np.isfinite(outs[True][2]).all()
    npt.assert_allclose(outs[True][0], outs[False][0], rtol=1e-9, atol=1e-10)
    npt.assert_array_equal(outs[True][1], outs[False][1])
    npt.assert_allclose(outs[True][2], outs[False][2], rtol=1e-8, atol=1e-9)
    ref = K.kalman_sweep(x0[0], delta, model.dynamics_factory, model.observations_factory,
                         model.log_likelihood_fn if kind == "lorenz" else oracle_target(model), True, eps_aux=noise["eps_aux"][0],
                         eps_samp=noise["eps_samp"][0], u_accept=noise["u_accept"][0])
    npt.assert_allclose(outs[True][0][0], ref["x"], rtol=1e-8, atol=1e-9)
    npt.assert_allclose(outs[True][2][0, 0], ref["log_alpha"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("d,T,C", [(1, 300, 40), (3, 129, 64)])
@pytest.mark.parametrize("share", [1, 0])
def test_sv_chain_minor_array_free_filter_missing_data_and_fp32(order, d, T, C, share):
    """The array-free SV filter (round 4: csrc/kalman_bodies.h::FilterOpFlySV -- both scan passes fold the steps in information form from (x, u, y), the
    log-determinants multiplied up per chunk; the log-density pass re-forms the pseudo-observations) where its special cases live: MISSING data (a NaN y_t makes the
    second-order pseudo-observation of that component NaN = unobserved; whole missing steps are pure prediction steps), the device step size (sweep_dd) and fp32
    (auxiliary block around the predicted mean).  fp64: every chain against the oracle's sweep and the dense-layout sweep; fp32: against the fp64 device sweep on
    the same noise."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d, seed=11)
    y = y.copy()
    y[5] = np.nan
    y[9, 0] = np.nan
    y[T - 1, d - 1] = np.nan
    model = SVModel(y, m0, P0, F, Q, b, order=order)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    rng = np.random.Generator(np.random.PCG64(70 + d))
    x0 = xtrue[None] + 0.2 * rng.standard_normal((C, T, d))
    noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
    h = _lib.default_handle()
    h.set_option(_lib.OPT_SHARE_MODEL, share)
    try:
        outs = {}
        for key, cmin, dt in (("cm64", True, np.float64), ("dense64", False, np.float64), ("cm32", True, np.float32)):
            chains = DeviceChains(h, x0.astype(dt), chain_minor=cmin)
            kernel(None, KalmanSampler(x=chains, updated=None), 0.3, noise=noise)
            outs[key] = (chains.to_host(), chains.accepted.to_host(), chains.logs.to_host())
    finally:
        h.set_option(_lib.OPT_SHARE_MODEL, 1)
    npt.assert_allclose(outs["cm64"][0], outs["dense64"][0], rtol=1e-9, atol=1e-10)
    npt.assert_array_equal(outs["cm64"][1], outs["dense64"][1])
    npt.assert_allclose(outs["cm64"][2][:, 1:], outs["dense64"][2][:, 1:], rtol=1e-9)
    for c in range(0, C, 9):
        ref = K.kalman_sweep(x0[c], 0.3, model.dynamics_factory, model.observations_factory, oracle_target(model), True,
                             eps_aux=noise["eps_aux"][c], eps_samp=noise["eps_samp"][c], u_accept=noise["u_accept"][c])
        npt.assert_allclose(outs["cm64"][2][c, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-9)
        assert bool(outs["cm64"][1][c]) == ref["accepted"]
        npt.assert_allclose(outs["cm64"][0][c], ref["x"], rtol=1e-9, atol=1e-10)
    # fp32: log alpha within 2e-2 of the fp64 sweep's (sums of T d terms of size ~1), the same decisions except at the margin, accepted trajectories to fp32 accuracy
    la64, la32 = outs["cm64"][2][:, 0], outs["cm32"][2][:, 0].astype(np.float64)
    assert np.abs(la64 - la32).max() < 5e-2, np.abs(la64 - la32).max()
    same = outs["cm64"][1] == outs["cm32"][1]
    assert same.mean() > 0.9
    both = same & (outs["cm64"][1] == 1)
    if both.any():
        npt.assert_allclose(outs["cm32"][0][both], outs["cm64"][0][both], rtol=2e-3, atol=2e-3)

"""Time-varying transition parameters (reference _primitives/csmc/csmc.py:103: `Mt.params` / `Gt.params` are scanned over time) and
gradient-informed independent proposals (csmc/independent.py:57-75 with gradient=True, :121-134, :173-190, :252-268).

CPU: the oracle's closed-form gradient against central finite differences of an independently written NumPy joint log-density (the
reference has no test of this branch: jax.grad is its definition).  GPU: the sweeps bit-exact against the C oracle; the exact-weights
variant leaves the AR(1) prior invariant (the reference's test_flat_potential criterion)."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import csmc as O


def _tv_model(T, d, rng):
    A = rng.standard_normal((T - 1, d, d))
    Q = A @ A.transpose(0, 2, 1) / d + 0.5 * np.eye(d)
    F = 0.8 * np.eye(d) + 0.1 * rng.standard_normal((T - 1, d, d))
    b = 0.2 * rng.standard_normal((T - 1, d))
    return F, b, Q


def _logpi_numpy(u, m0, P0, F, b, Q, potential, y, sig):
    """log M0(u_0) + G0(u_0) + sum_t [log N(u_{t+1}; F_t u_t + b_t, Q_t) + G(u_{t+1})], plain NumPy / float64"""
    from scipy.stats import multivariate_normal as mvn
    T, d = u.shape
    lp = mvn.logpdf(u[0], m0, P0)

    def pot(t):
        if potential == O.POT_FLAT:
            return 0.0
        if potential == O.POT_GAUSS_OBS:
            return float(np.sum(-0.5 * ((y[t] - u[t]) / sig) ** 2 - np.log(sig) - 0.5 * np.log(2 * np.pi)))
        return float(np.sum(-0.5 * (y[t] ** 2 * np.exp(-u[t]) + u[t]) - 0.5 * np.log(2 * np.pi)))

    lp += pot(0)
    for t in range(T - 1):
        Ft, bt, Qt = (F[t], b[t], Q[t]) if F.ndim == 3 else (F, b, Q)
        lp += mvn.logpdf(u[t + 1], Ft @ u[t] + bt, Qt) + pot(t + 1)
    return float(lp)


@pytest.mark.parametrize("d", [1, 2, 3])
@pytest.mark.parametrize("potential", [O.POT_FLAT, O.POT_GAUSS_OBS, O.POT_SV])
@pytest.mark.parametrize("tv", [False, True])
def test_oracle_gradient_equals_finite_differences(d, potential, tv):
    rng = np.random.default_rng(10 * d + potential + 100 * tv)
    T, sig = 7, 0.7
    m0, P0 = 0.1 * rng.standard_normal(d), 2.0 * np.eye(d)
    if tv:
        F, b, Q = _tv_model(T, d, rng)
        od = dict(F=F[0], b=b[0], chol_Q=np.linalg.cholesky(Q[0]), F_t=F, b_t=b, chol_Q_t=np.linalg.cholesky(Q))
    else:
        A = rng.standard_normal((d, d))
        F, b, Q = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d)), 0.1 * rng.standard_normal(d), A @ A.T / d + 0.5 * np.eye(d)
        od = dict(F=F, b=b, chol_Q=np.linalg.cholesky(Q))
    od.update(proposal=O.AUX_INDEPENDENT, potential=potential, m0=m0, chol_P0=np.linalg.cholesky(P0), sig_y=sig)
    y = rng.standard_normal((T, d))
    u = rng.standard_normal((T, d))
    g = O.grad_logpi(od, u, y if potential else None)
    fd = np.zeros_like(u)
    h = 1e-6
    for t in range(T):
        for k in range(d):
            e = np.zeros_like(u)
            e[t, k] = h
            fd[t, k] = (_logpi_numpy(u + e, m0, P0, F, b, Q, potential, y, sig) - _logpi_numpy(u - e, m0, P0, F, b, Q, potential, y, sig)) / (2 * h)
    npt.assert_allclose(g, fd, rtol=2e-6, atol=2e-6)


def test_oracle_gradient_lorenz_equals_finite_differences():
    from scipy.stats import multivariate_normal as mvn
    from tests.helpers import lorenz_setup
    T = 9
    M0, Mt, G0, Gt, x, y, sig_y = lorenz_setup(T, every=2, dt=0.01)
    F = np.zeros((3, 3))
    F[0] = Mt.theta
    od = dict(proposal=O.AUX_INDEPENDENT, potential=O.POT_GAUSS_OBS_MASKED, m0=M0.m0, chol_P0=M0.chol(), F=F, b=[Mt.dt, 0, 0], chol_Q=Mt.chol(),
              sig_y=sig_y, transition=O.TRANS_LORENZ63_EM)
    u = x + 0.1 * np.random.default_rng(0).standard_normal(x.shape)

    def logpi(u):
        lp = mvn.logpdf(u[0], M0.m0, M0.P0)
        for t in range(T):
            ok = np.isfinite(y[t])
            lp += float(np.sum(-0.5 * ((y[t][ok] - u[t][ok]) / sig_y) ** 2 - np.log(sig_y) - 0.5 * np.log(2 * np.pi)))
            if t + 1 < T:
                lp += mvn.logpdf(u[t + 1], Mt.mean(u[t]), Mt.chol() @ Mt.chol().T)
        return lp

    g = O.grad_logpi(od, u, y)
    h = 1e-6
    fd = np.array([[(logpi(u + h * (np.arange(3 * T).reshape(T, 3) == 3 * t + k)) - logpi(u - h * (np.arange(3 * T).reshape(T, 3) == 3 * t + k))) / (2 * h)
                    for k in range(3)] for t in range(T)])
    npt.assert_allclose(g, fd, rtol=1e-5, atol=1e-4)


def _pot(kind, y, sig=0.7):
    from aux_ssm_samplers_amd.csmc import FlatPotential, GaussianObsPotential, SVPotential
    if kind == O.POT_FLAT:
        return FlatPotential(), FlatPotential()
    if kind == O.POT_GAUSS_OBS:
        return GaussianObsPotential(sig=sig, y=y[0]), GaussianObsPotential(sig=sig, params=y[1:])
    return SVPotential(y=y[0]), SVPotential(params=y[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,N,T", [(1, 64, 9), (2, 100, 33), (3, 256, 20), (4, 65, 12), (5, 64, 11), (6, 25, 14), (17, 33, 8), (30, 25, 9), (32, 64, 5)])  # d > 4: csrc/csmc_wide.hip
@pytest.mark.parametrize("proposal", [O.BOOTSTRAP_LG, O.AUX_INDEPENDENT])
@pytest.mark.parametrize("backward", [True, False])
def test_time_varying_transitions_bit_exact_vs_oracle(dtype, d, N, T, proposal, backward):
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics
    rng = np.random.default_rng(7 * d + N + T)
    F, b, Q = _tv_model(T, d, rng)
    M0 = GaussianInit(m0=0.1 * rng.standard_normal(d), P0=2.0 * np.eye(d))
    Mt = LinearGaussianDynamics(F=F, b=b, Q=Q)
    assert Mt.time_varying
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(O.POT_GAUSS_OBS, y)
    x0 = rng.standard_normal((T, d)).astype(dtype)
    noise = dict(eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T))
    od = dict(proposal=proposal, potential=O.POT_GAUSS_OBS, m0=M0.m0, chol_P0=M0.chol(), F=F[0], b=b[0], chol_Q=np.linalg.cholesky(Q[0]),
              F_t=F, b_t=b, chol_Q_t=np.linalg.cholesky(Q), sig_y=0.7)
    delta, okw = None, {}
    if proposal == O.AUX_INDEPENDENT:
        delta = 0.3 + rng.random(T)
        noise["eps_aux"] = rng.standard_normal((T, d))
        fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
        okw = dict(sqrt_half_delta=np.sqrt(0.5 * delta), eps_aux=noise["eps_aux"])
    else:
        fk = _device.describe_bootstrap(M0, G0, Mt, Gt, Mt)
    nz = {k: np.asarray(v, dtype)[None] for k, v in noise.items()}
    x, anc, hist = _device.sweep(fk, x0, N, backward, noise=nz, delta=delta, want_history=True)
    ref = O.sweep(od, x0, N, backward, y=y, eps_prop=noise["eps_prop"], u_res=noise["u_res"], u_bwd=noise["u_bwd"], dtype=dtype, **okw)
    npt.assert_array_equal(hist["xs"], ref["xs"])
    npt.assert_array_equal(hist["log_ws"], ref["log_ws"])
    npt.assert_array_equal(hist["As"], ref["As"])
    npt.assert_array_equal(anc, ref["ancestors"])
    npt.assert_array_equal(x, ref["x"])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,N,T", [(1, 64, 9), (2, 100, 33), (3, 512, 20), (5, 33, 12), (8, 64, 7), (30, 25, 10), (32, 17, 6)])  # d > 4: the wide-state kernels (round 4)
@pytest.mark.parametrize("potential", [O.POT_FLAT, O.POT_GAUSS_OBS, O.POT_SV])
@pytest.mark.parametrize("gradient", [True, "exact"])
@pytest.mark.parametrize("tv", [False, True])
def test_gradient_proposals_bit_exact_vs_oracle(dtype, d, N, T, potential, gradient, tv):
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics
    from aux_ssm_samplers_amd import _lib
    rng = np.random.default_rng(3 * d + N + T + potential)
    M0 = GaussianInit(m0=0.1 * rng.standard_normal(d), P0=2.0 * np.eye(d))
    if tv:
        F, b, Q = _tv_model(T, d, rng)
        ext = dict(F=F[0], b=b[0], chol_Q=np.linalg.cholesky(Q[0]), F_t=F, b_t=b, chol_Q_t=np.linalg.cholesky(Q))
    else:
        A = rng.standard_normal((d, d))
        F, b, Q = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d)), 0.1 * rng.standard_normal(d), A @ A.T / d + 0.5 * np.eye(d)
        ext = dict(F=F, b=b, chol_Q=np.linalg.cholesky(Q))
    Mt = LinearGaussianDynamics(F=F, b=b, Q=Q)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(potential, y)
    gmode = _lib.GRAD_EXACT if gradient == "exact" else _lib.GRAD_REFERENCE
    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt, gmode)
    x0 = rng.standard_normal((T, d)).astype(dtype)
    delta = 0.05 + 0.1 * rng.random(T)
    noise = dict(eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T), eps_aux=rng.standard_normal((T, d)))
    od = dict(proposal=O.AUX_INDEPENDENT, potential=potential, m0=M0.m0, chol_P0=M0.chol(), sig_y=0.7, gradient=gmode, **ext)
    nz = {k: np.asarray(v, dtype)[None] for k, v in noise.items()}
    x, anc, hist = _device.sweep(fk, x0, N, True, noise=nz, delta=delta, want_history=True)
    ref = O.sweep(od, x0, N, True, y=y if potential else None, eps_prop=noise["eps_prop"], u_res=noise["u_res"], u_bwd=noise["u_bwd"], dtype=dtype,
                  sqrt_half_delta=np.sqrt(0.5 * delta), eps_aux=noise["eps_aux"])
    npt.assert_array_equal(hist["xs"], ref["xs"])
    npt.assert_array_equal(hist["log_ws"], ref["log_ws"])
    npt.assert_array_equal(anc, ref["ancestors"])
    npt.assert_array_equal(x, ref["x"])
    # the proposals really moved: without the gradient the particles differ
    fk0 = _device.describe_independent(M0, G0, Mt, Gt, Mt)
    _, _, h0 = _device.sweep(fk0, x0, N, True, noise=nz, delta=delta, want_history=True)
    assert np.max(np.abs(h0["xs"][:, 1:] - hist["xs"][:, 1:])) > 0


@pytest.mark.gpu
def test_gradient_exact_keeps_the_ar1_prior_invariant():
    """the criterion of the reference's test_flat_potential (test_csmc.py:18-69): mean 0, var 1, lag-1 cov rho, atol 0.05 -- here through
    get_independent_kernel(..., gradient="exact") (per-particle importance correction at every step)"""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.csmc import get_independent_kernel, GaussianInit, LinearGaussianDynamics, FlatPotential
    T, N, rho, C, M = 5, 32, 0.9, 2048, 40
    M0 = GaussianInit(m0=[0.0], P0=[[1.0]])
    Mt = LinearGaussianDynamics(F=[[rho]], b=[0.0], Q=[[1 - rho ** 2]])
    init, kernel = get_independent_kernel(M0, FlatPotential(), Mt, FlatPotential(), N=N, backward=True, Pt=Mt, gradient="exact")
    rng = np.random.default_rng(0)
    # the auxiliary kernel makes local moves (it mixes slowly at this delta), so the test is of INVARIANCE: start every chain from the AR(1)
    # law itself and check that the moments stay put.  (With gradient=True, the reference's own weights, they do not: its
    # GradientAuxiliaryGt correction is a per-step constant, independent.py:265-266 -- var drifts to ~0.7 within 12 sweeps in the oracle.)
    x0 = np.zeros((C, T, 1))
    x0[:, 0, 0] = rng.standard_normal(C)
    for t in range(1, T):
        x0[:, t, 0] = rho * x0[:, t - 1, 0] + np.sqrt(1 - rho ** 2) * rng.standard_normal(C)
    state = init(x0.astype(np.float32))
    keys = R.split(R.PRNGKey(0), M)
    out = []
    for it in range(M):
        state = kernel(keys[it], state, 0.4)
        if it >= M // 2:
            out.append(state.x[:, :, 0])
    xs = np.concatenate(out, axis=0)
    npt.assert_allclose(xs.mean(0), 0.0, atol=0.05)
    npt.assert_allclose(xs.var(0), 1.0, atol=0.05)
    npt.assert_allclose(np.mean(xs[:, 1:] * xs[:, :-1], axis=0), rho, atol=0.05)


@pytest.mark.gpu
def test_time_varying_and_gradient_argument_checks():
    from aux_ssm_samplers_amd.csmc import _device, get_independent_kernel, GaussianInit, LinearGaussianDynamics, FlatPotential
    M0 = GaussianInit(m0=[0.0], P0=[[1.0]])
    F = np.full((3, 1, 1), 0.9)
    Mt = LinearGaussianDynamics(F=F, b=np.zeros((3, 1)), Q=np.ones((3, 1, 1)))
    fk = _device.describe_bootstrap(M0, FlatPotential(), Mt, FlatPotential(), Mt)
    with pytest.raises(ValueError):  # 3 transition rows need T = 4
        _device.sweep(fk, np.zeros((6, 1), np.float32), 8, False, key=0)
    init, kern = get_independent_kernel(M0, FlatPotential(), Mt, FlatPotential(), N=8, gradient=True, parallel=True)  # (round 3: built)
    assert callable(kern)
    with pytest.raises(ValueError):
        get_independent_kernel(M0, FlatPotential(), Mt, FlatPotential(), N=8, gradient="maybe")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,N,T", [(1, 32, 9), (1, 64, 37), (2, 100, 33), (3, 256, 20), (4, 33, 16)])
@pytest.mark.parametrize("potential", [O.POT_FLAT, O.POT_GAUSS_OBS, O.POT_SV])
@pytest.mark.parametrize("gradient", [False, True])
@pytest.mark.parametrize("tv", [False, True])
def test_parallel_in_time_sweep_with_gradient_proposals_and_time_varying_transitions(dtype, d, N, T, potential, gradient, tv):
    """csmc/independent.py:78-118 with gradient=True (proposals mt = N(u + delta/2 grad, delta/2 I), leaf weights qt.logpdf - mt.logpdf:
    pit/csmc.py:83-91) and / or transitions read row by row: the tree kernels of csrc/pit.hip against the block-gathering oracle, bit for bit."""
    if not gradient and not tv:
        pytest.skip("covered by tests/test_gpu_pit.py")
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics
    from aux_ssm_samplers_amd import _lib
    rng = np.random.default_rng(11 * d + N + T + potential)
    M0 = GaussianInit(m0=0.1 * rng.standard_normal(d), P0=2.0 * np.eye(d))
    if tv:
        F, b, Q = _tv_model(T, d, rng)
        ext = dict(F=F[0], b=b[0], chol_Q=np.linalg.cholesky(Q[0]), F_t=F, b_t=b, chol_Q_t=np.linalg.cholesky(Q))
    else:
        A = rng.standard_normal((d, d))
        F, b, Q = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d)), 0.1 * rng.standard_normal(d), A @ A.T / d + 0.5 * np.eye(d)
        ext = dict(F=F, b=b, chol_Q=np.linalg.cholesky(Q))
    Mt = LinearGaussianDynamics(F=F, b=b, Q=Q)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(potential, y)
    gmode = _lib.GRAD_EXACT if gradient else _lib.GRAD_NONE
    fk = _device.describe_independent(M0, G0, Mt, Gt, None, gmode)
    x0 = rng.standard_normal((T, d)).astype(dtype)
    delta = 0.05 + 0.1 * rng.random(T)
    noise = dict(eps_aux=rng.standard_normal((T, d)), eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T, N)))
    noise = {k: np.asarray(v, dtype) for k, v in noise.items()}
    x, anc = _device.pit_sweep(fk, x0, N, noise={k: v[None] for k, v in noise.items()}, delta=delta)
    od = dict(proposal=O.AUX_INDEPENDENT, potential=potential, m0=M0.m0, chol_P0=M0.chol(), sig_y=0.7, gradient=gmode, **ext)
    ref = O.pit_sweep(od, x0, N, y=y if potential else None, sqrt_half_delta=np.sqrt(0.5 * delta), dtype=dtype, **noise)
    npt.assert_array_equal(anc, ref["ancestors"])
    npt.assert_array_equal(x, ref["x"])
    assert anc.any()
    if gradient:  # the shifted proposals really are different particles
        fk0 = _device.describe_independent(M0, G0, Mt, Gt, None)
        x_ng, _ = _device.pit_sweep(fk0, x0, N, noise={k: v[None] for k, v in noise.items()}, delta=delta)
        assert np.max(np.abs(x_ng - x)) > 0


def test_oracle_parallel_in_time_gradient_kernel_targets_the_smoother():
    """CPU: the parallel-in-time sweep with gradient-informed proposals (oracle restatement of independent.py:81-84 + pit/csmc.py:83-91) leaves the
    smoothing distribution of a linear-Gaussian model invariant -- the importance weights qt / mt per particle make up for the shifted proposals."""
    from tests.test_oracle_pit import _lg, smoother
    T, N, rho, sig_y, M, B = 6, 8, 0.9, 0.5, 30000, 1000
    model, xtrue, y = _lg(T, rho, sig_y, seed=3)
    model = dict(model, gradient=O.GRAD_EXACT)
    mean, var = smoother(T, rho, sig_y, y[:, 0])
    rng = np.random.default_rng(0)
    x = np.zeros((T, 1))
    acc, acc2, upd = np.zeros(T), np.zeros(T), np.zeros(T)
    shd = np.full(T, np.sqrt(0.5 * 0.15))  # (a gradient step this model's sharp likelihood, sig_y = 0.5, does not overshoot)
    for it in range(M):
        out = O.pit_sweep(model, x, N, y=y, sqrt_half_delta=shd, eps_aux=rng.standard_normal((T, 1)), eps_prop=rng.standard_normal((T, N, 1)),
                          u_res=rng.random((T, N)), dtype=np.float64)
        x = out["x"]
        if it >= B:
            acc += x[:, 0]
            acc2 += x[:, 0] ** 2
            upd += out["ancestors"] != 0
    n = M - B
    m_hat = acc / n
    v_hat = acc2 / n - m_hat ** 2
    assert upd.min() / n > 0.2
    npt.assert_allclose(m_hat, mean, atol=0.03)
    npt.assert_allclose(v_hat, var, rtol=0.08)

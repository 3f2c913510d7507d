"""extended / cubature / gauss_hermite (reference aux_samplers/_primitives/linearisation.py): the oracle's restatement against the reference's own known-answer
test (test_linearisation.py:13-48), and the device kernel (auxssm_linearise) against that test and against the oracle on the Lorenz-63 step."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import linearisation_np as O


def _reference_test_linear_inputs(seed=0):
    """the draws of test_linearisation.py:14-30 (np.random.randn in that order)"""
    rs = np.random.RandomState(seed)
    A, b = rs.randn(2, 4), rs.randn(2)
    q = rs.randn(2, 5)
    x_star = rs.randn(4)
    p_star = rs.randn(4, 10)
    return A, b, q @ q.T, x_star, p_star @ p_star.T


def _check_reference_known_answer(res, A, Q, b, **tol):
    (F_e, Q_e, b_e), (F_gh, Q_gh, b_gh), (F_c, Q_c, b_c) = res
    for F_, Q_, b_ in ((F_gh, Q_gh, b_gh), (F_c, Q_c, b_c)):      # test_linearisation.py:36-42
        npt.assert_allclose(F_e, F_, **tol)
        npt.assert_allclose(Q_e, Q_, **tol)
        npt.assert_allclose(b_e, b_, **tol)
    npt.assert_allclose(F_e, A, **tol)                               # :44-46
    npt.assert_allclose(Q_e, Q, **tol)
    npt.assert_allclose(b_e, b, **tol)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_reproduces_the_reference_known_answer(seed):
    A, b, Q, x_star, P_star = _reference_test_linear_inputs(seed)
    mean, cov = (lambda x, _: A @ x + b), (lambda *_: Q)
    res = (O.extended(mean, cov, None, x_star, P_star, jac=lambda x, _: A), O.gauss_hermite(mean, cov, None, x_star, P_star),
           O.cubature(mean, cov, None, x_star, P_star))
    _check_reference_known_answer(res, A, Q, b, rtol=1e-7, atol=1e-10)   # assert_allclose's default rtol, as the reference


def test_oracle_rules_and_host_rules_agree():
    """the reference's construction of the Gauss-Hermite rule (Hermite recurrence, np.roots, roll table) and the product's (hermgauss, itertools) are the
    same measure: same weighted point set, moments of N(0, I) exact to the rule's degree"""
    from aux_ssm_samplers_amd._primitives.linearisation import _cubature_rule, _gauss_hermite_rule
    for dim, order in [(1, 3), (2, 3), (3, 4), (2, 5)]:
        w, xi = O.gauss_hermite_points(dim, order)
        w2, xi2 = _gauss_hermite_rule(dim, order)
        a = sorted(map(tuple, np.round(np.column_stack([xi.T, w]), 10)))
        b = sorted(map(tuple, np.round(np.column_stack([xi2, w2]), 10)))
        npt.assert_allclose(a, b, atol=1e-9)
        npt.assert_allclose(w.sum(), 1.0, atol=1e-12)
        npt.assert_allclose((xi * w) @ xi.T, np.eye(dim), atol=1e-10)
    w, xi = O.cubature_points(3)
    w2, xi2 = _cubature_rule(3)
    npt.assert_allclose(xi.T, xi2)
    npt.assert_allclose(w, w2)


def _lorenz(theta, dt):
    def mean(x, _):
        return np.array([x[0] + dt * (theta[0] * (x[1] - x[0])), x[1] + dt * (theta[1] * x[0] - x[1] - x[0] * x[2]), x[2] + dt * (x[0] * x[1] - theta[2] * x[2])])

    def jac(x, _):
        return np.eye(3) + dt * np.array([[-theta[0], theta[0], 0.0], [theta[1] - x[2], -1.0, -x[0]], [x[1], x[0], -theta[2]]])
    return mean, jac


def test_host_path_with_python_callables_equals_oracle():
    """arbitrary callables stay on the host (NumPy): same numbers as the oracle's restatement on the Lorenz-63 step"""
    from aux_ssm_samplers_amd._primitives import linearisation as Lz
    theta, dt = np.array([10.0, 28.0, 8.0 / 3.0]), 0.02
    mean, jac = _lorenz(theta, dt)
    Q = 0.1 * np.eye(3)
    rng = np.random.default_rng(0)
    x, B = rng.standard_normal(3) * 5, rng.standard_normal((3, 6))
    P = B @ B.T / 6
    for got, want in ((Lz.cubature(mean, lambda *_: Q, None, x, P), O.cubature(mean, lambda *_: Q, None, x, P)),
                      (Lz.gauss_hermite(mean, lambda *_: Q, None, x, P, order=4), O.gauss_hermite(mean, lambda *_: Q, None, x, P, order=4)),
                      (Lz.extended(mean, lambda *_: Q, None, x, None, jac=jac), O.extended(mean, lambda *_: Q, None, x, None, jac=jac))):
        for g, w in zip(got, want):
            npt.assert_allclose(g, w, rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("seed", [0, 1])
def test_device_reproduces_the_reference_known_answer(seed, dtype):
    """test_linearisation.py:13-48 through auxssm_linearise: the affine map R^4 -> R^2 is recovered by all three methods"""
    from aux_samplers import extended, gauss_hermite, cubature
    from aux_ssm_samplers_amd._primitives.linearisation import AffineMean, ConstantCov
    A, b, Q, x_star, P_star = _reference_test_linear_inputs(seed)
    mean, cov = AffineMean(A.astype(dtype), b.astype(dtype)), ConstantCov(Q.astype(dtype))
    xs, Ps = x_star.astype(dtype), P_star.astype(dtype)
    res = (extended(mean, cov, None, xs, Ps), gauss_hermite(mean, cov, None, xs, Ps), cubature(mean, cov, None, xs, Ps))
    assert res[0][0].shape == (2, 4) and res[1][1].shape == (2, 2) and res[2][2].shape == (2,) and res[1][0].dtype == dtype
    tol = dict(rtol=1e-7, atol=1e-9) if dtype == np.float64 else dict(rtol=2e-3, atol=2e-3)
    _check_reference_known_answer(res, A, Q, b, **tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_vs_oracle_on_the_lorenz_step(dtype):
    """a trajectory's worth of linearisation points in one launch (the reference's vmap over x[:-1], examples/lorenz/auxiliary_kalman.py:26-28), shared and
    per-point P_star, Gauss-Hermite orders 2..5, against the oracle point by point; results kept in HBM on request"""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd._primitives import linearisation as Lz
    theta, dt = np.array([10.0, 28.0, 8.0 / 3.0]), 0.02
    mean_o, jac_o = _lorenz(theta, dt)
    Q = np.diag([0.1, 0.2, 0.3]) + 0.01
    rng = np.random.default_rng(3)
    n = 300
    x = (rng.standard_normal((n, 3)) * np.array([8.0, 9.0, 8.0]) + np.array([0.0, 0.0, 25.0])).astype(dtype)
    B = rng.standard_normal((n, 3, 5))
    Pn = (np.einsum("nij,nkj->nik", B, B) / 5 * 0.5).astype(dtype)
    mean, cov = Lz.Lorenz63Mean(theta, dt), Lz.ConstantCov(Q)
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == np.float64 else dict(rtol=5e-3, atol=5e-3)
    x64, P64 = x.astype(np.float64), Pn.astype(np.float64)
    sample = list(range(0, n, 37)) + [n - 1]

    def check(got, fn, P_of):
        F, Qo, b = got
        assert F.shape == (n, 3, 3) and Qo.shape == (n, 3, 3) and b.shape == (n, 3) and F.dtype == dtype
        for i in sample:
            w = fn(x64[i], P_of(i))
            npt.assert_allclose(F[i], w[0], **tol)
            npt.assert_allclose(Qo[i], w[1], **tol)
            npt.assert_allclose(b[i], w[2], **tol)

    cq = lambda *_: Q
    check(Lz.extended(mean, cov, None, x, None), lambda xi, P: O.extended(mean_o, cq, None, xi, None, jac=jac_o), lambda i: None)
    check(Lz.cubature(mean, cov, None, x, Pn[0]), lambda xi, P: O.cubature(mean_o, cq, None, xi, P), lambda i: P64[0])
    check(Lz.cubature(mean, cov, None, x, Pn), lambda xi, P: O.cubature(mean_o, cq, None, xi, P), lambda i: P64[i])
    for order in (2, 3, 5):
        check(Lz.gauss_hermite(mean, cov, None, x, Pn, order=order), lambda xi, P: O.gauss_hermite(mean_o, cq, None, xi, P, order=order), lambda i: P64[i])
    # one point, the reference's call shape; resident input and output
    F1, Q1, b1 = Lz.cubature(mean, cov, None, x[5], Pn[5])
    w = O.cubature(mean_o, cq, None, x64[5], P64[5])
    npt.assert_allclose(F1, w[0], **tol), npt.assert_allclose(Q1, w[1], **tol), npt.assert_allclose(b1, w[2], **tol)
    h = _lib.default_handle()
    Fd, Qd, bd = Lz.gauss_hermite(mean, cov, None, h.to_device(x), h.to_device(Pn), device_out=True)
    assert isinstance(Fd, _lib.DeviceArray)
    npt.assert_allclose(Fd.to_host()[7], O.gauss_hermite(mean_o, cq, None, x64[7], P64[7])[0], **tol)
    # a covariance that is not positive definite: NaN rows as jnp.linalg.cholesky gives, the other points untouched
    Pbad = Pn.copy()
    Pbad[3] = -np.eye(3)
    Fb, _, _ = Lz.cubature(mean, cov, None, x, Pbad)
    assert np.all(np.isnan(Fb[3])) and np.all(np.isfinite(Fb[4]))
    with pytest.raises(ValueError):
        Lz.cubature(mean, cov, None, x[:, :2], Pn)
    with pytest.raises(ValueError):
        Lz.gauss_hermite(mean, cov, None, x, Pn, order=9)


@pytest.mark.gpu
def test_lorenz_kalman_kernel_with_cubature_dynamics_vs_oracle_sweep():
    """the use the reference documents for these functions: a dynamics_factory built on a sigma-point linearisation handed to kalman.get_kernel
    (kalman/generic.py:53-106).  Device linearisation inside a Python factory, filter / sampler / log-densities through the primitives, against the
    oracle's sweep driven by the oracle's own cubature."""
    from aux_ssm_samplers_amd.kalman import get_kernel
    from aux_ssm_samplers_amd._primitives import linearisation as Lz
    from oracle import kalman_np as K
    from tests.helpers import lorenz_kalman_setup
    T = 60
    model, xtrue = lorenz_kalman_setup(T, seed=4)
    theta, dt = np.asarray(model.theta, np.float64), float(model.dt)
    Qm = np.asarray(model.Q)
    P_star = 0.01 * np.eye(3)
    mean, cov = Lz.Lorenz63Mean(theta, dt), Lz.ConstantCov(Qm)
    mean_o, _ = _lorenz(theta, dt)

    def dyn_device(x):
        Fs, Qs, bs = Lz.cubature(mean, cov, None, x[:-1], P_star)
        return model.m0, model.P0, Fs, Qs, bs

    def dyn_oracle(x):
        out = [O.cubature(mean_o, lambda *_: Qm, None, xi, P_star) for xi in x[:-1]]
        return model.m0, model.P0, np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), np.stack([o[2] for o in out])

    init, kernel = get_kernel(dyn_device, lambda z, u, dl: model.observations_factory(z, u, dl), lambda z: model.log_likelihood_fn(z), True)
    rng = np.random.default_rng(11)
    x = xtrue + 0.05 * rng.standard_normal((T, 3))
    noise = dict(eps_aux=rng.standard_normal((T, 3)), eps_samp=rng.standard_normal((T, 3)), u_accept=0.3)
    ref = K.kalman_sweep(x, 0.02, dyn_oracle, model.observations_factory, model.log_likelihood_fn, True, **noise)
    out = kernel(None, init(x), 0.02, noise=noise)
    npt.assert_allclose(out.log_alpha, ref["log_alpha"], rtol=1e-5, atol=1e-6)
    assert out.updated == ref["accepted"]
    npt.assert_allclose(out.x, ref["x"], rtol=1e-8, atol=1e-9)

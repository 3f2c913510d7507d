"""aux_samplers.mvn at the import surface (reference aux_samplers/__init__.py:3, _primitives/math/mvn/base.py).
GPU: `logpdf` through auxssm_mvn_logpdf against the reference's own doctest vector (mvn/base.py:37-45 -> scipy), against the oracle
(incl. the NaN / inf "numerically ignored" entries the filter relies on) and with broadcasting.  CPU: the host helpers."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K


def test_import_surface():
    import aux_samplers
    from aux_samplers import mvn
    from aux_samplers._primitives.math import mvn as mvn2, normalize, logsubexp, log1mexp  # noqa: F401
    from aux_samplers._primitives.math.mvn import logpdf, rvs  # noqa: F401  (reference mvn/__init__.py:1)
    assert mvn is mvn2 and hasattr(aux_samplers, "SamplerState") and hasattr(aux_samplers, "delta_adaptation")
    for name in ("logpdf", "rvs", "tril_log_det", "get_optimal_covariance"):
        assert callable(getattr(mvn, name))


def test_tril_log_det_ignores_nonfinite():
    from aux_ssm_samplers_amd._primitives.math import mvn
    L = np.array([[2.0, 0, 0], [1.0, np.nan, 0], [0.5, 0.1, -3.0]])
    npt.assert_allclose(mvn.tril_log_det(L), np.log(2.0) + np.log(3.0))
    npt.assert_allclose(mvn.tril_log_det(np.array([2.0, np.inf, 0.5])), 0.0, atol=1e-15)
    npt.assert_allclose(mvn.tril_log_det(L), K.tril_log_det(L))


def _optcov(impl):
    if impl == "oracle":
        from oracle.post_np import get_optimal_covariance
    else:
        from aux_ssm_samplers_amd._primitives.math.mvn import get_optimal_covariance
    return get_optimal_covariance


@pytest.mark.parametrize("impl", ["oracle", pytest.param("device", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("d", [1, 2, 3, 5, 30, 64])
def test_get_optimal_covariance_dominates_both(d, impl):
    """mvn/base.py:78-105 has no reference test; the defining property: Q = L L^T dominates both covariances, and equals P when P >= Sig."""
    import types
    mvn = types.SimpleNamespace(get_optimal_covariance=_optcov(impl))
    rng = np.random.default_rng(d)
    A, B = rng.standard_normal((d, d + 2)), rng.standard_normal((d, d + 2))
    P, S = A @ A.T, B @ B.T
    LQ = mvn.get_optimal_covariance(np.linalg.cholesky(P), np.linalg.cholesky(S))
    Q = LQ @ LQ.T
    if d == 1:
        npt.assert_allclose(LQ, np.maximum(np.linalg.cholesky(P), np.linalg.cholesky(S)))
        return
    tol = 1e-13 * d * np.linalg.norm(Q, 2) + 1e-10
    assert np.linalg.eigvalsh(Q - S).min() > -tol and np.linalg.eigvalsh(Q - P).min() > -tol
    big = P + S  # dominates S: the optimum is then S's dominating matrix = P + S itself? no: Q(P+S, S) must equal P + S
    LQ2 = mvn.get_optimal_covariance(np.linalg.cholesky(big), np.linalg.cholesky(S))
    npt.assert_allclose(LQ2 @ LQ2.T, big, rtol=1e-9, atol=1e-10 * d)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("d", [2, 4, 17, 64])
def test_get_optimal_covariance_device_vs_oracle(d, dtype):
    """auxssm_mvn_optimal_covariance (Jacobi eigen-decomposition in LDS) against oracle/post_np.py (LAPACK eigh): the factor itself -- the Cholesky factor of
    L L^T is unique, whatever the order and the signs of the eigenvectors -- incl. repeated eigenvalues (P = Sig), and the vector branch."""
    from aux_ssm_samplers_amd._primitives.math import mvn
    from oracle import post_np as O
    rng = np.random.default_rng(100 + d)
    A, B = rng.standard_normal((d, 2 * d)), rng.standard_normal((d, 2 * d))
    LP, LS = np.linalg.cholesky(A @ A.T / d).astype(dtype), np.linalg.cholesky(B @ B.T / d).astype(dtype)
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == np.float64 else dict(rtol=3e-3, atol=3e-4)
    got = mvn.get_optimal_covariance(LP, LS)
    assert got.dtype == dtype and np.all(np.triu(got, 1) == 0)
    npt.assert_allclose(got, O.get_optimal_covariance(LP.astype(np.float64), LS.astype(np.float64)), **tol)
    npt.assert_allclose(mvn.get_optimal_covariance(LP, LP), LP, **tol)          # Y = I: every eigenvalue is 1
    v1, v2 = np.abs(rng.standard_normal(d)).astype(dtype), np.abs(rng.standard_normal(d)).astype(dtype)
    npt.assert_array_equal(mvn.get_optimal_covariance(v1, v2), np.maximum(v1, v2))
    assert mvn.get_optimal_covariance(dtype(2.0), dtype(3.0)) == 3.0
    with pytest.raises(ValueError):
        mvn.get_optimal_covariance(np.eye(65), np.eye(65))


def test_log1mexp_logsubexp():
    from aux_ssm_samplers_amd._primitives.math import log1mexp, logsubexp
    x = np.array([-1e-8, -0.1, -0.7, -5.0, -40.0])
    npt.assert_allclose(log1mexp(x), np.log1p(-np.exp(x)), rtol=1e-7)
    npt.assert_allclose(logsubexp(np.log(5.0), np.log(3.0)), np.log(2.0), rtol=1e-12)
    npt.assert_allclose(logsubexp(np.log(3.0), np.log(5.0)), np.log(2.0), rtol=1e-12)


@pytest.mark.gpu
def test_logpdf_reference_doctest_vector_on_hip():
    # aux_samplers/_primitives/math/mvn/base.py:37-45
    from scipy.stats import multivariate_normal
    from aux_samplers import mvn
    z, mu = np.array([1.0, 2, 3]), np.array([2.0, 3, 4])
    L = np.array([[1, 0, 0], [0.2, 1.3, 0], [0.123, -0.5, 1.7]])
    want = multivariate_normal.logpdf(z, mu, L @ L.T)
    assert np.allclose(mvn.logpdf(z, mu, L), want)
    npt.assert_allclose(mvn.logpdf(z, mu, L), want, rtol=1e-13)
    npt.assert_allclose(mvn.logpdf(z.astype(np.float32), mu.astype(np.float32), L.astype(np.float32)), want, rtol=2e-6)
    # the same number through the Kalman entry point: T = 1, m0 = mu, P0 = L L^T, observation missing (K8 inside K7)
    import aux_ssm_samplers_amd._primitives.kalman as P
    lg = P.LGSSM(mu, L @ L.T, np.zeros((0, 3, 3)), np.zeros((0, 3, 3)), np.zeros((0, 3)), np.zeros((1, 1, 3)), np.ones((1, 1, 1)), np.zeros((1, 1)))
    npt.assert_allclose(P.prior_logpdf(z[None], lg), want, rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 2, 4, 7, 16, 64])
def test_logpdf_vs_oracle_with_broadcasting_and_masked_entries(dim):
    from aux_ssm_samplers_amd._primitives.math import mvn
    rng = np.random.default_rng(dim)
    nb = 37
    A = rng.standard_normal((nb, dim, dim + 3))
    L = np.linalg.cholesky(A @ A.transpose(0, 2, 1) + 0.1 * np.eye(dim))
    x, m = rng.standard_normal((nb, dim)), rng.standard_normal(dim)  # m broadcasts
    got = mvn.logpdf(x, m, L)
    want = np.array([K.mvn_logpdf(x[i], m, L[i]) for i in range(nb)])
    npt.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
    npt.assert_allclose(mvn.logpdf(x[3], m, L[3]), want[3], rtol=1e-10)       # no batch axis at all
    npt.assert_allclose(mvn.logpdf(x, m, L[0]), [K.mvn_logpdf(x[i], m, L[0]) for i in range(nb)], rtol=1e-10, atol=1e-10)
    if dim >= 2:
        # a masked component as sequential_update builds it (filtering.py:95-104): +inf on the diagonal, zero row / column, residual 0
        Lm = L.copy()
        k = dim // 2
        Lm[:, k, :] = 0.0
        Lm[:, :, k] = 0.0
        Lm[:, k, k] = np.inf
        xm = x.copy()
        xm[:, k] = m[k]
        got = mvn.logpdf(xm, m, Lm)
        want = np.array([K.mvn_logpdf(xm[i], m, Lm[i]) for i in range(nb)])
        npt.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
        Ln = Lm.copy()
        Ln[:, k, k] = np.nan
        npt.assert_allclose(mvn.logpdf(xm, m, Ln), want, rtol=1e-10, atol=1e-10)


@pytest.mark.gpu
def test_rvs_moments():
    from aux_ssm_samplers_amd._primitives.math import mvn
    L = np.array([[1.0, 0], [0.5, 2.0]])
    m = np.broadcast_to(np.array([1.0, -1.0]), (200000, 2))
    s = mvn.rvs(7, m, L)
    npt.assert_allclose(s.mean(0), [1.0, -1.0], atol=2e-2)
    npt.assert_allclose(np.cov(s.T), L @ L.T, atol=5e-2)

"""Pin the NumPy oracle (oracle/kalman_np.py) to the reference's own known-answer tests.

Mirrors aux_samplers/_primitives/test_kalman/test_filtering.py::test_vs_explicit (:20-55),
::test_batched_model (:58-107) and test_sampling.py (:23-127), with the explicit filter / RTS smoother
as the independent answer, on the reference's own seeded inputs.  CPU only.
"""
import numpy as np
import numpy.testing as npt
import pytest
from scipy.linalg import block_diag

from oracle import kalman_np as K
from tests.helpers import ref_lgssm_inputs, ref_batched_inputs


@pytest.mark.parametrize("seed", [0, 1234])
@pytest.mark.parametrize("T", [5, 7])
@pytest.mark.parametrize("dx", [1, 2])
@pytest.mark.parametrize("dy", [1, 3])
@pytest.mark.parametrize("parallel", [False, True])
@pytest.mark.parametrize("nan_index", [True, False])
def test_vs_explicit(seed, T, dx, dy, parallel, nan_index):
    # the reference only runs parallel=False with NaNs (test_filtering.py:24); we also pin the parallel path
    ys, lg = ref_lgssm_inputs(seed, T, dx, dy, nan_index)
    ms, Ps, ell = K.filtering(ys, lg, parallel)
    m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
    ems, ePs, eell = K.explicit_filter(ys, m0, P0, Hs, Rs, cs, Fs, Qs, bs)
    tol = dict(rtol=1e-7) if not parallel else dict(rtol=1e-6, atol=1e-9)
    npt.assert_allclose(ms, ems, **tol)
    npt.assert_allclose(Ps, ePs, **tol)
    npt.assert_allclose(ell, eell, **tol)


@pytest.mark.parametrize("seed", [0, 1234])
@pytest.mark.parametrize("T", [3, 5])
@pytest.mark.parametrize("dx", [1, 2])
@pytest.mark.parametrize("dy", [1, 3])
@pytest.mark.parametrize("parallel", [True, False])
def test_batched_model(seed, T, dx, dy, parallel):
    B = 3
    (bys, blg), (ys, lg) = ref_batched_inputs(seed, T, dx, dy, B)
    bms, bPs, bell = K.filtering(bys, blg, parallel)
    ms, Ps, ell = K.filtering(ys, lg, parallel)
    m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
    ems, ePs, eell = K.explicit_filter(ys, m0, P0, Hs, Rs, cs, Fs, Qs, bs)
    tol = dict(rtol=1e-6, atol=1e-9)
    npt.assert_allclose(ms, ems, **tol)
    npt.assert_allclose(Ps, ePs, **tol)
    npt.assert_allclose(ell, eell, **tol)
    npt.assert_allclose(np.reshape(bms, (T, B * dx)), ems, **tol)
    npt.assert_allclose(np.stack([block_diag(*p) for p in bPs]), ePs, **tol)
    npt.assert_allclose(bell, eell, **tol)


@pytest.mark.parametrize("seed", [42, 666])
@pytest.mark.parametrize("T", [3, 5])
@pytest.mark.parametrize("dx", [1, 2])
@pytest.mark.parametrize("dy", [1, 3])
def test_sampler_is_affine_in_noise_and_matches_smoother(seed, T, dx, dy):
    """test_sampling.py:23-68 estimates mean/cov from 500k draws.  The sampler is affine in eps, so the
    same two moments are available exactly: mean = sampling(eps=0); cov_t = J_t J_t^T with J the
    Jacobian w.r.t. eps.  Both must equal the RTS smoother."""
    ys, lg = ref_lgssm_inputs(seed, T, dx, dy)
    ms, Ps, _ = K.filtering(ys, lg, False)
    esm, esP = K.explicit_smoother(ms, Ps, lg[2], lg[3], lg[4])
    for parallel in (True, False):
        mean = K.sampling(np.zeros((T, dx)), ms, Ps, lg, parallel)
        npt.assert_allclose(mean, esm, rtol=1e-8, atol=1e-10)
        J = np.zeros((T, dx, T * dx))
        for k in range(T * dx):
            e = np.zeros(T * dx)
            e[k] = 1.0
            J[:, :, k] = K.sampling(e.reshape(T, dx), ms, Ps, lg, parallel) - mean
        cov = np.einsum("tik,tjk->tij", J, J)
        npt.assert_allclose(cov, esP, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("seed", [42, 666])
@pytest.mark.parametrize("mode", [True, False])
def test_sampler_batched_equals_block_diag(seed, mode):
    # test_sampling.py:71-127 : batched and block-diagonal dense models give the same samples
    T, dx, dy, B = 5, 2, 3, 3
    (bys, blg), (ys, lg) = ref_batched_inputs(seed, T, dx, dy, B)
    bms, bPs, _ = K.filtering(bys, blg, False)
    ms, Ps, _ = K.filtering(ys, lg, False)
    rng = np.random.default_rng(seed)
    for _ in range(5):
        eps = rng.standard_normal((T, B, dx))
        s = K.sampling(eps.reshape(T, B * dx), ms, Ps, lg, mode)
        bsamp = K.sampling(eps, bms, bPs, blg, mode)
        npt.assert_allclose(s, bsamp.reshape(T, B * dx), atol=1e-10, rtol=1e-10)


def test_parallel_equals_sequential_everything():
    ys, lg = ref_lgssm_inputs(3, 33, 2, 3, nan_index=True)
    a = K.filtering(ys, lg, True)
    b = K.filtering(ys, lg, False)
    for u, v in zip(a, b):
        npt.assert_allclose(u, v, rtol=1e-8, atol=1e-10)
    eps = np.random.default_rng(0).standard_normal((33, 2))
    npt.assert_allclose(K.sampling(eps, a[0], a[1], lg, True), K.sampling(eps, a[0], a[1], lg, False),
                        rtol=1e-8, atol=1e-10)


def test_mvn_logpdf_doctest():
    # aux_samplers/_primitives/math/mvn/base.py:37-45
    from scipy.stats import multivariate_normal
    z, mu = np.array([1., 2, 3]), np.array([2., 3, 4])
    L = np.array([[1, 0, 0], [0.2, 1.3, 0], [0.123, -0.5, 1.7]])
    npt.assert_allclose(K.mvn_logpdf(z, mu, L), multivariate_normal.logpdf(z, mu, L @ L.T))


def test_posterior_logpdf_is_a_normalised_density_ratio():
    """posterior_logpdf(x) = log p(x, y) - log p(y): check against the joint Gaussian written out densely."""
    from scipy.stats import multivariate_normal
    T, dx, dy = 4, 2, 3
    ys, lg = ref_lgssm_inputs(7, T, dx, dy)
    m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
    ms, Ps, ell = K.filtering(ys, lg, False)
    # dense joint of (x_0..x_{T-1}): mean / cov by forward recursion
    mean = np.zeros(T * dx)
    cov = np.zeros((T * dx, T * dx))
    mean[:dx] = m0
    cov[:dx, :dx] = P0
    for t in range(1, T):
        s, p = slice(t * dx, (t + 1) * dx), slice((t - 1) * dx, t * dx)
        mean[s] = Fs[t - 1] @ mean[p] + bs[t - 1]
        cov[s, :t * dx] = Fs[t - 1] @ cov[p, :t * dx]
        cov[:t * dx, s] = cov[s, :t * dx].T
        cov[s, s] = Fs[t - 1] @ cov[p, p] @ Fs[t - 1].T + Qs[t - 1]
    x = np.random.default_rng(1).standard_normal((T, dx))
    log_prior = multivariate_normal.logpdf(x.ravel(), mean, cov)
    log_lik = sum(multivariate_normal.logpdf(ys[t], Hs[t] @ x[t] + cs[t], Rs[t]) for t in range(T))
    npt.assert_allclose(K.posterior_logpdf(ys, x, ell, lg), log_prior + log_lik - ell, rtol=1e-9)
    # and ell itself is the marginal likelihood of y
    Hbig = block_diag(*Hs)
    my = Hbig @ mean + cs.ravel()
    Sy = Hbig @ cov @ Hbig.T + block_diag(*Rs)
    npt.assert_allclose(ell, multivariate_normal.logpdf(ys.ravel(), my, Sy), rtol=1e-9)


@pytest.mark.parametrize("seed,T,dx", [(42, 3, 1), (666, 5, 2), (7, 6, 3), (8, 9, 2)])
def test_dnc_sampler_restatement_targets_the_smoother(seed, T, dx):
    """The divide-and-conquer sampler (dnc_sampling.py:17-186, restated in oracle/kalman_np.py::dnc_sampling) draws from the same joint smoothing distribution as the
    reference's other two samplers: test_sampling.py:23-68 checks all three modes against `explicit_kalman_smoothing` (means and per-step covariances, atol 1e-2 over
    500 000 draws).  The restatement is affine in its noise given the tree, x = x(0) + A eps, so its law is known EXACTLY from T dx + 1 evaluations: x(0) must be the smoothed
    mean, the diagonal blocks of A A^T the smoothed covariances, and the whole of A A^T (cross-time blocks too, which a wrong mid-point conditional would break while
    leaving the marginals alone) the sequential sampler's joint covariance."""
    from tests.helpers import ref_lgssm_inputs
    ys, lg = ref_lgssm_inputs(seed, T, dx, 2)
    ms, Ps, _ = K.filtering(ys, lg, False)
    sm, sP = K.explicit_smoother(ms, Ps, lg[2], lg[3], lg[4])
    x0 = K.dnc_sampling(np.zeros((T, dx)), ms, Ps, lg)
    A = np.stack([K.dnc_sampling(np.eye(T * dx)[k].reshape(T, dx), ms, Ps, lg) - x0 for k in range(T * dx)], axis=-1).reshape(T * dx, T * dx)
    npt.assert_allclose(x0, sm, rtol=1e-9, atol=1e-10)                      # eps = 0 gives the smoothed mean exactly
    cov = A @ A.T                                                            # the sampler's exact joint covariance
    for t in range(T):
        npt.assert_allclose(cov[t * dx:(t + 1) * dx, t * dx:(t + 1) * dx], sP[t], rtol=1e-8, atol=1e-10)
    # joint law = the sequential sampler's (sampling.py:34-57): same affine-in-noise structure, same covariance
    y0 = K.sampling(np.zeros((T, dx)), ms, Ps, lg, False)
    B = np.stack([K.sampling(np.eye(T * dx)[k].reshape(T, dx), ms, Ps, lg, False) - y0 for k in range(T * dx)], axis=-1).reshape(T * dx, T * dx)
    npt.assert_allclose(cov, B @ B.T, rtol=1e-8, atol=1e-10)
    npt.assert_allclose(x0, y0, rtol=1e-9, atol=1e-10)

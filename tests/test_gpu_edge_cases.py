"""Edge cases of the boundary on the GPU: shortest series, all / mostly missing observations (the Lorenz pattern of
examples/lorenz/auxiliary_kalman.py:30-35: p = d + p_obs with the real observations NaN on most steps), both nan policies,
argument errors surfacing as ValueError exactly like the reference's construction-time checks."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-8, atol=1e-10)


def _rand_lgssm(rng, T, d, p):
    Fs = 0.6 * rng.standard_normal((T - 1, d, d)) / np.sqrt(d)
    A = rng.standard_normal((T - 1, d, 2 * d))
    Qs = A @ A.transpose(0, 2, 1) / (2 * d) + 0.2 * np.eye(d)
    bs = rng.standard_normal((T - 1, d))
    Hs = rng.standard_normal((T, p, d))
    B = rng.standard_normal((T, p, 2 * p))
    Rs = B @ B.transpose(0, 2, 1) / (2 * p) + 0.2 * np.eye(p)
    cs = rng.standard_normal((T, p))
    return (rng.standard_normal(d), np.eye(d) * 1.5, Fs, Qs, bs, Hs, Rs, cs)


@pytest.mark.parametrize("T", [1, 2, 3])
@pytest.mark.parametrize("d,p", [(1, 1), (2, 3), (4, 8)])
@pytest.mark.parametrize("parallel", [True, False])
def test_shortest_series(T, d, p, parallel):
    import aux_ssm_samplers_amd._primitives.kalman as P
    rng = np.random.default_rng(T * 100 + d)
    lg = _rand_lgssm(rng, T, d, p)
    ys = rng.standard_normal((T, p))
    ms, Ps, ell = P.filtering(ys, P.LGSSM(*lg), parallel)
    oms, oPs, oell = K.filtering(ys, lg, parallel)
    npt.assert_allclose(ms, oms, **TOL)
    npt.assert_allclose(Ps, oPs, **TOL)
    npt.assert_allclose(ell, oell, **TOL)
    eps = rng.standard_normal((T, d))
    npt.assert_allclose(P.sampling(None, oms, oPs, P.LGSSM(*lg), parallel, eps=eps), K.sampling(eps, oms, oPs, lg, parallel), **TOL)
    xs = rng.standard_normal((T, d))
    npt.assert_allclose(P.posterior_logpdf(ys, xs, oell, P.LGSSM(*lg)), K.posterior_logpdf(ys, xs, oell, lg), **TOL)


@pytest.mark.parametrize("parallel", [True, False])
def test_lorenz_like_sparse_observations(parallel):
    """d = 3, p = 5: auxiliary block always observed, the 2 real components observed every 20th step only (NaN elsewhere),
    a fully missing step, and an all-missing stretch of the real block.  Filter, sampler and both nan policies."""
    import aux_ssm_samplers_amd._primitives.kalman as P
    T, d, p = 400, 3, 5
    rng = np.random.default_rng(8)
    lg = _rand_lgssm(rng, T, d, p)
    ys = rng.standard_normal((T, p))
    ys[:, 3:] = np.nan
    ys[::20, 3:] = rng.standard_normal((len(range(0, T, 20)), 2))
    ys[0] = rng.standard_normal(p)
    ys[7] = np.nan                       # a completely missing step
    ms, Ps, ell = P.filtering(ys, P.LGSSM(*lg), parallel)
    oms, oPs, oell = K.filtering(ys, lg, parallel)
    npt.assert_allclose(ms, oms, **TOL)
    npt.assert_allclose(Ps, oPs, **TOL)
    npt.assert_allclose(ell, oell, **TOL)
    xs = rng.standard_normal((T, d))
    # reference policy: a partially observed step is dropped by nansum ...
    npt.assert_allclose(P.posterior_logpdf(ys, xs, oell, P.LGSSM(*lg)), K.posterior_logpdf(ys, xs, oell, lg), **TOL)
    # ... masked policy scores its finite components: equals the reference formula on the observed sub-vectors
    want = K.prior_logpdf(xs, lg) - oell
    from scipy.stats import multivariate_normal
    for t in range(T):
        k = np.isfinite(ys[t])
        if k.any():
            want += multivariate_normal.logpdf(ys[t][k], (lg[5][t] @ xs[t] + lg[7][t])[k], lg[6][t][np.ix_(k, k)])
    npt.assert_allclose(P.posterior_logpdf(ys, xs, oell, P.LGSSM(*lg), nan_policy="masked"), want, rtol=1e-8)


def test_all_observations_missing_is_the_prior():
    import aux_ssm_samplers_amd._primitives.kalman as P
    T, d, p = 50, 2, 2
    rng = np.random.default_rng(1)
    lg = _rand_lgssm(rng, T, d, p)
    ys = np.full((T, p), np.nan)
    ys[0] = rng.standard_normal(p)  # the reference's t = 0 update needs one finite y (filtering.py:52)
    for parallel in (True, False):
        ms, Ps, ell = P.filtering(ys, P.LGSSM(*lg), parallel)
        oms, oPs, oell = K.filtering(ys, lg, parallel)
        npt.assert_allclose(ms, oms, **TOL)
        npt.assert_allclose(Ps, oPs, **TOL)
        npt.assert_allclose(ell, oell, **TOL)


def test_argument_errors_are_value_errors():
    import aux_ssm_samplers_amd._primitives.kalman as P
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics, FlatPotential
    rng = np.random.default_rng(0)
    lg80 = _rand_lgssm(rng, 3, 80, 2)  # fp64 dx = 80 exceeds the LDS plan of the wide-state path: loud error, no fallback
    with pytest.raises(ValueError, match="LDS"):
        P.filtering(rng.standard_normal((3, 2)), P.LGSSM(*lg80), True)
    lg = _rand_lgssm(rng, 4, 2, 2)
    with pytest.raises(ValueError):
        P.filtering(rng.standard_normal((5, 2)), P.LGSSM(*lg), True)  # T mismatch
    M0 = GaussianInit(m0=[0.0], P0=[[1.0]])
    Mt = LinearGaussianDynamics(F=[[0.5]], b=[0.0], Q=[[1.0]])
    fk = _device.describe_bootstrap(M0, FlatPotential(), Mt, FlatPotential(), None)
    with pytest.raises(ValueError):
        _device.sweep(fk, np.zeros((3, 1)), 2048, False, key=0)  # N > 1024
    with pytest.raises(ValueError):
        _device.sweep(fk, np.zeros((3, 2)), 8, False, key=0)     # state dimension mismatch


@pytest.mark.parametrize("N", [2, 3, 64, 65, 1023, 1024])
def test_csmc_particle_count_edges(N):
    from oracle import csmc as O
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics, GaussianObsPotential
    T = 9
    rng = np.random.default_rng(N)
    y = rng.standard_normal((T, 1))
    M0 = GaussianInit(m0=[0.0], P0=[[1.0]])
    Mt = LinearGaussianDynamics(F=[[0.9]], b=[0.0], Q=[[0.19]])
    fk = _device.describe_bootstrap(M0, GaussianObsPotential(sig=0.5, y=y[0]), Mt, GaussianObsPotential(sig=0.5, params=y[1:]), Mt)
    x0 = rng.standard_normal((T, 1)).astype(np.float32)
    noise = dict(eps_prop=rng.standard_normal((1, T, N, 1)).astype(np.float32), u_res=rng.random((1, T - 1, N)).astype(np.float32),
                 u_bwd=rng.random((1, T)).astype(np.float32))
    for backward in (True, False):
        x, anc, hist = _device.sweep(fk, x0, N, backward, noise=noise, want_history=True)
        ref = O.sweep(dict(proposal=0, potential=1, m0=[0.0], chol_P0=[[1.0]], F=[[0.9]], b=[0.0], chol_Q=[[np.sqrt(0.19)]], sig_y=0.5),
                      x0, N, backward, y=y, eps_prop=noise["eps_prop"][0], u_res=noise["u_res"][0], u_bwd=noise["u_bwd"][0])
        npt.assert_array_equal(hist["As"], ref["As"])
        npt.assert_array_equal(anc, ref["ancestors"])
        npt.assert_array_equal(x, ref["x"])

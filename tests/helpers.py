"""Shared input recipes for the tests.

`ref_lgssm_inputs` / `ref_batched_inputs` reproduce the *inputs* of the reference's own known-answer
tests (aux_samplers/_primitives/test_kalman/test_filtering.py:20-47, 58-95 and test_sampling.py:23-50,
71-115): the legacy ``np.random.seed`` MT19937 stream is stable across NumPy versions, so drawing in
the same order gives bit-identical arrays without JAX.
"""
import numpy as np
from scipy.linalg import block_diag  # noqa: F401


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


# the workloads of the BASELINE configurations live in the package (bench.py and tools/ import them from there, not from the test package)
from aux_ssm_samplers_amd.workloads import rot, lg_model, sv_setup, lorenz_kalman_setup, lorenz_setup, c5_model, c5_batched_model  # noqa: E402,F401


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

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

"""Statistical / Taylor linearisation of a conditional mean and covariance (reference: aux_samplers/_primitives/linearisation.py).

    extended(mean, cov, params, x_star, P_star)          first-order Taylor expansion at x_star        (:11-44)
    cubature(mean, cov, params, x_star, P_star)          third-degree spherical cubature rule         (:77-102, :219-241)
    gauss_hermite(mean, cov, params, x_star, P_star, order=3)   tensor Gauss-Hermite rule             (:47-74, :127-190)

each returning (F, Q, b) with  E[f(X)] ~ F x + b  and residual covariance Q -- the (Fs, Qs, bs) a `dynamics_factory` hands to
kalman.get_kernel.  These are HOST helpers (NumPy): they produce the inputs of the hot path once per sweep on the host-factory path
(kalman/generic.py); the example models of the reference have closed-form device factories instead (kalman/models.py).

The reference differentiates `mean` with jax.jacfwd / jax.jacrev.  Without autodiff `extended` takes the Jacobian as `jac(x, params)`
or, by default, central differences with a step scaled to x_star (exact on linear maps up to rounding).
"""
import itertools
import math

import numpy as np


def _jacobian_fd(mean, params, x):
    x = np.asarray(x, np.float64)
    h = np.cbrt(np.finfo(np.float64).eps) * np.maximum(1.0, np.abs(x))
    cols = []
    for k in range(x.shape[0]):
        e = np.zeros_like(x)
        e[k] = h[k]
        cols.append((np.asarray(mean(x + e, params), np.float64) - np.asarray(mean(x - e, params), np.float64)) / (2.0 * h[k]))
    return np.stack(cols, axis=1)


def extended(mean, cov, params, x_star, _P_star=None, jac=None):
    """F = d mean / dx at x_star, Q = cov(x_star), b = mean(x_star) - F x_star."""
    x_star = np.asarray(x_star)
    m = np.asarray(mean(x_star, params))
    F = np.asarray(jac(x_star, params)) if jac is not None else _jacobian_fd(mean, params, x_star)
    return F, np.asarray(cov(x_star, params)), m - F @ x_star


def _cubature_rule(dim):
    """2 dim points +- sqrt(dim) e_i with equal weights"""
    pts = math.sqrt(dim) * np.concatenate([np.eye(dim), -np.eye(dim)], axis=0)
    return np.full(2 * dim, 0.5 / dim), pts


def _gauss_hermite_rule(dim, order):
    """tensor product of the `order`-point rule for N(0, 1): nodes sqrt(2) x_i, weights w_i / sqrt(pi) of the physicists' rule"""
    x1, w1 = np.polynomial.hermite.hermgauss(order)
    x1, w1 = math.sqrt(2.0) * x1, w1 / math.sqrt(math.pi)
    idx = np.array(list(itertools.product(range(order), repeat=dim)))
    return np.prod(w1[idx], axis=1), x1[idx]


def _sigma_point_linearisation(mean, cov, params, x_star, P_star, rule):
    x_star = np.asarray(x_star, np.float64)
    L = np.linalg.cholesky(np.asarray(P_star, np.float64))
    w, xi = rule(x_star.shape[0])
    pts = x_star[None, :] + xi @ L.T
    f = np.stack([np.asarray(mean(p, params), np.float64) for p in pts])
    m_f = w @ f
    dx, df = pts - x_star[None, :], f - m_f[None, :]
    Psi = (dx * w[:, None]).T @ df                      # Cov[X, f(X)]
    F = np.linalg.solve(L @ L.T, Psi).T                 # Cov[f(X), X] P^-1
    V = np.tensordot(w, np.stack([np.asarray(cov(p, params), np.float64) for p in pts]), axes=1)
    Phi = (df * w[:, None]).T @ df                      # Cov[f(X)]
    FL = F @ L
    return F, Phi - FL @ FL.T + V, m_f - F @ x_star


def cubature(mean, cov, params, x_star, P_star):
    return _sigma_point_linearisation(mean, cov, params, x_star, P_star, _cubature_rule)


def gauss_hermite(mean, cov, params, x_star, P_star, order=3):
    return _sigma_point_linearisation(mean, cov, params, x_star, P_star, lambda dim: _gauss_hermite_rule(dim, order))

"""Statistical / Taylor linearisation of a conditional mean and covariance (reference: aux_samplers/_primitives/linearisation.py).

    extended(mean, cov, params, x_star, P_star)          first-order Taylor expansion at x_star        (:11-44)
    cubature(mean, cov, params, x_star, P_star)          third-degree spherical cubature rule         (:77-102, :219-241)
    gauss_hermite(mean, cov, params, x_star, P_star, order=3)   tensor Gauss-Hermite rule             (:47-74, :127-190)

each returning (F, Q, b) with  E[f(X)] ~ F x + b  and residual covariance Q -- the (Fs, Qs, bs) a `dynamics_factory` hands to
kalman.get_kernel.  Two execution paths:
  * `mean` one of the DEVICE means below (`AffineMean`, `Lorenz63Mean`) with a `ConstantCov`: the linearisation of every row of a batched
    `x_star` (n, d) -- the reference's `jax.vmap` over the trajectory -- is one launch of auxssm_linearise (one lane per point, the sigma points
    enumerated in registers); pass `device_out=True` to keep (F, Q, b) in HBM;
  * arbitrary Python callables: HOST NumPy (they cannot run on the device), feeding the host-factory path of kalman/generic.py.

The reference differentiates `mean` with jax.jacfwd / jax.jacrev.  Without autodiff `extended` takes the Jacobian as `jac(x, params)`
or, by default, central differences with a step scaled to x_star (exact on linear maps up to rounding).
"""
import itertools
import math

import numpy as np

from .. import _lib

LIN_EXTENDED, LIN_CUBATURE, LIN_GAUSS_HERMITE = 0, 1, 2
FN_AFFINE, FN_LORENZ63 = 0, 1


class ConstantCov:
    """cov(x, params) = Q"""

    def __init__(self, Q):
        self.Q = np.asarray(Q)

    def __call__(self, _x=None, _params=None):
        return self.Q


class AffineMean:
    """mean(x, params) = A x + a, A (dy, dx): the map of the reference's own test of the three methods (test_linearisation.py:13-48)"""
    kind = FN_AFFINE

    def __init__(self, A, a):
        self.A, self.a = np.asarray(A), np.asarray(a)
        self.dim, self.dim_out = self.A.shape[1], self.A.shape[0]

    def __call__(self, x, _params=None):
        return self.A @ x + self.a

    def jac(self, _x, _params=None):
        return self.A

    def device_params(self, dtype):
        return np.ascontiguousarray(self.A, dtype), np.ascontiguousarray(self.a, dtype)


class Lorenz63Mean:
    """mean(x, params) = x + dt (phi_0(x) + theta * phi(x)) (examples/lorenz/model.py:10-25), the Euler step of the Lorenz-63 drift"""
    kind = FN_LORENZ63
    dim = dim_out = 3

    def __init__(self, theta, dt):
        self.theta, self.dt = np.asarray(theta, np.float64), float(dt)

    def __call__(self, x, _params=None):
        th, dt = self.theta, self.dt
        return np.array([x[0] + dt * (th[0] * (x[1] - x[0])), x[1] + dt * (th[1] * x[0] - x[1] - x[0] * x[2]), x[2] + dt * (x[0] * x[1] - th[2] * x[2])])

    def jac(self, x, _params=None):
        th, dt = self.theta, self.dt
        return np.eye(3) + dt * np.array([[-th[0], th[0], 0.0], [th[1] - x[2], -1.0, -x[0]], [x[1], x[0], -th[2]]])

    def device_params(self, dtype):
        return np.ascontiguousarray(np.concatenate([self.theta, [self.dt]]), dtype), None


def _on_device(mean, cov):
    return isinstance(mean, (AffineMean, Lorenz63Mean)) and isinstance(cov, ConstantCov)


def _device_linearise(method, mean, cov, x_star, P_star, order, handle, device_out):
    handle = handle or (x_star.handle if isinstance(x_star, _lib.DeviceArray) else _lib.default_handle())
    if isinstance(x_star, _lib.DeviceArray):
        xd, dtype, xshape = x_star, x_star.dtype, x_star.shape
    else:
        xh = np.asarray(x_star)
        dtype = np.dtype(np.float32) if xh.dtype == np.float32 else np.dtype(np.float64)
        xshape, xd = xh.shape, handle.to_device(np.ascontiguousarray(xh, dtype))
    dx, dy = mean.dim, mean.dim_out
    if xshape[-1] != dx or len(xshape) > 2:
        raise ValueError(f"x_star must be ({dx},) or (n, {dx}), got {xshape}")
    n = xshape[0] if len(xshape) == 2 else 1
    A, a = mean.device_params(dtype)
    Ad, ad = handle.to_device(A), (handle.to_device(a) if a is not None else None)
    Qd = handle.to_device(np.ascontiguousarray(cov.Q, dtype).reshape(dy, dy))
    Pd, sP = None, 0
    if method != LIN_EXTENDED:
        if isinstance(P_star, _lib.DeviceArray):
            Pd, pshape = P_star, P_star.shape
        else:
            Ph = np.ascontiguousarray(P_star, dtype)
            Pd, pshape = handle.to_device(Ph), Ph.shape
        if pshape not in ((dx, dx), (n, dx, dx)):
            raise ValueError(f"P_star must be ({dx}, {dx}) or ({n}, {dx}, {dx}), got {pshape}")
        sP = dx * dx if len(pshape) == 3 and len(xshape) == 2 else 0
    nodes = weights = None
    if method == LIN_GAUSS_HERMITE:
        x1, w1 = np.polynomial.hermite.hermgauss(order)
        nodes, weights = np.ascontiguousarray(math.sqrt(2.0) * x1), np.ascontiguousarray(w1 / math.sqrt(math.pi))
    F, Q, b = handle.empty((n, dy, dx), dtype), handle.empty((n, dy, dy), dtype), handle.empty((n, dy), dtype)
    import ctypes as C
    _lib.check(handle.lib.auxssm_linearise(handle.h, _lib.dtype_code(dtype), method, int(order), mean.kind, n, dx, dy, Ad.ptr, ad.ptr if ad is not None else None,
                                           Qd.ptr, nodes.ctypes.data_as(C.c_void_p) if nodes is not None else None,
                                           weights.ctypes.data_as(C.c_void_p) if weights is not None else None, xd.ptr,
                                           Pd.ptr if Pd is not None else None, sP, F.ptr, Q.ptr, b.ptr))
    if device_out:
        return F, Q, b
    F, Q, b = F.to_host(), Q.to_host(), b.to_host()
    return (F, Q, b) if len(xshape) == 2 else (F[0], Q[0], b[0])


def _jacobian_fd(mean, params, x):
    x = np.asarray(x, np.float64)
    h = np.cbrt(np.finfo(np.float64).eps) * np.maximum(1.0, np.abs(x))
    cols = []
    for k in range(x.shape[0]):
        e = np.zeros_like(x)
        e[k] = h[k]
        cols.append((np.asarray(mean(x + e, params), np.float64) - np.asarray(mean(x - e, params), np.float64)) / (2.0 * h[k]))
    return np.stack(cols, axis=1)


def extended(mean, cov, params, x_star, _P_star=None, jac=None, handle=None, device_out=False):
    """F = d mean / dx at x_star, Q = cov(x_star), b = mean(x_star) - F x_star."""
    if _on_device(mean, cov):
        return _device_linearise(LIN_EXTENDED, mean, cov, x_star, None, 0, handle, device_out)
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


def cubature(mean, cov, params, x_star, P_star, handle=None, device_out=False):
    if _on_device(mean, cov):
        return _device_linearise(LIN_CUBATURE, mean, cov, x_star, P_star, 0, handle, device_out)
    return _sigma_point_linearisation(mean, cov, params, x_star, P_star, _cubature_rule)


def gauss_hermite(mean, cov, params, x_star, P_star, order=3, handle=None, device_out=False):
    if _on_device(mean, cov):
        return _device_linearise(LIN_GAUSS_HERMITE, mean, cov, x_star, P_star, order, handle, device_out)
    return _sigma_point_linearisation(mean, cov, params, x_star, P_star, lambda dim: _gauss_hermite_rule(dim, order))

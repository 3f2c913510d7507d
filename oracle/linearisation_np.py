"""CPU restatement (NumPy) of the reference's linearisation methods -- TEST INFRASTRUCTURE ONLY (tests/).

Follows, as text, aux_samplers/_primitives/linearisation.py:
  * `extended`                 :11-44   (the Jacobian is passed in: there is no autodiff here)
  * `gauss_hermite`, `cubature` :47-104 on `_generic_sigma_points` :107-127 and `_cov` :130-133
  * `_gauss_hermite_points`    :136-189 with `_hermite_coeff` :192-217 (the physicists' polynomials by their recurrence, roots by np.roots,
                                         the roll table), `_cubature_points` :220-241

Parity pin: the reference's own known-answer test, test_linearisation.py:13-48 -- an affine map R^4 -> R^2 is recovered exactly by all three
methods -- reproduced on this file by tests/test_linearisation.py (and on the device kernel, which is then compared with this file on the
Lorenz-63 step).
"""
import math

import numpy as np
from scipy.linalg import cho_solve


def extended(mean, cov, params, x_star, _P_star, jac):
    b = mean(x_star, params)
    F = jac(x_star, params)
    Q = cov(x_star, params)
    return F, Q, b - F @ x_star


def _cov(wc, x_pts, x_mean, y_points, y_mean):
    one = (x_pts - x_mean[None, :]).T * wc[None, :]
    two = y_points - y_mean[None, :]
    return one @ two


def _generic_sigma_points(mean, cov, params, x_star, P_star, get_sigma_points):
    chol = np.linalg.cholesky(P_star)
    dim = x_star.shape[0]
    w, xi = get_sigma_points(dim)
    points = x_star[None, :] + (chol @ xi).T
    f_pts = np.stack([mean(p, params) for p in points])
    m_f = w @ f_pts
    Psi_x = _cov(w, points, x_star, f_pts, m_f)
    F_x = cho_solve((chol, True), Psi_x).T
    v_pts = np.stack([cov(p, params) for p in points])
    v_f = np.sum(w[:, None, None] * v_pts, 0)
    Phi = _cov(w, f_pts, m_f, f_pts, m_f)
    temp = F_x @ chol
    L = Phi - temp @ temp.T + v_f
    return F_x, L, m_f - F_x @ x_star


def _hermite_coeff(order):
    H = [np.array([1]), np.array([2, 0])]
    for i in range(2, order + 1):
        H.append(2 * np.append(H[i - 1], 0) - 2 * (i - 1) * np.pad(H[i - 2], (2, 0), "constant", constant_values=0))
    return H


def gauss_hermite_points(n_dim, order=3):
    n, p = n_dim, order
    hermite_coeff = _hermite_coeff(p)
    hermite_roots = np.flip(np.roots(hermite_coeff[-1]))
    table = np.zeros(shape=(n, p ** n))
    w_1d = np.zeros(shape=(p,))
    for i in range(p):
        w_1d[i] = (2 ** (p - 1) * math.factorial(p) * np.sqrt(np.pi) / (p ** 2 * (np.polyval(hermite_coeff[p - 1], hermite_roots[i])) ** 2))
    for i in range(n):
        base = np.ones(shape=(1, p ** (n - i - 1)))
        for j in range(1, p):
            base = np.concatenate([base, (j + 1) * np.ones(shape=(1, p ** (n - i - 1)))], axis=1)
        table[n - i - 1, :] = np.tile(base, (1, int(p ** i)))
    table = table.astype("int64") - 1
    s = 1 / (np.sqrt(np.pi) ** n)
    w = s * np.prod(w_1d[table], axis=0)
    xi = math.sqrt(2) * hermite_roots[table]
    return w, xi


def cubature_points(n_dim):
    w = np.ones(shape=(2 * n_dim,)) / (2 * n_dim)
    xi = np.concatenate([np.eye(n_dim), -np.eye(n_dim)], axis=0) * np.sqrt(n_dim)
    return w, xi.T


def gauss_hermite(mean, cov, params, x_star, P_star, order=3):
    return _generic_sigma_points(mean, cov, params, x_star, P_star, lambda dim: gauss_hermite_points(dim, order))


def cubature(mean, cov, params, x_star, P_star):
    return _generic_sigma_points(mean, cov, params, x_star, P_star, cubature_points)

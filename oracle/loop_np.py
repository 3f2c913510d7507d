"""CPU restatement (NumPy) of the MCMC loop around the sweeps -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline).

Follows, as text, the reference's
  * `loop` body: examples/stochastic_volatility/experiment.py:88-128, examples/lorenz/experiment.py:120-169
      stats_fn                :81-83   (squared jump, first and second moments)
      moving averages         :110-113 (avg_acceptance, window_avg_acceptance, running means of the stats)
      step-size schedule      :115-117 (lr = (n_iter - i) * lr / n_iter)
  * delta_adaptation: aux_samplers/common.py:4-32
  * theta_posterior_mean_and_chol, phi, phi_0: examples/lorenz/model.py:10-17, :59-79; gibbs_step: experiment.py:106-115

Parity pin: none of these has a test in the reference ("parity unpinned"); they are closed-form elementwise expressions,
pinned here by tests/test_oracle_loop.py against hand-computed values and a brute-force Bayesian linear regression.
"""
import numpy as np


def stats_fn(x_1, x_2):
    return (x_2 - x_1) ** 2, x_2, x_2 ** 2


def fold(i, u, v):
    """tree_map(lambda u, v: (i * u + v) / (i + 1), stats, next_stats)"""
    return (i * u + v) / (i + 1)


def accept_update(i, beta, updated, avg, window):
    """updated: bool array.  -> (avg, window) after folding sweep i"""
    f = np.asarray(updated).astype(avg.dtype)
    return (i * avg + f) / (i + 1), avg.dtype.type(beta) * f + avg.dtype.type(1.0 - beta) * window


def delta_adaptation(delta, target_rate, acceptance_rate, adaptation_rate, min_delta=1e-20, max_delta=1e20):
    rate = np.exp(adaptation_rate * (acceptance_rate - target_rate))
    return np.clip(delta * rate, min_delta, max_delta)


def pooled_delta_adaptation(delta, target_rate, window, adaptation_rate, min_delta=1e-20, max_delta=1e20):
    """window (C, m): the chains of one device sweep share delta (m,), so the rule runs on the chain-mean of the windowed acceptance
    (C = 1: the reference's rule unchanged)"""
    return delta_adaptation(delta, target_rate, np.mean(np.asarray(window), axis=0), adaptation_rate, min_delta, max_delta)


def phi_0(x):
    x1, x2, x3 = x[..., 0], x[..., 1], x[..., 2]
    return np.stack([np.zeros_like(x1), -x2 - x1 * x3, x1 * x2], -1)


def phi(x):
    x1, x2, x3 = x[..., 0], x[..., 1], x[..., 2]
    return np.stack([x2 - x1, x1, -x3], -1)


def theta_posterior_mean_and_chol(x, sigma_theta, dt, sigma_x):
    """x (T, 3) -> (mean (3,), chol (3,)): three independent scalar regressions (model.py:59-79)"""
    x = np.asarray(x, np.float64)
    X = dt * phi(x[:-1])
    Y = (x[1:] - x[:-1]) - dt * phi_0(x[:-1])
    Sigma = 1.0 / (np.einsum("ij,ij->j", X, X) + 1.0 / sigma_theta ** 2)
    return Sigma * np.einsum("ij,ij->j", X, Y), sigma_x * dt ** 0.5 * Sigma ** 0.5


def loop(kernel_fn, x0, updated0, init_delta, n_iter, noises, *, delta_fn=None, target_alpha=None, lr=None, beta=0.01):
    """The reference's `loop` for ONE chain with explicit per-sweep noise: kernel_fn(x, delta, noise) -> (x_next, updated).
    Returns (n_iter, (sq_jump, mean, sq_mean), x, delta, window_avg_acceptance, avg_acceptance)."""
    x = np.asarray(x0)
    stats = stats_fn(x, x)
    upd0 = np.asarray(updated0)
    avg = upd0.astype(x.dtype) * np.ones_like(upd0, dtype=x.dtype)
    window = avg.copy()
    delta = init_delta
    for i in range(n_iter):
        x_next, updated = kernel_fn(x, delta, noises[i])
        nstats = stats_fn(x, x_next)
        avg, window = accept_update(i, beta, updated, avg, window)
        stats = tuple(fold(i, u, v) for u, v in zip(stats, nstats))
        if delta_fn is not None:
            delta = delta_fn(delta, target_alpha, window, (n_iter - i) * lr / n_iter)
        x = x_next
    return n_iter, stats, x, delta, window, avg

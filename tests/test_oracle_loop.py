"""CPU: the NumPy restatement of the MCMC loop pieces (oracle/loop_np.py) against hand-computed values and brute force."""
import numpy as np

from oracle import loop_np as L
from aux_ssm_samplers_amd.common import delta_adaptation


def test_running_means_are_plain_means():
    rng = np.random.default_rng(0)
    xs = rng.standard_normal((6, 5, 2))
    stats = L.stats_fn(xs[0], xs[0])
    for i in range(5):
        stats = tuple(L.fold(i, u, v) for u, v in zip(stats, L.stats_fn(xs[i], xs[i + 1])))
    np.testing.assert_allclose(stats[0], np.mean((xs[1:] - xs[:-1]) ** 2, 0), rtol=1e-13)
    np.testing.assert_allclose(stats[1], np.mean(xs[1:], 0), rtol=1e-13)
    np.testing.assert_allclose(stats[2], np.mean(xs[1:] ** 2, 0), rtol=1e-13)


def test_accept_update_hand_values():
    avg = np.array([1.0, 1.0])
    win = np.array([1.0, 1.0])
    avg, win = L.accept_update(0, 0.25, np.array([True, False]), avg, win)
    np.testing.assert_array_equal(avg, [1.0, 0.0])
    np.testing.assert_array_equal(win, [1.0, 0.75])
    avg, win = L.accept_update(1, 0.25, np.array([False, True]), avg, win)
    np.testing.assert_array_equal(avg, [0.5, 0.5])
    np.testing.assert_allclose(win, [0.75, 0.25 + 0.75 * 0.75])


def test_delta_adaptation_is_the_package_rule_and_pools_over_chains():
    d = L.delta_adaptation(0.5, 0.25, 0.4, 0.1)
    assert abs(d - 0.5 * np.exp(0.1 * 0.15)) < 1e-15
    assert abs(d - delta_adaptation(0.5, 0.25, 0.4, 0.1)) < 1e-15
    assert L.delta_adaptation(1e-3, 0.5, 0.0, 50.0, min_delta=1e-4) == 1e-4
    win = np.array([[0.2, 0.9], [0.4, 0.5]])
    np.testing.assert_allclose(L.pooled_delta_adaptation(np.array([1.0, 2.0]), 0.3, win, 0.5),
                               [np.exp(0.0), 2.0 * np.exp(0.5 * 0.4)])
    np.testing.assert_allclose(L.pooled_delta_adaptation(np.array([1.0]), 0.3, win[:1, :1], 0.5), L.delta_adaptation(1.0, 0.3, 0.2, 0.5))


def test_theta_posterior_against_grid_integration():
    # component k: Y_t = theta_k X_t + sigma_Y e_t, prior theta_k ~ N(0, (sigma_Y sigma_theta)^2)  (the scaling model.py:59-79 implies)
    rng = np.random.default_rng(1)
    T, dt, sx, sth = 40, 0.01, 2.0, 3.0
    x = np.cumsum(rng.standard_normal((T, 3)), 0) * 0.3 + np.array([1.0, -1.0, 20.0])
    mean, chol = L.theta_posterior_mean_and_chol(x, sth, dt, sx)
    X = dt * L.phi(x[:-1])
    Y = x[1:] - x[:-1] - dt * L.phi_0(x[:-1])
    sy = sx * np.sqrt(dt)
    for k in range(3):
        grid = np.linspace(mean[k] - 12 * chol[k], mean[k] + 12 * chol[k], 20001)
        lp = -0.5 * np.sum((Y[:, k][None] - grid[:, None] * X[:, k][None]) ** 2, 1) / sy ** 2 - 0.5 * grid ** 2 / (sy * sth) ** 2
        w = np.exp(lp - lp.max())
        w /= w.sum()
        m = np.sum(w * grid)
        v = np.sum(w * (grid - m) ** 2)
        assert abs(m - mean[k]) < 1e-6 * max(1.0, abs(m))
        assert abs(np.sqrt(v) - chol[k]) < 1e-5 * chol[k]


def test_loop_restatement_on_a_toy_kernel():
    # kernel: deterministic shift accepted on even sweeps
    def kernel(x, delta, noise):
        return (x + noise * delta, True) if noise > 0 else (x, False)

    x0 = np.zeros((3, 1))
    noises = [1.0, -1.0, 1.0, 1.0]
    n, stats, x, delta, win, avg = L.loop(kernel, x0, True, 0.5, 4, noises, delta_fn=L.delta_adaptation, target_alpha=0.5, lr=1.0, beta=0.5)
    assert n == 4 and avg == 0.75
    # window: 1 -> 1 -> .5 -> .75 -> .875 ; delta: .5 e^{1*(1-.5)} -> * e^{.75*(.5-.5)} -> * e^{.5*(.75-.5)} -> * e^{.25*(.875-.5)}
    assert abs(win - 0.875) < 1e-15
    d = 0.5 * np.exp(0.5)
    xs = [0.5]
    d *= np.exp(0.0)
    xs.append(xs[-1])
    xs.append(xs[-1] + d)
    d *= np.exp(0.5 * 0.25)
    xs.append(xs[-1] + d)
    d *= np.exp(0.25 * 0.375)
    assert abs(delta - d) < 1e-14
    np.testing.assert_allclose(x, xs[-1])
    np.testing.assert_allclose(stats[1], np.mean(xs))

"""The cSMC parity chain closed ON THE GPU: the HIP sweep next to `oracle/csmc_np.py`, the LITERAL NumPy restatement of the
reference's arithmetic order (`_primitives/csmc/csmc.py:69-149`: normalised weights, plain left-to-right cumsum, `searchsorted`, `resamplings.py:32-37`), with
no contract oracle in between.

* fp64, the C3 stochastic-volatility model (F = 0.9, Q = P0 = 10.526, N = 1024, independent auxiliary proposals) and the C4 Lorenz-63 model (Euler-Maruyama
  transition, (x2, x3) observed sparsely, N = 512), both backward modes: resampling ancestors, backward indices and trajectories IDENTICAL, particles and
  log-weights to rounding (the contract's deviations from the literal order -- DESIGN section 2 -- change no index in fp64).
* fp32, the dtype C3 / C4 run in: the two orders round the cumulative weights differently, so a draw within a few ulps of a boundary picks the neighbouring
  particle.  Measured teacher-forced on the DEVICE's own stored log-weights: rate <= 2e-4 per draw, every miss an adjacent particle (or one across a run of
  particles whose fp32 weight is zero)."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import csmc_np as L

from tests.helpers import lorenz_setup

pytestmark = pytest.mark.gpu

Q_SV = 2.0 / (1.0 - 0.9 ** 2)  # tau / (1 - phi^2) = 10.526 (examples/stochastic_volatility/model.py:34-53)


def _sv_c3(T, rng):
    """C3's model on both sides: device family objects and literal protocol objects; data simulated as model.py:11-31"""
    from aux_ssm_samplers_amd.csmc import GaussianInit, LinearGaussianDynamics, SVPotential
    x = np.zeros((T, 1))
    x[0] = np.sqrt(Q_SV) * rng.standard_normal(1)
    for t in range(1, T):
        x[t] = 0.9 * x[t - 1] + np.sqrt(Q_SV) * rng.standard_normal(1)
    y = np.exp(0.5 * x) * rng.standard_normal((T, 1))
    M0, Mt = GaussianInit(m0=[0.0], P0=[[Q_SV]]), LinearGaussianDynamics(F=[[0.9]], b=[0.0], Q=[[Q_SV]])
    dev = (M0, SVPotential(y=y[0]), Mt, SVPotential(params=y[1:]))
    LQ = np.array([[np.sqrt(Q_SV)]])
    lit = (L.GaussianInit(np.zeros(1), LQ), L.ObsPotential("sv", y[0], first=True), L.LinearGaussianDynamics(np.array([[0.9]]), np.zeros(1), LQ, T),
           L.ObsPotential("sv", y[1:]))
    return dev, lit, x, y


def _lorenz_c4(T, seed):
    M0, Mt, G0, Gt, xtrue, y, sig_y = lorenz_setup(T, seed=seed)
    LQ = np.asarray(Mt.chol())
    Mo = L.GaussianInit(np.asarray(M0.m0, float), np.asarray(M0.chol(), float))
    lit = (Mo, L.ObsPotential("masked", y[0], sig_y, first=True), L.Lorenz63EM(np.asarray(Mt.theta, float), Mt.dt, LQ, T), L.ObsPotential("masked", y[1:], sig_y))
    return (M0, G0, Mt, Gt), lit, xtrue, y


@pytest.mark.parametrize("backward", [True, False])
@pytest.mark.parametrize("model", ["sv_c3", "lorenz_c4"])
def test_hip_sweep_fp64_equals_the_literal_restatement(model, backward):
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(20260 + backward)
    if model == "sv_c3":
        T, N, d = 160, 1024, 1
        dev, lit, xtrue, y = _sv_c3(T, rng)
        delta = np.full(T, 0.5)
        x0 = xtrue + 0.3 * rng.standard_normal((T, d))
    else:
        T, N, d = 130, 512, 3
        dev, lit, xtrue, y = _lorenz_c4(T, 4)
        delta = 0.05 + 0.05 * rng.random(T)
        x0 = xtrue + 0.1 * rng.standard_normal((T, d))
    nz = dict(eps_aux=rng.standard_normal((T, d)), eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T))
    fk = _device.describe_independent(dev[0], dev[1], dev[2], dev[3], dev[2])
    x, anc, hist = _device.sweep(fk, x0, N, backward, noise={k: v[None] for k, v in nz.items()}, delta=delta, want_history=True)
    _, kern = L.get_independent_kernel(lit[0], lit[1], lit[2], lit[3], N, backward=backward, Pt=lit[2])
    xl, Bl, lh = kern(L.Noise(**nz), x0, delta)
    npt.assert_array_equal(hist["As"], lh["As"])         # resampling ancestors of every step: identical
    npt.assert_array_equal(anc, Bl)                       # backward indices (ancestor trace or backward sampling): identical
    npt.assert_allclose(x, xl, rtol=1e-12, atol=1e-12)    # the trajectory: the same particles picked, particles to rounding
    npt.assert_allclose(hist["xs"], lh["xs"], rtol=1e-12, atol=1e-12)
    npt.assert_allclose(hist["log_ws"], lh["log_ws"], rtol=1e-10, atol=1e-10)
    assert np.all(hist["As"][:, 0] == 0) and np.array_equal(hist["xs"][:, 0], x0)


def test_fp32_ancestors_against_the_literal_order_tie_rate():
    """device fp32 ancestors vs the literal fp32 resampling (normalise -> cumsum -> searchsorted) redone from the DEVICE's stored log-weights and the same
    uniforms, step by step (teacher-forced: a miss does not propagate)"""
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(77)
    T, N, d = 1600, 1024, 1
    dev, lit, xtrue, y = _sv_c3(T, rng)
    x0 = (xtrue + 0.3 * rng.standard_normal((T, d))).astype(np.float32)
    nz = dict(eps_aux=rng.standard_normal((T, d)), eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T))
    nz = {k: v.astype(np.float32) for k, v in nz.items()}
    fk = _device.describe_independent(dev[0], dev[1], dev[2], dev[3], dev[2])
    x, anc, hist = _device.sweep(fk, x0, N, False, noise={k: v[None] for k, v in nz.items()}, delta=0.5, want_history=True)
    assert hist["log_ws"].dtype == np.float32
    bad = far = 0
    for t in range(1, T):
        w = L.normalize(hist["log_ws"][t - 1])
        assert w.dtype == np.float32
        A = L.multinomial(nz["u_res"][t - 1], w)
        miss = np.nonzero(A != hist["As"][t - 1])[0]
        bad += len(miss)
        for i in miss:
            a, b = sorted((int(A[i]), int(hist["As"][t - 1][i])))
            # a miss is the NEIGHBOURING particle, or one across particles whose whole normalised weight is below the rounding of the cumulative sums themselves
            # (a few ulps of 1: such particles are invisible to either order)
            if b - a > 1 and float(np.sum(w[a + 1:b], dtype=np.float64)) > 8 * np.finfo(np.float32).eps:
                far += 1
    rate = bad / ((T - 1) * (N - 1))
    print(f"device fp32 ancestors vs literal fp32 order: {bad} of {(T - 1) * (N - 1)} draws differ ({rate:.2e}), {far} farther than one visible particle")
    assert rate <= 2e-4, rate
    assert far == 0

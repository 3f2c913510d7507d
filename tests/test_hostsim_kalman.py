"""CPU check of the HIP path's per-lane math: the product's kernel bodies (csrc/kalman_bodies.h) compiled for the
host (tests/hostsim) vs the NumPy oracle, on the reference's own seeded test inputs and on chain-batched /
broadcast layouts.  The same comparisons run against the real HIP library in tests/test_gpu_kalman.py."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K
from tests import hostsim as H
from tests import hostsim as Hs
from tests.helpers import ref_lgssm_inputs, ref_batched_inputs, lg_model

TOL64 = dict(rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("seed", [0, 1234])
@pytest.mark.parametrize("T", [5, 7])
@pytest.mark.parametrize("dx", [1, 2])
@pytest.mark.parametrize("dy", [1, 3])
@pytest.mark.parametrize("E", [0, 2])  # 0 = sequential (one chunk), 2 = chunked scan
@pytest.mark.parametrize("nan_index", [True, False])
def test_filter_vs_oracle(seed, T, dx, dy, E, nan_index):
    ys, lg = ref_lgssm_inputs(seed, T, dx, dy, nan_index)
    ms, Ps, ell = H.filtering(ys, lg, E)
    oms, oPs, oell = K.filtering(ys, lg, E != 0)
    npt.assert_allclose(ms, oms, **TOL64)
    npt.assert_allclose(Ps, oPs, **TOL64)
    npt.assert_allclose(ell, oell, **TOL64)


@pytest.mark.parametrize("seed", [0, 1234])
@pytest.mark.parametrize("dx,dy", [(1, 1), (2, 3), (1, 3), (2, 1)])
@pytest.mark.parametrize("E", [0, 2])
def test_filter_batched(seed, dx, dy, E):
    T, B = 5, 3
    (bys, blg), (ys, lg) = ref_batched_inputs(seed, T, dx, dy, B)
    bms, bPs, bell = H.filtering(bys, blg, E)
    oms, oPs, oell = K.filtering(bys, blg, True)
    npt.assert_allclose(bms, oms, **TOL64)
    npt.assert_allclose(bPs, oPs, **TOL64)
    npt.assert_allclose(bell, oell, **TOL64)


@pytest.mark.parametrize("seed", [42, 666])
@pytest.mark.parametrize("T", [3, 5, 17])
@pytest.mark.parametrize("dx", [1, 2])
@pytest.mark.parametrize("E", [0, 3])
def test_sampler_vs_oracle(seed, T, dx, E):
    ys, lg = ref_lgssm_inputs(seed, T, dx, 3)
    ms, Ps, _ = K.filtering(ys, lg, False)
    eps = np.random.default_rng(seed).standard_normal((T, dx))
    xs = H.sampling(eps, ms, Ps, lg, E)
    npt.assert_allclose(xs, K.sampling(eps, ms, Ps, lg, True), **TOL64)


@pytest.mark.parametrize("seed", [42, 666])
def test_sampler_batched_equals_block_diag(seed):
    T, dx, dy, B = 5, 2, 3, 3
    (bys, blg), (ys, lg) = ref_batched_inputs(seed, T, dx, dy, B)
    bms, bPs, _ = K.filtering(bys, blg, False)
    eps = np.random.default_rng(seed).standard_normal((T, B, dx))
    xs = H.sampling(eps, bms, bPs, blg, 2)
    npt.assert_allclose(xs, K.sampling(eps, bms, bPs, blg, False), atol=1e-10, rtol=1e-10)


@pytest.mark.parametrize("nan_index", [True, False])
@pytest.mark.parametrize("dx,dy", [(1, 1), (2, 3), (1, 3), (2, 1)])
def test_joint_logpdf(nan_index, dx, dy):
    T = 7
    ys, lg = ref_lgssm_inputs(5, T, dx, dy, nan_index)
    xs = np.random.default_rng(0).standard_normal((T, dx))
    want = K.log_likelihood(ys, xs, lg) + K.prior_logpdf(xs, lg)
    npt.assert_allclose(H.joint_logpdf(ys, xs, lg), want, **TOL64)


@pytest.mark.parametrize("d", [1, 2, 4])
@pytest.mark.parametrize("E", [4, -4])  # negative: chain-minor internal buffers (the fused sweep's layout)
@pytest.mark.parametrize("dtype,tol", [(np.float64, TOL64), (np.float32, dict(rtol=2e-3, atol=2e-3))])
def test_chain_batched_broadcast_layout(d, E, dtype, tol):
    """C chains share time-invariant (stride-0) model parameters; only ys varies per chain."""
    T, C = 40, 3
    m = lg_model(T, d)
    delta = 0.5
    P = 2 * d
    rng = np.random.default_rng(3)
    u = rng.standard_normal((C, T, d))
    ys = np.concatenate([u, np.broadcast_to(m["y"], (C, T, d))], axis=-1)
    H_ = np.concatenate([np.eye(d), m["Hobs"]])
    R_ = np.zeros((P, P))
    R_[:d, :d] = 0.5 * delta * np.eye(d)
    R_[d:, d:] = m["Robs"]
    bt = np.broadcast_to
    lg = (m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)),
          bt(H_, (T, P, d)), bt(R_, (T, P, P)), bt(np.zeros(P), (T, P)))
    ms, Ps, ell = H.filtering(ys, lg, E, dtype=dtype, chains=True, chain_axis=False)
    eps = rng.standard_normal((C, T, d))
    xs = H.sampling(eps, ms, Ps, lg, E, dtype=dtype, chains=True)
    for c in range(C):
        oms, oPs, oell = K.filtering(ys[c], lg, True)
        npt.assert_allclose(ms[c], oms, **tol)
        npt.assert_allclose(Ps[c], oPs, **tol)
        npt.assert_allclose(ell[c], oell, rtol=tol["rtol"] * 10, atol=tol["atol"] * 10)
        npt.assert_allclose(xs[c], K.sampling(eps[c], oms, oPs, lg, True), rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)


@pytest.mark.parametrize("d,po", [(1, 1), (2, 2), (2, 3), (4, 4), (3, 2)])
@pytest.mark.parametrize("E", [0, 3, -3])
def test_block_diagonal_R_information_form(d, po, E):
    """R = blkdiag(R1 (d x d), R2 (po x po)) with the pblk hint: the information form (Lam = sum_b H_b^T R_b^-1 H_b,
    M = (I + Lam P)^-1 Lam, log|S| = log|R| + log|I + Lam P|) must reproduce the dense-Cholesky path / the oracle,
    including missing components inside either block and fully missing steps."""
    T, p = 60, d + po
    rng = np.random.default_rng(10 * d + po)
    Fs = 0.6 * rng.standard_normal((T - 1, d, d)) / np.sqrt(d)
    A = rng.standard_normal((T - 1, d, 2 * d))
    Qs = A @ A.transpose(0, 2, 1) / (2 * d) + 0.2 * np.eye(d)
    bs = rng.standard_normal((T - 1, d))
    Hs = rng.standard_normal((T, p, d))
    Rs = np.zeros((T, p, p))
    B1 = rng.standard_normal((T, d, 2 * d))
    B2 = rng.standard_normal((T, po, 2 * po))
    Rs[:, :d, :d] = B1 @ B1.transpose(0, 2, 1) / (2 * d) + 0.2 * np.eye(d)
    Rs[:, d:, d:] = B2 @ B2.transpose(0, 2, 1) / (2 * po) + 0.2 * np.eye(po)
    cs = rng.standard_normal((T, p))
    ys = rng.standard_normal((T, p))
    ys[rng.random((T, p)) < 0.15] = np.nan
    ys[5] = np.nan
    ys[0] = rng.standard_normal(p)
    lg = (rng.standard_normal(d), np.eye(d), Fs, Qs, bs, Hs, Rs, cs)
    ms, Ps, ell = H.filtering(ys, lg, E, pblk=d)
    oms, oPs, oell = K.filtering(ys, lg, True)
    npt.assert_allclose(ms, oms, rtol=1e-8, atol=1e-10)
    npt.assert_allclose(Ps, oPs, rtol=1e-8, atol=1e-10)
    npt.assert_allclose(ell, oell, rtol=1e-9)


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("d", [1, 2, 4])
@pytest.mark.parametrize("chain_minor", [False, True])
def test_sv_fused_logpdf_body(order, d, chain_minor):
    """The SV sweep's fused log-density body (jp_prop, jp_rev, lt_prop, lt_rev, corr per chain) against the oracle's pieces: the
    auxiliary model's joint log_likelihood + prior (base.py:99-166) and the target prior + potential, at both linearisation points."""
    from tests import hostsim as HS
    from tests.helpers import sv_setup
    from aux_ssm_samplers_amd.kalman.models import SVModel
    T, C, delta = 23, 3, 0.4
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d, seed=d)
    model = SVModel(y, m0, P0, F, Q, b, order=order)
    rng = np.random.default_rng(order * 10 + d)
    x = xtrue[None] + 0.3 * rng.standard_normal((C, T, d))
    xp = xtrue[None] + 0.3 * rng.standard_normal((C, T, d))
    u = x + np.sqrt(delta / 2) * rng.standard_normal((C, T, d))
    ys1, ys2, R1, R2 = (np.zeros((C, T, d)), np.zeros((C, T, d)), np.zeros((C, T, d, d)), np.zeros((C, T, d, d)))
    ref = np.zeros((5, C))
    lgp = (model.m0, model.P0, model.Fs, model.Qs, model.bs, None, None, None)
    for c in range(C):
        y1, Hm, Ra, cc = model.observations_factory(x[c], u[c], delta)
        y2, _, Rb, _ = model.observations_factory(xp[c], u[c], delta)
        ys1[c], ys2[c], R1[c], R2[c] = y1, y2, Ra, Rb
        lg1 = (model.m0, model.P0, model.Fs, model.Qs, model.bs, Hm, Ra, cc)
        lg2 = (model.m0, model.P0, model.Fs, model.Qs, model.bs, Hm, Rb, cc)
        ref[0, c] = K.log_likelihood(y1, xp[c], lg1) + K.prior_logpdf(xp[c], lg1)
        ref[1, c] = K.log_likelihood(y2, x[c], lg2) + K.prior_logpdf(x[c], lg2)
        ref[2, c] = K.prior_logpdf(xp[c], lgp) + model.log_potential(xp[c])
        ref[3, c] = K.prior_logpdf(x[c], lgp) + model.log_potential(x[c])
        ref[4, c] = np.sum(((xp[c] - u[c]) ** 2 - (x[c] - u[c]) ** 2) / delta)
    got = HS.sv_logpdf((model.m0, model.P0, model.Fs, model.Qs, model.bs), y, x, xp, u, ys1, ys2,
                            R1 if order == 2 else None, R2 if order == 2 else None, delta, chain_minor)
    npt.assert_allclose(got, ref, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("chain_minor", [False, True])
@pytest.mark.parametrize("per_chain_theta", [False, True])
def test_lorenz_fused_logpdf_body(chain_minor, per_chain_theta):
    """The Lorenz sweep's fused log-density body against the oracle's pieces: the joint of the auxiliary LGSSM linearised at x (resp. x')
    evaluated at x' (resp. x) -- base.py:99-166 on the NumPy factories of LorenzModel, NaN observation rows included -- and the target
    log_likelihood_fn at both points."""
    from tests import hostsim as HS
    from tests.helpers import lorenz_kalman_setup
    from aux_ssm_samplers_amd.kalman.models import LorenzModel
    T, C, delta = 41, 3, 0.02
    base, xtrue = lorenz_kalman_setup(T)
    rng = np.random.default_rng(2)
    theta = base.theta + (0.4 * rng.standard_normal((C, 3)) if per_chain_theta else np.zeros((C, 3)))
    x = xtrue[None] + 0.05 * rng.standard_normal((C, T, 3))
    xp = xtrue[None] + 0.05 * rng.standard_normal((C, T, 3))
    u = x + np.sqrt(delta / 2) * rng.standard_normal((C, T, 3))
    ref = np.zeros((5, C))
    for c in range(C):
        mc = LorenzModel(base.yobs, base.Hobs, base.Robs, base.cobs, base.m0, base.P0, theta[c], base.sigma_x, base.dt)
        for k, (lin, ev) in enumerate(((x[c], xp[c]), (xp[c], x[c]))):
            m0, P0, Fs, Qs, bs = mc.dynamics_factory(lin)
            ys, Hs, Rs, cs = mc.observations_factory(lin, u[c], delta)
            lg = (m0, P0, Fs, Qs, bs, Hs, Rs, cs)
            ref[k, c] = K.log_likelihood(ys, ev, lg) + K.prior_logpdf(ev, lg)
        ref[2, c] = mc.log_likelihood_fn(xp[c])
        ref[3, c] = mc.log_likelihood_fn(x[c])
        ref[4, c] = np.sum(((xp[c] - u[c]) ** 2 - (x[c] - u[c]) ** 2) / delta)
    par = np.concatenate([theta, np.full((C, 1), base.dt)], 1)
    n = T - 1
    lg = (base.m0, base.P0, np.broadcast_to(np.eye(3), (n, 3, 3)), base.Qs, np.zeros((n, 3)), base.Hobs, base.Robs, base.cobs)
    got = HS.lorenz_logpdf(lg, base.yobs, x, xp, u, par if per_chain_theta else par[0], delta, 0, chain_minor)
    npt.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)



@pytest.mark.parametrize("d", [1, 2, 3, 4])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fold_step_is_prefix_combined_with_the_steps_element(d, seed):
    """The general path never builds a step's scan element: it folds the raw step onto the prefix (kalman_math.h::filter_fold_step, 9 d^3 + one
    (d+1)-column LU).  That must be the reference operator applied to (prefix, element(step)) (filtering.py:163-183 on the element of :196-250)."""
    rng = np.random.default_rng(100 * d + seed)
    spd = lambda n, s=1.0: (lambda a: a @ a.T / (2 * n) + s * np.eye(n))(rng.standard_normal((n, 2 * n)))
    pack = lambda m: np.array([m[i, j] for i in range(d) for j in range(i, d)])
    p = d + 2
    Hm, R, y = rng.standard_normal((p, d)), spd(p, 0.3), rng.standard_normal(p)
    Ri = np.linalg.inv(R)
    Lam, g0, q0 = Hm.T @ Ri @ Hm, Hm.T @ Ri @ y, float(y @ Ri @ y)
    ldR = 0.5 * np.linalg.slogdet(R)[1]
    F, Q, bd = 0.7 * rng.standard_normal((d, d)), spd(d, 0.2), rng.standard_normal(d)
    acc = np.concatenate([rng.standard_normal(d * d), rng.standard_normal(d), pack(spd(d, 0.1)), rng.standard_normal(d), pack(spd(d, 0.05)), [rng.standard_normal()]])
    assert Hs.fold_check(F, Q, bd, pack(Lam), g0, q0, ldR, float(p), acc) < 1e-10


def test_auxiliary_block_around_the_predicted_mean_keeps_its_digits_in_fp32():
    """The general path's fold takes the step's observation in information form.  For the auxiliary block y = u, H = I, R = delta/2 I that form (|u|^2 / hd,
    u.m / hd, m.m / hd around the origin) cancels ~|x|^2 / hd down to the innovation: at Lorenz-63 scale (|x| ~ 25, delta = 1e-4: 3.7e7 -> ~3) fp32 lost every digit of
    the log-likelihood increment (C4: ell of 16 384 steps off by +390, round 4).  Kept apart and evaluated around the predicted mean (StepInfo::u / inv_hd) the fp32
    increment has the accuracy of its inputs; in fp64 the two forms agree to rounding."""
    rng = np.random.default_rng(12)
    d, delta, dt = 3, 1e-4, 1.25e-4
    hd = delta / 2
    pack = lambda S: np.array([S[i, j] for i in range(d) for j in range(i, d)])
    worst32_folded = worst32_apart = 0.0
    for rep in range(40):
        x = np.array([1.5, -1.5, 25.0]) + rng.standard_normal(3) * np.array([8.0, 8.0, 8.0])
        F = np.eye(d) + dt * rng.standard_normal((d, d)) * 20
        Q = 9 * dt * np.eye(d)
        bd = x - F @ x + dt * rng.standard_normal(d)                   # so that the predicted mean is x + O(dt)
        b0 = x + 0.007 * rng.standard_normal(d)
        C0 = pack(4.8e-5 * np.eye(d))
        u = x + np.sqrt(hd) * rng.standard_normal(d) + 0.03 * rng.standard_normal(d)
        obs = rep % 2 == 0                                              # every other step carries a real observation of (x2, x3) with variance 5
        H = np.array([[0, 1.0, 0], [0, 0, 1.0]])
        y = H @ x + np.sqrt(5.0) * rng.standard_normal(2)
        Lobs = pack(H.T @ H / 5.0) if obs else np.zeros(6)
        gobs = H.T @ y / 5.0 if obs else np.zeros(3)
        qobs = float(y @ y / 5.0) if obs else 0.0
        ldR = 0.5 * (d * np.log(hd) + (2 * np.log(5.0) if obs else 0.0))
        dim = d + (2 if obs else 0)
        za, zb, db, dC = Hs.fold_aux(np.float64, F, Q, bd, Lobs, gobs, qobs, u, 1 / hd, ldR, dim, b0, C0)
        assert abs(za - zb) < 1e-6 * max(1.0, abs(zb)) and db < 1e-9 and dC < 1e-12      # fp64: the same step (the folded form already loses 8 of its 16 digits)
        fa, fb, _, _ = Hs.fold_aux(np.float32, F, Q, bd, Lobs, gobs, qobs, u, 1 / hd, ldR, dim, b0, C0)
        worst32_folded = max(worst32_folded, abs(fa - zb))
        worst32_apart = max(worst32_apart, abs(fb - zb))
    assert worst32_apart < 5e-3, worst32_apart        # fp32 inputs of magnitude 25 carry 1e-6: an innovation of 0.035 to 1e-4 relative, its square over S likewise
    assert worst32_folded > 20 * worst32_apart, (worst32_folded, worst32_apart)   # what round 3 computed

"""The two cSMC oracles against each other (CPU only):

* `oracle/csmc_np.py` -- the LITERAL NumPy restatement of the reference's arithmetic order (normalised weights, plain cumsum,
  `searchsorted`), evaluating generic Python M0 / G0 / Mt / Gt / Pt objects;
* `oracle/csmc_ref.c` -- the co-designed CONTRACT oracle the HIP kernels reproduce bit for bit (unnormalised weights shifted by a
  bound, DPP-order cumsum, two-level search, closed model family).

fp64: identical ancestors (resampling `As`, backward indices `B`) and trajectories for every model of the closed family on identical
explicit noise.  fp32: the per-draw index-disagreement rate of the two orders (ties in the last ulps of the cumulative weights) is
measured teacher-forced and bounded.  Plus the reference's own statistical known answers (test_csmc.py:18-69 at its own size,
test_resamplings.py:11-24) on the literal restatement."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import csmc as O
from oracle import csmc_np as L


def _family(d, T, rng, potential, transition="linear", tv=False):
    """one model of the closed family as (contract-oracle dict, literal protocol objects, y)"""
    A = rng.standard_normal((d, d))
    Q = A @ A.T / d + 0.5 * np.eye(d)
    F = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d))
    b = 0.1 * rng.standard_normal(d)
    m0 = 0.1 * rng.standard_normal(d)
    LP0, LQ = np.linalg.cholesky(2.0 * np.eye(d)), np.linalg.cholesky(Q)
    sig = 0.7
    y = rng.standard_normal((T, d))
    od = dict(potential=potential, m0=m0, chol_P0=LP0, F=F, b=b, chol_Q=LQ, sig_y=sig)
    M0 = L.GaussianInit(m0, LP0)
    if transition == "lorenz":
        theta, dt = np.array([10.0, 28.0, 8.0 / 3.0]), 0.01
        Fl = np.zeros((3, 3))
        Fl[0] = theta
        LQ = 3.0 * np.sqrt(dt) * np.eye(3)
        od.update(F=Fl, b=[dt, 0, 0], chol_Q=LQ, transition=O.TRANS_LORENZ63_EM)
        Mt = L.Lorenz63EM(theta, dt, LQ, T)
    elif tv:
        Ft = F[None] + 0.05 * rng.standard_normal((T - 1, d, d))
        bt = b[None] + 0.1 * rng.standard_normal((T - 1, d))
        LQt = np.stack([np.linalg.cholesky(Q * (0.5 + rng.random())) for _ in range(T - 1)])
        od.update(F_t=Ft, b_t=bt, chol_Q_t=LQt)
        Mt = L.LinearGaussianDynamics(Ft, bt, LQt, T)
    else:
        Mt = L.LinearGaussianDynamics(F, b, LQ, T)
    if potential == O.POT_FLAT:
        G0, Gt, yy = L.FlatUnivariatePotential(), L.FlatPotential(), None
    else:
        kind = {O.POT_GAUSS_OBS: "gauss", O.POT_SV: "sv", O.POT_GAUSS_OBS_MASKED: "masked"}[potential]
        if kind == "masked":
            y[rng.random((T, d)) < 0.4] = np.nan
            y[1] = np.nan  # a whole missing step
        G0, Gt, yy = L.ObsPotential(kind, y[0], sig, first=True), L.ObsPotential(kind, y[1:], sig), y
    return od, (M0, G0, Mt, Gt), yy


def _noise(T, N, d, rng, aux):
    nz = dict(eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T))
    if aux:
        nz["eps_aux"] = rng.standard_normal((T, d))
    return nz


def _run_both(od, objs, y, x0, N, backward, proposal, nz, delta=None, gradient=0):
    M0, G0, Mt, Gt = objs
    T = x0.shape[0]
    key = L.Noise(**nz)
    if proposal == O.BOOTSTRAP_LG:
        # bootstrap: the FK model's own M0 / Mt propose, the weights are the potentials (test_csmc/common.py fixtures)
        _, kern = L.get_kernel(M0, G0, Mt, Gt, N, backward=backward, Pt=Mt)
        xl, Bl, hist = kern(key, x0)
        okw = {}
    else:
        _, kern = L.get_independent_kernel(M0, G0, Mt, Gt, N, backward=backward, Pt=Mt, gradient=gradient > 0,
                                           exact_gradient=gradient == O.GRAD_EXACT)
        xl, Bl, hist = kern(key, x0, delta)
        okw = dict(sqrt_half_delta=np.sqrt(0.5 * np.broadcast_to(delta, (T,))), eps_aux=nz["eps_aux"])
    ref = O.sweep(dict(od, proposal=proposal, gradient=gradient), x0, N, backward, y=y, eps_prop=nz["eps_prop"], u_res=nz["u_res"],
                  u_bwd=nz["u_bwd"], dtype=np.float64, **okw)
    return (xl, Bl, hist), ref


@pytest.mark.parametrize("backward", [True, False])
@pytest.mark.parametrize("proposal", [O.BOOTSTRAP_LG, O.AUX_INDEPENDENT])
@pytest.mark.parametrize("potential", [O.POT_FLAT, O.POT_GAUSS_OBS, O.POT_SV, O.POT_GAUSS_OBS_MASKED])
@pytest.mark.parametrize("d,N,T", [(1, 32, 25), (1, 1024, 40), (2, 100, 30), (3, 512, 12), (4, 65, 33), (8, 25, 20), (30, 25, 12)])  # (d > 4: csrc/csmc_wide.hip)
def test_contract_oracle_equals_literal_restatement_fp64(d, N, T, potential, proposal, backward):
    """identical explicit noise -> identical As, B and trajectory; particles / log-weights to rounding"""
    rng = np.random.default_rng(7919 * d + 13 * N + T + 101 * potential + proposal)
    od, objs, y = _family(d, T, rng, potential)
    x0 = rng.standard_normal((T, d))
    nz = _noise(T, N, d, rng, proposal == O.AUX_INDEPENDENT)
    delta = 0.5 + rng.random(T)
    (xl, Bl, hist), ref = _run_both(od, objs, y, x0, N, backward, proposal, nz, delta)
    npt.assert_array_equal(hist["As"], ref["As"])
    npt.assert_array_equal(Bl, ref["ancestors"])
    npt.assert_allclose(hist["xs"], ref["xs"], rtol=1e-12, atol=1e-12)
    npt.assert_allclose(hist["log_ws"], ref["log_ws"], rtol=1e-11, atol=1e-11)
    npt.assert_allclose(xl, ref["x"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("backward", [True, False])
@pytest.mark.parametrize("proposal", [O.BOOTSTRAP_LG, O.AUX_INDEPENDENT])
def test_lorenz_and_time_varying_members_fp64(proposal, backward):
    for transition, tv, d, N, T in (("lorenz", False, 3, 128, 20), ("linear", True, 2, 96, 24), ("linear", True, 1, 64, 30)):
        rng = np.random.default_rng(31 + d + N)
        pot = O.POT_GAUSS_OBS_MASKED if transition == "lorenz" else O.POT_SV
        od, objs, y = _family(d, T, rng, pot, transition, tv)
        x0 = rng.standard_normal((T, d)) + (np.array([1.5, -1.5, 25.0]) if transition == "lorenz" else 0.0)
        if transition == "lorenz":
            od["m0"], objs[0].m0 = np.array([1.5, -1.5, 25.0]), np.array([1.5, -1.5, 25.0])
            y = y + np.array([1.5, -1.5, 25.0])
            objs = (objs[0], L.ObsPotential("masked", y[0], 0.7, first=True), objs[2], L.ObsPotential("masked", y[1:], 0.7))
        nz = _noise(T, N, d, rng, proposal == O.AUX_INDEPENDENT)
        (xl, Bl, hist), ref = _run_both(od, objs, y, x0, N, backward, proposal, nz, 0.3)
        npt.assert_array_equal(hist["As"], ref["As"])
        npt.assert_array_equal(Bl, ref["ancestors"])
        npt.assert_allclose(xl, ref["x"], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("gradient", [O.GRAD_REFERENCE, O.GRAD_EXACT])
@pytest.mark.parametrize("backward", [True, False])
def test_gradient_proposals_fp64(gradient, backward):
    """independent.py:57-75 with gradient=True: the literal side differentiates `_log_pdf` (:121-134) numerically, the contract
    oracle uses the closed form -> proposals agree to ~1e-8, indices exactly.  GRAD_REFERENCE reproduces the reference's
    `jnp.sum` without axis (:265-266): the correction is a constant of the step and leaves every index unchanged."""
    for pot, d, N, T in ((O.POT_SV, 1, 64, 16), (O.POT_GAUSS_OBS, 2, 50, 12)):
        rng = np.random.default_rng(5 + d)
        od, objs, y = _family(d, T, rng, pot)
        x0 = rng.standard_normal((T, d))
        nz = _noise(T, N, d, rng, True)
        (xl, Bl, hist), ref = _run_both(od, objs, y, x0, N, backward, O.AUX_INDEPENDENT, nz, 0.4, gradient)
        npt.assert_array_equal(hist["As"], ref["As"])
        npt.assert_array_equal(Bl, ref["ancestors"])
        npt.assert_allclose(hist["xs"], ref["xs"], rtol=1e-7, atol=1e-7)
        npt.assert_allclose(xl, ref["x"], rtol=1e-7, atol=1e-7)


def test_fp32_index_disagreement_rate_is_bounded():
    """fp32: the literal order (normalise, left-to-right cumsum) and the contract order (shift by a bound, DPP-network cumsum,
    two-level search) round the cumulative weights differently, so a draw that lands within a few ulps of a boundary can pick the
    neighbouring particle.  Teacher-forced measurement: every step's resampling is redone in the literal order from the contract
    oracle's OWN stored log-weights and uniforms, and compared with the contract oracle's ancestors of that step."""
    rng = np.random.default_rng(2024)
    T, N, d = 400, 1024, 1
    od, objs, y = _family(d, T, rng, O.POT_SV)
    od.update(F=[[0.9]], b=[0.0], chol_Q=[[np.sqrt(10.526)]], chol_P0=[[np.sqrt(10.526)]], m0=[0.0])
    x0 = rng.standard_normal((T, d)).astype(np.float32)
    nz = _noise(T, N, d, rng, True)
    ref = O.sweep(dict(od, proposal=O.AUX_INDEPENDENT), x0, N, True, y=y, sqrt_half_delta=np.full(T, 0.5), eps_aux=nz["eps_aux"],
                  eps_prop=nz["eps_prop"], u_res=nz["u_res"], u_bwd=nz["u_bwd"], dtype=np.float32)
    u32 = nz["u_res"].astype(np.float32)
    bad = 0
    offby = 0
    for t in range(1, T):
        w = L.normalize(ref["log_ws"][t - 1])
        assert w.dtype == np.float32
        A = L.multinomial(u32[t - 1], w)
        diff = A != ref["As"][t - 1]
        bad += int(diff.sum())
        offby = max(offby, int(np.abs(A - ref["As"][t - 1]).max()))
    rate = bad / ((T - 1) * (N - 1))
    print(f"fp32 index disagreement rate {rate:.2e} ({bad} of {(T - 1) * (N - 1)} draws), largest index distance {offby}")
    assert rate < 2e-3          # measured 1e-4 .. 4e-4 at N = 1024
    assert offby <= 4           # a disagreement is a NEIGHBOURING particle (or one across a run of zero-weight particles)
    # the same comparison in fp64 on the same inputs: no disagreement at all
    ref64 = O.sweep(dict(od, proposal=O.AUX_INDEPENDENT), x0.astype(np.float64), N, True, y=y, sqrt_half_delta=np.full(T, 0.5),
                    eps_aux=nz["eps_aux"], eps_prop=nz["eps_prop"], u_res=nz["u_res"], u_bwd=nz["u_bwd"], dtype=np.float64)
    for t in range(1, T):
        A = L.multinomial(nz["u_res"][t - 1], L.normalize(ref64["log_ws"][t - 1]))
        npt.assert_array_equal(A, ref64["As"][t - 1])


def test_multinomial_resampling_known_answer_literal():
    """test_resamplings.py:11-24 on the literal restatement: index 0 kept, the others ~ weights (100_000 keys, atol 1e-3)"""
    rng = np.random.default_rng(42)
    w = rng.random(10)
    w /= w.sum()
    u = rng.random((100_000, 10))
    c = np.cumsum(w)
    idx = np.searchsorted(c, c[-1] * (1 - u))     # vectorised form of L.choice, checked against it below
    idx[:, 0] = 0
    for k in range(50):
        npt.assert_array_equal(L.multinomial(u[k], w), idx[k])
    bincount = np.bincount(idx[:, 1:].ravel(), minlength=10)
    npt.assert_allclose(bincount / bincount.sum(), w, atol=1e-3)
    assert np.all(idx[:, 0] == 0)


def _flat_chain(sweep_fn, M, T, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((T, 1))
    out = np.empty((M, T))
    for it in range(M):
        x = sweep_fn(x, rng)
        out[it] = x[:, 0]
    return out[M // 10:]


def _check_ar1_prior(xs, rho, atol=0.05):
    cov = np.cov(xs, rowvar=False)
    npt.assert_allclose(xs.mean(axis=0), 0.0, atol=atol)
    npt.assert_allclose(np.diag(cov), 1.0, atol=atol)
    npt.assert_allclose(np.diag(cov, 1), rho, atol=atol)


@pytest.mark.parametrize("backward", [True, False])
def test_flat_potential_reference_size_literal_and_contract(backward):
    """test_csmc.py:18-69 at the reference's OWN size (T = 5, N = 32, 50_000 iterations, burn-in 10 %, atol 0.05) on the literal
    restatement with the reference's fixture classes, and on the contract oracle; and the two chains are THE SAME chain (fp64,
    identical noise) -- every ancestor of every one of the 50_000 sweeps."""
    T, N, M, rho = 5, 32, 50_000, 0.9
    M0, G0, Gt, Mt = L.GaussianDistribution(0.0, 1.0), L.FlatUnivariatePotential(), L.FlatPotential(), L.GaussianDynamics(rho)
    _, kern = L.get_kernel(M0, G0, Mt, Gt, N, backward=backward, Pt=Mt)
    od = dict(proposal=O.BOOTSTRAP_LG, potential=O.POT_FLAT, m0=[0.0], chol_P0=[[1.0]], F=[[rho]], b=[0.0], chol_Q=[[(1 - rho ** 2) ** 0.5]])
    mism = [0]

    def both(x, rng):
        nz = _noise(T, N, 1, rng, False)
        xl, Bl, _ = kern(L.Noise(**nz), x)
        r = O.sweep(od, x, N, backward, dtype=np.float64, **nz)
        mism[0] += int(np.any(Bl != r["ancestors"]))
        npt.assert_allclose(xl, r["x"], rtol=1e-12, atol=1e-12)
        return xl

    xs = _flat_chain(both, M, T, 0)
    assert mism[0] == 0
    _check_ar1_prior(xs, rho)

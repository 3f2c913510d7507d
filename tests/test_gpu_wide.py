"""GPU parity tests of the wide-state Kalman path (csrc/wide.hip: one workgroup per time step / scan element, matrices in
LDS) -- every (dx, dy) beyond the register-resident per-lane kernels (dx > 4 or dy > 8), up to SURVEY config C5
(dx = dy = 64, fp32).  Oracle: oracle/kalman_np.py on the same seeded inputs, plus the reference's independent
explicit Kalman filter.  Tolerances: fp64 rtol 1e-8 / atol 1e-10 on well-conditioned models (1e-6 on the reference's
random ill-conditioned test models, whose own test uses 1e-7 at dx <= 2); fp32 stated per test.
"""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K
from tests.helpers import ref_lgssm_inputs, ref_batched_inputs

from tests.helpers import c5_model  # noqa: E402,F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import aux_ssm_samplers_amd._primitives.kalman as prim
    return prim


def stable_model(rng, T, d, p, nan=True):
    Fs = 0.5 * rng.standard_normal((T - 1, d, d)) / np.sqrt(d)
    A = rng.standard_normal((T - 1, d, 2 * d))
    Qs = A @ A.transpose(0, 2, 1) / (2 * d) + 0.1 * np.eye(d)
    bs = rng.standard_normal((T - 1, d))
    Hs = rng.standard_normal((T, p, d)) / np.sqrt(d)
    Bm = rng.standard_normal((T, p, 2 * p))
    Rs = Bm @ Bm.transpose(0, 2, 1) / (2 * p) + 0.1 * np.eye(p)
    cs = rng.standard_normal((T, p))
    ys = rng.standard_normal((T, p))
    if nan:
        ys[rng.random(T) < 0.1] = np.nan
        ys[rng.random((T, p)) < 0.05] = np.nan
        ys[0] = rng.standard_normal(p)
    m0 = rng.standard_normal(d)
    A0 = rng.standard_normal((d, 2 * d))
    P0 = A0 @ A0.T / (2 * d) + 0.5 * np.eye(d)
    return ys, (m0, P0, Fs, Qs, bs, Hs, Rs, cs)


@pytest.mark.parametrize("seed", [0, 1234])
@pytest.mark.parametrize("T", [1, 5, 7])
@pytest.mark.parametrize("dx,dy", [(5, 3), (2, 9), (6, 11), (8, 8), (16, 5)])
@pytest.mark.parametrize("parallel", [False, True])
@pytest.mark.parametrize("nan_index", [True, False])
def test_filter_reference_style_cases(P, seed, T, dx, dy, parallel, nan_index):
    """The reference's test_filtering.py recipe (random F, Q, H, R; NaN rows as at :43-47) at sizes it never reaches."""
    ys, lg = ref_lgssm_inputs(seed, max(T, 5), dx, dy, nan_index)
    if T < 5:
        m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
        ys, lg = ys[:T], (m0, P0, Fs[:T - 1], Qs[:T - 1], bs[:T - 1], Hs[:T], Rs[:T], cs[:T])
    ms, Ps, ell = P.filtering(ys, P.LGSSM(*lg), parallel)
    oms, oPs, oell = K.filtering(ys, lg, parallel)
    npt.assert_allclose(ms, oms, rtol=1e-6, atol=1e-8)
    npt.assert_allclose(Ps, oPs, rtol=1e-6, atol=1e-8)
    npt.assert_allclose(ell, oell, rtol=1e-8, atol=1e-8)
    m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
    ems, ePs, eell = K.explicit_filter(ys, m0, P0, Hs, Rs, cs, Fs, Qs, bs)
    npt.assert_allclose(ms, ems, rtol=1e-5, atol=1e-7)
    npt.assert_allclose(ell, eell, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("d,p,T", [(5, 5, 700), (8, 16, 300), (12, 7, 257), (16, 16, 130), (32, 32, 40)])
def test_all_scan_levels_fp64(P, d, p, T):
    """Time-varying stable model with NaN rows, T long enough for several chunks per sequence: parallel scan vs oracle,
    sequential == parallel, sampler and joint log-density."""
    rng = np.random.default_rng(d * 1000 + T)
    ys, lg64 = stable_model(rng, T, d, p)
    lg = P.LGSSM(*lg64)
    oms, oPs, oell = K.filtering(ys, lg64, True)
    ms, Ps, ell = P.filtering(ys, lg, True)
    npt.assert_allclose(ms, oms, rtol=1e-8, atol=1e-10)
    npt.assert_allclose(Ps, oPs, rtol=1e-8, atol=1e-10)
    npt.assert_allclose(ell, oell, rtol=1e-9)
    seq = P.filtering(ys, lg, False)
    npt.assert_allclose(seq[0], ms, rtol=1e-8, atol=1e-10)
    npt.assert_allclose(seq[2], ell, rtol=1e-9)
    eps = rng.standard_normal((T, d))
    for par in (True, False):
        xs = P.sampling(None, oms, oPs, lg, par, eps=eps)
        npt.assert_allclose(xs, K.sampling(eps, oms, oPs, lg64, True), rtol=1e-8, atol=1e-9)
    xr = rng.standard_normal((T, d))
    npt.assert_allclose(P.posterior_logpdf(ys, xr, oell, lg), K.posterior_logpdf(ys, xr, oell, lg64), rtol=1e-9)
    npt.assert_allclose(P.prior_logpdf(xr, lg), K.prior_logpdf(xr, lg64), rtol=1e-9)
    npt.assert_allclose(P.log_likelihood(ys, xr, lg), K.log_likelihood(ys, xr, lg64), rtol=1e-9)


@pytest.mark.parametrize("seed", [0, 7])
@pytest.mark.parametrize("parallel", [True, False])
def test_batched_model(P, seed, parallel):
    """B-axis semantics (base.py:40-49) on the wide path: batched == oracle batched (== block-diagonal dense)."""
    T, dx, dy, B = 5, 6, 3, 3
    (bys, blg), (ys, lg) = ref_batched_inputs(seed, T, dx, dy, B)
    bms, bPs, bell = P.filtering(bys, P.LGSSM(*blg), parallel)
    oms, oPs, oell = K.filtering(bys, blg, parallel)
    npt.assert_allclose(bms, oms, rtol=1e-6, atol=1e-8)
    npt.assert_allclose(bPs, oPs, rtol=1e-6, atol=1e-8)
    npt.assert_allclose(bell, oell, rtol=1e-8)
    eps = np.random.default_rng(seed).standard_normal((T, B, dx))
    bx = P.sampling(None, oms, oPs, P.LGSSM(*blg), parallel, eps=eps)
    dms, dPs, _ = K.filtering(ys, lg, False)
    npt.assert_allclose(bx.reshape(T, B * dx), K.sampling(eps.reshape(T, B * dx), dms, dPs, lg, parallel), rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("dtype,T", [(np.float32, 96), (np.float64, 40)])
def test_C5_dense_d64(P, dtype, T):
    """Config C5's model (d = p = 64; fp64 at d = 40, the largest the fp64 LDS plan holds) on a short horizon the fp64 oracle
    finishes in seconds: filter, sampler and log-density vs oracle.  fp32 tolerance 5e-4 (stated, achieved ~1e-5)."""
    d = 64 if dtype == np.float32 else 40
    u, lg64, x = c5_model(T, d)
    lg = P.LGSSM(*[np.ascontiguousarray(a, dtype) for a in lg64])
    oms, oPs, oell = K.filtering(u, lg64, True)
    ms, Ps, ell = P.filtering(u.astype(dtype), lg, True)
    tol = dict(rtol=1e-8, atol=1e-10) if dtype == np.float64 else dict(rtol=5e-4, atol=5e-4)
    npt.assert_allclose(ms, oms, **tol)
    npt.assert_allclose(Ps, oPs, **tol)
    npt.assert_allclose(ell, oell, rtol=tol["rtol"])
    eps = np.random.default_rng(1).standard_normal((T, d))
    xs = P.sampling(None, oms.astype(dtype), oPs.astype(dtype), lg, True, eps=eps.astype(dtype))
    npt.assert_allclose(xs, K.sampling(eps, oms, oPs, lg64, True), **tol)
    npt.assert_allclose(P.posterior_logpdf(u.astype(dtype), x.astype(dtype), oell, lg), K.posterior_logpdf(u, x, oell, lg64), rtol=tol["rtol"])


def test_C5_benchmarked_horizon_vs_fp64_sequential_fixture(P):
    """config C5 at the horizon bench.py times (d = p = 64, T = 8192, fp32, one sequence) -- 8191 fp32 combines of
    unpivoted blocked eliminations -- against the fp64 SEQUENTIAL filter / sampler of the oracle, committed at 32 time points
    (tests/golden/c5_T8192_known_answers.npz, tests/golden/make_c5_fixture.py).  Stated tolerance: 2e-3 (rtol and atol) on means,
    covariances and the sampled path, 1e-4 relative on the log-likelihood; the achieved errors are printed."""
    import os
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "c5_T8192_known_answers.npz"))
    T, d = 8192, 64
    u, lg64, _ = c5_model(T, d)
    eps = np.random.default_rng(1).standard_normal((T, d))
    npt.assert_allclose(u.sum(), ref["u_checksum"], rtol=1e-12)       # the synthetic inputs are the fixture's
    npt.assert_allclose(eps.sum(), ref["eps_checksum"], rtol=1e-12)
    lg = P.LGSSM(*[np.ascontiguousarray(a, np.float32) for a in lg64])
    ms, Ps, ell = P.filtering(u.astype(np.float32), lg, True)
    idx, full = ref["idx"], ref["full_idx"]
    e_m = np.max(np.abs(ms[idx] - ref["ms"]))
    e_p = np.max(np.abs(np.einsum("tii->ti", Ps[idx]) - ref["Ps_diag"]))
    e_f = np.max(np.abs(Ps[full] - ref["Ps_full"]))
    e_l = abs(float(ell) - float(ref["ell"])) / abs(float(ref["ell"]))
    xs = P.sampling(None, ms, Ps, lg, True, eps=eps.astype(np.float32))   # the device's own fp32 moments in, as a sweep would
    e_x = np.max(np.abs(xs[idx] - ref["xs"]))
    print(f"C5 T=8192 fp32 vs fp64 sequential: max|dm| {e_m:.2e} max|dP_diag| {e_p:.2e} max|dP| {e_f:.2e} rel|d ell| {e_l:.2e} max|dx| {e_x:.2e}")
    tol = dict(rtol=2e-3, atol=2e-3)
    npt.assert_allclose(ms[idx], ref["ms"], **tol)
    npt.assert_allclose(np.einsum("tii->ti", Ps[idx]), ref["Ps_diag"], **tol)
    npt.assert_allclose(Ps[full], ref["Ps_full"], **tol)
    npt.assert_allclose(xs[idx], ref["xs"], **tol)
    assert e_l < 1e-4


def test_too_large_for_lds_fails_loudly(P):
    """fp64 d = 64 exceeds the 160 KB LDS plan: a clear ValueError, never a silent fallback."""
    T, d = 4, 64
    u, lg64, _ = c5_model(T, d)
    with pytest.raises((ValueError, RuntimeError), match="LDS"):
        P.filtering(u, P.LGSSM(*lg64), True)


def lg_concat_wide(T, d, po, seed=0):
    """Stable LG model of state size d with po real observations, as an LGConcatModel (the sweep's device model)."""
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    rng = np.random.default_rng(seed)
    F = 0.9 * np.eye(d) + 0.04 * (np.eye(d, k=1) + np.eye(d, k=-1))
    Q = 0.2 * np.eye(d)
    Hobs = rng.standard_normal((po, d)) / np.sqrt(d)
    Robs = 0.5 * np.eye(po)
    x = np.zeros((T, d))
    x[0] = rng.standard_normal(d)
    for t in range(1, T):
        x[t] = F @ x[t - 1] + np.sqrt(0.2) * rng.standard_normal(d)
    y = x @ Hobs.T + np.sqrt(0.5) * rng.standard_normal((T, po))
    bt = np.broadcast_to
    m0, P0 = np.zeros(d), np.eye(d)
    model = LGConcatModel(m0, P0, bt(F, (T - 1, d, d)), bt(Q, (T - 1, d, d)), bt(np.zeros(d), (T - 1, d)),
                          bt(Hobs, (T, po, d)), bt(Robs, (T, po, po)), bt(np.zeros(po), (T, po)), y)
    return model, x, y, (m0, P0)


@pytest.mark.parametrize("d,po,T", [(6, 6, 300), (8, 3, 130), (16, 16, 60), (3, 7, 100)])
@pytest.mark.parametrize("parallel", [True, False])
def test_device_sweep_wide_vs_oracle_sweep(d, po, T, parallel):
    """kalman/generic.py:53-106 through auxssm_kalman_sweep on the wide-state path (p = d + po up to 32) vs the oracle's
    sweep on identical explicit noise; the exact LG proposal makes log alpha == 0 (SURVEY 8c known answer)."""
    from aux_ssm_samplers_amd.kalman import get_kernel
    model, xt, y, (m0, P0) = lg_concat_wide(T, d, po)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, parallel)
    rng = np.random.Generator(np.random.PCG64(1000))
    x = xt + 0.3 * rng.standard_normal((T, d))
    noise = dict(eps_aux=rng.standard_normal((T, d)), eps_samp=rng.standard_normal((T, d)), u_accept=rng.random())
    lgo = (m0, P0, model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
    ref = K.kalman_sweep(x, 0.5, model.dynamics_factory, model.observations_factory,
                         lambda z: K.log_likelihood(y, z, lgo) + K.prior_logpdf(z, lgo), parallel, **noise)
    out = kernel(None, init(x), 0.5, noise=noise)
    npt.assert_allclose(out.x, ref["x"], rtol=1e-8, atol=1e-9)
    assert out.updated == ref["accepted"]
    npt.assert_allclose(out.logs[0, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-9)
    assert abs(out.log_alpha) < 1e-6 and out.updated


def test_device_sweep_wide_multichain_resident():
    """Several chains resident on the device (dense layout for dx > 4), device Threefry noise: every chain accepts with
    log alpha == 0 (exact LG proposal) and moves."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains
    T, d, po, C = 80, 8, 8, 5
    model, xt, y, _ = lg_concat_wide(T, d, po)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    rng = np.random.default_rng(3)
    x0 = xt[None] + 0.3 * rng.standard_normal((C, T, d))
    h = _lib.default_handle()
    chains = DeviceChains(h, x0)
    assert not chains.chain_minor
    st = kernel(7, init(chains), 0.5)
    xa = st.x.to_host()
    assert np.all(st.x.accepted.to_host() == 1)
    assert np.all(np.abs(st.x.logs.to_host()[:, 0]) < 1e-6)
    assert np.abs(xa - x0).max() > 1e-3 and np.all(np.isfinite(xa))


@pytest.mark.parametrize("d,p,T", [(12, 7, 257), (33, 40, 60)])
def test_diagonal_observation_noise_with_missing_data(P, d, p, T):
    """wk_obs_info takes a diagonal R without elimination (row scaling): same numbers as the dense route, NaN entries and all-NaN rows included."""
    rng = np.random.default_rng(d + T)
    ys, (m0, P0, Fs, Qs, bs, Hs, Rs, cs) = stable_model(rng, T, d, p)
    Rs = np.stack([np.diag(0.2 + rng.random(p)) for _ in range(T)])
    lg64 = (m0, P0, Fs, Qs, bs, Hs, Rs, cs)
    oms, oPs, oell = K.filtering(ys, lg64, True)
    for par in (True, False):
        ms, Ps, ell = P.filtering(ys, P.LGSSM(*lg64), par)
        npt.assert_allclose(ms, oms, rtol=1e-8, atol=1e-10)
        npt.assert_allclose(Ps, oPs, rtol=1e-8, atol=1e-10)
        npt.assert_allclose(ell, oell, rtol=1e-9)
    # the joint log-density takes diagonal covariances without elimination too (wk_logpdf / gauss2): diagonal Q as well, NaN rows kept
    Qd = np.stack([np.diag(0.1 + rng.random(d)) for _ in range(T - 1)])
    lgd = (m0, P0, Fs, Qd, bs, Hs, Rs, cs)
    xr = rng.standard_normal((T, d))
    npt.assert_allclose(P.log_likelihood(ys, xr, P.LGSSM(*lgd)), K.log_likelihood(ys, xr, lgd), rtol=1e-9)
    npt.assert_allclose(P.prior_logpdf(xr, P.LGSSM(*lgd)), K.prior_logpdf(xr, lgd), rtol=1e-9)
    npt.assert_allclose(P.posterior_logpdf(ys, xr, oell, P.LGSSM(*lgd)), K.posterior_logpdf(ys, xr, oell, lgd), rtol=1e-9)


def test_element_path_kept_for_sizes_the_fold_does_not_hold():
    """AUXSSM_WIDE_NO_FOLD=1 routes the filter through the element build + general combine (what d > 64 or a too-small LDS budget falls back
    to); it must give the oracle's numbers too.  Own process: the switch is read once per process."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, numpy.testing as npt\n"
            "import aux_ssm_samplers_amd._primitives.kalman as P\n"
            "from oracle import kalman_np as K\n"
            "from tests.test_gpu_wide import stable_model\n"
            "rng = np.random.default_rng(3)\n"
            "for d, p, T in ((8, 16, 300), (32, 32, 40)):\n"
            "    ys, lg64 = stable_model(rng, T, d, p)\n"
            "    o = K.filtering(ys, lg64, True)\n"
            "    r = P.filtering(ys, P.LGSSM(*lg64), True)\n"
            "    for a, b in zip(r, o): npt.assert_allclose(a, b, rtol=1e-8, atol=1e-10)\n"
            "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, AUXSSM_WIDE_NO_FOLD="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]

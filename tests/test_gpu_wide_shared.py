"""The wide-state filter for SEVERAL sequences that share every model parameter (csrc/wide_shared.h: one matrix recursion on sequence 0, the
gain-form affine mean recursion of all sequences as the columns of one matrix) against
  (1) the per-sequence path of the same library (AUXSSM_OPT_SHARE_MODEL = 0: every sequence runs the full associative scan of filtering.py:163-183),
  (2) the NumPy oracle (oracle/kalman_np.py::filtering, filtering.py:18-250 line by line) sequence by sequence,
including time-varying parameters, missing observations (one pattern for all sequences: the shared form; differing patterns: the library must
notice and run the per-sequence path), steps with nothing observed, more sequences than one column block, and C5's sizes.
Tolerances: fp64 rtol 1e-8 / atol 1e-10; fp32 5e-4 (as tests/test_gpu_wide.py)."""
import ctypes as C

import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K
from tests.helpers import c5_model
from tests.test_gpu_wide import stable_model

pytestmark = pytest.mark.gpu


def device_filter(lg, ys, dtype, share=True):
    """auxssm_kalman_filter on S sequences ys (S, T, p) with ONE parameter set (chain stride 0); returns ms, Ps, ell and the launch groups that ran"""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd._primitives.kalman.base import DeviceLGSSM
    h = _lib.default_handle()
    S, T, p = ys.shape
    d = lg[0].shape[-1]
    dl = DeviceLGSSM(h, tuple(lg), 1, T, 1, d, p, False, dtype)
    yd = h.to_device(np.ascontiguousarray(ys, dtype))
    yarr = yd.arr(T * p, p, 0)
    ms, Ps, ell = h.empty((S, T, d), dtype), h.empty((S, T, d, d), dtype), h.empty((S,), dtype)
    dims = _lib.Dims(S, T, 1, d, p)
    h.set_option(_lib.OPT_SHARE_MODEL, 1 if share else 0)
    try:
        h.prof_enable(_lib.K_ALL, 64)
        _lib.check(h.lib.auxssm_kalman_filter(h.h, _lib.dtype_code(dtype), C.byref(dims), C.byref(dl.c), C.byref(yarr), 1, ms.ptr, Ps.ptr, ell.ptr))
        groups = h.prof_read_groups()
    finally:
        h.prof_disable()
        h.set_option(_lib.OPT_SHARE_MODEL, 1)
    return ms.to_host(), Ps.to_host(), ell.to_host(), groups


def check(lg, ys, dtype, expect_shared=True, tol=None):
    tol = tol or (dict(rtol=1e-8, atol=1e-10) if dtype == np.float64 else dict(rtol=5e-4, atol=5e-4))
    ms, Ps, ell, groups = device_filter(lg, ys, dtype, True)
    assert ("filter_tab" in groups) == expect_shared, groups
    ms1, Ps1, ell1, g1 = device_filter(lg, ys, dtype, False)
    assert "filter_tab" not in g1
    npt.assert_allclose(ms, ms1, **tol)
    npt.assert_allclose(Ps, Ps1, **tol)
    npt.assert_allclose(ell, ell1, rtol=tol["rtol"], atol=tol["atol"])
    for s in range(ys.shape[0]):
        oms, oPs, oell = K.filtering(ys[s], lg, True)
        npt.assert_allclose(ms[s], oms, **tol)
        npt.assert_allclose(Ps[s], oPs, **tol)
        npt.assert_allclose(ell[s], oell, rtol=tol["rtol"], atol=tol["atol"])


@pytest.mark.parametrize("d,p,T,S", [(8, 8, 50, 3), (16, 5, 130, 16), (12, 20, 257, 5), (6, 11, 300, 70), (33, 40, 60, 20), (5, 3, 9, 2)])
def test_shared_vs_per_sequence_and_oracle_fp64(d, p, T, S):
    rng = np.random.default_rng(d * 1000 + T)
    y0, lg = stable_model(rng, T, d, p, nan=False)
    ys = y0[None] + rng.standard_normal((S, T, p))
    check(lg, ys, np.float64)


@pytest.mark.parametrize("d,p,T,S", [(8, 8, 64, 4), (12, 7, 257, 17), (16, 16, 40, 3)])
def test_missing_observations_one_pattern(d, p, T, S):
    """NaN rows (nothing observed at that step), scattered NaN components: the pattern of sequence 0 for all -> still the shared form"""
    rng = np.random.default_rng(T + S)
    y0, lg = stable_model(rng, T, d, p, nan=True)
    ys = y0[None] + rng.standard_normal((S, T, p))  # NaN + x = NaN: same pattern everywhere
    assert np.isnan(ys).any() and np.isnan(ys[0]).all(1).any()
    check(lg, ys, np.float64)


def test_differing_patterns_fall_back_to_the_per_sequence_path():
    rng = np.random.default_rng(3)
    d, p, T, S = 8, 6, 40, 3
    y0, lg = stable_model(rng, T, d, p, nan=True)
    ys = y0[None] + rng.standard_normal((S, T, p))
    ys[1, 7, 2] = np.nan if np.isfinite(ys[0, 7, 2]) else 0.3
    check(lg, ys, np.float64, expect_shared=False)


def test_batch_axis_sequences_share_the_recursion_too():
    """C chains x B batch members (base.py:40-49: the log-likelihood of a chain is the SUM over its batch members) with one parameter set: all C B sequences ride in
    the shared form; ms, Ps per (chain, batch member) and ell per chain equal the per-sequence path and the oracle."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd._primitives.kalman.base import DeviceLGSSM
    rng = np.random.default_rng(21)
    d, p, T, Cn, B = 7, 9, 60, 3, 2
    y0, lg = stable_model(rng, T, d, p, nan=False)
    ys = y0[None, :, None, :] + rng.standard_normal((Cn, T, B, p))
    h = _lib.default_handle()
    out = {}
    for share in (1, 0):
        dl = DeviceLGSSM(h, tuple(lg), 1, T, 1, d, p, False, np.float64)
        yd = h.to_device(np.ascontiguousarray(ys))
        yarr = yd.arr(T * B * p, B * p, p)
        ms, Ps, ell = h.empty((Cn, T, B, d), np.float64), h.empty((Cn, T, B, d, d), np.float64), h.empty((Cn,), np.float64)
        dims = _lib.Dims(Cn, T, B, d, p)
        h.set_option(_lib.OPT_SHARE_MODEL, share)
        try:
            h.prof_enable(_lib.K_ALL, 64)
            _lib.check(h.lib.auxssm_kalman_filter(h.h, _lib.F64, C.byref(dims), C.byref(dl.c), C.byref(yarr), 1, ms.ptr, Ps.ptr, ell.ptr))
            groups = h.prof_read_groups()
        finally:
            h.prof_disable()
            h.set_option(_lib.OPT_SHARE_MODEL, 1)
        assert ("filter_tab" in groups) == bool(share)
        out[share] = (ms.to_host(), Ps.to_host(), ell.to_host())
    for a, b in zip(out[1], out[0]):
        npt.assert_allclose(a, b, rtol=1e-8, atol=1e-10)
    for c in range(Cn):
        tot = 0.0
        for b in range(B):
            oms, oPs, oell = K.filtering(ys[c, :, b], lg, True)
            npt.assert_allclose(out[1][0][c, :, b], oms, rtol=1e-8, atol=1e-10)
            npt.assert_allclose(out[1][1][c, :, b], oPs, rtol=1e-8, atol=1e-10)
            tot += oell
        npt.assert_allclose(out[1][2][c], tot, rtol=1e-8)


@pytest.mark.parametrize("order", [1, 2])
def test_wide_sv_sweep_with_several_chains_shared_equals_per_chain(order):
    """Inside a sweep: the stochastic-volatility sampler at D = 6 (wide path) with 4 dense chains.  First order: H = I, R = delta/2 I and the dynamics are the same for
    every chain, so both filters of the sweep take the shared form; second order: R depends on the chain's state, so they must not.  Either way the keyed sweep equals
    the one with AUXSSM_OPT_SHARE_MODEL off (same keys, same draws) to rounding."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from tests.helpers import sv_setup
    T, D, Cn = 40, 6, 4
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, D)
    model = SVModel(y, m0, P0, F, Q, b, order=order)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    h = _lib.default_handle()
    x0 = xtrue[None] + 0.2 * np.random.default_rng(9).standard_normal((Cn, T, D))
    res = {}
    for share in (1, 0):
        chains = DeviceChains(h, x0, chain_minor=False)
        h.set_option(_lib.OPT_SHARE_MODEL, share)
        try:
            h.prof_enable(_lib.K_ALL, 256)
            for i in range(3):
                kernel(R.PRNGKey(60 + i), KalmanSampler(x=chains, updated=None), 0.3)
            groups = h.prof_read_groups()
        finally:
            h.prof_disable()
            h.set_option(_lib.OPT_SHARE_MODEL, 1)
        assert ("filter_tab" in groups) == (share == 1 and order == 1), groups
        res[share] = (chains.to_host(), chains.logs.to_host(), chains.accepted.to_host())
    npt.assert_allclose(res[1][0], res[0][0], rtol=1e-8, atol=1e-9)
    npt.assert_allclose(res[1][1][:, 1:], res[0][1][:, 1:], rtol=1e-9)
    npt.assert_allclose(res[1][1][:, 0], res[0][1][:, 0], atol=1e-6)
    npt.assert_array_equal(res[1][2], res[0][2])


def test_sequence_dependent_parameters_are_not_shared():
    """a per-sequence parameter array (chain stride != 0) must never take the shared form"""
    from aux_ssm_samplers_amd import _lib
    import aux_ssm_samplers_amd._primitives.kalman as P
    rng = np.random.default_rng(5)
    d, p, T = 6, 9, 30
    y0, lg = stable_model(rng, T, d, p, nan=False)
    h = _lib.default_handle()
    h.prof_enable(_lib.K_ALL, 64)
    try:
        P.filtering(y0, P.LGSSM(*lg), True)
        assert "filter_tab" not in h.prof_read_groups()
    finally:
        h.prof_disable()


@pytest.mark.parametrize("dtype,d,T,S", [(np.float32, 64, 96, 16), (np.float64, 40, 40, 4), (np.float32, 64, 40, 70)])
def test_C5_sizes(dtype, d, T, S):
    u, lg64, x = c5_model(T, d)
    rng = np.random.default_rng(2)
    ys = u[None] + 0.3 * rng.standard_normal((S, T, d))
    check([np.ascontiguousarray(a) for a in lg64], ys, dtype)


def test_C5_benchmarked_horizon_16_sequences():
    """T = 8192, d = p = 64, fp32, 16 sequences (what bench.py's C5 leg times): sequence 0 carries the fixture's observations and is checked against the
    committed fp64 sequential answers (tests/golden/c5_T8192_known_answers.npz, tolerance 2e-3 as in tests/test_gpu_wide.py); the other sequences
    against the per-sequence path of the library at a few time points."""
    import os
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "c5_T8192_known_answers.npz"))
    T, d, S = 8192, 64, 16
    u, lg64, _ = c5_model(T, d)
    rng = np.random.default_rng(4)
    ys = u[None] + np.concatenate([np.zeros((1, T, d)), 0.3 * rng.standard_normal((S - 1, T, d))])
    lg = [np.ascontiguousarray(a) for a in lg64]
    ms, Ps, ell, groups = device_filter(lg, ys, np.float32, True)
    assert "filter_tab" in groups
    idx = ref["idx"]
    tol = dict(rtol=2e-3, atol=2e-3)
    npt.assert_allclose(ms[0][idx], ref["ms"], **tol)
    npt.assert_allclose(np.einsum("tii->ti", Ps[0][idx]), ref["Ps_diag"], **tol)
    assert abs(float(ell[0]) - float(ref["ell"])) / abs(float(ref["ell"])) < 1e-4
    e_m = np.max(np.abs(ms[0][idx] - ref["ms"]))
    ms1, Ps1, ell1, _ = device_filter(lg, ys[[0, 5, 15]], np.float32, False)
    for k, s in enumerate([0, 5, 15]):
        npt.assert_allclose(ms[s][idx], ms1[k][idx], **tol)
        npt.assert_allclose(Ps[s][idx], Ps1[k][idx], **tol)
        assert abs(float(ell[s]) - float(ell1[k])) / abs(float(ell1[k])) < 1e-4
    print(f"shared wide filter, T=8192 x 16: max|dm| vs fp64 fixture {e_m:.2e}; groups {groups}")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("d,po,T,C", [(8, 8, 80, 5), (12, 7, 257, 3), (33, 20, 60, 18), (30, 8, 40, 4), (16, 5, 700, 70)])
def test_wide_lg_sweep_one_covariance_copy_and_shared_sampler_tables(d, po, T, C, dtype):
    """The LG_CONCAT sweep at dx > 4 with several dense chains on one model: ONE copy of the filtered covariances (the shared filter's
    broadcast to the chains' slots is gone, its pattern read-back replaced by a carrier), and the pathwise sampler builds its gains / Cholesky factors once per time
    step, the chains riding as columns of d x CB products (wide.hip::run_sample_shared).  Against the per-chain path (AUXSSM_OPT_SHARE_MODEL = 0) on the same explicit
    noise, and against the oracle's sweep chain by chain (fp64); missing observations included."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from tests.test_gpu_wide import lg_concat_wide
    from oracle import kalman_np as K
    model, xt, y, _ = lg_concat_wide(T, d, po, seed=d)
    y = y.copy()
    y[3] = np.nan
    y[5, : po // 2] = np.nan
    model = LGConcatModel(model.m0, model.P0, model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs, y)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    rng = np.random.default_rng(d + T)
    x0 = (xt[None] + 0.3 * rng.standard_normal((C, T, d))).astype(dtype)
    noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
    h = _lib.default_handle()
    res = {}
    for share in (1, 0):
        chains = DeviceChains(h, x0)
        assert not chains.chain_minor
        h.set_option(_lib.OPT_SHARE_MODEL, share)
        try:
            kernel(None, KalmanSampler(x=chains, updated=None), 0.4, noise=noise)
        finally:
            h.set_option(_lib.OPT_SHARE_MODEL, 1)
        res[share] = (chains.to_host(), chains.logs.to_host(), chains.accepted.to_host())
    tol = dict(rtol=1e-8, atol=1e-9) if dtype == np.float64 else dict(rtol=2e-3, atol=2e-3)
    npt.assert_allclose(res[1][0], res[0][0], **tol)
    if dtype == np.float64:
        npt.assert_allclose(res[1][1][:, 1:], res[0][1][:, 1:], rtol=1e-9)
        npt.assert_array_equal(res[1][2], res[0][2])
        # (log alpha is not 0 here: under the reference's NaN policy a step with a missing COMPONENT drops its auxiliary term too -- base.py:159-166)
        for c in range(0, C, 4):
            lgo = (model.m0, model.P0, model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
            ref = K.kalman_sweep(x0[c].astype(np.float64), 0.4, model.dynamics_factory, model.observations_factory,
                                 lambda z: K.log_likelihood(y, z, lgo) + K.prior_logpdf(z, lgo), True,
                                 eps_aux=noise["eps_aux"][c], eps_samp=noise["eps_samp"][c], u_accept=noise["u_accept"][c])
            assert bool(res[1][2][c]) == ref["accepted"]
            npt.assert_allclose(res[1][0][c], ref["x"], rtol=1e-8, atol=1e-9)
            npt.assert_allclose(res[1][1][c, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-9)


@pytest.mark.parametrize("d,po,T,C,dtype", [(62, 6, 40, 3, np.float32), (64, 4, 33, 2, np.float32), (34, 34, 30, 3, np.float64), (20, 50, 25, 2, np.float64)])
def test_more_than_64_observations_split_elimination(d, po, T, C, dtype):
    """d + po > 64 concatenated observations (a d = 64 sweep with any real observation): the SPD eliminations of the gain table, the t = 0 update and the information
    rows are split once by a Schur complement so that both halves run the blocked (n <= 64) elimination (wide.hip::spd_solve).  The sweep against the oracle's, chain
    by chain, on explicit noise -- fp64 at 1e-8, fp32 at its accuracy -- with a missing observation row and a missing component."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from tests.test_gpu_wide import lg_concat_wide
    from oracle import kalman_np as K
    model, xt, y, _ = lg_concat_wide(T, d, po, seed=d + po)
    y = y.copy()
    y[2] = np.nan
    y[4, 0] = np.nan
    model = LGConcatModel(model.m0, model.P0, model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs, y)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    rng = np.random.default_rng(d * po)
    x0 = (xt[None] + 0.3 * rng.standard_normal((C, T, d))).astype(dtype)
    noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
    h = _lib.default_handle()
    try:
        chains = DeviceChains(h, x0)
        kernel(None, KalmanSampler(x=chains, updated=None), 0.4, noise=noise)
    except ValueError as e:
        if "LDS" in str(e):
            pytest.skip(str(e))
        raise
    xs, logs, acc = chains.to_host(), chains.logs.to_host(), chains.accepted.to_host()
    lgo = (model.m0, model.P0, model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
    tol = dict(rtol=1e-8, atol=1e-9) if dtype == np.float64 else dict(rtol=5e-3, atol=5e-3)
    for c in range(C):
        ref = K.kalman_sweep(x0[c].astype(np.float64), 0.4, model.dynamics_factory, model.observations_factory,
                             lambda z: K.log_likelihood(y, z, lgo) + K.prior_logpdf(z, lgo), True,
                             eps_aux=noise["eps_aux"][c], eps_samp=noise["eps_samp"][c], u_accept=noise["u_accept"][c])
        if dtype == np.float64:
            assert bool(acc[c]) == ref["accepted"]
            npt.assert_allclose(logs[c, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-9)
        if bool(acc[c]) == ref["accepted"]:
            npt.assert_allclose(xs[c], ref["x"], **tol)

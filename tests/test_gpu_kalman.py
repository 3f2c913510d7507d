"""GPU parity tests of the auxiliary-Kalman path: HIP kernels through the C ABI (ctypes) vs the NumPy oracle on
the same seeded inputs -- the reference's own test parametrisations, chain-batched layouts, sizes that exercise
every level of the chunked scan, and size-independent properties at the BASELINE C2 size.

Tolerances: fp64 rtol 1e-8 / atol 1e-10 (SURVEY 8d asks 1e-9/1e-10 for x_prop; asserted where stated);
fp32 rtol 2e-3 on filter moments of ill-conditioned random test models, 2e-4 on the well-conditioned C2 model.
"""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K
from tests.helpers import ref_lgssm_inputs, ref_batched_inputs, lg_model

pytestmark = pytest.mark.gpu
TOL64 = dict(rtol=1e-8, atol=1e-10)


@pytest.fixture(scope="module")
def P():
    import aux_ssm_samplers_amd._primitives.kalman as prim
    return prim


@pytest.mark.parametrize("seed", [0, 1234])
@pytest.mark.parametrize("T", [5, 7])
@pytest.mark.parametrize("dx", [1, 2])
@pytest.mark.parametrize("dy", [1, 3])
@pytest.mark.parametrize("parallel", [False, True])
@pytest.mark.parametrize("nan_index", [True, False])
def test_filter_vs_oracle_reference_cases(P, seed, T, dx, dy, parallel, nan_index):
    ys, lg = ref_lgssm_inputs(seed, T, dx, dy, nan_index)
    ms, Ps, ell = P.filtering(ys, P.LGSSM(*lg), parallel)
    oms, oPs, oell = K.filtering(ys, lg, parallel)
    npt.assert_allclose(ms, oms, **TOL64)
    npt.assert_allclose(Ps, oPs, **TOL64)
    npt.assert_allclose(ell, oell, **TOL64)
    # and against the reference's own independent answer (test_filtering.py:52-55, rtol 1e-7)
    m0, P0, Fs, Qs, bs, Hs, Rs, cs = lg
    ems, ePs, eell = K.explicit_filter(ys, m0, P0, Hs, Rs, cs, Fs, Qs, bs)
    npt.assert_allclose(ms, ems, rtol=1e-6, atol=1e-9)
    npt.assert_allclose(ell, eell, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("seed", [0, 1234])
@pytest.mark.parametrize("T", [3, 5])
@pytest.mark.parametrize("dx,dy", [(1, 1), (2, 3), (1, 3), (2, 1)])
@pytest.mark.parametrize("parallel", [True, False])
def test_filter_batched_model(P, seed, T, dx, dy, parallel):
    B = 3
    (bys, blg), (ys, lg) = ref_batched_inputs(seed, T, dx, dy, B)
    bms, bPs, bell = P.filtering(bys, P.LGSSM(*blg), parallel)
    oms, oPs, oell = K.filtering(bys, blg, parallel)
    npt.assert_allclose(bms, oms, **TOL64)
    npt.assert_allclose(bPs, oPs, **TOL64)
    npt.assert_allclose(bell, oell, **TOL64)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("T,B", [(1, 300), (2, 256), (7, 300), (130, 257), (1000, 64)])
@pytest.mark.parametrize("dx,dy", [(1, 1), (2, 3), (3, 1)])
@pytest.mark.parametrize("parallel", [True, False])
def test_filter_wide_batch_axis_runs_one_sequence_per_lane(P, T, B, dx, dy, parallel, dtype):
    """B >= 32 independent models side by side (the reference's spatial example: 64 scalar LGSSMs, examples/spatial/model.py:103-112): `auxssm_kalman_filter` maps
    lanes to (c, b) sequences -- no element buffer, composites from elements built in registers, then the SEQUENTIAL recursion from every chunk's prefix
    (kernels.hip.h::run_filter, kalman_bodies.h::FilterOpSeqWalk).  Against the oracle's batched filter, with missing observations and several sequences of
    observations on shared parameters."""
    Cn = 2 if T <= 130 else 4
    (bys, blg), _ = ref_batched_inputs(5 + T + B, max(T, 2), dx, dy, B)
    blg = list(blg)
    blg[2] = 0.6 * blg[2] / np.sqrt(dx)                      # contracting transitions: T = 1000 steps stay in range
    blg = [a[:T - 1] if k in (2, 3, 4) else (a[:T] if k >= 5 else a) for k, a in enumerate(blg)]
    rng = np.random.default_rng(T * B)
    ys = bys[None, :T] + 0.3 * rng.standard_normal((Cn, T, B, dy))
    ys[rng.random(ys.shape) < 0.1] = np.nan                 # missing components, whole missing observations among them
    if T > 2:
        ys[:, 2] = np.nan
    import ctypes as C
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd._primitives.kalman.base import DeviceLGSSM
    h = _lib.default_handle()
    dl = DeviceLGSSM(h, tuple(blg), 1, T, B, dx, dy, True, dtype)           # parameters shared by the Cn sequences of observations (chain stride 0)
    yd = h.to_device(ys.astype(dtype))
    msd, Psd, elld = h.empty((Cn, T, B, dx), dtype), h.empty((Cn, T, B, dx, dx), dtype), h.empty((Cn,), dtype)
    dims = _lib.Dims(Cn, T, B, dx, dy)
    yarr = yd.arr(T * B * dy, B * dy, dy)
    _lib.check(h.lib.auxssm_kalman_filter(h.h, _lib.dtype_code(dtype), C.byref(dims), C.byref(dl.c), C.byref(yarr), int(parallel), msd.ptr, Psd.ptr, elld.ptr))
    ms, Ps, ell = msd.to_host(), Psd.to_host(), elld.to_host()
    tol = TOL64 if dtype == np.float64 else dict(rtol=2e-3, atol=2e-3)
    for c in range(Cn):
        oms, oPs, oell = K.filtering(ys[c], blg, parallel)
        npt.assert_allclose(ms[c], oms, **tol)
        npt.assert_allclose(Ps[c], oPs, **tol)
        npt.assert_allclose(ell[c], oell, rtol=tol["rtol"], atol=tol["atol"] * T * B * 0.05 + tol["atol"])


@pytest.mark.parametrize("seed", [42, 666])
@pytest.mark.parametrize("T", [3, 5, 300])
@pytest.mark.parametrize("dx", [1, 2])
@pytest.mark.parametrize("parallel", [True, False])
def test_sampler_vs_oracle(P, seed, T, dx, parallel):
    ys, lg = ref_lgssm_inputs(seed, T, dx, 3)
    ms, Ps, _ = K.filtering(ys, lg, False)
    eps = np.random.default_rng(seed).standard_normal((T, dx))
    xs = P.sampling(None, ms, Ps, P.LGSSM(*lg), parallel, eps=eps)
    npt.assert_allclose(xs, K.sampling(eps, ms, Ps, lg, parallel), rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("seed", [42, 666])
@pytest.mark.parametrize("parallel", [True, False])
def test_sampler_batched_equals_block_diag(P, seed, parallel):
    T, dx, dy, B = 5, 2, 3, 3
    (bys, blg), (ys, lg) = ref_batched_inputs(seed, T, dx, dy, B)
    bms, bPs, _ = K.filtering(bys, blg, False)
    ms, Ps, _ = K.filtering(ys, lg, False)
    eps = np.random.default_rng(seed).standard_normal((T, B, dx))
    bx = P.sampling(None, bms, bPs, P.LGSSM(*blg), parallel, eps=eps)
    npt.assert_allclose(bx.reshape(T, B * dx), K.sampling(eps.reshape(T, B * dx), ms, Ps, lg, parallel), atol=1e-10, rtol=1e-10)


@pytest.mark.parametrize("nan_index", [True, False])
@pytest.mark.parametrize("dx,dy", [(1, 1), (2, 3), (1, 3), (2, 1)])
def test_posterior_logpdf(P, nan_index, dx, dy):
    T = 7
    ys, lg = ref_lgssm_inputs(5, T, dx, dy, nan_index)
    xs = np.random.default_rng(0).standard_normal((T, dx))
    _, _, ell = K.filtering(ys, lg, False)
    npt.assert_allclose(P.posterior_logpdf(ys, xs, ell, P.LGSSM(*lg)), K.posterior_logpdf(ys, xs, ell, lg), **TOL64)
    npt.assert_allclose(P.prior_logpdf(xs, P.LGSSM(*lg)), K.prior_logpdf(xs, lg), **TOL64)
    npt.assert_allclose(P.log_likelihood(ys, xs, P.LGSSM(*lg)), K.log_likelihood(ys, xs, lg), rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("d,T", [(1, 1000), (2, 4097), (3, 2500), (4, 20000)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_all_scan_levels_time_varying(P, d, T, dtype):
    """Random stable time-varying model, T large enough for many chunks per sequence and a multi-lane aggregate
    scan; NaN rows sprinkled in.  Oracle = its parallel (associative_scan) path in fp64."""
    rng = np.random.default_rng(d * 1000 + T)
    p = d + 1
    Fs = 0.5 * rng.standard_normal((T - 1, d, d)) / np.sqrt(d)
    A = rng.standard_normal((T - 1, d, 2 * d))
    Qs = A @ A.transpose(0, 2, 1) / (2 * d) + 0.1 * np.eye(d)
    bs = rng.standard_normal((T - 1, d))
    Hs = rng.standard_normal((T, p, d))
    Bm = rng.standard_normal((T, p, 2 * p))
    Rs = Bm @ Bm.transpose(0, 2, 1) / (2 * p) + 0.1 * np.eye(p)
    cs = rng.standard_normal((T, p))
    ys = rng.standard_normal((T, p))
    ys[rng.random(T) < 0.1] = np.nan
    ys[rng.random((T, p)) < 0.05] = np.nan
    ys[0] = rng.standard_normal(p)
    m0 = rng.standard_normal(d)
    P0 = np.eye(d)
    lg64 = (m0, P0, Fs, Qs, bs, Hs, Rs, cs)
    lg = P.LGSSM(*[a.astype(dtype) for a in lg64])
    oms, oPs, oell = K.filtering(ys, lg64, True)
    ms, Ps, ell = P.filtering(ys.astype(dtype), lg, True)
    tol = TOL64 if dtype == np.float64 else dict(rtol=2e-3, atol=2e-3)
    npt.assert_allclose(ms, oms, **tol)
    npt.assert_allclose(Ps, oPs, **tol)
    npt.assert_allclose(ell, oell, rtol=tol["rtol"])
    eps = rng.standard_normal((T, d))
    xs = P.sampling(None, oms.astype(dtype), oPs.astype(dtype), lg, True, eps=eps.astype(dtype))
    npt.assert_allclose(xs, K.sampling(eps, oms, oPs, lg64, True), **tol)
    if dtype == np.float64:
        seq = P.filtering(ys, lg, False)
        npt.assert_allclose(seq[0], ms, rtol=1e-9, atol=1e-10)
        npt.assert_allclose(seq[2], ell, rtol=1e-10)


@pytest.mark.parametrize("d,T", [(2, 4097), (4, 3000)])
def test_tile_scan_of_few_sequences_equals_the_chunked_scan(P, d, T):
    """Few sequences take the Kogge-Stone tile scan (kernels.hip.h::k_ks_tile: at most two tiles of 256 elements per CU), many the chunked three-launch scan: the same
    model as ONE sequence (tile scan) and as 48 copies on the chain axis (chunked scan: 48 x 17 tiles > 512) gives the same filter and the same pathwise sample to rounding,
    NaN rows included; the sequential recursion agrees with both."""
    rng = np.random.default_rng(T + d)
    p = d + 1
    Fs = 0.6 * rng.standard_normal((T - 1, d, d)) / np.sqrt(d)
    A = rng.standard_normal((T - 1, d, 2 * d))
    Qs = A @ A.transpose(0, 2, 1) / (2 * d) + 0.1 * np.eye(d)
    bs = rng.standard_normal((T - 1, d))
    Hs = rng.standard_normal((T, p, d))
    Rs = np.broadcast_to(0.3 * np.eye(p), (T, p, p))
    cs = rng.standard_normal((T, p))
    ys = rng.standard_normal((T, p))
    ys[rng.random(T) < 0.1] = np.nan
    ys[0] = rng.standard_normal(p)
    lg = P.LGSSM(rng.standard_normal(d), np.eye(d), Fs, Qs, bs, Hs, Rs, cs)
    ms1, Ps1, ell1 = P.filtering(ys, lg, True)                               # one sequence: tiles
    seq = P.filtering(ys, lg, False)
    npt.assert_allclose(ms1, seq[0], rtol=1e-9, atol=1e-10)
    npt.assert_allclose(Ps1, seq[1], rtol=1e-9, atol=1e-10)
    npt.assert_allclose(ell1, seq[2], rtol=1e-10)
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd._primitives.kalman.base import DeviceLGSSM
    import ctypes as C
    h = _lib.default_handle()
    Cn = 48
    dl = DeviceLGSSM(h, tuple(lg), 1, T, 1, d, p, False, np.float64)
    yd = h.to_device(np.ascontiguousarray(np.broadcast_to(ys, (Cn, T, p))))
    yarr = yd.arr(T * p, p, 0)
    ms, Ps, ell = h.empty((Cn, T, 1, d), np.float64), h.empty((Cn, T, 1, d, d), np.float64), h.empty((Cn,), np.float64)
    dims = _lib.Dims(Cn, T, 1, d, p)
    h.set_option(_lib.OPT_SHARE_MODEL, 0)                                    # (per-sequence matrix recursions: the general scan, not the gain form)
    try:
        _lib.check(h.lib.auxssm_kalman_filter(h.h, _lib.F64, C.byref(dims), C.byref(dl.c), C.byref(yarr), 1, ms.ptr, Ps.ptr, ell.ptr))
    finally:
        h.set_option(_lib.OPT_SHARE_MODEL, 1)
    msh, Psh, ellh = ms.to_host(), Ps.to_host(), ell.to_host()
    for c in (0, 17, 47):
        npt.assert_allclose(msh[c, :, 0], ms1, rtol=1e-9, atol=1e-10)
        npt.assert_allclose(Psh[c, :, 0], Ps1, rtol=1e-9, atol=1e-10)
        npt.assert_allclose(ellh[c], ell1, rtol=1e-10)
    eps = rng.standard_normal((T, d))
    x1 = P.sampling(None, ms1, Ps1, lg, True, eps=eps)
    npt.assert_allclose(x1, P.sampling(None, ms1, Ps1, lg, False, eps=eps), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("T,dx", [(1, 2), (2, 1), (3, 2), (5, 3), (8, 4), (33, 2), (100, 1), (257, 4), (1000, 3)])
def test_dnc_sampler_device_vs_oracle(T, dx, dtype):
    """auxssm_kalman_dnc_sample (csrc/dnc.hip: the reference's divide-and-conquer sampler, dnc_sampling.py:17-186) against its NumPy restatement on explicit noise:
    every tree shape incl. odd interval counts on several levels; same warning, batched input refused as the reference does (:42-43)."""
    import warnings
    import aux_ssm_samplers_amd._primitives.kalman as P
    from aux_ssm_samplers_amd._primitives.kalman import dnc_sampling
    rng = np.random.default_rng(T * 10 + dx)
    Fs = 0.7 * rng.standard_normal((max(T - 1, 1), dx, dx)) / np.sqrt(dx)
    A = rng.standard_normal((max(T - 1, 1), dx, 2 * dx))
    Qs = A @ A.transpose(0, 2, 1) / (2 * dx) + 0.2 * np.eye(dx)
    bs = rng.standard_normal((max(T - 1, 1), dx))
    Hs, Rs, cs = rng.standard_normal((T, 2, dx)), np.broadcast_to(0.5 * np.eye(2), (T, 2, 2)), np.zeros((T, 2))
    lg = (rng.standard_normal(dx), np.eye(dx), Fs[:T - 1], Qs[:T - 1], bs[:T - 1], Hs, Rs, cs)
    ms, Ps, _ = K.filtering(rng.standard_normal((T, 2)), lg, False)
    eps = rng.standard_normal((T, dx))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        xs = dnc_sampling.sampling(None, ms.astype(dtype), Ps.astype(dtype), P.LGSSM(*[a.astype(dtype) for a in lg]), eps=eps.astype(dtype))
    assert any("proof-of-concept" in str(x.message) for x in w)
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == np.float64 else dict(rtol=5e-4, atol=5e-4)
    npt.assert_allclose(xs, K.dnc_sampling(eps, ms, Ps, lg), **tol)
    with pytest.raises(ValueError):
        dnc_sampling.sampling(None, np.zeros((3, 2, 2)), np.zeros((3, 2, 2, 2)), P.LGSSM(*lg))


def test_dnc_sampler_reference_statistical_test():
    """test_sampling.py::test_parallel_vs_sequential with mode = "dnc" (:23-68) on the device: 200 000 chains in one call (the C entry point takes a chain axis), keyed
    device noise, against the RTS smoother (`explicit_kalman_smoothing`) at the reference's tolerance (atol = rtol = 1e-2)."""
    import ctypes as C
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd._primitives.kalman.base import DeviceLGSSM
    T, dx, n = 5, 2, 200_000
    ys, lg = ref_lgssm_inputs(42, T, dx, 3)
    ms, Ps, _ = K.filtering(ys, lg, False)
    sm, sP = K.explicit_smoother(ms, Ps, lg[2], lg[3], lg[4])
    h = _lib.default_handle()
    dl = DeviceLGSSM(h, list(lg[:5]) + [None, None, None], 1, T, 1, dx, 1, False, np.float64)
    msd = h.to_device(np.ascontiguousarray(np.broadcast_to(ms, (n, T, dx))))
    Psd = h.to_device(np.ascontiguousarray(np.broadcast_to(Ps, (n, T, dx, dx))))
    eps = h.rng_normal(R.PRNGKey(3), 0, (n, T, dx), np.float64)
    xs = h.empty((n, T, dx), np.float64)
    dims = _lib.Dims(n, T, 1, dx, 1)
    _lib.check(h.lib.auxssm_kalman_dnc_sample(h.h, _lib.F64, C.byref(dims), C.byref(dl.c), msd.ptr, Psd.ptr, eps.ptr, xs.ptr))
    X = xs.to_host()
    npt.assert_allclose(X.mean(0), sm, atol=1e-2, rtol=1e-2)
    for t in range(T):
        npt.assert_allclose(np.cov(X[:, t].T), sP[t], atol=1e-2, rtol=1e-2)


def _lg_concat(T, d, dtype=np.float64):
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    m = lg_model(T, d, dtype=dtype)
    bt = np.broadcast_to
    return m, LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)),
                            bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), m["y"])


@pytest.mark.parametrize("d,T", [(2, 1024), (4, 600), (1, 50)])
@pytest.mark.parametrize("parallel", [True, False])
def test_device_sweep_vs_oracle_sweep(d, T, parallel):
    """BASELINE config C1 (T=1024, d=2) and friends: the fused device sweep vs the oracle's restatement of
    kalman/generic.py:53-106 on identical explicit noise; x_prop at fp64 rtol 1e-9 / atol 1e-10."""
    from aux_ssm_samplers_amd.kalman import get_kernel
    m, model = _lg_concat(T, d)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, parallel)
    rng = np.random.Generator(np.random.PCG64(1000))
    x = m["x_true"] + 0.3 * rng.standard_normal((T, d))
    delta = 0.5
    noise = dict(eps_aux=rng.standard_normal((T, d)), eps_samp=rng.standard_normal((T, d)), u_accept=rng.random())
    lgo = (m["m0"], m["P0"], model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
    ref = K.kalman_sweep(x, delta, model.dynamics_factory, model.observations_factory,
                         lambda z: K.log_likelihood(m["y"], z, lgo) + K.prior_logpdf(z, lgo), parallel, **noise)
    out = kernel(None, init(x), delta, noise=noise)
    npt.assert_allclose(out.x, ref["x"], rtol=1e-9, atol=1e-10)
    assert out.updated == ref["accepted"]
    npt.assert_allclose(out.logs[0, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-9)
    # linear-Gaussian model with exact proposal: the MH ratio is identically 1 (SURVEY 8c known answer)
    assert abs(out.log_alpha) < 1e-7 and abs(ref["log_alpha"]) < 1e-7
    assert out.updated


def test_host_factory_path_equals_device_sweep():
    from aux_ssm_samplers_amd.kalman import get_kernel
    T, d = 200, 2
    m, model = _lg_concat(T, d)
    rng = np.random.default_rng(5)
    x = m["x_true"] + 0.3 * rng.standard_normal((T, d))
    noise = dict(eps_aux=rng.standard_normal((T, d)), eps_samp=rng.standard_normal((T, d)), u_accept=0.3)
    init, kdev = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    # wrapping the factories in lambdas hides the device model -> generic host-factory path
    init2, khost = get_kernel(lambda z: model.dynamics_factory(z), lambda z, u, dl: model.observations_factory(z, u, dl),
                              lambda z: model.log_likelihood_fn(z), True)
    a = kdev(None, init(x), 0.5, noise=noise)
    b = khost(None, init2(x), 0.5, noise=noise)
    npt.assert_allclose(a.x, b.x, rtol=1e-9, atol=1e-10)
    assert a.updated == b.updated
    assert abs(b.log_alpha) < 1e-7


def test_rng_device_matches_host_threefry():
    from aux_ssm_samplers_amd import _lib, random as R
    h = _lib.default_handle()
    key = R.PRNGKey(123456789012345)
    n = 4096
    u = h.rng_uniform(key, 7, (n,), np.float64).to_host()
    a, b = R.threefry2x32(key[0], key[1], np.arange(n // 2, dtype=np.uint32), np.full(n // 2, 7, np.uint32))
    npt.assert_array_equal(u, np.stack([a, b], axis=1).reshape(-1).astype(np.float64) * 2.3283064365386963e-10)  # block i>>1, word i&1
    z = h.rng_normal(key, 3, (200000,), np.float32).to_host()
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, None)])
def test_full_size_C2_properties(dtype, tol):
    """BASELINE config C2 (T=65536, d=4), 4 chains: size-independent properties.
    (i) log alpha == 0 and every chain accepts; (ii) parallel == sequential trajectories; (iii) the proposal is
    exact: resampling with eps_samp = 0 returns the smoother mean, which is a fixed point of a second sweep."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel
    T, d, C = 65536, 4, 4
    m, model = _lg_concat(T, d, dtype)
    rng = np.random.default_rng(11)
    x = (m["x_true"][None] + 0.3 * rng.standard_normal((C, T, d))).astype(dtype)
    noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
    init, kpar = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    _, kseq = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, False)
    a = kpar(None, init(x), 0.5, noise=noise)
    b = kseq(None, init(x), 0.5, noise=noise)
    if dtype == np.float64:
        assert np.all(np.abs(a.log_alpha) < tol), a.log_alpha
        assert a.updated.all() and b.updated.all()
        npt.assert_allclose(a.x, b.x, rtol=1e-9, atol=1e-9)
    else:
        # fp32: log-densities are sums of 65536 O(1) terms -> absolute error O(1e-1); trajectories agree to 2e-4
        assert np.all(np.abs(a.log_alpha) < 5.0), a.log_alpha
        npt.assert_allclose(kpar(None, init(x), 0.5, noise=dict(noise, u_accept=np.zeros(C))).x,
                            kseq(None, init(x), 0.5, noise=dict(noise, u_accept=np.zeros(C))).x, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("seed", [42, 666])
@pytest.mark.parametrize("T,dx,dy", [(3, 1, 1), (5, 2, 3)])
@pytest.mark.parametrize("parallel", [True, False])
def test_sampler_moments_reference_statistical_test(seed, T, dx, dy, parallel):
    """aux_samplers/_primitives/test_kalman/test_sampling.py::test_parallel_vs_sequential (:23-68) on the HIP path: the
    empirical mean / covariance of 500 000 pathwise samples (device Threefry noise, one launch over 500 000 chains) match
    the RTS smoother at atol = rtol = 1e-2."""
    import ctypes as C
    from aux_ssm_samplers_amd import _lib, _layout, random as R
    from aux_ssm_samplers_amd._primitives.kalman.base import DeviceLGSSM
    ys, lg = ref_lgssm_inputs(seed, T, dx, dy)
    ms, Ps, _ = K.filtering(ys, lg, False)
    esm, esP = K.explicit_smoother(ms, Ps, lg[2], lg[3], lg[4])
    h = _lib.default_handle()
    NS = 500_000
    dl = DeviceLGSSM(h, list(lg[:5]) + [None, None, None], 1, T, 1, dx, 1, False, np.float64)  # chain-shared (stride 0)
    msd = h.to_device(np.broadcast_to(ms, (NS, T, dx)).copy())
    Psd = h.to_device(np.broadcast_to(Ps, (NS, T, dx, dx)).copy())
    eps = h.rng_normal(R.PRNGKey(seed), 0, (NS, T, 1, dx), np.float64)
    xs = h.empty((NS, T, 1, dx), np.float64)
    dims = _lib.Dims(NS, T, 1, dx, 1)
    _lib.check(h.lib.auxssm_kalman_sample(h.h, _lib.F64, C.byref(dims), C.byref(dl.c), msd.ptr, Psd.ptr, eps.ptr, int(parallel), xs.ptr))
    s = xs.to_host()[:, :, 0, :]
    npt.assert_allclose(s.mean(0), esm, atol=1e-2, rtol=1e-2)
    cov = np.einsum("nti,ntj->tij", s - s.mean(0), s - s.mean(0)) / (NS - 1)
    npt.assert_allclose(cov, esP, atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("d,T,time_varying", [(4, 700, False), (2, 333, False), (1, 200, False), (4, 260, True), (2, 150, True)])
@pytest.mark.parametrize("parallel", [True, False])
def test_chain_minor_sweep_equals_dense_sweep_and_oracle(d, T, parallel, time_varying):
    """The layout bench.py runs (>= 32 chains: state, noise and every internal buffer chain-minor, lanes over chains) against the
    dense layout chain by chain and against the oracle sweep, on identical explicit noise, missing observation rows included."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import get_kernel, LGConcatModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    C = 70  # more than one wave of chains, not a multiple of 64
    m = lg_model(T, d)
    y = m["y"].copy()
    rng = np.random.default_rng(d * 100 + T)
    y[rng.random(T) < 0.15] = np.nan      # whole steps missing
    if d > 1:
        y[rng.random((T, d)) < 0.05] = np.nan  # single components missing
    bt = np.broadcast_to
    if time_varying:  # chain-shared but different at every time step (materialised arrays, time stride != 0)
        Fs = m["F"][None] * (1 + 0.1 * rng.standard_normal((T - 1, 1, 1))) + 0.02 * rng.standard_normal((T - 1, d, d))
        A = rng.standard_normal((T - 1, d, 2 * d))
        Qs = 0.05 * A @ A.transpose(0, 2, 1) / d + 0.05 * np.eye(d)
        bs = 0.1 * rng.standard_normal((T - 1, d))
        Hs = np.eye(d)[None] + 0.2 * rng.standard_normal((T, d, d))
        Bm = rng.standard_normal((T, d, 2 * d))
        Rs = 0.2 * Bm @ Bm.transpose(0, 2, 1) / d + 0.2 * np.eye(d)
        cs = 0.1 * rng.standard_normal((T, d))
        model = LGConcatModel(m["m0"], m["P0"], Fs, Qs, bs, Hs, Rs, cs, y)
    else:
        model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)),
                              bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), y)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, parallel)
    x0 = m["x_true"][None] + 0.3 * rng.standard_normal((C, T, d))
    noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
    h = _lib.default_handle()
    outs = {}
    for cm in (True, False):
        chains = DeviceChains(h, x0, chain_minor=cm)
        assert chains.chain_minor == cm
        st = kernel(None, KalmanSampler(x=chains, updated=None), 0.5, noise=noise)
        outs[cm] = (chains.to_host(), chains.accepted.to_host(), chains.logs.to_host())
    npt.assert_allclose(outs[True][0], outs[False][0], rtol=1e-9, atol=1e-10)
    npt.assert_array_equal(outs[True][1], outs[False][1])
    npt.assert_allclose(outs[True][2][:, 1:], outs[False][2][:, 1:], rtol=1e-9)
    # the chain-minor run above hoisted the chain-shared model parameters (AUXSSM_OPT_SHARE_MODEL, default): the general per-chain
    # path of the same layout must give the same sweep
    h.set_option(_lib.OPT_SHARE_MODEL, 0)
    try:
        chains = DeviceChains(h, x0, chain_minor=True)
        kernel(None, KalmanSampler(x=chains, updated=None), 0.5, noise=noise)
        gen = (chains.to_host(), chains.accepted.to_host(), chains.logs.to_host())
    finally:
        h.set_option(_lib.OPT_SHARE_MODEL, 1)
    npt.assert_allclose(outs[True][0], gen[0], rtol=1e-9, atol=1e-10)
    npt.assert_array_equal(outs[True][1], gen[1])
    npt.assert_allclose(outs[True][2][:, 1:], gen[2][:, 1:], rtol=1e-9)
    lgo = (m["m0"], m["P0"], model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
    for c in (0, 33, C - 1):
        ref = K.kalman_sweep(x0[c], 0.5, model.dynamics_factory, model.observations_factory,
                             lambda z: K.log_likelihood(y, z, lgo) + K.prior_logpdf(z, lgo), parallel,
                             eps_aux=noise["eps_aux"][c], eps_samp=noise["eps_samp"][c], u_accept=noise["u_accept"][c])
        npt.assert_allclose(outs[True][0][c], ref["x"], rtol=1e-9, atol=1e-10)
        npt.assert_allclose(outs[True][2][c, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-9)
        assert bool(outs[True][1][c]) == ref["accepted"]


@pytest.mark.parametrize("dtype,nan_policy", [(np.float64, "masked"), (np.float32, "reference"), (np.float32, "masked")])
def test_shared_model_mode_equals_general_path(dtype, nan_policy):
    """AUXSSM_OPT_SHARE_MODEL on vs off (chain-minor layout), fp32 and the masked NaN policy included: same proposals, same
    acceptances, same log terms up to rounding (fp64 1e-9; fp32: the two paths order 65k-term sums differently -> 2e-3 relative)."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler, _get_device_kernel
    d, T, C = 4, 500, 96
    m = lg_model(T, d)
    y = m["y"].copy()
    rng = np.random.default_rng(17)
    y[rng.random(T) < 0.1] = np.nan
    y[rng.random((T, d)) < 0.05] = np.nan
    bt = np.broadcast_to
    model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)),
                          bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), y)
    init, kernel = _get_device_kernel(model, True, nan_policy)
    x0 = (m["x_true"][None] + 0.3 * rng.standard_normal((C, T, d))).astype(dtype)
    noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
    h = _lib.default_handle()
    res = {}
    try:
        for share in (1, 0):
            h.set_option(_lib.OPT_SHARE_MODEL, share)
            chains = DeviceChains(h, x0, chain_minor=True)
            kernel(None, KalmanSampler(x=chains, updated=None), 0.5, noise=noise)
            res[share] = (chains.to_host(), chains.accepted.to_host(), chains.logs.to_host())
    finally:
        h.set_option(_lib.OPT_SHARE_MODEL, 1)
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == np.float64 else dict(rtol=2e-3, atol=2e-3)
    # compare the proposals through the accepted chains and the log terms; acceptance may flip only where log alpha ~ log u in fp32
    npt.assert_allclose(res[1][2][:, 1:], res[0][2][:, 1:], rtol=tol["rtol"])
    same = res[1][1] == res[0][1]
    assert same.mean() > (0.999 if dtype == np.float64 else 0.9)
    npt.assert_allclose(res[1][0][same], res[0][0][same], **tol)
    assert np.all(np.isfinite(res[1][0]))


@pytest.mark.parametrize("layout", ["cm_shared", "cm_general", "dense"])
def test_fp32_sweep_log_alpha_uses_fp64_totals(layout):
    """fp32 sweep at T = 32768: the five per-chain log-density totals are ~1e5, where an fp32 ulp is 0.008-0.016, and log alpha differences six of
    them -- held in fp32 that was |log alpha| ~ 0.1 and 1-2 % spurious rejections of an exact proposal (C2 in fp32: 0.13, acceptance 0.988).  The
    device sweep accumulates, reduces and compares them in fp64 (csrc/smallmat.h::Acc): on a linear-Gaussian model, where log alpha == 0 exactly,
    what is left is the rounding of the individual fp32 terms."""
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler, _get_device_kernel
    d, T, C = 4, 32768, 64
    m = lg_model(T, d)
    bt = np.broadcast_to
    model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)),
                          bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), m["y"])
    init, kernel = _get_device_kernel(model, True, "reference")
    rng = np.random.default_rng(5)
    x0 = (m["x_true"][None] + 0.3 * rng.standard_normal((C, T, d))).astype(np.float32)
    noise = dict(eps_aux=rng.standard_normal((C, T, d)), eps_samp=rng.standard_normal((C, T, d)), u_accept=rng.random(C))
    h = _lib.default_handle()
    try:
        h.set_option(_lib.OPT_SHARE_MODEL, 1 if layout == "cm_shared" else 0)
        chains = DeviceChains(h, x0, chain_minor=layout != "dense")
        kernel(None, KalmanSampler(x=chains, updated=None), 0.5, noise=noise)
        logs, acc = chains.logs.to_host(), chains.accepted.to_host()
    finally:
        h.set_option(_lib.OPT_SHARE_MODEL, 1)
    assert np.abs(logs[:, 1:]).min() > 1e4  # the totals are large ...
    assert np.abs(logs[:, 0]).max() < 5e-3, np.abs(logs[:, 0]).max()  # ... and their combination is not rounded to their ulp
    assert acc.all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,nu", [(1, 1), (7, 3), (4096, 64), (100001, 257)])
def test_fused_sweep_noise_draw_equals_the_three_fills(dtype, n, nu):
    """auxssm_kalman_draw (one launch) == auxssm_rng_normal x 2 + auxssm_rng_uniform (stream 0) on the three child keys, bit for bit."""
    from aux_ssm_samplers_amd import _lib, random as R
    h = _lib.default_handle()
    ka, ks, kc = R.split(R.PRNGKey(n + nu), 3)
    ea, es, ua = h.empty((n,), dtype), h.empty((n,), dtype), h.empty((nu,), dtype)
    h.kalman_draw(ka, ks, kc, ea, es, ua)
    npt.assert_array_equal(ea.to_host(), h.rng_normal(ka, 0, (n,), dtype).to_host())
    npt.assert_array_equal(es.to_host(), h.rng_normal(ks, 0, (n,), dtype).to_host())
    npt.assert_array_equal(ua.to_host(), h.rng_uniform(kc, 0, (nu,), dtype).to_host())


@pytest.mark.parametrize("parallel", [True, False])
@pytest.mark.parametrize("order", [1, 2])
def test_batched_model_through_get_kernel_as_the_spatial_example(parallel, order):
    """The reference's spatial example (examples/spatial/auxiliary_kalman.py:10-66) runs the generic sampler on a BATCHED model: the state is (T, d, 1), every array of the
    factories carries the batch axis d (d independent scalar LGSSMs coupled only through the potential).  The same call shapes through `get_kernel` with Python
    factories (the host-factory path: filter / sampler / log-density primitives on the batch axis, here wide enough -- 300 -- for the one-sequence-per-lane filter),
    first- and second-order observation factories with a spatially coupled potential, against the oracle's sweep on the same noise."""
    from aux_ssm_samplers_amd.kalman import get_kernel
    T, d = 30, 300
    rng = np.random.default_rng(17 + order)
    y = rng.standard_normal((T, d))
    lam, s2 = 0.3, 0.8
    m0, P0 = np.zeros((d, 1)), np.ones((d, 1, 1))
    F, Q, b = np.full((d, 1, 1), 0.9), np.full((d, 1, 1), 0.5), np.zeros((d, 1))
    eyes, zeros = np.ones((T, d, 1, 1)), np.zeros((T, d, 1))

    def log_potential(x):          # x (T, d): Gaussian observations + a nearest-neighbour coupling across the batch ("space")
        return float(np.sum(-0.5 * (y - x) ** 2 / s2) - lam * np.sum((x[:, 1:] - x[:, :-1]) ** 2))

    def grad_potential(x):
        g = (y - x) / s2
        dx = x[:, 1:] - x[:, :-1]
        g[:, 1:] -= 2 * lam * dx
        g[:, :-1] += 2 * lam * dx
        return g

    hess_diag = -(1.0 / s2 + 4 * lam) * np.ones(d)            # the example's diagonal Hessian approximation (:44)

    def dynamics_factory(_x):
        return m0, P0, np.tile(F[None], (T - 1, 1, 1, 1)), np.tile(Q[None], (T - 1, 1, 1, 1)), np.tile(b[None], (T - 1, 1, 1))

    def first_order(x, u, delta):
        g = np.nan_to_num(grad_potential(x.reshape(-1, d))).reshape(T, d, 1)
        return u + 0.5 * delta * g, eyes, 0.5 * delta * eyes, zeros

    def second_order(x, u, delta):
        g = grad_potential(x.reshape(-1, d)).reshape(T, d, 1)
        Om = 1.0 / (-hess_diag[None, :, None, None] + 2 * eyes / delta)
        return Om[..., 0] * (2 * u / delta + g - hess_diag[None, :, None] * x), eyes, Om, zeros

    def log_likelihood_fn(x):
        out = np.sum(-0.5 * (x[0] - m0) ** 2 / P0[:, 0] - 0.5 * np.log(2 * np.pi * P0[:, 0]))
        pred = F[None, :, 0] * x[:-1] + b[None]
        out += np.sum(-0.5 * (x[1:] - pred) ** 2 / Q[None, :, 0] - 0.5 * np.log(2 * np.pi * Q[None, :, 0]))
        return float(out) + log_potential(x.reshape(-1, d))

    obs = first_order if order == 1 else second_order
    init, kernel = get_kernel(dynamics_factory, obs, log_likelihood_fn, parallel)
    x = rng.standard_normal((T, d, 1))
    noise = dict(eps_aux=rng.standard_normal((T, d, 1)), eps_samp=rng.standard_normal((T, d, 1)), u_accept=rng.random())
    out = kernel(None, init(x), 0.2, noise=noise)
    ref = K.kalman_sweep(x, 0.2, dynamics_factory, obs, log_likelihood_fn, parallel, **noise)
    assert out.updated == ref["accepted"]
    npt.assert_allclose(out.x, ref["x"], rtol=1e-9, atol=1e-10)
    npt.assert_allclose(out.log_alpha, ref["log_alpha"], rtol=1e-7, atol=1e-7)

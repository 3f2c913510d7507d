"""The chain-shared LG_CONCAT sweep in three streaming passes on a lazy state (csrc/fused_shared.h, auxssm_kalman_sweep_fused) against
(1) the keyed sweep it replaces -- same keys, same draws: proposals to rounding, log terms to the rounding of sums of ~T d terms;
(2) the NumPy oracle sweep (oracle/kalman_np.py: kalman/generic.py:53-106 line by line) on the noise the device drew;
(3) its own bookkeeping: a state scattered over the ping-pong pair, rejected chains, resolve, device-resident step size, plain (sel = NULL) mode."""
import ctypes as C

import numpy as np
import numpy.testing as npt
import pytest

from oracle import kalman_np as K
from tests.helpers import lg_model

pytestmark = pytest.mark.gpu


def _setup(T, d, dtype, Cn, seed=0):
    from aux_ssm_samplers_amd.kalman import LGConcatModel, get_kernel
    m = lg_model(T, d, dtype=dtype)
    bt = np.broadcast_to
    model = LGConcatModel(m["m0"], m["P0"], bt(m["F"], (T - 1, d, d)), bt(m["Q"], (T - 1, d, d)), bt(m["b"], (T - 1, d)),
                          bt(m["Hobs"], (T, d, d)), bt(m["Robs"], (T, d, d)), bt(m["cobs"], (T, d)), m["y"])
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    rng = np.random.default_rng(seed)
    x0 = (m["x_true"][None] + 0.3 * rng.standard_normal((Cn, T, d))).astype(dtype)
    return m, model, kernel, x0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("T,d,Cn", [(4096, 4, 64), (1000, 2, 130), (257, 1, 2), (64, 4, 66), (3001, 4, 256),
                                    # few chains: one wave walks several chunks side by side (the PK instantiations of the two passes, csrc/fused_shared.h):
                                    # chain counts that are / are not powers of two, T that is not a multiple of a workgroup's chunks, the shortest T, 2..32 chains
                                    (1000, 4, 8), (257, 2, 6), (4096, 4, 16), (333, 2, 32), (64, 4, 4), (2048, 4, 30), (513, 4, 2), (777, 1, 12), (1500, 4, 64), (999, 2, 100), (2048, 4, 128)])
def test_fused_equals_keyed_sweep(dtype, T, d, Cn):
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    h = _lib.default_handle()
    m, model, kernel, x0 = _setup(T, d, dtype, Cn)
    a = DeviceChains(h, x0, chain_minor=True)               # fused, lazy state
    b = DeviceChains(h, x0, chain_minor=True, fused=False)  # the keyed sweep
    tol = dict(rtol=1e-9, atol=1e-10) if dtype == np.float64 else dict(rtol=2e-3, atol=2e-3)
    for i, delta in enumerate([0.5, 0.5, 0.2]):
        key = R.PRNGKey(40 + i)
        kernel(key, KalmanSampler(x=a, updated=None), delta)
        kernel(key, KalmanSampler(x=b, updated=None), delta)
        assert a.fused is True and b.fused is False
        la, lb = a.logs.to_host(), b.logs.to_host()
        npt.assert_array_equal(a.accepted.to_host(), b.accepted.to_host())
        npt.assert_allclose(a.to_host(), b.to_host(), **tol)
        # the four totals are O(T d); log alpha is their difference (identically 0 for this model)
        scale = np.abs(lb[:, 1:]).max()
        npt.assert_allclose(la[:, 1:], lb[:, 1:], rtol=0, atol=(1e-12 if dtype == np.float64 else 2e-6) * scale)
        assert np.abs(la[:, 0]).max() < (1e-7 if dtype == np.float64 else 5e-2)
    assert a.accepted.to_host().all()


def test_fused_vs_oracle_sweep():
    """x' and the log terms against oracle/kalman_np.py::kalman_sweep driven by the NumPy factories on the noise the device draws for these keys"""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    h = _lib.default_handle()
    T, d, Cn, dtype = 700, 4, 4, np.float64
    m, model, kernel, x0 = _setup(T, d, dtype, Cn)
    a = DeviceChains(h, x0, chain_minor=True)
    key = R.PRNGKey(7)
    kernel(key, KalmanSampler(x=a, updated=None), 0.5)
    assert a.fused is True
    k_aux, k_samp, k_acc = R.split(key, 3)
    b = DeviceChains(h, x0, chain_minor=True, fused=False)
    h.kalman_draw(k_aux, k_samp, k_acc, b.eps_aux, b.eps_samp, b.u_acc)
    ea, es, ua = b.stats_to_host(b.eps_aux), b.stats_to_host(b.eps_samp), b.u_acc.to_host()
    xa, logs = a.to_host(), a.logs.to_host()
    lgo = (m["m0"], m["P0"], model.Fs, model.Qs, model.bs, model.Hobs, model.Robs, model.cobs)
    for c in range(Cn):
        ref = K.kalman_sweep(x0[c], 0.5, model.dynamics_factory, model.observations_factory,
                             lambda z: K.log_likelihood(m["y"], z, lgo) + K.prior_logpdf(z, lgo), True, eps_aux=ea[c], eps_samp=es[c], u_accept=ua[c])
        npt.assert_allclose(xa[c], ref["x"], rtol=1e-9, atol=1e-10)
        assert abs(logs[c, 0] - ref["log_alpha"]) < 1e-7
        npt.assert_allclose(logs[c, 1:], [ref["lp_prop"], ref["lp_rev"], ref["lt_prop"], ref["lt_rev"]], rtol=1e-10)


def _raw_sweep(h, model, chains, key, delta, sel, x, x_alt, dtype):
    from aux_ssm_samplers_amd import _lib, random as R
    dl, ybuf, yarr = model.device(h, dtype)
    dims = _lib.Dims(chains.C, chains.T, 1, chains.dx, model.p_obs)
    keys = R.split(key, 3)
    k6 = (C.c_uint32 * 6)(*[int(v) for k in keys for v in np.asarray(k, np.uint32).reshape(2)])
    dev = isinstance(delta, _lib.DeviceArray)
    rc = h.lib.auxssm_kalman_sweep_fused(h.h, _lib.dtype_code(dtype), model.kmodel, C.byref(dims), C.byref(dl.c), C.byref(yarr), 1.0 if dev else float(delta),
                                         delta.ptr if dev else None, k6, 1, _lib.NAN_REFERENCE, chains.layout, x.ptr, x_alt.ptr, None if sel is None else sel.ptr,
                                         chains.u_acc.ptr, chains.accepted.ptr, chains.logs.ptr)
    _lib.check(rc)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lazy_state_bookkeeping(dtype):
    """A state SCATTERED over the ping-pong pair by a random selector (the other buffer's slots hold garbage), one chain with a NaN (its
    log alpha is NaN: rejected, selector unchanged), a device-resident step size, and the plain (sel = NULL) mode: all give the sweep of the
    gathered state; resolve gathers and zeroes the selector."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    h = _lib.default_handle()
    T, d, Cn = 1500, 2, 64
    m, model, kernel, x0 = _setup(T, d, dtype, Cn, seed=3)
    x0[5, 700, 1] = np.nan
    rng = np.random.default_rng(1)
    ref = DeviceChains(h, x0, chain_minor=True)                      # lazy from a clean start
    key = R.PRNGKey(99)
    kernel(key, KalmanSampler(x=ref, updated=None), 0.4)
    acc = ref.accepted.to_host()
    assert acc[5] == 0 and acc.sum() == Cn - 1
    npt.assert_array_equal(ref.sel.to_host(), acc)                   # accepted chains now live in the other buffer
    xr, lr = ref.to_host(), ref.logs.to_host()
    assert np.isnan(xr[5, 700, 1]) and np.array_equal(np.nan_to_num(xr[5]), np.nan_to_num(x0[5]))
    npt.assert_array_equal(ref.sel.to_host(), 0)                     # resolve zeroed it
    # scattered start
    sel0 = rng.integers(0, 2, Cn).astype(np.int32)
    cm = lambda a: np.ascontiguousarray(np.asarray(a, dtype).transpose(1, 2, 0))
    junk = rng.standard_normal(x0.shape).astype(dtype) * 100
    xa = np.where(sel0[:, None, None] == 0, x0, junk)
    xb = np.where(sel0[:, None, None] == 1, x0, junk)
    ch = DeviceChains(h, x0, chain_minor=True)
    ch.x.copy_from_host(cm(xa))
    ch.x_alt = h.to_device(cm(xb), dtype)
    ch.sel = h.to_device(sel0, np.int32)
    _raw_sweep(h, model, ch, key, 0.4, ch.sel, ch.x, ch.x_alt, dtype)
    ch._lazy_dirty = True
    npt.assert_array_equal(ch.sel.to_host(), sel0 ^ acc)
    npt.assert_array_equal(np.nan_to_num(ch.to_host()), np.nan_to_num(xr))
    npt.assert_array_equal(np.nan_to_num(ch.logs.to_host()), np.nan_to_num(lr))
    # device-resident step size, plain mode
    for mode in ("dd", "plain"):
        c2 = DeviceChains(h, x0, chain_minor=True)
        c2.x_alt = h.empty(c2.x.shape, dtype)
        delta = h.to_device(np.full(1, 0.4, dtype), dtype) if mode == "dd" else 0.4
        sel = h.zeros((Cn,), np.int32) if mode == "dd" else None
        c2.sel = sel
        _raw_sweep(h, model, c2, key, delta, sel, c2.x, c2.x_alt, dtype)
        c2._lazy_dirty = sel is not None
        npt.assert_array_equal(np.nan_to_num(c2.to_host()), np.nan_to_num(xr), err_msg=mode)
        npt.assert_array_equal(np.nan_to_num(c2.logs.to_host()), np.nan_to_num(lr), err_msg=mode)


def test_x_property_never_shows_stale_rows_after_a_fused_sweep():
    """DeviceChains.x gathers the lazy state before handing out the DeviceArray: after a fused sweep that accepted, chains.x.to_host() is
    the state to_host() returns, and a reset through chains.x.copy_from_host takes effect for every chain."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    h = _lib.default_handle()
    m, model, kernel, x0 = _setup(400, 2, np.float64, 64, seed=5)
    ch = DeviceChains(h, x0, chain_minor=True)
    kernel(R.PRNGKey(3), KalmanSampler(x=ch, updated=None), 0.5)
    assert ch.fused is True and ch.accepted.to_host().sum() > 0 and ch.sel.to_host().sum() > 0   # accepted chains live in x_alt now
    raw = ch.x.to_host()                                                                          # the property resolves first
    npt.assert_array_equal(ch.sel.to_host(), 0)
    npt.assert_array_equal(np.ascontiguousarray(raw.transpose(2, 0, 1)), ch.to_host())
    assert not np.array_equal(ch.to_host(), x0)
    kernel(R.PRNGKey(4), KalmanSampler(x=ch, updated=None), 0.5)
    ch.x.copy_from_host(np.ascontiguousarray(x0.transpose(1, 2, 0)))                              # a reset through the public attribute
    npt.assert_array_equal(ch.to_host(), x0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lazy_state_moments_fold_accepted_and_rejected_chains(dtype):
    """Running moments attached to the state of a FUSED sweep are folded by it (k_fs_stats), for accepted chains and for a rejected one (a NaN chain: zero
    jump, unchanged state) alike: bit for bit what auxssm_stats_update gives on the resolved states before / after each sweep."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    h = _lib.default_handle()
    T, d, Cn = 700, 2, 64
    m, model, kernel, x0 = _setup(T, d, dtype, Cn, seed=8)
    x0[7, 300, 0] = np.nan                                            # log alpha NaN: rejected in every sweep
    ch = DeviceChains(h, x0, chain_minor=True)
    ref = DeviceChains(h, x0, chain_minor=True)
    stats = tuple(h.zeros(ch._x.shape, dtype) for _ in range(3))
    want = tuple(h.zeros(ch._x.shape, dtype) for _ in range(3))
    prev = h.empty(ch._x.shape, dtype)
    h.stats_attach(stats, 0, ch.x)
    try:
        for i in range(4):
            key = R.PRNGKey(50 + i)
            kernel(key, KalmanSampler(x=ch, updated=None), 0.4)       # folds the moments itself
            assert ch.fused is True
            prev.copy_from(ref.x)
            h.stats_attach(None, 0)                                   # (the reference chains are another state: detach while they sweep)
            kernel(key, KalmanSampler(x=ref, updated=None), 0.4)
            h.stats_update(i, prev, ref.x, want)
            h.stats_attach(stats, i + 1, ch._x)
            acc = ch.accepted.to_host()
            assert acc[7] == 0 and acc.sum() >= Cn - 2
    finally:
        h.stats_attach(None, 0)
    npt.assert_array_equal(np.nan_to_num(ch.to_host()), np.nan_to_num(ref.to_host()))
    for a, b, name in zip(stats, want, ("sq_jump", "mean", "sq_mean")):
        npt.assert_array_equal(np.nan_to_num(a.to_host()), np.nan_to_num(b.to_host()), err_msg=name)
    sq = ch.stats_to_host(stats[0])
    assert np.all(np.nan_to_num(sq[7]) == 0) and float(np.nanmax(sq)) > 0   # the rejected chain never jumped


def test_refusals_fall_back_to_the_keyed_sweep():
    """odd chain counts and dense layouts are refused before anything is enqueued; kernel() then runs the keyed sweep (attached running moments are NOT a
    refusal: the fused sweep folds them, test_lazy_state_moments_fold_accepted_and_rejected_chains)"""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    h = _lib.default_handle()
    m, model, kernel, x0 = _setup(300, 2, np.float64, 33)
    a = DeviceChains(h, x0, chain_minor=True)
    kernel(R.PRNGKey(1), KalmanSampler(x=a, updated=None), 0.5)
    assert a.fused is False and a.x_alt is None
    b = DeviceChains(h, x0, chain_minor=True, fused=False)
    kernel(R.PRNGKey(1), KalmanSampler(x=b, updated=None), 0.5)
    npt.assert_array_equal(a.to_host(), b.to_host())


def test_full_size_C2_log_alpha_is_zero_and_lazy_chain_of_sweeps():
    """BASELINE config C2 (T = 65536, d = 4, fp64) through size-independent properties: for the exact linear-Gaussian model log alpha == 0, every
    proposal is accepted, and 4 lazy sweeps in a row equal 4 keyed sweeps to rounding."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    h = _lib.default_handle()
    m, model, kernel, x0 = _setup(65536, 4, np.float64, 64)
    a = DeviceChains(h, x0, chain_minor=True)
    b = DeviceChains(h, x0, chain_minor=True, fused=False)
    for i in range(4):
        kernel(R.PRNGKey(500 + i), KalmanSampler(x=a, updated=None), 0.5)
        kernel(R.PRNGKey(500 + i), KalmanSampler(x=b, updated=None), 0.5)
        assert np.abs(a.logs.to_host()[:, 0]).max() < 1e-7
    assert a.fused is True
    npt.assert_allclose(a.to_host(), b.to_host(), rtol=1e-9, atol=1e-10)

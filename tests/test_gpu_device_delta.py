"""auxssm_kalman_sweep_dd: the sweep with a DEVICE-RESIDENT step size (boundary gap: the adaptation loop of
examples/*/experiment.py with common.py:4-32 delta_adaptation read C floats back per burn-in sweep).  Same kernels, delta read from device
memory: results must be bit-identical to the host-scalar sweep for every device model and layout."""
import numpy as np
import numpy.testing as npt
import pytest

from aux_ssm_samplers_amd import _lib
from aux_ssm_samplers_amd.kalman import get_kernel, DeviceChains, KalmanSampler
from tests.helpers import lg_model

pytestmark = pytest.mark.gpu


def _models(dtype):
    from aux_ssm_samplers_amd.kalman import LGConcatModel, SVModel
    from tests.helpers import sv_setup, lorenz_kalman_setup
    T, d = 400, 2
    m = lg_model(T, d, dtype=dtype)
    full = lambda a, n: np.ascontiguousarray(np.broadcast_to(a, (n,) + a.shape))
    yield "lg", LGConcatModel(m["m0"], m["P0"], full(m["F"], T - 1), full(m["Q"], T - 1), full(m["b"], T - 1), full(m["Hobs"], T), full(m["Robs"], T),
                              full(m["cobs"], T), m["y"]), 0.4, T, d
    y, _, (m0, P0, F, Q, b) = sv_setup(64, 2)
    for order in (1, 2):
        yield f"sv{order}", SVModel(y, m0, P0, F, Q, b, order=order), 0.05, 64, 2
    yield "lorenz", lorenz_kalman_setup(160, every=8, dt=1e-3)[0], 1e-3, 160, 3


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("C,chain_minor", [(3, False), (64, True)])
def test_device_delta_sweep_is_bit_identical_to_the_host_scalar_sweep(dtype, C, chain_minor):
    h = _lib.default_handle()
    for name, model, delta, T, d in _models(dtype):
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        rng = np.random.default_rng(7)
        x0 = rng.standard_normal((C, T, d)).astype(dtype) * 0.3
        noise = dict(eps_aux=rng.standard_normal((C, T, d)).astype(dtype), eps_samp=rng.standard_normal((C, T, d)).astype(dtype),
                     u_accept=rng.random(C).astype(dtype))
        cm = chain_minor and not getattr(model, "dense_only", False)
        a = DeviceChains(h, x0, chain_minor=cm)
        b = DeviceChains(h, x0, chain_minor=cm)
        kernel(None, KalmanSampler(x=a, updated=None), delta, noise=noise)
        dd = h.to_device(np.full(1, delta, dtype), dtype)
        kernel(None, KalmanSampler(x=b, updated=None), dd, noise=noise)
        npt.assert_array_equal(a.to_host(), b.to_host(), err_msg=name)
        npt.assert_array_equal(a.logs.to_host(), b.logs.to_host(), err_msg=name)
        npt.assert_array_equal(a.accepted.to_host(), b.accepted.to_host(), err_msg=name)


def test_wrong_dtype_device_delta_is_refused():
    h = _lib.default_handle()
    name, model, delta, T, d = next(_models(np.float64))
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    ch = DeviceChains(h, np.zeros((2, T, d)), chain_minor=False)
    with pytest.raises(ValueError):
        kernel(0, KalmanSampler(x=ch, updated=None), h.to_device(np.full(1, 0.3, np.float32), np.float32))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("C,chain_minor,share", [(3, False, 1), (64, True, 1), (64, True, 0), (130, True, 1)])
def test_keyed_sweep_equals_draw_then_sweep(dtype, C, chain_minor, share):
    """auxssm_kalman_sweep_keyed: kernel(key, state, delta) with the KEYS of the three draws.  Where the chain-shared affine scans run the noise
    is generated inside their reduce passes (lane pairs sharing Threefry blocks); it must be bit for bit what auxssm_kalman_draw puts into the
    buffers, and the sweep's results must be those of draw + sweep."""
    from aux_ssm_samplers_amd import random as R
    h = _lib.default_handle()
    h.set_option(_lib.OPT_SHARE_MODEL, share)
    try:
        for name, model, delta, T, d in _models(dtype):
            if name != "lg" and C == 130:
                continue
            T2 = T if name != "lg" else T  # (short horizons: few chunks; the gen path still needs more than one chunk)
            init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
            rng = np.random.default_rng(11)
            x0 = rng.standard_normal((C, T, d)).astype(dtype) * 0.3
            cm = chain_minor and not getattr(model, "dense_only", False)
            a = DeviceChains(h, x0, chain_minor=cm, fused=False)  # (the keyed sweep proper: the fused one keeps no noise buffers, tests/test_gpu_fused.py)
            b = DeviceChains(h, x0, chain_minor=cm, fused=False)
            key = R.PRNGKey(123)
            kernel(key, KalmanSampler(x=a, updated=None), delta)                     # keyed
            k_aux, k_samp, k_acc = R.split(key, 3)
            h.kalman_draw(k_aux, k_samp, k_acc, b.eps_aux, b.eps_samp, b.u_acc)      # the three fills ...
            ea, es, ua = b.eps_aux.to_host(), b.eps_samp.to_host(), b.u_acc.to_host()
            npt.assert_array_equal(a.eps_aux.to_host(), ea, err_msg=name)
            npt.assert_array_equal(a.eps_samp.to_host(), es, err_msg=name)
            npt.assert_array_equal(a.u_acc.to_host(), ua, err_msg=name)
            to_ctd = (lambda e: np.ascontiguousarray(e.transpose(2, 0, 1))) if cm else (lambda e: e)
            kernel(None, KalmanSampler(x=b, updated=None), delta, noise=dict(eps_aux=to_ctd(ea), eps_samp=to_ctd(es), u_accept=ua))  # ... then the sweep
            npt.assert_array_equal(a.to_host(), b.to_host(), err_msg=name)
            npt.assert_array_equal(a.logs.to_host(), b.logs.to_host(), err_msg=name)
    finally:
        h.set_option(_lib.OPT_SHARE_MODEL, 1)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_model_stage_on_the_side_stream_equals_the_single_stream_sweep(dtype):
    """AUXSSM_OPT_OVERLAP_MODEL_STAGE: the chain-independent model stage of a chain-shared keyed sweep (concatenated observation model, matrix filter,
    gain / sampler / log-density tables) runs on a second stream with a double-buffered slab and overlaps the previous sweep.  Sequences of sweeps --
    step size changing from sweep to sweep, two resident states taking turns on the handle, the data (yobs) replaced on the device between two
    sweeps by an asynchronous device-to-device copy (the fence behind a foreign call), a general-path sweep in between -- must give bit for bit the
    single-stream results."""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    h = _lib.default_handle()
    T, d, C = 1500, 2, 64
    m = lg_model(T, d, dtype=dtype)
    full = lambda a, n: np.ascontiguousarray(np.broadcast_to(a, (n,) + a.shape))
    rng = np.random.default_rng(3)
    y2 = (m["y"] + 0.5 * rng.standard_normal(m["y"].shape)).astype(m["y"].dtype)
    x0 = rng.standard_normal((2, C, T, d)).astype(dtype) * 0.3
    deltas = [0.4, 0.4, 0.25, 0.25, 0.6, 0.4, 0.4, 0.4]

    def run(overlap):
        h.set_option(_lib.OPT_OVERLAP_MODEL_STAGE, overlap)
        model = LGConcatModel(m["m0"], m["P0"], full(m["F"], T - 1), full(m["Q"], T - 1), full(m["b"], T - 1), full(m["Hobs"], T), full(m["Robs"], T),
                              full(m["cobs"], T), m["y"])
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        chains = [DeviceChains(h, x0[0], chain_minor=True), DeviceChains(h, x0[1], chain_minor=True)]
        _, ybuf, _ = model.device(h, dtype)
        ynew = _lib.DeviceArray(h, ybuf.shape, ybuf.dtype)
        ynew.copy_from_host(np.ascontiguousarray(y2, dtype=dtype).reshape(ybuf.shape))
        out, junk = [], None
        for i, dl in enumerate(deltas):
            key = R.PRNGKey(1000 + i)
            st = chains[i % 2]
            if i == 4:  # new data, enqueued asynchronously on the handle's stream BEHIND a few milliseconds of other work: the next model stage must
                # still come after it (without the fence of ctx.h::SideStage it would read the old data while the fill below is running)
                junk = h.rng_normal(R.PRNGKey(5), 9, (1 << 28,), np.float32)  # (kept alive: freeing it would synchronise the device)
                ybuf.copy_from(ynew)
            if i == 6:  # a general-path sweep in between (no model stage)
                h.set_option(_lib.OPT_SHARE_MODEL, 0)
            kernel(key, KalmanSampler(x=st, updated=None), dl)
            if i == 6:
                h.set_option(_lib.OPT_SHARE_MODEL, 1)
            if i in (1, 5, 7):
                out.append((st.to_host(), st.accepted.to_host(), st.logs.to_host()))
        out.append((chains[0].to_host(), chains[1].to_host()))
        return out

    try:
        a = run(1)
        b = run(0)
    finally:
        h.set_option(_lib.OPT_OVERLAP_MODEL_STAGE, 1)
        h.set_option(_lib.OPT_SHARE_MODEL, 1)
    for u, v in zip(a, b):
        for p, q in zip(u, v):
            npt.assert_array_equal(p, q)
    assert not np.array_equal(a[-1][0], x0[0])


def test_model_stage_memo_is_exact(monkeypatch):
    """The fused sweep's model stage is MEMOISED per slab (ctx.h::SideStage, round 4): rebuilt only when its inputs differ byte for byte from the snapshot the slab's
    tables were built from -- decided on the device, no host synchronisation.  A run of sweeps with a fixed step size (every stage after the first three is skipped),
    then a new step size, then the DATA and a MODEL MATRIX rewritten in place on the device (same pointers: only the comparison kernel can notice), then the old data
    back: bit for bit the results of the same run with the memo switched off (AUXSSM_STAGE_MEMO=0, a separate library instance would read the variable once, so the
    switch here is the single-stream sweep, which has no stage to memoise)."""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    h = _lib.default_handle()
    dtype = np.float64
    T, d, C = 1200, 2, 64
    m = lg_model(T, d, dtype=dtype)
    full = lambda a, n: np.ascontiguousarray(np.broadcast_to(a, (n,) + a.shape))
    rng = np.random.default_rng(4)
    y2 = (m["y"] + 0.5 * rng.standard_normal(m["y"].shape)).astype(dtype)
    x0 = rng.standard_normal((C, T, d)).astype(dtype) * 0.3
    deltas = [0.4] * 7 + [0.3] * 4 + [0.4] * 9

    def run(overlap):
        h.set_option(_lib.OPT_OVERLAP_MODEL_STAGE, overlap)
        model = LGConcatModel(m["m0"], m["P0"], full(m["F"], T - 1), full(m["Q"], T - 1), full(m["b"], T - 1), full(m["Hobs"], T), full(m["Robs"], T),
                              full(m["cobs"], T), m["y"])
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        ch = DeviceChains(h, x0, chain_minor=True)
        dl_, ybuf, _ = model.device(h, dtype)
        yold, ynew = _lib.DeviceArray(h, ybuf.shape, ybuf.dtype), _lib.DeviceArray(h, ybuf.shape, ybuf.dtype)
        yold.copy_from(ybuf)
        ynew.copy_from_host(np.ascontiguousarray(y2, dtype=dtype).reshape(ybuf.shape))
        Fbuf = dl_.bufs["Fs"]
        F2 = _lib.DeviceArray(h, Fbuf.shape, Fbuf.dtype)
        F2.copy_from_host((0.9 * Fbuf.to_host()).astype(dtype))
        out = []
        for i, dl in enumerate(deltas):
            if i == 12:
                ybuf.copy_from(ynew)      # new data behind the same pointer
            if i == 15:
                Fbuf.copy_from(F2)        # a model matrix rewritten in place
            if i == 17:
                ybuf.copy_from(yold)      # ... and the old data back
            kernel(R.PRNGKey(2000 + i), KalmanSampler(x=ch, updated=None), dl)
            assert ch.fused is True
            if i in (2, 6, 8, 11, 12, 13, 15, 16, 17, 19):
                out.append((ch.to_host(), ch.accepted.to_host(), ch.logs.to_host()))
        return out

    try:
        a = run(1)
        b = run(0)
    finally:
        h.set_option(_lib.OPT_OVERLAP_MODEL_STAGE, 1)
    for u, v in zip(a, b):
        for p, q in zip(u, v):
            npt.assert_array_equal(p, q)
    assert not np.array_equal(a[3][2], a[4][2])


def test_sv_first_order_model_stage_on_the_side_stream_equals_the_single_stream_sweep():
    """The first-order SV factory with chain-shared dynamics: covariances and gain rows depend on the model and the step size only, so the sweep builds
    them once (the reverse filter reuses the proposal filter's rows) on the second stream -- bit for bit the single-stream sweep, over changing step sizes."""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.kalman import SVModel
    from tests.helpers import sv_setup
    h = _lib.default_handle()
    T, d, C = 3000, 2, 64
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d)
    x0 = np.repeat(xtrue[None], C, axis=0) + 0.1 * np.random.default_rng(2).standard_normal((C, T, d))

    def run(overlap):
        h.set_option(_lib.OPT_OVERLAP_MODEL_STAGE, overlap)
        model = SVModel(y, m0, P0, F, Q, b, order=1)
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        st = DeviceChains(h, x0, chain_minor=True)
        for i, dl in enumerate([0.05, 0.05, 0.03, 0.03, 0.08, 0.05]):
            kernel(R.PRNGKey(77 + i), KalmanSampler(x=st, updated=None), dl)
        return st.to_host(), st.accepted.to_host(), st.logs.to_host()

    try:
        a, b_ = run(1), run(0)
    finally:
        h.set_option(_lib.OPT_OVERLAP_MODEL_STAGE, 1)
    for p, q in zip(a, b_):
        npt.assert_array_equal(p, q)
    assert 0 < a[1].mean() < 1  # (some chains moved, some did not)


def test_model_array_rewritten_through_the_raw_stream_between_two_sweeps():
    """a caller that holds auxssm_stream() can queue a write of a model array on it that the library never sees.  Once the stream
    has been handed out every model stage waits for the tail of the stream, so the sweep after such a write reads the NEW data: bit for bit the
    single-stream result.  (The write here is a hipMemcpyAsync issued straight through the HIP runtime behind milliseconds of queued work.)"""
    import ctypes
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.kalman import LGConcatModel
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    T, d, C, dtype = 1500, 2, 64, np.float64
    m = lg_model(T, d, dtype=dtype)
    full = lambda a, n: np.ascontiguousarray(np.broadcast_to(a, (n,) + a.shape))
    rng = np.random.default_rng(11)
    y2 = (m["y"] + 0.5 * rng.standard_normal(m["y"].shape)).astype(dtype)
    x0 = rng.standard_normal((C, T, d)).astype(dtype) * 0.3

    def run(overlap, raw):
        h = _lib.Handle()
        try:
            h.set_option(_lib.OPT_OVERLAP_MODEL_STAGE, overlap)
            model = LGConcatModel(m["m0"], m["P0"], full(m["F"], T - 1), full(m["Q"], T - 1), full(m["b"], T - 1), full(m["Hobs"], T),
                                  full(m["Robs"], T), full(m["cobs"], T), m["y"])
            init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
            st = DeviceChains(h, x0, chain_minor=True)
            _, ybuf, _ = model.device(h, dtype)
            ynew = _lib.DeviceArray(h, ybuf.shape, ybuf.dtype)
            ynew.copy_from_host(np.ascontiguousarray(y2).reshape(ybuf.shape))
            stream = h.stream() if raw else None
            junk = None
            for i in range(6):
                if i == 3:
                    junk = h.rng_normal(R.PRNGKey(5), 9, (1 << 28,), np.float32)  # milliseconds of queued work in front of the write
                    if raw:
                        assert hip.hipMemcpyAsync(ybuf.ptr, ynew.ptr, ybuf.nbytes, 3, ctypes.c_void_p(stream)) == 0  # hipMemcpyDeviceToDevice
                    else:
                        ybuf.copy_from(ynew)
                kernel(R.PRNGKey(2000 + i), KalmanSampler(x=st, updated=None), 0.4)
            return st.to_host(), st.accepted.to_host(), st.logs.to_host()
        finally:
            h.sync()
            h.close()

    a, b, c = run(1, True), run(0, True), run(0, False)
    for p, q, r in zip(a, b, c):
        npt.assert_array_equal(p, q)
        npt.assert_array_equal(p, r)


def test_c_abi_default_is_the_single_stream_sweep():
    """the C ABI ships AUXSSM_OPT_OVERLAP_MODEL_STAGE off (include/auxssm.h); only the Python layer's Handle opts in"""
    import ctypes
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.auxssm_create(0, ctypes.byref(h)) == 0
    try:
        v = ctypes.c_int(-1)
        assert lib.auxssm_get_option(h, _lib.OPT_OVERLAP_MODEL_STAGE, ctypes.byref(v)) == 0 and v.value == 0
    finally:
        lib.auxssm_destroy(h)
    import os
    assert _lib.Handle().get_option(_lib.OPT_OVERLAP_MODEL_STAGE) == (0 if os.environ.get("AUXSSM_OVERLAP_TAB") == "0" else 1)  # (the measurement switch keeps it off)

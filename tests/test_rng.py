"""Threefry-2x32-20: Random123 known-answer vectors for both host implementations (product
aux_ssm_samplers_amd/random.py, oracle/rng_np.py); device fills vs the oracle on the GPU."""
import numpy as np
import numpy.testing as npt
import pytest

KATS = [((0, 0), (0, 0), (0x6b200159, 0x99ba4efe)),
        ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
        ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0))]


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_threefry_known_answers(which):
    if which == "product":
        from aux_ssm_samplers_amd.random import threefry2x32
    else:
        from oracle.rng_np import threefry2x32
    for key, ctr, want in KATS:
        a, b = threefry2x32(np.uint32(key[0]), np.uint32(key[1]), np.uint32(ctr[0]), np.uint32(ctr[1]))
        assert (int(a), int(b)) == want


def test_split_is_deterministic_and_distinct():
    from aux_ssm_samplers_amd import random as R
    k = R.split(R.PRNGKey(7), 1000)
    assert len({tuple(x) for x in k}) == 1000
    npt.assert_array_equal(k, R.split(7, 1000))


def test_rng_tables_are_current_and_symmetric():
    """csrc/rng_tables.h is what tools/gen_rng_tables.py writes (60-digit decimal evaluation, correctly rounded), the turn table maps onto itself under a
    quarter turn EXACTLY (normals_cm's lane-pair split relies on it), the log table's end points are exact and every entry is within half an ulp of libm's value + 1 ulp."""
    import math
    import os
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    sys.path.insert(0, tools)
    try:
        import gen_rng_tables as G
    finally:
        sys.path.remove(tools)
    hdr = os.path.join(os.path.dirname(tools), "aux_ssm_samplers_amd", "csrc", "rng_tables.h")
    assert open(hdr).read() == G.render()
    logt, turn = G.tables()
    assert logt[0] == (2.0, -2.0 * math.log(0.5)) and logt[256] == (1.0, 0.0)
    for j in range(257):
        F = (256 + j) / 512
        assert abs(logt[j][0] - 1 / F) <= 2.3e-16 * (1 / F) and abs(logt[j][1] + 2 * math.log(F)) <= 4.5e-16
    for j in range(256):
        c, s = turn[j]
        c2, s2 = turn[(j + 64) % 256]
        assert c2 == -s and s2 == c
        assert abs(c - math.cos(2 * math.pi * j / 256)) < 1e-15 and abs(s - math.sin(2 * math.pi * j / 256)) < 1e-15
        assert abs(c * c + s * s - 1) < 3e-16


@pytest.mark.gpu
def test_fp64_normals_edge_words_and_the_lane_pair_split():
    """The table-driven fp64 transform at the words where its branches meet (u1 next to 0 and 1: the largest radius and the radii whose table terms cancel exactly;
    u2 on sector boundaries and quarter turns) against an 80-bit libm evaluation, and the chain-minor in-kernel draws (lane pairs, the odd lane a quarter turn back)
    bit-identical to the fill kernel -- the latter through the keyed fused sweep's tests (tests/test_gpu_device_delta.py); here the fill itself."""
    from aux_ssm_samplers_amd import _lib, random as R
    from oracle import rng_np as O
    h = _lib.default_handle()
    key = R.PRNGKey(99)
    n = 1 << 20
    z = h.rng_normal(key, 3, (n,), np.float64).to_host()
    b0, b1 = O._bits(key, 3, n // 2)
    u1 = (b0.astype(np.longdouble) + np.longdouble(0.5)) / np.longdouble(2) ** 32
    u2 = (b1.astype(np.longdouble) + np.longdouble(0.5)) / np.longdouble(2) ** 32
    r = np.sqrt(-2 * np.log(u1))
    tau = 2 * np.longdouble("3.14159265358979323846264338327950288")
    ref = np.stack([r * np.cos(tau * u2), r * np.sin(tau * u2)], axis=1).reshape(-1)
    err = np.abs(z.astype(np.longdouble) - ref)
    assert float(err.max()) < 1e-14, float(err.max())  # |z| <= 6.8: a few ulp
    # the extreme words of this sample really exercise the ends of the tables
    assert b0.min() < 2 ** 16 and b0.max() > 2 ** 32 - 2 ** 16 and ((b1 + np.uint32(0x800000)) >> np.uint32(24)).min() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_device_fill_vs_oracle(dtype):
    from aux_ssm_samplers_amd import _lib, random as R
    from oracle import rng_np as O
    h = _lib.default_handle()
    key = R.PRNGKey(2**40 + 12345)
    n = 100_000
    npt.assert_array_equal(h.rng_uniform(key, 5, (n,), dtype).to_host(), O.uniform(key, 5, n, dtype))
    z = h.rng_normal(key, 9, (n,), dtype).to_host()
    npt.assert_allclose(z, O.normal(key, 9, n, dtype), rtol=2e-5 if dtype == np.float32 else 1e-12, atol=4e-6 if dtype == np.float32 else 1e-13)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02


# ---- jax.random compatibility (threefry2x32, non-partitionable layout) ----------------------------------------------------------------------------------------
# Known answers: the values JAX's documentation prints ("Pseudorandom numbers" / "The Sharp Bits": PRNGKey(0), PRNGKey(42)); float32.
JAX_KAT = dict(split0=[[4146024105, 967050713], [2718843009, 1272950319]], uniform0=0.41845703, normal0=-0.20584226, normal0_subkey=-1.2515389,
               normal42_3=[0.18693547, -1.2806505, -1.5593132], normal42_individually=[-0.04838839, 0.10796146, -1.2226542])


def _jax_known_answers(split, uniform, normal):
    k0, k42 = np.array([0, 0], np.uint32), np.array([0, 42], np.uint32)
    npt.assert_array_equal(split(k0, 2), np.array(JAX_KAT["split0"], np.uint32))
    npt.assert_allclose(np.ravel(uniform(k0, 1))[0], JAX_KAT["uniform0"], rtol=0, atol=6e-8)            # (printed with 8 digits)
    npt.assert_allclose(np.ravel(normal(k0, 1))[0], JAX_KAT["normal0"], rtol=0, atol=2e-7)
    npt.assert_allclose(np.ravel(normal(split(k0, 2)[1], 1))[0], JAX_KAT["normal0_subkey"], rtol=0, atol=2e-7)
    npt.assert_allclose(np.ravel(normal(k42, 3)), JAX_KAT["normal42_3"], rtol=0, atol=2e-7)              # odd count: one counter appended
    npt.assert_allclose([np.ravel(normal(k, 1))[0] for k in split(k42, 3)], JAX_KAT["normal42_individually"], rtol=0, atol=2e-7)


def test_jax_streams_oracle_reproduces_the_documented_values():
    from oracle import rng_np as O
    _jax_known_answers(O.jax_split, lambda k, n: O.jax_uniform(k, n, np.float32), lambda k, n: O.jax_normal(k, n, np.float32))
    # the host key arithmetic of the product is the same function
    from aux_ssm_samplers_amd import random as R
    for key, num in [((0, 0), 2), ((123, 456), 5), ((7, 9), 1)]:
        npt.assert_array_equal(R.jax_split(np.array(key, np.uint32), num), O.jax_split(np.array(key, np.uint32), num))


@pytest.mark.gpu
def test_jax_streams_device_reproduces_the_documented_values_and_the_oracle():
    from aux_ssm_samplers_amd import random as R
    from oracle import rng_np as O
    _jax_known_answers(R.jax_split, lambda k, n: R.jax_uniform(k, (n,), np.float32), lambda k, n: R.jax_normal(k, (n,), np.float32))
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 64, 1001):
        keys = rng.integers(0, 2 ** 32, (5, 2), dtype=np.uint64).astype(np.uint32)
        for dtype in (np.float32, np.float64):
            u = R.jax_uniform(keys, (n,), dtype)
            z = R.jax_normal(keys, (n,), dtype)
            ub = R.jax_uniform(keys, (n,), dtype, minval=-2.0, maxval=3.0)
            assert u.shape == (5, n) and u.dtype == dtype
            for c in range(5):
                npt.assert_array_equal(u[c], O.jax_uniform(keys[c], n, dtype))                      # integer / exact float operations: bit for bit
                npt.assert_array_equal(ub[c], O.jax_uniform(keys[c], n, dtype, -2.0, 3.0))
                tol = dict(rtol=3e-6, atol=3e-7) if dtype == np.float32 else dict(rtol=1e-13, atol=1e-14)   # (erfinv: log1p / sqrt of the platform)
                npt.assert_allclose(z[c], O.jax_normal(keys[c], n, dtype), **tol)
    # shapes: scalar, matrix; one key
    assert np.shape(R.jax_uniform(keys[0], ())) == () and R.jax_normal(keys[0], (7, 3), np.float64).shape == (7, 3)
    npt.assert_array_equal(R.jax_normal(keys[0], (7, 3), np.float64).ravel(), R.jax_normal(keys[0], (21,), np.float64))
    # a large float64 draw reaches the tails (|u| > 1 - 6e-8, where a float32 start of erfinv would be infinite): all finite, the tail counts as expected, and the
    # extreme values against SciPy's erfinv on the oracle's uniforms
    kb, nb = np.array([1, 2], np.uint32), 60_000_000
    big = R.jax_normal(kb, (nb,), np.float64)
    assert np.isfinite(big).all() and abs(big.mean()) < 6e-4 and abs(big.std() - 1) < 4e-4
    n5 = int((np.abs(big) > 5.0).sum())
    assert 10 <= n5 <= 65, n5                                   # expected 34.4
    ext = np.argsort(-np.abs(big))[:50]
    from scipy.special import erfinv
    ub = O.jax_uniform(kb, nb, np.float64, np.nextafter(-1.0, 0.0), 1.0)
    npt.assert_allclose(big[ext], np.sqrt(2.0) * erfinv(ub[ext]), rtol=1e-13)


@pytest.mark.gpu
def test_kernels_draw_what_the_reference_draws_from_the_key_in_jax_mode():
    """random.set_compat("jax"): kalman.get_kernel's kernel(key, state, delta) and the auxiliary particle-Gibbs kernel consume split(key, ...) / normal / uniform exactly
    as the reference's code does (kalman/generic.py:58-73; csmc/generic.py:64-67 + _primitives/csmc/csmc.py:53, :71-85, :111, :129-138), so the result equals the
    explicit-noise sweep on arrays assembled HERE from the oracle's key arithmetic and the device's jax-compatible fills."""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.kalman import get_kernel
    from aux_ssm_samplers_amd.csmc import get_independent_kernel, GaussianInit, LinearGaussianDynamics, SVPotential
    from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup, sv_setup
    from oracle import rng_np as O
    from oracle import kalman_np as K
    key = np.array([2023, 7], np.uint32)
    prev = R.set_compat("jax")
    try:
        # --- Kalman: Lorenz model, one chain, fp64: against the oracle's sweep on the oracle's jax draws
        T = 50
        model, xtrue = lorenz_kalman_setup(T, seed=3)
        init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
        x = xtrue + 0.05 * np.random.default_rng(1).standard_normal((T, 3))
        out = kernel(key, init(x), 0.02)
        ka, ks, kc = O.jax_split(key, 3)
        noise = dict(eps_aux=O.jax_normal(ka, 3 * T, np.float64).reshape(T, 3), eps_samp=O.jax_normal(ks, 3 * T, np.float64).reshape(T, 3),
                     u_accept=float(O.jax_uniform(kc, 1, np.float64)[0]))
        ref = K.kalman_sweep(x, 0.02, model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True, **noise)
        npt.assert_allclose(out.x, ref["x"], rtol=1e-8, atol=1e-9)
        assert out.updated == ref["accepted"]
        # several chains: one key per chain (what jax.vmap(kernel) is given), and the host-factory path draws the same
        keys = O.jax_split(key, 3)
        xs = np.stack([x, x + 0.01, x - 0.01])
        outs = kernel(keys, init(xs), 0.02)
        npt.assert_allclose(outs.x[0], kernel(keys[0], init(xs[0]), 0.02).x, rtol=1e-9, atol=1e-10)
        # resident chains in either layout: the draws are written straight into the chains' noise buffers on the device
        from aux_ssm_samplers_amd import _lib
        from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
        for cm in (False, True):
            ch = DeviceChains(_lib.default_handle(), xs, chain_minor=cm)
            kernel(keys, KalmanSampler(x=ch, updated=None), 0.02)
            npt.assert_allclose(ch.to_host(), outs.x, rtol=1e-9, atol=1e-10)
            for c in range(3):   # every chain against the oracle's sweep on the oracle's draws for ITS key
                ka_, ks_, kc_ = O.jax_split(keys[c], 3)
                nz_ = dict(eps_aux=O.jax_normal(ka_, 3 * T, np.float64).reshape(T, 3), eps_samp=O.jax_normal(ks_, 3 * T, np.float64).reshape(T, 3),
                           u_accept=float(O.jax_uniform(kc_, 1, np.float64)[0]))
                rf = K.kalman_sweep(xs[c], 0.02, model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True, **nz_)
                npt.assert_allclose(ch.to_host()[c], rf["x"], rtol=1e-8, atol=1e-9)
        ih, kh = get_kernel(lambda z: model.dynamics_factory(z), lambda z, u, dl: model.observations_factory(z, u, dl), lambda z: model.log_likelihood_fn(z), True)
        npt.assert_allclose(kh(key, ih(x), 0.02).x, out.x, rtol=1e-8, atol=1e-9)
        # --- auxiliary particle Gibbs, independent proposals, both backward modes, fp32
        Tc, N, d = 20, 32, 1
        y, xtrue, (m0, P0, F, Q, b) = sv_setup(Tc, d, seed=2, rho=0.0)
        M0, Mt = GaussianInit(m0=m0, P0=P0), LinearGaussianDynamics(F=F, b=b, Q=Q)
        x0 = xtrue.astype(np.float32)
        for backward in (True, False):
            init_c, kern_c = get_independent_kernel(M0, SVPotential(y=y[0]), Mt, SVPotential(params=y[1:]), N, backward=backward, Pt=Mt)
            got = kern_c(key, init_c(x0), 0.5)
            aux_key, k = O.jax_split(key, 2)
            k_fwd, k_bwd = O.jax_split(k, 2)
            kt = O.jax_split(k_fwd, Tc)
            f32 = np.float32
            eps_prop = np.stack([R.jax_normal(kt[0], (N, d), f32)] + [R.jax_normal(O.jax_split(kt[t], 2)[1], (N, d), f32) for t in range(1, Tc)])
            u_res = np.stack([R.jax_uniform(O.jax_split(kt[t], 2)[0], (N,), f32) for t in range(1, Tc)])
            u_bwd = np.zeros(Tc, f32)
            if backward:
                kb = O.jax_split(k_bwd, Tc)
                for t in range(Tc):
                    u_bwd[t] = R.jax_uniform(kb[Tc - 1 - t], (), f32)
            else:
                u_bwd[Tc - 1] = R.jax_uniform(k_bwd, (), f32)
            R.set_compat(None)
            want = kern_c(None, init_c(x0), 0.5, noise=dict(eps_aux=R.jax_normal(aux_key, (Tc, d), f32), eps_prop=eps_prop, u_res=u_res, u_bwd=u_bwd))
            R.set_compat("jax")
            npt.assert_array_equal(got.x, want.x)
            npt.assert_array_equal(got.updated, want.updated)
            assert got.updated.any()
        # --- the plain cSMC kernel (bootstrap proposals): key_fwd, key_bwd = split(key) with no auxiliary split (_primitives/csmc/csmc.py:52-59)
        from aux_ssm_samplers_amd._primitives.csmc import get_kernel as get_csmc_kernel
        from aux_ssm_samplers_amd.csmc import _device
        init_b, kern_b = get_csmc_kernel(M0, SVPotential(y=y[0]), Mt, SVPotential(params=y[1:]), N, backward=True, Pt=Mt)
        got = kern_b(key, init_b(x0))
        k_fwd, k_bwd = O.jax_split(key, 2)
        kt, kb = O.jax_split(k_fwd, Tc), O.jax_split(k_bwd, Tc)
        noise = dict(eps_prop=np.stack([R.jax_normal(kt[0], (N, d), f32)] + [R.jax_normal(O.jax_split(kt[t], 2)[1], (N, d), f32) for t in range(1, Tc)]),
                     u_res=np.stack([R.jax_uniform(O.jax_split(kt[t], 2)[0], (N,), f32) for t in range(1, Tc)]),
                     u_bwd=np.array([R.jax_uniform(kb[Tc - 1 - t], (), f32) for t in range(Tc)], f32))
        R.set_compat(None)
        xw, ancw, _ = _device.sweep(_device.describe_bootstrap(M0, SVPotential(y=y[0]), Mt, SVPotential(params=y[1:]), Mt), x0, N, True, noise=noise)
        R.set_compat("jax")
        npt.assert_array_equal(got.x, xw)
        npt.assert_array_equal(got.updated, ancw != 0)
        # --- the parallel-in-time kernel (csmc/independent.py:105-110 + pit/csmc.py:70-75 + pit/operator.py:76-81): T not a power of two, the root stitch's single draw
        Tp = 21
        yp, xtp, (m0p, P0p, Fp, Qp, bp) = sv_setup(Tp, d, seed=5, rho=0.0)
        M0p, Mtp = GaussianInit(m0=m0p, P0=P0p), LinearGaussianDynamics(F=Fp, b=bp, Q=Qp)
        init_p, kern_p = get_independent_kernel(M0p, SVPotential(y=yp[0]), Mtp, SVPotential(params=yp[1:]), N, parallel=True)
        xp0 = xtp.astype(np.float32)
        got = kern_p(key, init_p(xp0), 0.5)
        aux_key, k = O.jax_split(key, 2)
        sk, rk = O.jax_split(k, 2)
        sks, rks = O.jax_split(sk, Tp), O.jax_split(rk, Tp)
        u_res = np.stack([R.jax_uniform(rks[t], (N,), f32) for t in range(Tp)])
        u_res[16, 0] = R.jax_uniform(rks[16], (), f32)                       # the last stitch of the tree: boundary 2^(ceil(log2 21) - 1) = 16, one draw of shape ()
        noise = dict(eps_aux=R.jax_normal(aux_key, (Tp, d), f32), eps_prop=np.stack([R.jax_normal(sks[t], (N, d), f32) for t in range(Tp)]), u_res=u_res)
        R.set_compat(None)
        want = kern_p(None, init_p(xp0), 0.5, noise=noise)
        R.set_compat("jax")
        npt.assert_array_equal(got.x, want.x)
        npt.assert_array_equal(got.ancestors, want.ancestors)
        assert got.ancestors.any()
    finally:
        R.set_compat(prev)


@pytest.mark.gpu
def test_loop_in_jax_mode_splits_keys_as_the_reference_loop_does():
    """loop() (examples/stochastic_volatility/experiment.py:88-128: keys = jax.random.split(key, n_iter), one sweep per key) on one resident Kalman chain in jax mode
    == the same sweeps issued by hand with jax_split keys; resident particle chains refuse the mode (their draws are made inside the kernels)."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.kalman import get_kernel, SVModel
    from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler
    from aux_ssm_samplers_amd.loop import loop
    from aux_ssm_samplers_amd.workloads import sv_setup
    T = 40
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, 1, seed=3, rho=0.0)
    model = SVModel(y, m0, P0, F, Q, b, order=2)
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    h = _lib.default_handle()
    key = np.array([5, 6], np.uint32)
    prev = R.set_compat("jax")
    try:
        a = DeviceChains(h, xtrue[None])
        loop(key, 0.5, KalmanSampler(x=a, updated=None), kernel, None, 5)
        bch = DeviceChains(h, xtrue[None])
        for k in R.jax_split(key, 5):
            kernel(k, KalmanSampler(x=bch, updated=None), 0.5)
        npt.assert_array_equal(a.to_host(), bch.to_host())
        # the module's own split / normal / uniform follow jax.random while the mode is on
        npt.assert_array_equal(R.split(np.array([0, 0], np.uint32), 2), np.array(JAX_KAT["split0"], np.uint32))
        npt.assert_allclose(R.normal(np.array([0, 0], np.uint32), (1,), np.float32)[0], JAX_KAT["normal0"], atol=2e-7)
        npt.assert_allclose(R.uniform(np.array([0, 0], np.uint32), (1,), np.float32)[0], JAX_KAT["uniform0"], atol=6e-8)
        from aux_ssm_samplers_amd.csmc import get_independent_kernel, CsmcChains, CSMCState, GaussianInit, LinearGaussianDynamics, SVPotential
        ic, kc = get_independent_kernel(GaussianInit(m0=m0, P0=P0), SVPotential(y=y[0]), LinearGaussianDynamics(F=F, b=b, Q=Q), SVPotential(params=y[1:]), 16)
        with pytest.raises(NotImplementedError):
            kc(key, CSMCState(x=CsmcChains(h, xtrue[None].astype(np.float32), delta=0.5), updated=None), None)
    finally:
        R.set_compat(prev)

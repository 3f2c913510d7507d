"""GPU parity tests of the conditional-SMC path: HIP kernels through the C ABI vs the plain-C oracle
(oracle/csmc_ref.c) on identical explicit noise.  The contract (BASELINE north_star): ancestor indices BIT-EXACT;
here particles, log-weights and trajectories are bit-exact too, because both sides use the same fixed reduction
orders and the same bit-reproducible exp/log.  Plus the reference's own statistical tests on the HIP path."""
import numpy as np
import numpy.testing as npt
import pytest

from oracle import csmc as O

from tests.helpers import lorenz_setup  # noqa: E402,F401

pytestmark = pytest.mark.gpu


def _models(d, rng):
    from aux_ssm_samplers_amd.csmc import GaussianInit, LinearGaussianDynamics
    A = rng.standard_normal((d, d))
    Q = A @ A.T / d + 0.5 * np.eye(d)
    F = 0.9 * np.eye(d) + 0.05 * rng.standard_normal((d, d))
    b = 0.1 * rng.standard_normal(d)
    P0 = 2.0 * np.eye(d)
    m0 = 0.1 * rng.standard_normal(d)
    return GaussianInit(m0=m0, P0=P0), LinearGaussianDynamics(F=F, b=b, Q=Q)


def _odesc(proposal, potential, M0, Mt, sig_y=1.0):
    return dict(proposal=proposal, potential=potential, m0=M0.m0, chol_P0=M0.chol(), F=Mt.F, b=Mt.b, chol_Q=Mt.chol(), sig_y=sig_y)


def _pot(kind, y, sig=0.7):
    from aux_ssm_samplers_amd.csmc import FlatPotential, GaussianObsPotential, SVPotential
    if kind == O.POT_FLAT:
        return FlatPotential(), FlatPotential()
    if kind == O.POT_GAUSS_OBS:
        return GaussianObsPotential(sig=sig, y=y[0]), GaussianObsPotential(sig=sig, params=y[1:])
    return SVPotential(y=y[0]), SVPotential(params=y[1:])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,N,T", [(1, 32, 5), (1, 1024, 300), (2, 100, 64), (3, 512, 40), (4, 65, 33)])
@pytest.mark.parametrize("proposal", [O.BOOTSTRAP_LG, O.AUX_INDEPENDENT])
@pytest.mark.parametrize("potential", [O.POT_FLAT, O.POT_GAUSS_OBS, O.POT_SV])
@pytest.mark.parametrize("backward", [True, False])
def test_sweep_bit_exact_vs_oracle(dtype, d, N, T, proposal, potential, backward):
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(1000 * d + N + T)
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(potential, y)
    sig = 0.7
    x0 = rng.standard_normal((T, d)).astype(dtype)
    noise = dict(eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T))
    delta = None
    okw = {}
    if proposal == O.AUX_INDEPENDENT:
        delta = 0.5 + rng.random(T)  # time-varying delta (csmc/generic.py:61-63 accepts a vector)
        noise["eps_aux"] = rng.standard_normal((T, d))
        fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
        okw = dict(sqrt_half_delta=np.sqrt(0.5 * delta), eps_aux=noise["eps_aux"])
    else:
        fk = _device.describe_bootstrap(M0, G0, Mt, Gt, Mt)
    noise32 = {k: np.asarray(v, dtype) for k, v in noise.items()}
    x, anc, hist = _device.sweep(fk, x0, N, backward, noise={k: v[None] for k, v in noise32.items()}, delta=delta, want_history=True)
    ref = O.sweep(_odesc(proposal, potential, M0, Mt, sig), x0, N, backward, y=y if potential else None, eps_prop=noise["eps_prop"],
                  u_res=noise["u_res"], u_bwd=noise["u_bwd"], dtype=dtype, **okw)
    npt.assert_array_equal(hist["xs"], ref["xs"])
    npt.assert_array_equal(hist["log_ws"], ref["log_ws"])
    npt.assert_array_equal(hist["As"], ref["As"])          # resampling ancestors: bit-exact
    npt.assert_array_equal(anc, ref["ancestors"])          # backward indices: bit-exact
    npt.assert_array_equal(x, ref["x"])
    assert np.all(hist["As"][:, 0] == 0) and np.all(hist["xs"][:, 0] == x0)


@pytest.mark.parametrize("T", [50, 51, 1, 2])
def test_multichain_equals_single_chain_and_threefry_equals_explicit(T):
    """C chains in one launch == C single launches; in-kernel Threefry noise == the same draws materialised by the fill
    kernels (_device.key_noise: streams 1..4, two time steps per Threefry block for the forward pass) fed back as
    explicit arrays."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(7)
    d, N, C = 2, 128, 5
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(O.POT_SV, y)
    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
    x0 = rng.standard_normal((C, T, d)).astype(np.float32)
    key = R.PRNGKey(99)
    h = _lib.default_handle()
    xa, anca, _ = _device.sweep(fk, x0, N, True, key=key, delta=0.5)
    noise = _device.key_noise(h, key, C, T, N, d, np.float32)
    xb, ancb, _ = _device.sweep(fk, x0, N, True, noise=noise, delta=0.5)
    npt.assert_array_equal(xa, xb)
    npt.assert_array_equal(anca, ancb)
    for c in range(C):
        xc, ancc, _ = _device.sweep(fk, x0[c], N, True, noise={k: v[c:c + 1] for k, v in noise.items()}, delta=0.5)
        npt.assert_array_equal(xc, xb[c])
        npt.assert_array_equal(ancc, ancb[c])


def test_c3_shape_instantiation_equals_the_generic_kernel_bit_for_bit(monkeypatch):
    """config C3's shape (SV potential, d = 1, N = 1024, fp32, independent auxiliary proposals, backward sampling, draws generated inside the forward kernel)
    runs k_csmc_fwd<float, 1, false, false, 16, SP = 1>, whose model switches are folded at compile time; the same key as explicit arrays runs the generic
    instantiation: trajectories and ancestors must be identical.  (Few chains would otherwise take the pre-generated draws, i.e. the generic kernel.)"""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd import _lib
    from aux_ssm_samplers_amd.csmc import _device
    monkeypatch.setenv("AUXSSM_CSMC_NO_PREGEN", "1")
    rng = np.random.default_rng(17)
    d, N, C, T = 1, 1024, 3, 301
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    y[5] = 0.0       # (no finite bound of the potential at y = 0: the exact-maximum branch)
    G0, Gt = _pot(O.POT_SV, y)
    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
    x0 = rng.standard_normal((C, T, d)).astype(np.float32)
    key = R.PRNGKey(123)
    xa, anca, _ = _device.sweep(fk, x0, N, True, key=key, delta=0.3)
    noise = _device.key_noise(_lib.default_handle(), key, C, T, N, d, np.float32)
    xb, ancb, _ = _device.sweep(fk, x0, N, True, noise=noise, delta=0.3)
    npt.assert_array_equal(xa, xb)
    npt.assert_array_equal(anca, ancb)
    assert (anca != 0).mean() > 0.5


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,N,T,C", [(1, 64, 2, 3), (1, 100, 7, 2), (3, 512, 8, 4), (2, 33, 65, 5), (4, 1024, 5, 2)])
@pytest.mark.parametrize("proposal", ["independent", "bootstrap"])
def test_draws_generated_ahead_of_the_forward_pass_equal_in_kernel_draws(dtype, d, N, T, C, proposal, monkeypatch):
    """Fewer chains than CUs: the forward pass's Threefry draws are written out by a separate full-chip kernel first (csmc_dev.h::k_csmc_pregen) and read back
    as explicit arrays.  Same trajectories, ancestors, particle systems and log-weights bit for bit as with the draws made inside the pass
    (AUXSSM_CSMC_NO_PREGEN=1), for odd and even T, partial last waves and every state dimension of the register kernels."""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(d * 100 + T)
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(O.POT_SV, y)
    fk = (_device.describe_independent if proposal == "independent" else _device.describe_bootstrap)(M0, G0, Mt, Gt, Mt)
    x0 = rng.standard_normal((C, T, d)).astype(dtype)
    kw = dict(key=R.PRNGKey(31), delta=0.5) if proposal == "independent" else dict(key=R.PRNGKey(31))
    monkeypatch.delenv("AUXSSM_CSMC_NO_PREGEN", raising=False)
    xa, anca, exa = _device.sweep(fk, x0, N, True, want_history=True, **kw)
    monkeypatch.setenv("AUXSSM_CSMC_NO_PREGEN", "1")
    xb, ancb, exb = _device.sweep(fk, x0, N, True, want_history=True, **kw)
    npt.assert_array_equal(xa, xb)
    npt.assert_array_equal(anca, ancb)
    for k in exa:
        npt.assert_array_equal(exa[k], exb[k])
    assert len({xa[c].tobytes() for c in range(C)}) == C


@pytest.mark.parametrize("backward", [True, False])
@pytest.mark.parametrize("mode", ["threefry", "explicit"])
def test_chain_batched_sweep_equals_one_launch(backward, mode, monkeypatch):
    """A chain's particle system is T N (D + 1) reals (537 MB at C3), so a sweep over more chains than the device holds runs the forward + backward pair
    batch by batch (csrc/csmc.hip, CsmcArgs::c0).  Forced here with AUXSSM_CSMC_BATCH = 2 on 5 chains: the same trajectories and ancestors bit for bit,
    for in-kernel Threefry noise (streams indexed by the global chain) and explicit arrays, backward sampling and ancestor tracing (As in the workspace)."""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(8)
    d, N, C, T = 2, 192, 5, 37
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(O.POT_GAUSS_OBS, y)
    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
    x0 = rng.standard_normal((C, T, d)).astype(np.float32)
    key = R.PRNGKey(123)
    kw = dict(key=key) if mode == "threefry" else dict(noise=_device.key_noise(_lib.default_handle(), key, C, T, N, d, np.float32))
    monkeypatch.delenv("AUXSSM_CSMC_BATCH", raising=False)
    xa, anca, _ = _device.sweep(fk, x0, N, backward, delta=0.5, **kw)
    for cb in ("2", "1", "4"):
        monkeypatch.setenv("AUXSSM_CSMC_BATCH", cb)
        xb, ancb, _ = _device.sweep(fk, x0, N, backward, delta=0.5, **kw)
        npt.assert_array_equal(xa, xb)
        npt.assert_array_equal(anca, ancb)
    assert len({xa[c].tobytes() for c in range(C)}) == C  # (the chains do differ)


@pytest.mark.parametrize("backward", [True, False])
def test_flat_potential_reference_statistical_test(backward):
    """aux_samplers/_primitives/test_csmc/test_csmc.py::test_flat_potential (:18-69) on the HIP path: AR(1) prior is
    invariant: mean 0, var 1, lag-1 cov rho, atol 0.05.  2048 chains x 40 sweeps instead of 1 chain x 50_000."""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd._primitives.csmc import get_kernel
    from aux_ssm_samplers_amd.csmc import GaussianInit, LinearGaussianDynamics, FlatPotential
    T, N, rho, C, M = 5, 32, 0.9, 2048, 40
    M0 = GaussianInit(m0=[0.0], P0=[[1.0]])
    Mt = LinearGaussianDynamics(F=[[rho]], b=[0.0], Q=[[1 - rho ** 2]])
    init, kernel = get_kernel(M0, FlatPotential(), Mt, FlatPotential(), N=N, backward=backward, Pt=Mt)
    rng = np.random.default_rng(0)
    state = init(rng.standard_normal((C, T, 1)).astype(np.float32))
    keys = R.split(R.PRNGKey(0), M)
    out = []
    for it in range(M):
        state = kernel(keys[it], state)
        if it >= M // 4:
            out.append(state.x[:, :, 0])
    xs = np.concatenate(out)
    cov = np.cov(xs, rowvar=False)
    npt.assert_allclose(xs.mean(0), 0.0, atol=0.05)
    npt.assert_allclose(np.diag(cov), 1.0, atol=0.05)
    npt.assert_allclose(np.diag(cov, 1), rho, atol=0.05)


@pytest.mark.parametrize("backward", [True, False])
def test_flat_potential_at_the_reference_protocol(backward):
    """the same known answer at the reference's OWN protocol (test_csmc.py:18-69): ONE chain, T = 5, N = 32, 50_000 sweeps, the first
    10 % discarded, atol 0.05 -- the literal restatement and the contract oracle run it in tests/test_oracle_csmc_literal.py."""
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd._primitives.csmc import get_kernel
    from aux_ssm_samplers_amd.csmc import GaussianInit, LinearGaussianDynamics, FlatPotential
    T, N, rho, M = 5, 32, 0.9, 50_000
    M0 = GaussianInit(m0=[0.0], P0=[[1.0]])
    Mt = LinearGaussianDynamics(F=[[rho]], b=[0.0], Q=[[1 - rho ** 2]])
    init, kernel = get_kernel(M0, FlatPotential(), Mt, FlatPotential(), N=N, backward=backward, Pt=Mt)
    state = init(np.random.default_rng(0).standard_normal((T, 1)).astype(np.float32))
    keys = R.split(R.PRNGKey(0), M)
    xs = np.empty((M, T))
    for it in range(M):
        state = kernel(keys[it], state)
        xs[it] = state.x[:, 0]
    xs = xs[M // 10:]
    cov = np.cov(xs, rowvar=False)
    npt.assert_allclose(xs.mean(0), 0.0, atol=0.05)
    npt.assert_allclose(np.diag(cov), 1.0, atol=0.05)
    npt.assert_allclose(np.diag(cov, 1), rho, atol=0.05)


def test_independent_kernel_api_and_C3_shape_smoke():
    """csmc.get_independent_kernel on the SV model of BASELINE config C3 (d=1, N=1024, backward sampling), reduced T.
    Checks the API surface (init/kernel/CSMCState) and bit-exactness against the oracle given the same noise."""
    from aux_ssm_samplers_amd.csmc import get_independent_kernel, GaussianInit, LinearGaussianDynamics, SVPotential
    T, N = 2048, 1024
    phi, tau = 0.9, 2.0
    q = tau / (1 - phi ** 2)  # examples/stochastic_volatility/model.py:34-53
    rng = np.random.default_rng(5)
    xtrue = np.zeros(T)
    xtrue[0] = np.sqrt(q) * rng.standard_normal()
    for t in range(1, T):
        xtrue[t] = phi * xtrue[t - 1] + np.sqrt(q) * rng.standard_normal()
    y = (np.exp(0.5 * xtrue) * rng.standard_normal(T))[:, None]
    M0 = GaussianInit(m0=[0.0], P0=[[q]])
    Mt = LinearGaussianDynamics(F=[[phi]], b=[0.0], Q=[[q]])
    init, kernel = get_independent_kernel(M0, SVPotential(y=y[0]), Mt, SVPotential(params=y[1:]), N, backward=True, Pt=Mt)
    x0 = xtrue[:, None].astype(np.float32)
    noise = dict(eps_aux=rng.standard_normal((1, T, 1)), eps_prop=rng.standard_normal((1, T, N, 1)),
                 u_res=rng.random((1, T - 1, N)), u_bwd=rng.random((1, T)))
    noise = {k: v.astype(np.float32) for k, v in noise.items()}
    st = kernel(None, init(x0), 0.5, noise=noise)
    assert st.x.shape == (T, 1) and st.updated.shape == (T,) and st.updated.dtype == bool
    ref = O.sweep(dict(proposal=O.AUX_INDEPENDENT, potential=O.POT_SV, m0=[0.0], chol_P0=[[np.sqrt(q)]], F=[[phi]], b=[0.0],
                       chol_Q=[[np.sqrt(q)]]), x0, N, True, y=y, sqrt_half_delta=np.full(T, np.sqrt(0.25)),
                  eps_aux=noise["eps_aux"][0], eps_prop=noise["eps_prop"][0], u_res=noise["u_res"][0], u_bwd=noise["u_bwd"][0])
    npt.assert_array_equal(st.ancestors, ref["ancestors"])
    npt.assert_array_equal(st.x, ref["x"])
    assert st.updated.mean() > 0.5  # backward sampling moves most time steps


def test_unsupported_python_models_fail_loudly():
    from aux_ssm_samplers_amd._primitives.csmc import get_kernel
    from aux_ssm_samplers_amd.csmc import get_independent_kernel, get_generic_kernel, GaussianInit, LinearGaussianDynamics, FlatPotential

    class MyDyn:
        def sample(self, key, x, p):
            return x

    M0 = GaussianInit(m0=[0.0], P0=[[1.0]])
    Mt = LinearGaussianDynamics(F=[[0.5]], b=[0.0], Q=[[1.0]])
    with pytest.raises(NotImplementedError):
        get_kernel(M0, FlatPotential(), MyDyn(), FlatPotential(), N=8)
    # (gradient proposals in the parallel-in-time sweep: built in round 3, tests/test_csmc_gradient_timevarying.py)
    assert callable(get_independent_kernel(M0, FlatPotential(), Mt, FlatPotential(), 8, gradient=True, parallel=True)[1])
    with pytest.raises(NotImplementedError):
        get_generic_kernel(lambda u, s: None, 8)
    with pytest.raises(ValueError):
        get_generic_kernel(lambda u, s: None, 8, backward=True)  # csmc/generic.py:44-45


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N", [2, 10, 64, 100, 1024])
def test_normalize_and_multinomial_primitives_bit_exact(dtype, N):
    """math/utils.py::normalize and resamplings.py::multinomial as standalone device primitives vs the C oracle."""
    from aux_ssm_samplers_amd._primitives.math import normalize
    from aux_ssm_samplers_amd._primitives.csmc.resamplings import multinomial
    rng = np.random.default_rng(N)
    lw = (5 * rng.standard_normal((7, N))).astype(dtype)
    w = normalize(lw)
    u = rng.random((7, N)).astype(dtype)
    idx = multinomial(None, w, u=u)
    for r in range(7):
        wr = O.normalize(lw[r], dtype)
        npt.assert_array_equal(w[r], wr)
        npt.assert_array_equal(idx[r], O.multinomial(wr, u[r], dtype))
    assert np.all(idx[:, 0] == 0)


def test_multinomial_resampling_reference_statistical_test():
    # test_csmc/test_resamplings.py:11-24 on the HIP path: index 0 always 0, the others ~ weights (atol 1e-3)
    from aux_ssm_samplers_amd import random as R
    from aux_ssm_samplers_amd._primitives.csmc.resamplings import multinomial
    rng = np.random.default_rng(42)
    w = rng.random(10)
    w /= w.sum()
    idx = multinomial(R.PRNGKey(42), np.broadcast_to(w, (100_000, 10)).copy())
    assert np.all(idx[:, 0] == 0)
    cnt = np.bincount(idx[:, 1:].ravel(), minlength=10)
    npt.assert_allclose(cnt / cnt.sum(), w, atol=2e-3)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N,T", [(64, 40), (512, 130), (1000, 33)])
@pytest.mark.parametrize("proposal", [O.BOOTSTRAP_LG, O.AUX_INDEPENDENT])
@pytest.mark.parametrize("backward", [True, False])
def test_lorenz63_sweep_bit_exact_vs_oracle(dtype, N, T, proposal, backward):
    """BASELINE config C4's Feynman-Kac model (stochastic Lorenz-63, Euler-Maruyama transition, x2/x3 observed sparsely, NaN = missing)
    through the same kernels: particles, log-weights, resampling ancestors and backward indices bit-exact vs oracle/csmc_ref.c."""
    from aux_ssm_samplers_amd.csmc import _device
    M0, Mt, G0, Gt, xtrue, y, sig_y = lorenz_setup(T, seed=N + T)
    rng = np.random.default_rng(77 + N)
    d = 3
    x0 = (xtrue + 0.1 * rng.standard_normal((T, d))).astype(dtype)
    noise = dict(eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T))
    delta, okw = None, {}
    if proposal == O.AUX_INDEPENDENT:
        delta = 0.05 + 0.05 * rng.random(T)
        noise["eps_aux"] = rng.standard_normal((T, d))
        fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
        okw = dict(sqrt_half_delta=np.sqrt(0.5 * delta), eps_aux=noise["eps_aux"])
    else:
        fk = _device.describe_bootstrap(M0, G0, Mt, Gt, Mt)
    noise32 = {k: np.asarray(v, dtype) for k, v in noise.items()}
    x, anc, hist = _device.sweep(fk, x0, N, backward, noise={k: v[None] for k, v in noise32.items()}, delta=delta, want_history=True)
    F = np.zeros((3, 3))
    F[0] = Mt.theta
    od = dict(proposal=proposal, potential=O.POT_GAUSS_OBS_MASKED, m0=M0.m0, chol_P0=M0.chol(), F=F, b=[Mt.dt, 0, 0], chol_Q=Mt.chol(),
              sig_y=sig_y, transition=O.TRANS_LORENZ63_EM)
    ref = O.sweep(od, x0, N, backward, y=y, eps_prop=noise["eps_prop"], u_res=noise["u_res"], u_bwd=noise["u_bwd"], dtype=dtype, **okw)
    npt.assert_array_equal(hist["xs"], ref["xs"])
    npt.assert_array_equal(hist["log_ws"], ref["log_ws"])
    npt.assert_array_equal(hist["As"], ref["As"])
    npt.assert_array_equal(anc, ref["ancestors"])
    npt.assert_array_equal(x, ref["x"])
    assert np.all(np.isfinite(hist["log_ws"]))


def test_lorenz63_particle_gibbs_tracks_the_truth():
    """Statistical sanity of the Lorenz model through the public kernel API (csmc.get_kernel machinery is model-agnostic): a few
    bootstrap particle-Gibbs sweeps from a poor start move the trajectory to the observations (x2, x3 RMSE well under the prior
    spread) -- a wrong drift or a mishandled NaN row cannot do that."""
    from aux_ssm_samplers_amd.csmc import _device
    T, N = 200, 1024
    M0, Mt, G0, Gt, xtrue, y, sig_y = lorenz_setup(T, seed=5, every=4)
    fk = _device.describe_bootstrap(M0, G0, Mt, Gt, Mt)
    x = np.repeat(np.array([[1.5, -1.5, 25.0]]), T, axis=0)
    for k in range(6):
        x, anc, _ = _device.sweep(fk, x, N, True, key=100 + k)
    rmse = np.sqrt(np.mean((x[:, 1:] - xtrue[:, 1:]) ** 2))
    assert rmse < 3.0, rmse
    assert (anc != 0).mean() > 0.5


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("M,N", [(10, 10), (100, 100), (1000, 100), (50, 200), (1024, 1024), (1, 5)])
def test_systematic_resampling_bit_exact_vs_oracle(dtype, M, N):
    """conditional systematic resampling (resamplings.py:40-86) on the device vs the C oracle (itself equal to Chopin & Singh's Algorithm 4,
    tests/test_oracle_csmc.py), rows of independent weight vectors in one launch, explicit (U, V, W)."""
    from aux_ssm_samplers_amd._primitives.csmc.resamplings import systematic
    rng = np.random.default_rng(M * 7 + N)
    rows = 64
    w = (rng.random((rows, M)) ** 2).astype(dtype)
    w[3] = 0
    w[3, 0] = 1                      # all the mass on the conditioned particle
    if M > 1:
        w[5, 0] = 0                  # a conditioned particle of zero weight
    w = (w / w.sum(1, keepdims=True)).astype(dtype)
    uvw = rng.random((rows, 3)).astype(dtype)
    got = systematic(None, w, N, uvw=uvw)
    assert got.shape == (rows, N)
    for r in range(rows):
        npt.assert_array_equal(got[r], O.systematic(w[r], uvw[r], N, dtype=dtype))
    ok = np.ones(rows, bool)
    if M > 1:
        ok[5] = False
    assert np.all(got[ok, 0] == 0)
    one = systematic(7, w[0], N)     # keyed draw, single vector
    assert one.shape == (N,) and one[0] == 0 and one.max() < M


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N", [64, 512, 1024])
def test_backward_weights_fall_back_to_the_exact_maximum_after_underflow(dtype, N):
    """The backward pass shifts its weights by a reduction-free BOUND of their maximum (forward block maximum + transition log-normaliser, sweep
    contract).  With a transition far tighter than the independent proposal every such weight underflows and the step must fall back to the exact
    maximum -- in the kernels and in the oracle alike (bit-exact), and the draw must still pick the particle the transition favours."""
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics
    d, T = 1, 24
    rng = np.random.default_rng(5 + N)
    M0 = GaussianInit(m0=np.zeros(d), P0=np.eye(d))
    Mt = LinearGaussianDynamics(F=0.9 * np.eye(d), b=np.zeros(d), Q=1e-8 * np.eye(d))   # sd 1e-4 against proposals of sd 0.5
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(O.POT_GAUSS_OBS, y)
    x0 = rng.standard_normal((T, d)).astype(dtype)
    delta = np.full(T, 0.5)
    noise = dict(eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T), eps_aux=rng.standard_normal((T, d)))
    fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
    noise32 = {k: np.asarray(v, dtype) for k, v in noise.items()}
    x, anc, hist = _device.sweep(fk, x0, N, True, noise={k: v[None] for k, v in noise32.items()}, delta=delta, want_history=True)
    ref = O.sweep(_odesc(O.AUX_INDEPENDENT, O.POT_GAUSS_OBS, M0, Mt, 0.7), x0, N, True, y=y, eps_prop=noise["eps_prop"], u_res=noise["u_res"],
                  u_bwd=noise["u_bwd"], dtype=dtype, sqrt_half_delta=np.sqrt(0.5 * delta), eps_aux=noise["eps_aux"])
    npt.assert_array_equal(anc, ref["ancestors"])
    npt.assert_array_equal(x, ref["x"])
    # the bound really underflows on some steps: log p(x_{t+1} | x_t^i) - c_t = -z_i^2 / 2 with z of order 0.5 / 1e-4; a step falls back when
    # even the nearest particle is beyond the exp underflow of the dtype
    xs = hist["xs"].astype(np.float64)
    lim = 2 * (88.0 if dtype == np.float32 else 746.0)
    zmin = [(((float(x[k + 1, 0]) - 0.9 * xs[k, :, 0]) / 1e-4) ** 2).min() for k in range(T - 1)]
    assert sum(z > lim for z in zmin) >= 1, zmin


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("N", [100, 512, 1024])
@pytest.mark.parametrize("case", ["tight_obs", "sv_zero_obs"])
def test_forward_weights_bound_and_its_fallbacks(dtype, N, case):
    """The forward pass shifts the weights of a step by sup_x G_t(x) (+ the transition's log-normaliser) instead of their block maximum (sweep
    contract).  tight_obs: observation noise far below the particles' spread, so every weight underflows under the bound and the NEXT step must
    redo it with the exact maximum; sv_zero_obs: a stochastic-volatility potential with y_t = 0 has no upper bound (that step keeps the block
    maximum).  Bit-exact against the oracle either way."""
    from aux_ssm_samplers_amd.csmc import _device
    d, T = 1, 20
    rng = np.random.default_rng(17 + N)
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    if case == "tight_obs":
        pot, sig = O.POT_GAUSS_OBS, 1e-6
    else:
        pot, sig = O.POT_SV, 0.7
        y[[3, 9]] = 0.0
    G0, Gt = _pot(pot, y, sig)
    x0 = rng.standard_normal((T, d)).astype(dtype)
    noise = dict(eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T))
    fk = _device.describe_bootstrap(M0, G0, Mt, Gt, Mt)
    noise32 = {k: np.asarray(v, dtype) for k, v in noise.items()}
    x, anc, hist = _device.sweep(fk, x0, N, True, noise={k: v[None] for k, v in noise32.items()}, want_history=True)
    ref = O.sweep(_odesc(O.BOOTSTRAP_LG, pot, M0, Mt, sig), x0, N, True, y=y, eps_prop=noise["eps_prop"], u_res=noise["u_res"], u_bwd=noise["u_bwd"],
                  dtype=dtype)
    npt.assert_array_equal(hist["xs"], ref["xs"])
    npt.assert_array_equal(hist["log_ws"], ref["log_ws"])
    npt.assert_array_equal(hist["As"], ref["As"])
    npt.assert_array_equal(anc, ref["ancestors"])
    npt.assert_array_equal(x, ref["x"])
    if case == "tight_obs":  # the bound c_obs is far above every log-weight: exp(lw - bound) = 0 for all particles on (nearly) every step
        lw = hist["log_ws"].astype(np.float64)
        c_obs = -np.log(sig) - 0.5 * np.log(2 * np.pi)
        under = (lw[1:-1] - c_obs).max(axis=1) < (-104 if dtype == np.float32 else -746)
        assert under.sum() >= 3, under


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("d,N,T", [(5, 25, 12), (8, 64, 20), (30, 25, 16), (32, 33, 9), (16, 2, 7)])
@pytest.mark.parametrize("proposal", [O.BOOTSTRAP_LG, O.AUX_INDEPENDENT])
@pytest.mark.parametrize("potential", [O.POT_FLAT, O.POT_GAUSS_OBS, O.POT_SV])
@pytest.mark.parametrize("backward", [True, False])
def test_wide_state_sweep_bit_exact_vs_oracle(dtype, d, N, T, proposal, potential, backward):
    """4 < dx <= 32 with few particles (csrc/csmc_wide.hip: one wave per chain, components in LDS rows) -- the reference's own stochastic-volatility
    protocol is D = 30, N = 25 (examples/stochastic_volatility/experiment.sh:1-10): particles, log-weights, ancestors and the trajectory bit-exact
    against the same oracle as the register kernels."""
    from aux_ssm_samplers_amd.csmc import _device
    rng = np.random.default_rng(100 * d + N + T)
    M0, Mt = _models(d, rng)
    y = rng.standard_normal((T, d))
    G0, Gt = _pot(potential, y)
    x0 = rng.standard_normal((T, d)).astype(dtype)
    noise = dict(eps_prop=rng.standard_normal((T, N, d)), u_res=rng.random((T - 1, N)), u_bwd=rng.random(T))
    delta, okw = None, {}
    if proposal == O.AUX_INDEPENDENT:
        delta = 0.5 + rng.random(T)
        noise["eps_aux"] = rng.standard_normal((T, d))
        fk = _device.describe_independent(M0, G0, Mt, Gt, Mt)
        okw = dict(sqrt_half_delta=np.sqrt(0.5 * delta), eps_aux=noise["eps_aux"])
    else:
        fk = _device.describe_bootstrap(M0, G0, Mt, Gt, Mt)
    nz = {k: np.asarray(v, dtype) for k, v in noise.items()}
    x, anc, hist = _device.sweep(fk, x0, N, backward, noise={k: v[None] for k, v in nz.items()}, delta=delta, want_history=True)
    ref = O.sweep(_odesc(proposal, potential, M0, Mt, 0.7), x0, N, backward, y=y if potential else None, eps_prop=noise["eps_prop"],
                  u_res=noise["u_res"], u_bwd=noise["u_bwd"], dtype=dtype, **okw)
    npt.assert_array_equal(hist["xs"], ref["xs"])
    npt.assert_array_equal(hist["log_ws"], ref["log_ws"])
    npt.assert_array_equal(hist["As"], ref["As"])
    npt.assert_array_equal(anc, ref["ancestors"])
    npt.assert_array_equal(x, ref["x"])
    assert np.all(hist["As"][:, 0] == 0) and np.all(hist["xs"][:, 0] == x0)


def test_wide_state_threefry_equals_explicit_and_chains_are_independent():
    """the SV protocol's shape (D = 30, N = 25, T = 250) for several chains: the keyed sweep equals the explicit sweep on key_noise(wide=True),
    a multi-chain launch equals single launches, refusals are loud"""
    from aux_ssm_samplers_amd import _lib, random as R
    from aux_ssm_samplers_amd.csmc import _device, GaussianInit, LinearGaussianDynamics, SVPotential
    from tests.helpers import sv_setup
    h = _lib.default_handle()
    T, d, N, C = 250, 30, 25, 3
    y, xtrue, (m0, P0, F, Q, b) = sv_setup(T, d)
    M0, Mt = GaussianInit(m0=m0, P0=P0), LinearGaussianDynamics(F=F, b=b, Q=Q)
    fk = _device.describe_independent(M0, SVPotential(y=y[0]), Mt, SVPotential(params=y[1:]), Mt)
    x0 = (xtrue[None] + 0.1 * np.random.default_rng(0).standard_normal((C, T, d))).astype(np.float32)
    key = R.PRNGKey(77)
    xa, anca, _ = _device.sweep(fk, x0, N, True, key=key, delta=0.05)
    nz = _device.key_noise(h, key, C, T, N, d, np.float32)
    xb, ancb, _ = _device.sweep(fk, x0, N, True, noise=nz, delta=0.05)
    npt.assert_array_equal(xa, xb)
    npt.assert_array_equal(anca, ancb)
    for c in range(C):
        xc, ancc, _ = _device.sweep(fk, x0[c], N, True, noise={k: v[c:c + 1] for k, v in nz.items()}, delta=0.05)
        npt.assert_array_equal(xc, xa[c])
    assert (anca != 0).mean() > 0.3
    with pytest.raises(ValueError):  # more than one wave of particles at dx > 4
        _device.sweep(fk, x0[0], 128, True, key=key, delta=0.05)

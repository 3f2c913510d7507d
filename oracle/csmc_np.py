"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product package).

LITERAL NumPy restatement of the reference's conditional-SMC path, function by function in the reference's own
arithmetic ORDER (normalised weights, plain left-to-right cumsum, `r = c[-1] (1 - u)`, `searchsorted`), evaluating
GENERIC Python `M0 / G0 / Mt / Gt / Pt` objects that follow the reference's protocol (`_primitives/csmc/base.py:18-71`).
It stands beside `oracle/csmc_ref.c`, the co-designed *contract* oracle of the HIP kernels (unnormalised weights, DPP-order
cumsum, two-level search, closed model family): `tests/test_oracle_csmc_literal.py` drives both with the same explicit
noise and demands identical ancestor indices in fp64, a bounded tie rate in fp32.

Reference map (all paths relative to /root/reference/aux_samplers/):
    _primitives/csmc/csmc.py        get_kernel :16-66, _csmc :69-107, _backward_scanning_pass :110-124,
                                    _backward_sampling_pass :127-149
    _primitives/csmc/resamplings.py multinomial :14-37
    _primitives/math/utils.py       normalize :23-39
    csmc/generic.py                 kernel :56-72, init :74-77
    csmc/independent.py             _get_classical_kernel :57-75, _log_pdf :121-134, AuxiliaryM0 :143-158,
                                    AuxiliaryG0 :163-169, GradientAuxiliaryG0 :173-190, AuxiliaryMtDynamics :192-198,
                                    AuxiliaryGt :238-248, GradientAuxiliaryGt :252-268
    _primitives/test_csmc/common.py the reference's test fixtures (GaussianDynamics, GaussianDistribution, ...)
Third-party semantics restated from their published algorithms [ext] (JAX is not installed here: "parity unpinned against JAX
bits", pinned by the reference's statistical known answers, tests/test_oracle_csmc_literal.py):
    jax.random.choice(key, M, p=w, shape)  -> p_cuml = cumsum(p); r = p_cuml[-1] * (1 - uniform(key, shape)); searchsorted(p_cuml, r)
    jax.scipy.special.logsumexp            -> amax = max(a); amax = where(isfinite(amax), amax, 0); log(sum(exp(a - amax))) + amax
    jax.scipy.stats.norm.logpdf            -> -(log(2 pi scale^2) + (x - loc)^2 / scale^2) / 2
    jax.lax.scan                           -> a Python loop;  jax.grad -> central differences (grad_fd below)

PRNG.  "Identical PRNG inputs" = explicit noise arrays.  Where the reference splits a key, this file indexes a
`Noise` record with the same tree shape:  csmc.py:53 split(key) -> (fwd, bwd);  :71 split(fwd, T): keys[0] -> M0.sample -> eps_prop[0],
keys[t] -> split -> (resampling: u_res[t-1], sampling: eps_prop[t]) (:85);  backward scanning: choice(key_bwd) -> u_bwd[T-1] (:111);
backward sampling: split(bwd, T), keys[0] -> B_T -> u_bwd[T-1], keys[1:] paired with REVERSED time (:129-146) -> step t uses u_bwd[t];
csmc/generic.py:64-67 normal(auxiliary_key, x.shape) -> eps_aux.  A `sample(key, ...)` method therefore receives the standard-normal
array the reference would have drawn from that key.
"""
import numpy as np

LOG_2PI = float(np.log(2.0 * np.pi))


# ---- third-party semantics [ext] --------------------------------------------------------------------------------------
def logsumexp(a):
    amax = np.max(a)
    if not np.isfinite(amax):
        amax = a.dtype.type(0)
    with np.errstate(divide="ignore"):
        return np.log(np.sum(np.exp(a - amax))) + amax


def norm_logpdf(x, loc, scale):
    x, loc, scale = np.asarray(x), np.asarray(loc), np.asarray(scale)
    scale_sqrd = scale * scale
    log_normalizer = np.log(x.dtype.type(2.0 * np.pi) * scale_sqrd)
    quadratic = (x - loc) * (x - loc) / scale_sqrd
    return -(log_normalizer + quadratic) / x.dtype.type(2)


def choice(u, p):
    """jax.random.choice(key, len(p), p=p, shape=u.shape) given the uniforms `u` the key would have produced"""
    p_cuml = np.cumsum(p)                               # sequential left-to-right in p's dtype
    r = p_cuml[-1] * (p.dtype.type(1) - np.asarray(u, p.dtype))
    return np.searchsorted(p_cuml, r, side="left").astype(np.int64)


def grad_fd(fn, u, h=1e-5):
    """central differences of the scalar fn at u (stands in for jax.grad, independent.py:60,82)"""
    u = np.array(u, np.float64)
    g = np.zeros_like(u)
    it = np.nditer(u, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        up, um = u.copy(), u.copy()
        up[i] += h
        um[i] -= h
        g[i] = (fn(up) - fn(um)) / (2 * h)
    return g


# ---- math/utils.py:23-39 ------------------------------------------------------------------------------------------------
def normalize(log_weights):
    log_weights = log_weights - logsumexp(log_weights)
    return np.exp(log_weights)


# ---- resamplings.py:14-37 -----------------------------------------------------------------------------------------------
def multinomial(key_u, weights, N=None):
    M = weights.shape[0]
    N = M if N is None else N
    indices = choice(np.asarray(key_u)[:N], weights)
    indices[0] = 0
    return indices


# ---- _primitives/csmc/base.py:18-71: the model protocol -------------------------------------------------------------------
class Distribution:
    def sample(self, key, N):
        raise NotImplementedError

    def logpdf(self, x):
        raise NotImplementedError


class Dynamics:
    params = None  # pytree with leading axis T - 1, scanned (csmc.py:103)

    def sample(self, key, x_t, params):
        raise NotImplementedError

    def logpdf(self, x_t_p_1, x_t, params):
        raise NotImplementedError


class Noise:
    """the explicit draws of one sweep (see the module docstring for the key tree they replace)"""

    def __init__(self, eps_prop, u_res, u_bwd, eps_aux=None):
        self.eps_prop, self.u_res, self.u_bwd, self.eps_aux = eps_prop, u_res, u_bwd, eps_aux


def _tree_index(params, t):
    if params is None:
        return None
    if isinstance(params, (tuple, list)):
        return tuple(_tree_index(p, t) for p in params)
    return params[t]


# ---- _primitives/csmc/csmc.py -----------------------------------------------------------------------------------------
def get_kernel(M0, G0, Mt, Gt, N, backward=False, Pt=None):
    """csmc.py:16-66; kernel(noise, x) -> (x, ancestors, history); updated = ancestors != 0 (:59)"""
    if backward and Pt is None:
        Pt = Mt
    elif backward and not hasattr(Pt, "logpdf"):
        raise ValueError("When `backward` is True, `Pt` must implement a valid logpdf method.")

    def kernel(key, x_star):
        w_T, xs, log_ws, As = _csmc(key, x_star, M0, G0, Mt, Gt, N, multinomial)
        if not backward:
            x, ancestors = _backward_scanning_pass(key, w_T, xs, As)
        else:
            x, ancestors = _backward_sampling_pass(key, Pt, w_T, xs, log_ws)
        return x, ancestors, dict(xs=xs, log_ws=log_ws, As=As, w_T=w_T)

    def init(x_star):
        return x_star, np.ones(x_star.shape[0], bool)  # csmc.py:61-64 (ancestors == 0 -> all True)

    return init, kernel


def _csmc(key, x_star, M0, G0, Mt, Gt, N, resampling):
    T = x_star.shape[0]
    x0 = np.array(M0.sample(key.eps_prop[0], N))        # :74
    x0[0] = x_star[0]                                   # :76
    log_w0 = G0(x0)                                     # :79
    w0 = normalize(log_w0)                              # :80
    xs, log_ws, As = [x0], [log_w0], []
    w_t_m_1, x_t_m_1 = w0, x0
    for t in range(1, T):                               # lax.scan :103
        Mt_params, Gt_params = _tree_index(Mt.params, t - 1), _tree_index(Gt.params, t - 1)
        A_t = resampling(key.u_res[t - 1], w_t_m_1)     # :87
        x_t_m_1 = np.take(x_t_m_1, A_t, axis=0)         # :88
        x_t = np.array(Mt.sample(key.eps_prop[t], x_t_m_1, Mt_params))  # :91
        x_t[0] = x_star[t]                              # :92
        log_w_t = Gt(x_t, x_t_m_1, Gt_params)           # :95
        w_t = normalize(log_w_t)                        # :96
        w_t_m_1, x_t_m_1 = w_t, x_t
        xs.append(x_t), log_ws.append(log_w_t), As.append(A_t)
    As = np.array(As, np.int64).reshape(T - 1, N)
    return w_t_m_1, np.array(xs), np.array(log_ws), As


def _backward_scanning_pass(key, w_T, xs, As):
    T = xs.shape[0]
    B_T = int(choice(key.u_bwd[T - 1], w_T))            # :111
    x_out, Bs = [xs[-1, B_T]], [B_T]
    B_t = B_T
    for t in range(T - 2, -1, -1):                      # scan over (xs[-2::-1], As[::-1]) :121
        B_t = int(As[t][B_t])                           # :116
        x_out.append(xs[t][B_t]), Bs.append(B_t)
    return np.array(x_out[::-1]), np.array(Bs[::-1], np.int64)


def _backward_sampling_pass(key, Mt, w_T, xs, log_ws):
    T = xs.shape[0]
    B_T = int(choice(key.u_bwd[T - 1], w_T))            # keys[0] :131
    x_t = xs[-1, B_T]
    x_out, Bs = [x_t], [B_T]
    for t in range(T - 2, -1, -1):                      # :146, params reversed :141
        log_w = Mt.logpdf(x_t, xs[t], _tree_index(Mt.params, t)) + log_ws[t]  # :136
        w = normalize(log_w)                            # :137
        B = int(choice(key.u_bwd[t], w))                # :138
        x_t = xs[t][B]
        x_out.append(x_t), Bs.append(B)
    return np.array(x_out[::-1]), np.array(Bs[::-1], np.int64)


# ---- csmc/generic.py:56-72 ----------------------------------------------------------------------------------------------
def get_generic_kernel(factory, N, backward=False, Pt=None):
    if backward and Pt is None:
        raise ValueError("If backward is True, the true dynamics `Pt` must be provided.")
    elif backward and not hasattr(Pt, "logpdf"):
        raise ValueError("`Pt` must implement a valid logpdf method.")

    def kernel(key, x, delta):
        T = x.shape[0]
        sqrt_half_delta = np.sqrt(x.dtype.type(0.5) * np.asarray(delta, x.dtype))
        if np.ndim(sqrt_half_delta) == 0:
            sqrt_half_delta = sqrt_half_delta * np.ones((T,), x.dtype)
        u = x + sqrt_half_delta[:, None] * key.eps_aux      # :67
        m0, g0, mt, gt = factory(u, sqrt_half_delta)
        _, auxiliary_kernel = get_kernel(m0, g0, mt, gt, N, backward=backward, Pt=Pt)
        return auxiliary_kernel(key, x)

    def init(x):
        return x, np.zeros(x.shape[0], bool)                # generic.py:74-77 (ancestors != 0 -> all False)

    return init, kernel


# ---- csmc/independent.py (classical, sequential) --------------------------------------------------------------------------
def get_independent_kernel(M0, G0, Mt, Gt, N, backward=False, Pt=None, gradient=False, exact_gradient=False):
    """independent.py:57-75.  `exact_gradient` is NOT the reference: it applies GradientAuxiliaryGt's correction per particle
    (the `axis=-1` the reference's `jnp.sum` at :265-266 lacks) -- the contract's AUXSSM_GRAD_EXACT."""

    def factory(u, scale):
        if gradient:
            grad_pi = grad_fd(lambda v: float(_log_pdf(v.astype(u.dtype), M0, G0, Mt, Gt)), u).astype(u.dtype)
        else:
            grad_pi = 0.0 * u
        m0 = AuxiliaryM0(u[0], scale[0], grad_pi[0])
        mt = AuxiliaryMtDynamics((u[1:], scale[1:], grad_pi[1:]))
        if gradient:
            g0 = GradientAuxiliaryG0(M0, G0, u[0], scale[0], grad_pi[0])
            gt = GradientAuxiliaryGt(Mt, Gt, (u[1:], scale[1:], grad_pi[1:]), exact_gradient)
        else:
            g0 = AuxiliaryG0(M0, G0)
            gt = AuxiliaryGt(Mt, Gt)
        return m0, g0, mt, gt

    return get_generic_kernel(factory, N, backward, Pt)


def _log_pdf(u, M0, G0, Mt, Gt):
    """independent.py:121-134"""
    log_pdf = M0.logpdf(u[0]) + G0(u[0])
    for t in range(u.shape[0] - 1):
        out = Gt(u[t + 1], u[t], _tree_index(Gt.params, t))
        out = out + Mt.logpdf(u[t + 1], u[t], _tree_index(Mt.params, t))
        log_pdf = log_pdf + out
    return log_pdf


class AuxiliaryM0(Distribution):  # :143-158
    def __init__(self, u, sqrt_half_delta, grad):
        self.u, self.sqrt_half_delta, self.grad = u, sqrt_half_delta, grad

    def logpdf(self, x):
        half_delta = self.sqrt_half_delta ** 2
        mean = self.u + half_delta * self.grad
        return np.sum(norm_logpdf(x, mean, self.sqrt_half_delta), axis=-1)

    def sample(self, key, N):
        half_delta = self.sqrt_half_delta ** 2
        mean = self.u + half_delta * self.grad
        return mean[None, ...] + self.sqrt_half_delta * key


class AuxiliaryG0:  # :163-169
    def __init__(self, M0, G0):
        self.M0, self.G0 = M0, G0

    def __call__(self, x):
        return self.G0(x) + self.M0.logpdf(x)


class GradientAuxiliaryG0:  # :173-190
    def __init__(self, M0, G0, u, sqrt_half_delta, grad):
        self.M0, self.G0, self.u, self.sqrt_half_delta, self.grad = M0, G0, u, sqrt_half_delta, grad

    def __call__(self, x):
        half_delta = self.sqrt_half_delta ** 2
        mean = self.u + half_delta * self.grad
        out = self.G0(x) + self.M0.logpdf(x)
        out = out + np.sum(norm_logpdf(x, self.u, self.sqrt_half_delta), axis=-1)
        out = out - np.sum(norm_logpdf(x, mean, self.sqrt_half_delta), axis=-1)
        return out


class AuxiliaryMtDynamics(Dynamics):  # :192-198 (the proposal ignores the parent)
    def __init__(self, params):
        self.params = params

    def sample(self, key, x_t, params):
        u_t, sqrt_half_delta, grad_t = params
        half_delta = sqrt_half_delta ** 2
        mean = u_t[None, :] + half_delta * grad_t[None, :]
        return mean + sqrt_half_delta * key


class AuxiliaryGt:  # :238-248
    def __init__(self, Mt, Gt):
        self.Mt, self.Gt = Mt, Gt
        self.params = (Mt.params, Gt.params)

    def __call__(self, x_t_p_1, x_t, params):
        Mt_params, Gt_params = params
        return self.Mt.logpdf(x_t_p_1, x_t, Mt_params) + self.Gt(x_t_p_1, x_t, Gt_params)


class GradientAuxiliaryGt:  # :252-268
    def __init__(self, Mt, Gt, params, exact=False):
        self.Mt, self.Gt, self.exact = Mt, Gt, exact
        self.params = (params, Mt.params, Gt.params)

    def __call__(self, x_t_p_1, x_t, params):
        (u_t, sqrt_half_delta, grad_t), Mt_params, Gt_params = params
        half_delta = sqrt_half_delta ** 2
        mean = u_t + half_delta * grad_t
        out_1 = self.Mt.logpdf(x_t_p_1, x_t, Mt_params) + self.Gt(x_t_p_1, x_t, Gt_params)
        axis = -1 if self.exact else None           # the reference sums over ALL particles (:265-266): a constant of the step
        out_2 = np.sum(norm_logpdf(x_t_p_1, u_t, sqrt_half_delta), axis=axis)
        out_2 = out_2 - np.sum(norm_logpdf(x_t_p_1, mean, sqrt_half_delta), axis=axis)
        return out_1 + out_2


# ---- the reference's test fixtures (_primitives/test_csmc/common.py) -------------------------------------------------------
class GaussianDynamics(Dynamics):  # common.py:11-31 (also a Potential)
    def __init__(self, rho=0.9):
        self.rho, self.sig = rho, (1 - rho ** 2) ** 0.5

    def logpdf(self, x_t_p_1, x_t, _params):
        return np.sum(norm_logpdf(x_t_p_1, self.rho * x_t, x_t.dtype.type(self.sig)), axis=-1)

    def sample(self, key, x_t, params):
        return self.rho * x_t + self.sig * key

    def __call__(self, x_t_p_1, x_t, params):
        return self.logpdf(x_t_p_1, x_t, params)


class GaussianDistribution(Distribution):  # common.py:34-49
    def __init__(self, mu=0.0, sig=1.0):
        self.mu, self.sig = mu, sig

    def sample(self, key, N):
        return self.mu + self.sig * key

    def logpdf(self, x):
        return np.sum(norm_logpdf(x, x.dtype.type(self.mu), x.dtype.type(self.sig)), axis=-1)

    def __call__(self, x):
        return self.logpdf(x)


class GaussianObservationPotential:  # common.py:52-58
    def __init__(self, params, sig=1.0):
        self.params, self.sig = params, sig

    def __call__(self, x_t_p_1, _x_t, params):
        return norm_logpdf(np.asarray(params, x_t_p_1.dtype), x_t_p_1, x_t_p_1.dtype.type(self.sig)).ravel()


class FlatUnivariatePotential:  # common.py:61-68
    def __call__(self, x):
        return np.zeros(x.shape[:1], dtype=x.dtype)


class FlatPotential:  # common.py:71-75
    params = None

    def __call__(self, x_t_p_1, _x_t, _params):
        return np.zeros(x_t_p_1.shape[:1], dtype=x_t_p_1.dtype)


# ---- the closed Feynman-Kac family of include/auxssm.h, written as GENERIC protocol objects -------------------------------
def _mvn_chol_logpdf(x, mean, L):
    """log N(x; mean, L L^T) for x (..., d) (math/mvn/base.py:15-58: solve_triangular, -sum log diag - d/2 log 2 pi)"""
    from scipy.linalg import solve_triangular
    d = L.shape[-1]
    r = np.atleast_2d(x - mean)
    z = solve_triangular(L, r.T, lower=True).T
    out = -0.5 * np.sum(z * z, axis=-1) - np.sum(np.log(np.diag(L))) - 0.5 * d * LOG_2PI
    return (out if np.ndim(x - mean) > 1 else out[0]).astype(np.result_type(x, mean))


class GaussianInit(Distribution):
    def __init__(self, m0, chol_P0):
        self.m0, self.L = np.asarray(m0), np.asarray(chol_P0)

    def sample(self, key, N):
        return self.m0[None, :] + key @ self.L.T

    def logpdf(self, x):
        return _mvn_chol_logpdf(x, self.m0, self.L)


class LinearGaussianDynamics(Dynamics):
    """X' ~ N(F X + b, L L^T); params = (F_t, b_t, L_t) with leading axis T - 1 (time-invariant models pass broadcast views)"""

    def __init__(self, F, b, chol_Q, T):
        F, b, L = np.asarray(F), np.asarray(b), np.asarray(chol_Q)
        bt = lambda a, nd: a if a.ndim == nd + 1 else np.broadcast_to(a, (T - 1,) + a.shape)
        self.params = (bt(F, 2), bt(b, 1), bt(L, 2))

    def mean(self, x, params):
        F, b, _ = params
        return x @ F.T + b

    def sample(self, key, x_t, params):
        return self.mean(x_t, params) + key @ params[2].T

    def logpdf(self, x_t_p_1, x_t, params):
        return _mvn_chol_logpdf(x_t_p_1, self.mean(x_t, params), params[2])


class Lorenz63EM(LinearGaussianDynamics):
    """Euler-Maruyama step of Lorenz-63 (examples/lorenz/model.py:10-25): mean x + dt f(x; theta), covariance L L^T"""

    def __init__(self, theta, dt, chol_Q, T):
        self.theta, self.dt = np.asarray(theta), dt
        L = np.asarray(chol_Q)
        self.params = (np.zeros((T - 1, 0)), np.zeros((T - 1, 0)), np.broadcast_to(L, (T - 1,) + L.shape))

    def mean(self, x, params):
        th, x1, x2, x3 = self.theta, x[..., 0], x[..., 1], x[..., 2]
        f = np.stack([th[0] * (x2 - x1), th[1] * x1 - x2 - x1 * x3, x1 * x2 - th[2] * x3], axis=-1)
        return x + x.dtype.type(self.dt) * f


class ObsPotential:
    """g_t(x_t) as a `Potential` with params = y[1:] and as the `UnivariatePotential` of y[0] (kinds: 'gauss', 'masked', 'sv')"""

    def __init__(self, kind, y, sig=1.0, first=False):
        self.kind, self.sig, self.first = kind, sig, first
        self.params = None if first else np.asarray(y)
        self.y0 = np.asarray(y) if first else None

    def _g(self, x, y):
        x2 = np.atleast_2d(x)
        if self.kind == "gauss":
            out = np.sum(norm_logpdf(np.asarray(y, x.dtype), x2, x.dtype.type(self.sig)), axis=-1)
        elif self.kind == "masked":  # finite components only (examples/lorenz/model.py:43-56: nansum over the observed entries)
            out = np.nansum(norm_logpdf(np.asarray(y, x.dtype), x2, x.dtype.type(self.sig)), axis=-1)
        else:  # stochastic volatility: y_k ~ N(0, exp(x_k)) (examples/stochastic_volatility/model.py:56-63), NaN terms dropped
            y = np.asarray(y, x.dtype)
            with np.errstate(over="ignore", invalid="ignore"):
                v = -0.5 * (y * y * np.exp(-x2) + x2) - x.dtype.type(0.5 * LOG_2PI)
            out = np.nansum(v, axis=-1)
        return out.astype(x.dtype) if x.ndim > 1 else out.astype(x.dtype)[0]

    def __call__(self, *a):
        if self.first:
            return self._g(a[0], self.y0)
        return self._g(a[0], a[2])

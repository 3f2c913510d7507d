"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU (NumPy) restatement of the auxiliary-Kalman hot path of AdrienCorenflos/aux-ssm-samplers.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / reported baseline.

Every function is written "batched by design": the trailing axes are the matrix/vector axes the
reference's ``jnp.vectorize`` signatures name, and any number of leading axes broadcast.  The
reference's ``jax.vmap`` over time therefore becomes "call the function on the whole (n, ...)
stack", and its sequential ``lax.scan`` becomes a Python loop calling the same functions on
un-stacked inputs.  One implementation serves both paths.

Pinning (see tests/test_oracle_kalman.py): the sequential and parallel filters are checked against
`explicit_filter` / `explicit_smoother` below -- independent textbook restatements of the reference's
own NumPy test oracles (aux_samplers/_primitives/test_kalman/common.py:5-79) -- on the reference's own
``np.random.seed`` inputs (test_filtering.py:20-107, test_sampling.py:23-127), incl. missing data.
JAX is not installed here, so nothing PRNG-bit-dependent is pinned: noise enters as explicit arrays.

Third-party semantics restated here (the arithmetic of the reference lives in JAX/XLA/LAPACK,
jax>=0.3.25 per reference setup.py:21-23, not in /root/reference):
  * ``jnp.linalg.cholesky``: LAPACK potrf; on failure (a pivot <= 0 or NaN) the *whole* factor is NaN.
    An ``inf`` pivot is not a failure (sqrt(inf) = inf).
  * ``jax.lax.associative_scan``: recursive odd/even formulation, see `associative_scan`.
"""
import math

import numpy as np

_LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------
# small batched linear algebra with IEEE inf/NaN behaviour spelled out
# ----------------------------------------------------------------------------------------------

def _mm(a, b):
    return np.matmul(a, b)


def _mv(a, x):
    return np.einsum("...ij,...j->...i", a, x)


def _T(a):
    return np.swapaxes(a, -1, -2)


def _sym(a):
    return 0.5 * (a + _T(a))


def cholesky(a):
    """Lower Cholesky factor, batched; LAPACK/JAX semantics: any pivot <= 0 or NaN -> all-NaN factor."""
    a = np.asarray(a)
    n = a.shape[-1]
    L = np.zeros_like(a)
    fail = np.zeros(a.shape[:-2], dtype=bool)
    with np.errstate(all="ignore"):
        for j in range(n):
            s = a[..., j, j] - np.sum(L[..., j, :j] ** 2, axis=-1)
            fail |= ~(s > 0)  # catches <= 0 and NaN, lets +inf through
            ljj = np.sqrt(s)
            L[..., j, j] = ljj
            for i in range(j + 1, n):
                t = a[..., i, j] - np.sum(L[..., i, :j] * L[..., j, :j], axis=-1)
                L[..., i, j] = t / ljj
    L = np.where(fail[..., None, None], np.nan, L)
    return L


def solve_lower(L, b):
    """Solve L z = b (L lower triangular), b (..., n) or (..., n, k)."""
    vec = b.ndim == L.ndim - 1
    if vec:
        b = b[..., None]
    n = L.shape[-1]
    z = np.zeros(np.broadcast_shapes(L.shape[:-2], b.shape[:-2]) + b.shape[-2:], dtype=np.result_type(L, b))
    with np.errstate(all="ignore"):
        for i in range(n):
            acc = b[..., i, :] - np.einsum("...j,...jk->...k", L[..., i, :i], z[..., :i, :])
            z[..., i, :] = acc / L[..., i, i][..., None]
    return z[..., 0] if vec else z


def solve_upper_from_lower_T(L, b):
    """Solve L^T z = b."""
    vec = b.ndim == L.ndim - 1
    if vec:
        b = b[..., None]
    n = L.shape[-1]
    z = np.zeros(np.broadcast_shapes(L.shape[:-2], b.shape[:-2]) + b.shape[-2:], dtype=np.result_type(L, b))
    with np.errstate(all="ignore"):
        for i in range(n - 1, -1, -1):
            acc = b[..., i, :] - np.einsum("...j,...jk->...k", L[..., i + 1:, i], z[..., i + 1:, :])
            z[..., i, :] = acc / L[..., i, i][..., None]
    return z[..., 0] if vec else z


def cho_solve(L, b):
    return solve_upper_from_lower_T(L, solve_lower(L, b))


def solve_general(a, b):
    """jax.scipy.linalg.solve (LU with partial pivoting), batched."""
    return np.linalg.solve(a, b)


# ----------------------------------------------------------------------------------------------
# mvn.logpdf / tril_log_det   (aux_samplers/_primitives/math/mvn/base.py:15-58, 108-128)
# ----------------------------------------------------------------------------------------------

def tril_log_det(chol):
    # base.py:123-128 : non-finite diagonal entries are replaced by 1 and so ignored by the log
    d = np.diagonal(chol, axis1=-2, axis2=-1)
    d = np.where(np.isfinite(d), d, 1.0)
    with np.errstate(all="ignore"):
        return np.nansum(np.log(np.abs(d)), axis=-1)


def mvn_logpdf(x, m, chol):
    # base.py:49-58 ; _INF = 1e500 is +inf so the final clip is the identity
    cd = np.diagonal(chol, axis1=-2, axis2=-1)
    dim = np.sum(np.isfinite(cd), axis=-1)
    chol_clip = np.where(np.isfinite(chol), chol, np.inf)
    with np.errstate(all="ignore"):
        y = solve_lower(chol_clip, x - m)
        const = tril_log_det(chol) + 0.5 * dim * _LOG_2PI
        return -0.5 * np.sum(y * y, axis=-1) - const


def norm_logpdf(x, loc, scale):
    with np.errstate(all="ignore"):
        z = (x - loc) / scale
        return -0.5 * z * z - np.log(scale) - 0.5 * _LOG_2PI


# ----------------------------------------------------------------------------------------------
# filtering.py
# ----------------------------------------------------------------------------------------------

def _mask_obs(y, H, R, c):
    """filtering.py:89-100 / :204-213 -- zero missing rows of H, c, rows/cols of R, +inf on R's diagonal."""
    dy = y.shape[-1]
    nan = ~np.isfinite(y)
    diag_R = np.where(nan, np.inf, np.diagonal(R, axis1=-2, axis2=-1))
    R_ = np.where(nan[..., None, :], 0.0, R)
    R_ = np.where(nan[..., :, None], 0.0, R_)
    R_ = R_.copy()
    idx = np.arange(dy)
    R_[..., idx, idx] = diag_R
    H_ = np.where(nan[..., :, None], 0.0, H)
    c_ = np.where(nan, 0.0, c)
    return nan, H_, R_, c_


def sequential_update(y, m, P, H, c, R):
    """filtering.py:83-130.  Returns (m, P, ell_inc)."""
    y, m, P, H, c, R = map(np.asarray, (y, m, P, H, c, R))
    dy = y.shape[-1]
    with np.errstate(all="ignore"):
        nan, H_, R_, c_ = _mask_obs(y, H, R, c)
        y_hat = _mv(H_, m) + c_
        y_ = np.where(np.isnan(y), y_hat, y)  # nan_to_num(y, nan=y_hat): only NaN is replaced
        y_diff = y_ - y_hat
        S = R_ + _mm(_mm(H_, P), _T(H_))
        if dy == 1:  # :108-111
            chol_S = S ** 0.5
            ell_inc = norm_logpdf(y_[..., 0], y_hat[..., 0], chol_S[..., 0, 0])
            G = _mm(P, _T(H_)) / S
        else:  # :112-117
            chol_S = cholesky(S)
            ell_inc = mvn_logpdf(y_, y_hat, chol_S)
            fi = np.finfo(chol_S.dtype)
            chol_S = np.where(np.isnan(chol_S), fi.max, chol_S)
            chol_S = np.where(np.isposinf(chol_S), fi.max, chol_S)
            chol_S = np.where(np.isneginf(chol_S), fi.min, chol_S)
            G = _T(cho_solve(chol_S, _mm(H_, P)))
        m_new = m + _mv(G, y_diff)
        S0 = np.where(np.isfinite(S), S, 0.0)
        P_new = P - _mm(_mm(G, S0), _T(G))
        P_new = _sym(P_new)
        ell_inc = np.where(np.isnan(ell_inc), 0.0, ell_inc)
    # :127-130  lax.cond(any(isfinite(y)), _update, _passthrough)
    any_obs = np.any(np.isfinite(y), axis=-1)
    m_out = np.where(any_obs[..., None], m_new, m)
    P_out = np.where(any_obs[..., None, None], P_new, P)
    ell_out = np.where(any_obs, ell_inc, 0.0)
    return m_out, P_out, ell_out


def sequential_predict(m, P, F, b, Q):
    """filtering.py:134-139"""
    m = _mv(F, m) + b
    P = Q + _mm(_mm(F, P), _T(F))
    return m, _sym(P)


def sequential_predict_update(m, P, F, b, Q, y, H, c, R):
    """filtering.py:143-147"""
    m, P = sequential_predict(m, P, F, b, Q)
    return sequential_update(y, m, P, H, c, R)


def filtering_op(e1, e2):
    """filtering.py:163-183 -- e1 = earlier prefix, e2 = later element."""
    A1, b1, C1, eta1, J1 = e1
    A2, b2, C2, eta2, J2 = e2
    dim = b1.shape[-1]
    I = np.eye(dim, dtype=A1.dtype)
    IpCJ = I + _mm(C1, J2)
    IpJC = I + _mm(J2, C1)
    if dim == 1:
        AIpCJ_inv = A2 / IpCJ
        AIpJC_inv = A1 / IpJC
    else:
        AIpCJ_inv = _T(solve_general(_T(IpCJ), _T(A2)))
        AIpJC_inv = _T(solve_general(_T(IpJC), A1))
    A = _mm(AIpCJ_inv, A1)
    b = _mv(AIpCJ_inv, b1 + _mv(C1, eta2)) + b2
    C = _mm(AIpCJ_inv, _mm(C1, _T(A2))) + C2
    eta = _mv(AIpJC_inv, eta2 - _mv(J2, b1)) + eta1
    J = _mm(AIpJC_inv, _mm(J2, A1)) + J1
    return A, b, _sym(C), eta, _sym(J)


def filtering_init_one(F, Q, b, H, R, c, y, m, P):
    """filtering.py:196-250 -- one scan element (A, b, C, eta, J) per transition; batched over leading axes."""
    dy = y.shape[-1]
    with np.errstate(all="ignore"):
        m_ = _mv(F, m) + b
        P_ = _mm(_mm(F, P), _T(F)) + Q
        nan, H_, R_, c_ = _mask_obs(y, H, R, c)
        S = _mm(_mm(H_, P_), _T(H_)) + R_
        if dy == 1:
            S_invH_T = _T(H_) / S[..., 0:1, 0:1]
        else:
            chol = cholesky(S)
            chol = np.where(np.isfinite(chol), chol, np.finfo(chol.dtype).max)
            S_invH_T = _T(cho_solve(chol, H_))
        K = _mm(P_, S_invH_T)
        A = F - _mm(_mm(K, H_), F)
        y_diff_b = np.where(nan, 0.0, y - _mv(H_, b) - c_)
        y_diff_m = np.where(nan, 0.0, y - _mv(H_, m_) - c_)
        b_std = m_ + _mv(K, y_diff_m)
        S0 = np.where(np.isfinite(S), S, 0.0)
        C = P_ - _mm(_mm(K, S0), _T(K))
        temp = _mm(_T(F), S_invH_T)
        eta = _mv(temp, y_diff_b)
        J = _mm(_mm(temp, H_), F)
        upd = (A, b_std, _sym(C), eta, _sym(J))
        # passthrough :239-248
        pas = (np.broadcast_to(F, A.shape), m_, _sym(P_), np.zeros_like(b_std), np.zeros_like(A))
    any_obs = np.any(np.isfinite(y), axis=-1)
    out = []
    for u, p in zip(upd, pas):
        sel = any_obs.reshape(any_obs.shape + (1,) * (u.ndim - any_obs.ndim))
        out.append(np.where(sel, u, p))
    return tuple(out)


def associative_scan(fn, elems, reverse=False):
    """jax.lax.associative_scan [third-party, restated]: inclusive scan along axis 0 by the recursive
    odd/even scheme; fn(a, b) gets a = lower-index combination.  reverse=True flips, scans, flips."""
    if reverse:
        elems = tuple(e[::-1] for e in elems)
    n = elems[0].shape[0]

    def _scan(el):
        k = el[0].shape[0]
        if k < 2:
            return el
        reduced = fn(tuple(e[0:-1:2] for e in el), tuple(e[1::2] for e in el))
        odd = _scan(reduced)
        if k % 2 == 0:
            even = fn(tuple(o[:-1] for o in odd), tuple(e[2::2] for e in el))
        else:
            even = fn(odd, tuple(e[2::2] for e in el))
        even = tuple(np.concatenate([e[0:1], ev], axis=0) for e, ev in zip(el, even))
        out = []
        for ev, od in zip(even, odd):
            r = np.empty((k,) + ev.shape[1:], dtype=ev.dtype)
            r[0::2] = ev
            r[1::2] = od
            out.append(r)
        return tuple(out)

    res = _scan(tuple(elems)) if n > 0 else tuple(elems)
    if reverse:
        res = tuple(r[::-1] for r in res)
    return res


def filtering(ys, lgssm, parallel):
    """filtering.py:18-79.  lgssm = (m0, P0, Fs, Qs, bs, Hs, Rs, cs).  Returns ms, Ps, ell."""
    m0, P0, Fs, Qs, bs, Hs, Rs, cs = [np.asarray(a) for a in lgssm]
    ys = np.asarray(ys)
    T = ys.shape[0]
    m0f, P0f, ell0 = sequential_update(ys[0], m0, P0, Hs[0], cs[0], Rs[0])
    if parallel:
        # _filtering_init :188-192 -- (m, P) = (m0+, P0+) for the first element, zeros for the others
        n = T - 1
        ms_in = np.concatenate([m0f[None], np.zeros((n - 1,) + m0f.shape, m0f.dtype)]) if n > 0 else m0f[None][:0]
        Ps_in = np.concatenate([P0f[None], np.zeros((n - 1,) + P0f.shape, P0f.dtype)]) if n > 0 else P0f[None][:0]
        elems = filtering_init_one(Fs, Qs, bs, Hs[1:], Rs[1:], cs[1:], ys[1:], ms_in, Ps_in)
        _, ms, Ps, _, _ = associative_scan(filtering_op, elems)
        ms = np.concatenate([m0f[None], ms])
        Ps = np.concatenate([P0f[None], Ps])
        *_, ell_inc = sequential_predict_update(ms[:-1], Ps[:-1], Fs, bs, Qs, ys[1:], Hs[1:], cs[1:], Rs[1:])
        ell = ell0 + np.nansum(ell_inc, axis=0)
    else:
        ms = np.empty((T,) + m0f.shape, m0f.dtype)
        Ps = np.empty((T,) + P0f.shape, P0f.dtype)
        ms[0], Ps[0] = m0f, P0f
        m, P, ell = m0f, P0f, ell0
        for t in range(1, T):
            m, P, inc = sequential_predict_update(m, P, Fs[t - 1], bs[t - 1], Qs[t - 1], ys[t], Hs[t], cs[t], Rs[t])
            ms[t], Ps[t] = m, P
            ell = ell + inc
    if np.ndim(ell) == 1:  # batched case :43-45
        ell = np.sum(ell)
    return ms, Ps, ell


# ----------------------------------------------------------------------------------------------
# sampling.py
# ----------------------------------------------------------------------------------------------

def mean_and_chol(F, Q, b, m, P):
    """sampling.py:60-105"""
    dim = m.shape[-1]
    with np.errstate(all="ignore"):
        S = _sym(_mm(_mm(F, P), _T(F)) + Q)
        if dim == 1:
            gain = P * F / S
        else:
            gain = _mm(P, _T(cho_solve(cholesky(S), F)))  # solve(S, F, assume_a="pos")
        inc_Sig = _sym(P - _mm(_mm(gain, S), _T(gain)))
        inc_m = m - _mv(gain, _mv(F, m) + b)
        L = np.sqrt(inc_Sig) if dim == 1 else cholesky(inc_Sig)
        L = np.nan_to_num(L)  # nan -> 0, +-inf -> +-finfo.max
    return inc_m, L, gain


def sampling_init(eps, ms, Ps, Fs, Qs, bs):
    """sampling.py:127-136 with the N(0, I) draws `eps` (shape ms.shape) given explicitly."""
    inc_m, L, gains = mean_and_chol(Fs, Qs, bs, ms[:-1], Ps[:-1])
    incs = inc_m + _mv(L, eps[:-1])
    with np.errstate(all="ignore"):
        if Ps.shape[-1] == 1:
            Ll = np.sqrt(Ps[-1])
        else:
            Ll = cholesky(Ps[-1])
        Ll = np.nan_to_num(Ll)
    last_inc = ms[-1] + _mv(Ll, eps[-1])
    gains = np.concatenate([gains, np.zeros_like(Ps[-1])[None]])
    incs = np.concatenate([incs, last_inc[None]])
    return gains, incs


def sampling_op(e1, e2):
    """sampling.py:51-55 -- e1 = accumulated later times, e2 = current."""
    G1, e1v = e1
    G2, e2v = e2
    return _mm(G2, G1), _mv(G2, e1v) + e2v


def sampling(eps, ms, Ps, lgssm, parallel):
    """sampling.py:11-40 with explicit noise."""
    Fs, Qs, bs = [np.asarray(a) for a in lgssm[2:5]]
    gains, incs = sampling_init(np.asarray(eps), np.asarray(ms), np.asarray(Ps), Fs, Qs, bs)
    if parallel:
        _, samples = associative_scan(sampling_op, (gains, incs), reverse=True)
        return samples
    T = ms.shape[0]
    out = np.empty_like(incs)
    out[-1] = incs[-1]
    x = incs[-1]
    for t in range(T - 2, -1, -1):
        x = _mv(gains[t], x) + incs[t]
        out[t] = x
    return out


# ----------------------------------------------------------------------------------------------
# base.py
# ----------------------------------------------------------------------------------------------

def log_likelihood(ys, xs, lgssm):
    """base.py:137-166 (incl. the per-step-NaN-is-dropped behaviour of nansum)."""
    Hs, Rs, cs = [np.asarray(a) for a in lgssm[5:8]]
    with np.errstate(all="ignore"):
        pred = _mv(Hs, xs) + cs
        if cs.shape[-1] == 1:
            out = norm_logpdf(ys[..., 0], pred[..., 0], np.sqrt(Rs)[..., 0, 0])
        else:
            out = mvn_logpdf(ys, pred, cholesky(Rs))
    return np.nansum(out)


def prior_logpdf(xs, lgssm):
    """base.py:99-134"""
    m0, P0, Fs, Qs, bs = [np.asarray(a) for a in lgssm[:5]]
    with np.errstate(all="ignore"):
        pred = _mv(Fs, xs[:-1]) + bs
        if m0.shape[-1] == 1:
            out = np.nansum(norm_logpdf(xs[0, ..., 0], m0[..., 0], np.sqrt(P0)[..., 0, 0]))
            tr = norm_logpdf(xs[1:, ..., 0], pred[..., 0], np.sqrt(Qs)[..., 0, 0])
        else:
            out = np.nansum(mvn_logpdf(xs[0], m0, cholesky(P0)))
            tr = mvn_logpdf(xs[1:], pred, cholesky(Qs))
    return out + np.nansum(tr)


def posterior_logpdf(ys, xs, ell, lgssm):
    """base.py:72-96"""
    return log_likelihood(ys, xs, lgssm) - ell + prior_logpdf(xs, lgssm)


# ----------------------------------------------------------------------------------------------
# kalman/generic.py -- one auxiliary-Kalman MH sweep with explicit noise
# ----------------------------------------------------------------------------------------------

def get_alpha(lp_prop, lp_rev, lt_prop, lt_rev, sqrt_delta, u, x, x_prop):
    """generic.py:98-106 ; returns (alpha, log_alpha)."""
    log_alpha = lt_prop - lt_rev
    log_alpha += lp_rev - lp_prop
    dp, dc = (x_prop - u) / sqrt_delta, (x - u) / sqrt_delta
    log_alpha -= np.sum(dp ** 2 - dc ** 2)
    # jnp.minimum(0, nan) = nan (generic.py:105): a NaN ratio gives alpha = nan and bernoulli(key, nan) rejects -- Python's min() would return 0.0
    return (float("nan") if math.isnan(log_alpha) else math.exp(min(0.0, log_alpha))), log_alpha


def kalman_sweep(x, delta, dynamics_factory, observations_factory, log_likelihood_fn, parallel,
                 eps_aux, eps_samp, u_accept):
    """generic.py:53-90.  eps_aux, eps_samp ~ N(0,I) of x.shape, u_accept ~ U[0,1).
    Returns dict(x, accepted, x_prop, log_alpha, u, and the four log terms)."""
    x = np.asarray(x)
    u = x + math.sqrt(0.5 * delta) * eps_aux

    def do_one(xlin, x_prop=None):
        m0, P0, Fs, Qs, bs, *_ = dynamics_factory(xlin)
        ys, Hs, Rs, cs, *_ = observations_factory(xlin, u, delta)
        lg = (m0, P0, Fs, Qs, bs, Hs, Rs, cs)
        ms, Ps, ell = filtering(ys, lg, parallel)
        if x_prop is None:
            x_prop = sampling(eps_samp, ms, Ps, lg, parallel)
        return posterior_logpdf(ys, x_prop, ell, lg), log_likelihood_fn(x_prop), x_prop

    lp_prop, lt_prop, x_prop = do_one(x)
    lp_rev, lt_rev, _ = do_one(x_prop, x)
    alpha, log_alpha = get_alpha(lp_prop, lp_rev, lt_prop, lt_rev, math.sqrt(delta), u, x, x_prop)
    accepted = bool(u_accept < alpha)  # jax.random.bernoulli(key, p) == uniform(key) < p
    return dict(x=x_prop if accepted else x, accepted=accepted, x_prop=x_prop, log_alpha=log_alpha, u=u,
                lp_prop=lp_prop, lp_rev=lp_rev, lt_prop=lt_prop, lt_rev=lt_rev)


# ----------------------------------------------------------------------------------------------
# independent textbook filter / RTS smoother: the pin for everything above
# (restates what aux_samplers/_primitives/test_kalman/common.py:5-79 checks the reference against)
# ----------------------------------------------------------------------------------------------

def explicit_filter(ys, m0, P0, Hs, Rs, cs, Fs, Qs, bs):
    """Covariance-form Kalman filter that *deletes* missing observation rows (common.py:27-79).
    X_0~N(m0,P0); X_t = F_{t-1} X_{t-1} + b_{t-1} + N(0,Q_{t-1}); Y_t = H_t X_t + c_t + N(0,R_t)."""
    from scipy.stats import multivariate_normal
    T = len(ys)
    dx = Hs.shape[2]
    ms = np.zeros((T, dx))
    Ps = np.zeros((T, dx, dx))
    ell = 0.0
    m, P = np.array(m0, float), np.array(P0, float)
    for t in range(T):
        if t > 0:
            m = Fs[t - 1] @ m + bs[t - 1]
            P = Fs[t - 1] @ P @ Fs[t - 1].T + Qs[t - 1]
        keep = np.isfinite(ys[t]) if t > 0 else np.ones(len(ys[t]), bool)  # common.py never masks t = 0
        if keep.any():
            H, R, c, y = Hs[t][keep], Rs[t][keep][:, keep], cs[t][keep], ys[t][keep]
            S = H @ P @ H.T + R
            r = y - (H @ m + c)
            ell += multivariate_normal.logpdf(r, np.zeros(len(r)), S)
            K = P @ H.T @ np.linalg.inv(S)
            m = m + K @ r
            P = P - K @ (S @ K.T if t == 0 else H @ P)
        ms[t], Ps[t] = m, P
    return ms, Ps, ell


def explicit_smoother(ms, Ps, Fs, Qs, bs):
    """RTS smoother (common.py:5-24)."""
    T = ms.shape[0]
    sm, sP = ms.copy(), Ps.copy()
    for t in range(T - 2, -1, -1):
        Pp = Fs[t] @ Ps[t] @ Fs[t].T + Qs[t]
        mp = Fs[t] @ ms[t] + bs[t]
        K = Ps[t] @ Fs[t].T @ np.linalg.inv(Pp)
        sm[t] = ms[t] + K @ (sm[t + 1] - mp)
        sP[t] = Ps[t] + K @ (sP[t + 1] - Pp) @ K.T
    return sm, sP


# ---- divide-and-conquer pathwise sampler (reference: _primitives/kalman/dnc_sampling.py) -------------------------------------------------
def _dnc_init(m, P, F, Q, b):
    """_init_elems :128-137"""
    E = np.linalg.solve(F @ P @ F.T + Q, F @ P).T
    return E, m - E @ (F @ m + b), P - E @ F @ P


def _dnc_combine(e1, e2):
    """_combination_operator_impl :104-118"""
    E1, g1, L1 = e1
    E2, g2, L2 = e2
    E, g, L = E1 @ E2, g1 + E1 @ g2, L1 + E1 @ L2 @ E1.T
    G = np.linalg.solve(L, E1 @ L2).T
    return (E, g, L), (G, E2 - G @ E, g2 - G @ g, L2 - G @ L @ G.T)


def dnc_sampling(eps, ms, Ps, lgssm):
    """sampling(key, ms, Ps, lgssm) of dnc_sampling.py:17-77 with explicit noise: time index t is drawn with eps[t] (the reference splits its key per tree level).
    Tree: make_dnc_tree :172-186 / _combine_elements :140-169 (pairs of neighbouring intervals, the odd last interval carried up unchanged)."""
    m0, P0, Fs, Qs, bs = lgssm[:5]
    ms, Ps, eps = np.asarray(ms, np.float64), np.asarray(Ps, np.float64), np.asarray(eps, np.float64)
    T = ms.shape[0]
    xs = np.zeros_like(ms)
    xs[-1] = ms[-1] + np.linalg.cholesky(Ps[-1]) @ eps[-1]
    if T == 1:
        return xs
    elems = [_dnc_init(ms[t], Ps[t], np.asarray(Fs[t], np.float64), np.asarray(Qs[t], np.float64), np.asarray(bs[t], np.float64)) for t in range(T - 1)]
    iv = [(t, t + 1) for t in range(T - 1)]
    tree = []
    while len(elems) > 1:
        ne, nxt_e, nxt_iv, level = len(elems), [], [], []
        for p in range(ne // 2):
            new, aux = _dnc_combine(elems[2 * p], elems[2 * p + 1])
            nxt_e.append(new)
            nxt_iv.append((iv[2 * p][0], iv[2 * p + 1][1]))
            level.append((iv[2 * p][0], iv[2 * p][1], iv[2 * p + 1][1], aux))
        if ne % 2:
            nxt_e.append(elems[-1])
            nxt_iv.append(iv[-1])
        elems, iv = nxt_e, nxt_iv
        tree.append(level)
    E, g, L = elems[0]
    xs[0] = E @ xs[-1] + g + np.linalg.cholesky(L) @ eps[0]
    for level in tree[::-1]:
        for left, mid, right, (G, Gm, w, V) in level:
            xs[mid] = G @ xs[left] + Gm @ xs[right] + w + np.linalg.cholesky(V) @ eps[mid]
    return xs

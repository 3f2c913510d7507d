"""CPU restatement (NumPy) of the two post-processing primitives of SURVEY 8(f) rank 4 -- TEST INFRASTRUCTURE ONLY (tests/).

Follows, as text, the reference's
  * `get_optimal_covariance`: aux_samplers/_primitives/math/mvn/base.py:78-105 (the dominating covariance of Section 3 of the paper)
  * `effective_sample_size`:  aux_samplers/examples/rare_event/ess.py:28-160 (BlackJAX's estimator with the option of dividing by the TRUE variance)

Parity pin: neither has a test in the reference ("parity unpinned").  Pinned here by their defining properties
(tests/test_mvn.py: the result dominates both covariances and equals P when P >= Sig; tests/test_host_helpers.py: iid and AR(1)
known answers, the hand-computed clamped-gather case of ess.py:156) and, for the device kernels, by these functions.
"""
import numpy as np
from scipy.fft import next_fast_len
from scipy.linalg import solve_triangular


def get_optimal_covariance(chol_P, chol_Sig):
    """mvn/base.py:78-105"""
    chol_P, chol_Sig = np.asarray(chol_P), np.asarray(chol_Sig)
    if (chol_P.ndim < 2 and chol_Sig.ndim < 2) or chol_P.shape[0] == 1:      # :94-95
        return np.maximum(chol_P, chol_Sig)
    right_Y = solve_triangular(chol_P, chol_Sig, lower=True)                 # :98
    w, v = np.linalg.eigh(right_Y.T @ right_Y)                               # :99
    w = np.minimum(w, 1.0)                                                   # :100
    left_Q = chol_Sig @ (v * (1.0 / np.sqrt(w))[None, :])                    # :101-103
    return np.linalg.cholesky(left_Q @ left_Q.T)                             # :104


def effective_sample_size(input_array, var=None, chain_axis=0, sample_axis=1):
    """ess.py:28-160: ESS = M N / tau, tau = -1 + 2 sum_t P_t over Geyer's initial positive, monotone sequence of paired
    autocorrelations P_t = rho_2t + rho_2t+1; autocovariances by FFT (:64-70), averaged over the M chains (:71); `var` replaces the
    empirical variance estimates (:84-88)."""
    a = np.moveaxis(np.asarray(input_array, np.float64), (chain_axis, sample_axis), (0, 1))
    M, N = a.shape[:2]
    rest = a.shape[2:]
    a = a.reshape(M, N, -1)
    chain_mean = a.mean(axis=1, keepdims=True)
    c = a - chain_mean
    m = next_fast_len(2 * N)
    f = np.fft.rfft(c, n=m, axis=1)
    acov = np.fft.irfft(f * np.conj(f), n=m, axis=1)[:, :N] / N        # biased autocovariance per chain
    acov = acov.mean(axis=0)                                           # (N, K)
    var0 = acov[0] * N / (N - 1.0)
    wvar = var0 * (N - 1.0) / N
    if M > 1:
        wvar = wvar + chain_mean[:, 0].var(axis=0, ddof=1)
    if var is not None:
        wvar = np.broadcast_to(np.asarray(var, np.float64).reshape(-1), wvar.shape).copy()
        var0 = wvar.copy()
    n_even = N - N % 2
    rho = np.concatenate([np.ones((1, a.shape[2])), 1.0 - (var0[None] - acov[1:n_even]) / wvar[None]], axis=0)
    even, odd = rho[0::2].copy(), rho[1::2].copy()
    ess = np.empty(a.shape[2])
    for k in range(a.shape[2]):
        e, o = even[:, k], odd[:, k]
        pos = (e + o) > 0.0
        L = len(pos) if pos.all() else int(np.argmin(pos))            # length of the initial positive run (:107-119)
        last = max(L - 1, 0)                                          # its last index (0 when the run is empty, as the reference's scan)
        o[L:] = 0.0
        keep = np.zeros(len(e), bool)
        keep[:L] = True
        if last + 1 < len(e):
            keep[last + 1] = e[last + 1] > 0                          # "improve estimation" (:125): the scatter drops an out-of-range index
        e[~keep] = 0.0
        s = e + o
        run = np.minimum.accumulate(s)                                # initial monotone sequence (:129-141)
        upd = s > np.concatenate([[s[0]], run[:-1]])
        e_f = np.where(upd, run / 2.0, e)
        o_f = np.where(upd, run / 2.0, o)
        extra = e_f[min(last + 1, len(e) - 1)]                        # (:156: the gather clamps an out-of-range index to the last even term)
        tau = -1.0 + 2.0 * np.sum(e_f + o_f) - extra
        tau = max(tau, 1.0 / np.log10(M * N))
        ess[k] = M * N / tau
    return ess.reshape(rest) if rest else float(ess[0])

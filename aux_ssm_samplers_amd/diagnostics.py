"""Post-processing of sampler output on the host (NumPy): the effective sample size the reference's rare-event experiment reports
(aux_samplers/examples/rare_event/ess.py:28-160, a BlackJAX derivative with the option of dividing by the TRUE variance) and the
result files its experiment scripts write (examples/stochastic_volatility/experiment.py:238-246, rare_event/experiment.py:323-328).
Nothing here is on the sampler path."""
import os

import numpy as np
from scipy.fft import next_fast_len


def effective_sample_size(input_array, var=None, chain_axis=0, sample_axis=1):
    """ESS = M N / tau, tau = -1 + 2 sum_t P_t over Geyer's initial positive, monotone sequence of paired autocorrelations
    P_t = rho_{2t} + rho_{2t+1}; autocovariances by FFT, averaged over the M chains; `var` replaces the empirical variance estimate
    (ess.py:28-160; same estimator as Stan's).  Returns the array with the chain and sample axes removed."""
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
        L = len(pos) if pos.all() else int(np.argmin(pos))            # length of the initial positive run
        last = max(L - 1, 0)                                          # its last index (0 when the run is empty, as the reference's scan)
        o[L:] = 0.0
        keep = np.zeros(len(e), bool)
        keep[:L] = True
        if last + 1 < len(e):
            keep[last + 1] = e[last + 1] > 0                          # "improve estimation": one more even term if it is positive
        e[~keep] = 0.0
        s = e + o
        run = np.minimum.accumulate(s)                                # initial monotone sequence
        upd = s > np.concatenate([[s[0]], run[:-1]])
        e_f = np.where(upd, run / 2.0, e)
        o_f = np.where(upd, run / 2.0, o)
        extra = e_f[min(last + 1, len(e) - 1)]                        # (ess.py:156: the gather clamps an out-of-range index to the last even term)
        tau = -1.0 + 2.0 * np.sum(e_f + o_f) - extra
        tau = max(tau, 1.0 / np.log10(M * N))
        ess[k] = M * N / tau
    return ess.reshape(rest) if rest else float(ess[0])


def save_experiment_npz(directory, style, D, T, N, parallel, gradient, *, ejsd_per_key, acceptance_rate_per_key, delta_per_key, time_per_key):
    """the file the stochastic-volatility experiment writes, same name pattern and keys (experiment.py:238-246), so that the reference's
    analysis scripts (results_analysis_cpu.py:11-16) read it unchanged.  Returns the path."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{style}-{D}-{T}-{N}-{parallel}-{gradient}.npz")
    np.savez(path, ejsd_per_key=np.asarray(ejsd_per_key), acceptance_rate_per_key=np.asarray(acceptance_rate_per_key),
             delta_per_key=np.asarray(delta_per_key), time_per_key=np.asarray(time_per_key))
    return path


def save_rare_event_csv(directory, style, T, N, parallel, gradient, results, true_values=None):
    """the rare-event experiment's result tables (rare_event/experiment.py:318-328): `results` / `true_values` map
    (rho, r2, timestep, statistic) -> value (any pandas-Series-like with that 4-level index, or a dict); written with the reference's
    file names and index names.  Returns the paths."""
    import pandas as pd
    os.makedirs(directory, exist_ok=True)
    names = ["rho", "r2", "timestep", "statistic"]

    def series(obj):
        s = obj if isinstance(obj, pd.Series) else pd.Series(obj)
        s.index = pd.MultiIndex.from_tuples(list(s.index), names=names) if not isinstance(s.index, pd.MultiIndex) else s.index.set_names(names)
        return s

    p1 = os.path.join(directory, f"{style}-{T}-{N}-{parallel}-{gradient}.csv")
    series(results).to_csv(p1)
    out = [p1]
    if true_values is not None:
        p2 = os.path.join(directory, f"{T}-true.csv")
        series(true_values).to_csv(p2)
        out.append(p2)
    return out

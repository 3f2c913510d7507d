"""Post-processing of sampler output: the effective sample size the reference's rare-event experiment reports
(aux_samplers/examples/rare_event/ess.py:28-160, a BlackJAX derivative with the option of dividing by the TRUE variance) -- computed on the
device (auxssm_ess), from a host array or from draws already resident in HBM -- and the result files its experiment scripts write
(examples/stochastic_volatility/experiment.py:238-246, rare_event/experiment.py:323-328; host I/O)."""
import os

import numpy as np


def effective_sample_size(input_array, var=None, chain_axis=0, sample_axis=1, handle=None):
    """ESS = M N / tau, tau = -1 + 2 sum_t P_t over Geyer's initial positive, monotone sequence of paired autocorrelations
    P_t = rho_{2t} + rho_{2t+1}; autocovariances averaged over the M chains; `var` replaces the empirical variance estimate
    (ess.py:28-160; same estimator as Stan's) -- on the device (auxssm_ess: means, autocovariances of every lag by direct sums in double, one
    lane per series for Geyer's sequences).  `input_array`: a NumPy array or a resident `DeviceArray` (then chain_axis = 0, sample_axis = 1).
    Returns the array with the chain and sample axes removed."""
    from . import _lib
    if isinstance(input_array, _lib.DeviceArray):
        if (chain_axis, sample_axis) != (0, 1):
            raise ValueError("a resident array must be laid out (chains, draws, ...)")
        handle, a, shape, dtype = input_array.handle, input_array, input_array.shape, input_array.dtype
    else:
        handle = handle or _lib.default_handle()
        h = np.asarray(input_array)
        dtype = np.dtype(np.float32) if h.dtype == np.float32 else np.dtype(np.float64)
        h = np.ascontiguousarray(np.moveaxis(h, (chain_axis, sample_axis), (0, 1)), dtype)
        shape, a = h.shape, handle.to_device(h)
    M, N = shape[:2]
    rest = shape[2:]
    K = int(np.prod(rest, dtype=np.int64)) if rest else 1
    vd = None
    if var is not None:
        vd = handle.to_device(np.ascontiguousarray(np.broadcast_to(np.asarray(var, dtype).reshape(-1), (K,))))
    out = handle.empty((K,), dtype)
    _lib.check(handle.lib.auxssm_ess(handle.h, _lib.dtype_code(dtype), M, N, K, a.ptr, vd.ptr if vd is not None else None, out.ptr))
    ess = out.to_host().astype(np.float64)
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

"""The MCMC loop around the sweeps, on the device (reference: `loop` of examples/stochastic_volatility/experiment.py:88-128 and
examples/lorenz/experiment.py:120-169; `gibbs_step` :106-115; `theta_posterior_mean_and_chol` examples/lorenz/model.py:59-79).

    loop(key, init_delta, init_state, kernel_fn, delta_fn, n_iter, ...) ->
        (n_iter, (sq_jump, mean, sq_mean), state, delta, window_avg_acceptance, avg_acceptance)

same argument order and return order as the reference.  The chains stay resident in HBM (`state.x` a kalman.DeviceChains or a
csmc.CsmcChains); per sweep the loop enqueues: the sweep, the running squared-jump / first / second moments (folded into the Kalman
sweep's accept step, a separate pass for cSMC), the acceptance averages and, while adapting, the step-size rule -- all HIP kernels on
the handle's stream (include/auxssm.h, "the MCMC loop around the sweeps").  A sampling run (delta_fn None) never synchronises with
the host.  While adapting with the reference's rule the Kalman step size is a device scalar as well (auxssm_kalman_sweep_dd): no read-back per sweep;
a user-supplied rule runs on the host and reads the C windowed acceptances back each burn-in sweep.

Chains of one call share delta: the adaptation rule sees the chain-mean of the windowed acceptance (one chain: the reference exactly).
"""
import functools

import numpy as np

from . import _lib, random as _random
from .common import delta_adaptation


def _is_reference_rule(delta_fn):
    """-> (min_delta, max_delta) if delta_fn is common.delta_adaptation (or a functools.partial of it fixing only the bounds)"""
    if delta_fn is delta_adaptation:
        return 1e-20, 1e20
    if isinstance(delta_fn, functools.partial) and delta_fn.func is delta_adaptation and not delta_fn.args and \
            set(delta_fn.keywords) <= {"min_delta", "max_delta"}:
        return delta_fn.keywords.get("min_delta", 1e-20), delta_fn.keywords.get("max_delta", 1e20)
    return None


class LorenzThetaStep:
    """theta | x of the stochastic Lorenz-63 Gibbs sampler (experiment.py:106-115), one theta per chain, drawn and written on the device
    into the parameter rows the LORENZ63_EXT sweep reads (`model` a kalman.LorenzModel built with theta of shape (C, 3))."""

    def __init__(self, model, sigma_theta):
        self.model, self.sigma_theta = model, float(sigma_theta)
        self._eps = None

    def __call__(self, key, chains):
        handle = chains.handle
        par = self.model.par_device(handle, chains.dtype, chains.C)
        if self._eps is None or self._eps.handle is not handle or self._eps.shape != (chains.C, 3) or self._eps.dtype != chains.dtype:
            self._eps = handle.empty((chains.C, 3), chains.dtype)
        if _random.compat() == "jax":   # examples/lorenz/experiment.py:115: jax.random.normal(key_theta, (3,)) -- one key per chain (`key` (C, 2) or split(key, C))
            kk = np.asarray(key, np.uint32)
            keys = kk if kk.ndim == 2 else (_random.as_key(key)[None] if chains.C == 1 else _random.jax_split(_random.as_key(key), chains.C))
            self._eps.copy_from_host(_random.jax_normal(keys, (3,), chains.dtype, handle))
        else:
            handle.rng_normal_into(key, 0, self._eps)
        handle.lorenz_theta_update(chains.x, self.sigma_theta, self.model.sigma_x, self._eps, par, layout=chains.layout)

    def theta(self, chains):
        return self.model.par_device(chains.handle, chains.dtype, chains.C).to_host()[:, :3]


def loop(key, init_delta, init_state, kernel_fn, delta_fn, n_iter, target_alpha=None, lr=None, beta=0.01, theta_step=None,
         callback=None):
    """See module docstring.  init_state.x: DeviceChains (kernel_fn(key, state, delta) from kalman.get_kernel) or CsmcChains (from
    csmc.get_kernel / get_independent_kernel).  theta_step: a LorenzThetaStep run after every sweep (Kalman chains).
    callback(i, state) is called after every sweep (e.g. to keep thinned samples; it may synchronise).

    Returns (n_iter, stats, state, delta, window_avg_acceptance, avg_acceptance): stats = (sq_jump, mean, sq_mean) as DeviceArrays
    in the chains' resident layout (chains.stats_to_host(a) -> (C, T, dx)); the acceptance averages are DeviceArrays (C,) [Kalman] or
    (C, T) [cSMC]; delta a float [Kalman] or the chains' device array (T,) [cSMC]."""
    from .kalman.generic import DeviceChains
    from .csmc._device import CsmcChains

    chains = init_state.x
    kalman = isinstance(chains, DeviceChains)
    if not kalman and not isinstance(chains, CsmcChains):
        raise ValueError("loop() runs on resident chains: init_state.x must be a kalman.DeviceChains or a csmc.CsmcChains")
    if delta_fn is not None and (target_alpha is None or lr is None):
        raise ValueError("target_alpha and lr are required with delta_fn")
    if theta_step is not None and not kalman:
        raise ValueError("theta_step needs Kalman chains")
    handle, dtype = chains.handle, chains.dtype
    split = _random.jax_split if _random.compat() == "jax" else _random.split   # (experiment.py:90: keys = jax.random.split(key, n_iter))
    keys = split(key, n_iter)
    stats = tuple(handle.zeros(chains.x.shape, dtype) for _ in range(3))  # fold 0 overwrites: (0 u + v) / 1 = v, as stats_fn(x, x) would
    flags = chains.accepted if kalman else chains.ancestors
    m = 1 if kalman else chains.T
    upd = init_state.updated if init_state.updated is not None else True
    if hasattr(upd, "to_host"):  # the state a previous loop returned: flags still on the device
        upd = upd.to_host().reshape(chains.C, m) != 0
    upd0 = np.broadcast_to(np.asarray(upd, dtype), (chains.C, m))
    avg = handle.to_device(upd0, dtype)
    window = handle.to_device(upd0, dtype)
    rule = _is_reference_rule(delta_fn) if delta_fn is not None else None
    if kalman:
        delta = float(init_delta)
        # the reference's rule while adapting: delta stays on the DEVICE (one scalar, updated by auxssm_delta_adapt from the chain-mean of the
        # windowed acceptance and read by auxssm_kalman_sweep_dd) -- no host round trip per sweep.  A user rule still runs on the host.
        delta_dev = handle.to_device(np.full(1, delta, dtype), dtype) if rule is not None else None
    else:
        if init_delta is not None:
            chains.set_delta(init_delta)
        delta = None
        delta_dev = None
        x_prev = handle.empty(chains.x.shape, dtype)
    state = init_state
    if kalman:
        handle.stats_attach(stats, 0, chains.x)
    try:
        for i in range(n_iter):
            if kalman:
                k_sweep, k_theta = (keys[i], None) if theta_step is None else split(keys[i], 2)
                state = kernel_fn(k_sweep, state, delta if delta_dev is None else delta_dev)  # folds the moments in its accept step
                if theta_step is not None:
                    theta_step(k_theta, chains)
            else:
                x_prev.copy_from(chains.x)
                state = kernel_fn(keys[i], state, None)
                handle.stats_update(i, x_prev, chains.x, stats)
            handle.accept_update(i, beta, flags, avg, window)
            if delta_fn is not None:
                lr_i = (n_iter - i) * lr / n_iter
                if kalman and delta_dev is not None:
                    handle.delta_adapt(window, target_alpha, lr_i, delta_dev, None, rule[0], rule[1])
                elif kalman:
                    delta = float(delta_fn(delta, target_alpha, float(np.mean(window.to_host())), lr_i))
                elif rule is not None:
                    handle.delta_adapt(window, target_alpha, lr_i, chains.delta, chains.sqrt_half_delta, rule[0], rule[1])
                else:
                    chains.set_delta(delta_fn(chains.delta.to_host(), target_alpha, np.mean(window.to_host(), axis=0), lr_i))
            if callback is not None:
                callback(i, state)
    finally:
        if kalman:
            handle.stats_attach(None, 0)
    if kalman and delta_dev is not None:
        delta = float(delta_dev.to_host()[0])  # one read at the END of the run (the return value is a float, as the reference's)
    return n_iter, stats, state, (delta if kalman else chains.delta), window, avg

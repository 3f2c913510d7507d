"""Auxiliary Kalman sampler (reference: aux_samplers/kalman/generic.py).

get_kernel(dynamics_factory, observations_factory, log_likelihood_fn, parallel) -> (init, kernel), the same
names, argument order and (init, kernel) return order as the reference (generic.py:19-46, :95).

Two execution paths, same semantics:
  * device sweep: the three callables are the bound methods of one built-in device model (models.py); the whole
    sweep -- auxiliary draw, proposal LGSSM, filter scan, pathwise sample scan, log-densities, MH accept -- is one
    auxssm_kalman_sweep call, for C chains at once, state resident in HBM.
  * host-factory path: arbitrary NumPy factories run on the host each sweep; filtering / sampling / log-densities
    run on the GPU through the primitives.
"""
import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Any

import numpy as np

from .. import _lib, random as _random
from .._primitives.base import SamplerState
from .._primitives.kalman import LGSSM, filtering, sampling, posterior_logpdf


@dataclass
class KalmanSampler(SamplerState):
    x: Any
    updated: Any


def get_kernel(dynamics_factory, observations_factory, log_likelihood_fn, parallel):
    """See module docstring.  Returns (init, kernel)."""
    model = _same_device_model(dynamics_factory, observations_factory, log_likelihood_fn)
    if model is not None:
        return _get_device_kernel(model, parallel)
    return _get_host_kernel(dynamics_factory, observations_factory, log_likelihood_fn, parallel)


def _same_device_model(*fns):
    owners = [getattr(f, "__self__", None) for f in fns]
    names = [getattr(f, "__name__", "") for f in fns]
    from .models import LGConcatModel, SVModel, LorenzModel
    if (isinstance(owners[0], (LGConcatModel, SVModel, LorenzModel)) and all(o is owners[0] for o in owners)
            and names == ["dynamics_factory", "observations_factory", "log_likelihood_fn"]):
        return owners[0]
    return None


# ------------------------------------------------------------------------------------------------
# host-factory path (generic.py:53-106 line by line, heavy lifting on the GPU)
# ------------------------------------------------------------------------------------------------
def _get_host_kernel(dynamics_factory, observations_factory, log_likelihood_fn, parallel):
    def kernel(key, state, delta, noise=None):
        """noise: optional dict(eps_aux, eps_samp, u_accept) of explicit draws (parity tests)."""
        x = np.asarray(state.x)
        handle = _lib.default_handle()
        if noise is None and _random.compat() == "jax":
            nz = _random.jax_kalman_noise(_random.as_key(key)[None], x.shape[0], x.shape[1], np.float32 if x.dtype == np.float32 else np.float64, handle)
            noise = dict(eps_aux=nz["eps_aux"][0], eps_samp=nz["eps_samp"][0], u_accept=float(nz["u_accept"][0]))
        k_aux, k_samp, k_acc = _random.split(key, 3) if noise is None else (None, None, None)
        if noise is None:
            eps_aux = handle.rng_normal(k_aux, 0, x.shape, x.dtype).to_host()
            eps_samp = handle.rng_normal(k_samp, 0, x.shape, x.dtype).to_host()
            u_acc = _random.uniform_scalar(k_acc)
        else:
            eps_aux, eps_samp, u_acc = noise["eps_aux"], noise["eps_samp"], noise["u_accept"]
        u = x + math.sqrt(0.5 * delta) * eps_aux

        def do_one(xlin, x_prop=None):
            m0, P0, Fs, Qs, bs, *_ = dynamics_factory(xlin)
            ys, Hs, Rs, cs, *_ = observations_factory(xlin, u, delta)
            lgssm = LGSSM(m0, P0, Fs, Qs, bs, Hs, Rs, cs)
            ms, Ps, ell = filtering(ys, lgssm, parallel)
            if x_prop is None:
                x_prop = sampling(None, ms, Ps, lgssm, parallel, eps=eps_samp)
            return posterior_logpdf(ys, x_prop, ell, lgssm), log_likelihood_fn(x_prop), x_prop

        lp_prop, lt_prop, x_prop = do_one(x)
        lp_rev, lt_rev, _ = do_one(x_prop, x)
        log_alpha = _log_alpha(lp_prop, lp_rev, lt_prop, lt_rev, math.sqrt(delta), u, x, x_prop)
        alpha = math.exp(min(0.0, log_alpha)) if not math.isnan(log_alpha) else float("nan")
        accept = bool(u_acc < alpha)
        out = KalmanSampler(x=x_prop if accept else x, updated=accept)
        out.log_alpha = log_alpha
        return out

    def init(x):
        return KalmanSampler(x=x, updated=True)

    return init, kernel


def _log_alpha(lp_prop, lp_rev, lt_prop, lt_rev, sqrt_delta, u, x, x_prop):
    # generic.py:98-106
    la = lt_prop - lt_rev
    la += lp_rev - lp_prop
    dp, dc = (x_prop - u) / sqrt_delta, (x - u) / sqrt_delta
    la -= float(np.sum(dp ** 2 - dc ** 2))
    return float(la)


# ------------------------------------------------------------------------------------------------
# device sweep
# ------------------------------------------------------------------------------------------------
class DeviceChains:
    """C chains' trajectories resident in HBM.  With >= 32 chains (or chain_minor=True) the state and the per-sweep noise
    are stored chain-minor, (T, dx, C): the sweep's lanes then run over chains (AUXSSM_LAYOUT_CHAIN_MINOR, include/auxssm.h)."""

    def __init__(self, handle, x, dtype=None, chain_minor=None, fused=None, model=None):
        """fused=False: keyed sweeps never take the fused three-pass path (auxssm_kalman_sweep_fused), e.g. to read the drawn noise back from
        eps_aux / eps_samp afterwards -- the fused sweep has no such buffers.
        model: the device model the chains will be swept with (a layout hint only): a chain-shared linear-Gaussian model takes the chain-minor layout -- and with
        it the fused sweep -- from 4 chains on, not 32 (measured at C2's sizes, profiles/r04_i_low_chain_layout.txt: 8 chains 9.4k -> 17.1k sweeps/s, 16 chains
        11.3k -> 36.7k; 2 chains are faster time-minor)."""
        x = np.asarray(x)
        if x.ndim == 2:
            x = x[None]
        self.handle = handle
        self.C, self.T, self.dx = x.shape
        env = os.environ.get("AUXSSM_CM")
        auto = self.C >= 32
        if (model is not None and getattr(model, "kmodel", None) == _lib.KMODEL_LG_CONCAT and self.C >= 4 and self.C % 2 == 0 and self.T >= 64
                and self.dx <= 4 and 1 <= getattr(model, "p_obs", 0) <= 4):
            auto = True   # (the conditions of csrc/api.hip::fused_refusal; anything else keeps the time-minor general path below 32 chains)
        self.chain_minor = bool(int(env)) if env is not None and chain_minor is None else (auto if chain_minor is None else bool(chain_minor))
        if self.dx > 4 and chain_minor is None:
            self.chain_minor = False  # dx > 4 runs the wide-state kernels (csrc/wide.hip): a workgroup per time step, dense layout
        self.layout = _lib.LAYOUT_CHAIN_MINOR if self.chain_minor else _lib.LAYOUT_DENSE
        self._x = handle.to_device(self._to_layout(x), dtype or x.dtype)
        self.dtype = self._x.dtype
        self.accepted = handle.zeros((self.C,), np.int32)
        self.logs = handle.zeros((self.C, 5), self.dtype)
        # per-sweep noise, allocated once: no hipMalloc / hipFree (and no stream sync) inside the sweep loop
        self._eps_aux = self._eps_samp = None
        self.u_acc = handle.empty((self.C,), self.dtype)
        # lazy state of the fused chain-shared sweep (auxssm_kalman_sweep_fused): chain c lives in x_alt where sel[c] != 0; allocated on first use
        self.x_alt = None
        self.sel = None
        if fused is None and os.environ.get("AUXSSM_FUSED") == "0":   # measurement switch: the keyed sweep of rounds 1-2
            fused = False
        self.fused = None if fused is None or fused else False  # None: not tried yet; False: refused (by the library or the caller): keyed sweeps

    # per-sweep noise buffers of the unfused sweeps (allocated once, on first use: the fused sweep draws inside its passes and needs none)
    @property
    def eps_aux(self):
        if self._eps_aux is None:
            self._eps_aux = self.handle.empty(self._x.shape, self.dtype)
        return self._eps_aux

    @property
    def eps_samp(self):
        if self._eps_samp is None:
            self._eps_samp = self.handle.empty(self._x.shape, self.dtype)
        return self._eps_samp

    @property
    def x(self):
        """the resident state as ONE DeviceArray.  After a fused sweep the chains with sel[c] != 0 live in x_alt: reading `x` gathers them first (resolve), so
        `chains.x.to_host()`, `chains.x.copy_from_host(...)` or handing `chains.x.ptr` to another auxssm_* call never sees stale rows.  The
        fused sweep itself works on the private pair (_x, x_alt) and does not come through here."""
        self.resolve()
        return self._x

    def resolve(self):
        """gather the lazy state into x (auxssm_kalman_state_resolve): before anything but a fused sweep reads x"""
        if self.sel is not None and self._lazy_dirty:
            dims = _lib.Dims(self.C, self.T, 1, self.dx, 0)
            _lib.check(self.handle.lib.auxssm_kalman_state_resolve(self.handle.h, _lib.dtype_code(self.dtype), C.byref(dims), self._x.ptr, self.x_alt.ptr, self.sel.ptr))
            self._lazy_dirty = False

    _lazy_dirty = False

    def _to_layout(self, a):
        """(C, T, dx) -> the resident layout"""
        a = np.asarray(a).reshape(self.C, self.T, self.dx)
        return np.ascontiguousarray(a.transpose(1, 2, 0)) if self.chain_minor else a

    def to_host(self):
        """trajectories as (C, T, dx)"""
        self.resolve()
        return self.stats_to_host(self.x)

    def stats_to_host(self, a):
        """any DeviceArray in the resident layout of x (e.g. the running moments of loop.loop) as (C, T, dx)"""
        a = a.to_host()
        return np.ascontiguousarray(a.transpose(2, 0, 1)) if self.chain_minor else a


def _get_device_kernel(model, parallel, nan_policy="reference"):
    pol = {"reference": _lib.NAN_REFERENCE, "masked": _lib.NAN_MASKED}[nan_policy]

    def sweep(handle, chains, delta, eps_aux, eps_samp, u_acc, keys=None):
        """One auxssm_kalman_sweep on resident buffers (all DeviceArray). Asynchronous.  keys (three Threefry keys): the keyed sweep -- the
        library draws the noise itself (into the three buffers), inside its first consumer where it can."""
        dl, ybuf, yarr = model.device(handle, chains.dtype)
        if "lorenz_par" in dl.bufs and dl.bufs["lorenz_par"].shape[0] not in (1, chains.C):
            raise ValueError(f"the model holds {dl.bufs['lorenz_par'].shape[0]} theta rows, the chains are {chains.C}")
        dims = _lib.Dims(chains.C, chains.T, 1, chains.dx, model.p_obs)
        if keys is not None:
            dev = isinstance(delta, _lib.DeviceArray)
            if dev and (delta.dtype != np.dtype(chains.dtype) or delta.size < 1):
                raise ValueError("a device-resident delta must be a DeviceArray of one scalar of the chains' dtype")
            k6 = (C.c_uint32 * 6)(*[int(v) for k in keys for v in np.asarray(k, np.uint32).reshape(2)])
            # chain-shared linear-Gaussian model on chain-minor resident chains: the sweep in two streaming passes on a LAZY state (no select
            # pass, no noise buffers).  The library refuses -- before enqueueing anything -- what it cannot run fused; keyed sweeps from then on.
            if chains.fused is not False and eps_aux is None and model.kmodel == _lib.KMODEL_LG_CONCAT and chains.chain_minor:
                if chains.x_alt is None:
                    chains.x_alt = handle.empty(chains._x.shape, chains.dtype)
                    chains.sel = handle.zeros((chains.C,), np.int32)
                rc = handle.lib.auxssm_kalman_sweep_fused(
                    handle.h, _lib.dtype_code(chains.dtype), model.kmodel, C.byref(dims), C.byref(dl.c), C.byref(yarr),
                    1.0 if dev else float(delta), delta.ptr if dev else None, k6, int(bool(parallel)), pol, chains.layout, chains._x.ptr,
                    chains.x_alt.ptr, chains.sel.ptr, u_acc.ptr, chains.accepted.ptr, chains.logs.ptr)
                if rc == _lib.ERR_UNSUPPORTED:  # (nothing was enqueued) e.g. AUXSSM_OPT_SHARE_MODEL switched off, odd chain count, per-chain model
                    chains.resolve()
                    if chains.fused is None:    # never ran fused: stop trying, drop the partner buffer
                        chains.fused = False
                        chains.x_alt = chains.sel = None
                else:
                    _lib.check(rc)
                    chains.fused = True
                    chains._lazy_dirty = True
                    return
            chains.resolve()
            if eps_aux is None:
                eps_aux, eps_samp = chains.eps_aux, chains.eps_samp
            _lib.check(handle.lib.auxssm_kalman_sweep_keyed(
                handle.h, _lib.dtype_code(chains.dtype), model.kmodel, C.byref(dims), C.byref(dl.c), C.byref(yarr),
                1.0 if dev else float(delta), delta.ptr if dev else None, k6, int(bool(parallel)), pol, chains.layout, chains.x.ptr,
                eps_aux.ptr, eps_samp.ptr, u_acc.ptr, chains.accepted.ptr, chains.logs.ptr))
            return
        chains.resolve()
        if isinstance(delta, _lib.DeviceArray):  # device-resident step size (one scalar of the chains' dtype): no host round trip
            if delta.dtype != np.dtype(chains.dtype) or delta.size < 1:
                raise ValueError("a device-resident delta must be a DeviceArray of one scalar of the chains' dtype")
            _lib.check(handle.lib.auxssm_kalman_sweep_dd(
                handle.h, _lib.dtype_code(chains.dtype), model.kmodel, C.byref(dims), C.byref(dl.c), C.byref(yarr),
                delta.ptr, int(bool(parallel)), pol, chains.layout, chains.x.ptr, eps_aux.ptr, eps_samp.ptr, u_acc.ptr,
                chains.accepted.ptr, chains.logs.ptr))
            return
        _lib.check(handle.lib.auxssm_kalman_sweep(
            handle.h, _lib.dtype_code(chains.dtype), model.kmodel, C.byref(dims), C.byref(dl.c), C.byref(yarr),
            float(delta), int(bool(parallel)), pol, chains.layout, chains.x.ptr, eps_aux.ptr, eps_samp.ptr, u_acc.ptr,
            chains.accepted.ptr, chains.logs.ptr))

    def draw(handle, key, chains):
        k_aux, k_samp, k_acc = _random.split(key, 3)
        handle.kalman_draw(k_aux, k_samp, k_acc, chains.eps_aux, chains.eps_samp, chains.u_acc)  # one launch, the values of the three fills
        return chains.eps_aux, chains.eps_samp, chains.u_acc

    def kernel(key, state, delta, noise=None):
        """state.x: ndarray (T, dx) [one chain], ndarray (C, T, dx) or DeviceChains (resident, updated in place)."""
        resident = isinstance(state.x, DeviceChains)
        handle = state.x.handle if resident else _lib.default_handle()  # resident chains carry their device
        if noise is None and _random.compat() == "jax":
            # the reference's own draws from this key (random.jax_kalman_noise: split(key, 3), two normals of x's shape, the acceptance uniform) as explicit arrays;
            # several chains: one key per chain -- `key` (C, 2), as jax.vmap(kernel) takes them, or split(key, C)
            Cn = state.x.C if resident else (1 if np.ndim(state.x) == 2 else np.shape(state.x)[0])
            Tn, dn = (state.x.T, state.x.dx) if resident else np.shape(state.x)[-2:]
            dt = state.x.dtype if resident else (np.float32 if np.asarray(state.x).dtype == np.float32 else np.float64)
            kk = np.asarray(key, np.uint32)
            keys_c = kk if kk.ndim == 2 else (_random.as_key(key)[None] if Cn == 1 else _random.jax_split(_random.as_key(key), Cn))
            if keys_c.shape[0] != Cn:
                raise ValueError(f"{keys_c.shape[0]} keys for {Cn} chains")
            jax_keys = _random.jax_split(keys_c, 3)     # (C, 3, 2): auxiliary_key, sampling_key, accept_key of every chain (kalman/generic.py:58)
        else:
            jax_keys = None
        chains = state.x if resident else DeviceChains(handle, state.x, chain_minor=False if model.dense_only else None,
                                                       model=model if parallel and noise is None and jax_keys is None else None)
        keys = None
        if jax_keys is not None:  # the draws go straight into the chains' noise buffers, in their layout (dense (C, T, d): a key's values are contiguous; chain-minor (T, d, C))
            eps_aux, eps_samp, u_acc = chains.eps_aux, chains.eps_samp, chains.u_acc
            n = chains.T * chains.dx
            ks, es = (1, chains.C) if chains.chain_minor else (n, 1)
            _random._jax_fill(1, jax_keys[:, 0], n, chains.dtype, 0.0, 1.0, handle, out=eps_aux, key_stride=ks, elem_stride=es)
            _random._jax_fill(1, jax_keys[:, 1], n, chains.dtype, 0.0, 1.0, handle, out=eps_samp, key_stride=ks, elem_stride=es)
            _random._jax_fill(0, jax_keys[:, 2], 1, chains.dtype, 0.0, 1.0, handle, out=u_acc, key_stride=1, elem_stride=1)
        elif noise is None:  # the keyed sweep: same values as draw() + sweep(), in one call (eps buffers: taken on demand, the fused sweep has none)
            keys = _random.split(key, 3)
            eps_aux, eps_samp, u_acc = None, None, chains.u_acc
        else:
            shape = (chains.C, chains.T, chains.dx)
            eps_aux, eps_samp, u_acc = chains.eps_aux, chains.eps_samp, chains.u_acc
            eps_aux.copy_from_host(chains._to_layout(np.asarray(noise["eps_aux"], chains.dtype)).reshape(eps_aux.shape))
            eps_samp.copy_from_host(chains._to_layout(np.asarray(noise["eps_samp"], chains.dtype)).reshape(eps_samp.shape))
            u_acc.copy_from_host(np.asarray(noise["u_accept"], chains.dtype).reshape(chains.C))
        sweep(handle, chains, delta, eps_aux, eps_samp, u_acc, keys)
        if resident:
            return KalmanSampler(x=chains, updated=chains.accepted)
        acc = chains.accepted.to_host().astype(bool)
        x = chains.to_host()
        out = KalmanSampler(x=x[0] if np.ndim(state.x) == 2 else x, updated=bool(acc[0]) if np.ndim(state.x) == 2 else acc)
        logs = chains.logs.to_host()
        out.log_alpha = float(logs[0, 0]) if np.ndim(state.x) == 2 else logs[:, 0]
        out.logs = logs
        return out

    kernel.sweep = sweep
    kernel.draw = draw

    def init(x):
        return KalmanSampler(x=x, updated=True)

    return init, kernel

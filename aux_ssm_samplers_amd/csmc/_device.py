"""Host driver of auxssm_csmc_sweep: model description, noise, buffers."""
import ctypes as C

import numpy as np

from .. import _lib, random as _random
from .models import (GaussianInit, LinearGaussianDynamics, FlatPotential, GaussianObsPotential, SVPotential, Lorenz63Dynamics,
                     MaskedGaussianObsPotential)

_UNSUPPORTED = ("{what} is a Python object the HIP kernels cannot evaluate. The cSMC kernels run the closed model family "
                "of aux_ssm_samplers_amd.csmc.models (GaussianInit, LinearGaussianDynamics, Lorenz63Dynamics, FlatPotential, "
                "GaussianObsPotential, MaskedGaussianObsPotential, SVPotential) in-kernel; there is no CPU fallback.")


class FkDesc:
    def __init__(self, proposal, potential, m0, chol_P0, F, b, chol_Q, y, sig_y, transition=_lib.TRANS_LINEAR, gradient=_lib.GRAD_NONE):
        self.proposal, self.potential, self.sig_y, self.transition = proposal, potential, float(sig_y), int(transition)
        self.gradient = int(gradient)
        self.m0 = np.ascontiguousarray(m0, np.float64).reshape(-1)
        self.dx = self.m0.shape[0]
        d = self.dx
        self.chol_P0 = np.ascontiguousarray(chol_P0, np.float64).reshape(d, d)
        F, b, chol_Q = np.asarray(F, np.float64), np.asarray(b, np.float64), np.asarray(chol_Q, np.float64)
        self.tv = None  # time-varying transitions: (F_t (T-1,d,d), b_t (T-1,d), chol_Q_t (T-1,d,d)), uploaded per handle / dtype
        if F.ndim == 3:
            n = F.shape[0]
            self.tv = (np.ascontiguousarray(F.reshape(n, d, d)), np.ascontiguousarray(np.broadcast_to(b, (n, d))),
                       np.ascontiguousarray(np.broadcast_to(chol_Q, (n, d, d))))
            F, b, chol_Q = self.tv[0][0], self.tv[1][0], self.tv[2][0]  # (the invariant slots are not read then)
        self.F = np.ascontiguousarray(F, np.float64).reshape(d, d)
        self.b = np.ascontiguousarray(b, np.float64).reshape(d)
        self.chol_Q = np.ascontiguousarray(chol_Q, np.float64).reshape(d, d)
        self.y = None if y is None else np.asarray(y)
        self._ydev = {}
        self._tvdev = {}

    def tvdev(self, handle, dtype, T):
        """device copies of the time-varying transition arrays (or None)"""
        if self.tv is None:
            return None
        if self.tv[0].shape[0] != T - 1:
            raise ValueError(f"time-varying dynamics have {self.tv[0].shape[0]} rows, the state has T - 1 = {T - 1} transitions")
        key = (id(handle), np.dtype(dtype).str)
        if key not in self._tvdev:
            self._tvdev[key] = tuple(handle.to_device(a, dtype) for a in self.tv)
        return self._tvdev[key]

    def struct(self, handle, dtype, T):
        """the auxssm_fk_model of this description on `handle` (keeps the device arrays alive through self)"""
        m = _lib.FkModel(self.proposal, self.potential, self.dx, self.transition, self.m0.ctypes.data, self.chol_P0.ctypes.data,
                         self.F.ctypes.data, self.b.ctypes.data, self.chol_Q.ctypes.data, None, self.sig_y, None, None, None, self.gradient, 0)
        yd = self.ydev(handle, dtype)
        if yd is not None:
            if yd.shape[0] != T:
                raise ValueError(f"observations have {yd.shape[0]} time steps, state has {T}")
            m.y = yd.ptr.value
        tv = self.tvdev(handle, dtype, T)
        if tv is not None:
            m.F_t, m.b_t, m.chol_Q_t = tv[0].ptr.value, tv[1].ptr.value, tv[2].ptr.value
        return m

    def ydev(self, handle, dtype):
        if self.y is None:
            return None
        key = (id(handle), np.dtype(dtype).str)
        if key not in self._ydev:
            self._ydev[key] = handle.to_device(self.y, dtype)
        return self._ydev[key]


def _potential(G0, Gt, d):
    if type(G0) is not type(Gt) and not (isinstance(G0, GaussianInit) and isinstance(Gt, GaussianObsPotential)):
        raise NotImplementedError(_UNSUPPORTED.format(what=f"G0={type(G0).__name__} with Gt={type(Gt).__name__}"))
    if isinstance(Gt, FlatPotential):
        return _lib.POT_FLAT, None, 1.0
    if isinstance(Gt, MaskedGaussianObsPotential):
        if G0.y is None or Gt.params is None:
            raise ValueError("the potential needs y (G0.y = ys[0]) and params (Gt.params = ys[1:])")
        y = np.concatenate([np.reshape(G0.y, (1, d)), np.reshape(Gt.params, (-1, d))], axis=0)
        if abs(G0.sig - Gt.sig) > 1e-12 * Gt.sig:
            raise NotImplementedError("G0 and Gt must share the observation noise scale")
        return _lib.POT_GAUSS_OBS_MASKED, y, Gt.sig
    if isinstance(Gt, (GaussianObsPotential, SVPotential)):
        y0 = G0.m0 if isinstance(G0, GaussianInit) else G0.y
        if y0 is None or Gt.params is None:
            raise ValueError("the potential needs y (G0.y = ys[0]) and params (Gt.params = ys[1:])")
        y = np.concatenate([np.reshape(y0, (1, d)), np.reshape(Gt.params, (-1, d))], axis=0)
        if isinstance(Gt, SVPotential):
            return _lib.POT_SV, y, 1.0
        sig = Gt.sig
        if isinstance(G0, GaussianInit):
            s0 = float(np.sqrt(np.reshape(G0.P0, -1)[0]))
            if abs(s0 - sig) > 1e-12 * sig:
                raise NotImplementedError("G0 and Gt must share the observation noise scale")
        return _lib.POT_GAUSS_OBS, y, sig
    raise NotImplementedError(_UNSUPPORTED.format(what=f"Gt={type(Gt).__name__}"))


def _dyn(M0, Mt):
    if not isinstance(M0, GaussianInit):
        raise NotImplementedError(_UNSUPPORTED.format(what=f"M0={type(M0).__name__}"))
    if not isinstance(Mt, (LinearGaussianDynamics, Lorenz63Dynamics)):
        raise NotImplementedError(_UNSUPPORTED.format(what=f"Mt={type(Mt).__name__}"))
    return M0, Mt


def _trans(Mt):
    """(transition kind, F, b) as the C ABI wants them (include/auxssm.h, auxssm_fk_transition)"""
    if isinstance(Mt, Lorenz63Dynamics):
        F = np.zeros((3, 3))
        F[0] = np.asarray(Mt.theta, np.float64).reshape(3)
        return _lib.TRANS_LORENZ63_EM, F, np.array([float(Mt.dt), 0.0, 0.0])
    return _lib.TRANS_LINEAR, Mt.F, Mt.b


def describe_bootstrap(M0, G0, Mt, Gt, Pt):
    """_primitives.csmc.get_kernel: M0/Mt are the proposals, G0/Gt the potentials."""
    M0, Mt = _dyn(M0, Mt)
    if Pt is not None and Pt is not Mt and not (isinstance(Pt, LinearGaussianDynamics) and isinstance(Mt, LinearGaussianDynamics)
                                                   and np.array_equal(Pt.F, Mt.F) and np.array_equal(Pt.Q, Mt.Q) and np.array_equal(Pt.b, Mt.b)):
        raise NotImplementedError("backward sampling with Pt != Mt is not supported by the bootstrap device kernel")
    d = np.size(M0.m0)
    pot, y, sig = _potential(G0, Gt, d)
    tk, F, b = _trans(Mt)
    return FkDesc(_lib.PROP_BOOTSTRAP_LG, pot, M0.m0, M0.chol(), F, b, Mt.chol(), y, sig, tk)


def describe_independent(M0, G0, Mt, Gt, Pt, gradient=_lib.GRAD_NONE):
    """csmc.get_independent_kernel (classical): proposals N(u_t [+ delta_t/2 grad_t], delta_t/2 I); M0/Mt enter the weights."""
    M0, Mt = _dyn(M0, Mt)
    if Pt is not None and Pt is not Mt:
        raise NotImplementedError("Pt must be the model dynamics Mt")
    d = np.size(M0.m0)
    pot, y, sig = _potential(G0, Gt, d)
    tk, F, b = _trans(Mt)
    return FkDesc(_lib.PROP_AUX_INDEPENDENT, pot, M0.m0, M0.chol(), F, b, Mt.chol(), y, sig, tk, gradient)


def key_noise(handle, key, Cn, T, N, d, dtype, wide=None):
    """The explicit noise arrays a THREEFRY sweep with `key` draws in-kernel (index map: csrc/csmc.hip::k_csmc_fwd): an
    EXPLICIT sweep on these arrays is bit-identical to the keyed one.  Debug / test utility.  wide (default: d > 4): the wide-state kernels
    (csrc/csmc_wide.hip) index their draws by the natural flat position in the explicit arrays."""
    if wide is None:
        wide = d > 4
    if wide:
        return dict(eps_aux=handle.rng_normal(key, 1, (Cn, T, d), dtype).to_host(), eps_prop=handle.rng_normal(key, 2, (Cn, T, N, d), dtype).to_host(),
                    u_res=handle.rng_uniform(key, 3, (Cn, max(T - 1, 0), N), dtype).to_host() if T > 1 else np.zeros((Cn, 0, N), dtype),
                    u_bwd=handle.rng_uniform(key, 4, (Cn, T), dtype).to_host())
    T2 = (T + 1) // 2
    ep = handle.rng_normal(key, 2, (Cn, T2, N, d, 2), dtype).to_host()
    ur = handle.rng_uniform(key, 3, (Cn, T2, N, 2), dtype).to_host()
    return dict(eps_aux=handle.rng_normal(key, 1, (Cn, T, d), dtype).to_host(),
                eps_prop=np.ascontiguousarray(np.moveaxis(ep, 4, 2).reshape(Cn, 2 * T2, N, d)[:, :T]),
                u_res=np.ascontiguousarray(np.moveaxis(ur, 3, 2).reshape(Cn, 2 * T2, N)[:, :max(T - 1, 0)]),
                u_bwd=handle.rng_uniform(key, 4, (Cn, T), dtype).to_host())


class CsmcChains:
    """C chains' reference trajectories resident in HBM, (C, T, d), with the sweep's other per-chain buffers: a run of sweeps on them
    (kernel(key, state, delta) with state.x a CsmcChains) never returns to the host.  delta: per-time-step step sizes (T,) kept on the
    device beside sqrt(delta / 2), the array the sweep reads (csmc/generic.py:61-63)."""

    def __init__(self, handle, x, delta=None, dtype=None):
        x = np.asarray(x)
        if x.ndim == 2:
            x = x[None]
        self.handle = handle
        self.C, self.T, self.dx = x.shape
        self.dtype = np.dtype(dtype or (np.float32 if x.dtype == np.float32 else np.float64))
        self.x = handle.to_device(x, self.dtype)
        self.ancestors = handle.zeros((self.C, self.T), np.int32)
        self.delta = self.sqrt_half_delta = None
        if delta is not None:
            self.set_delta(delta)

    def set_delta(self, delta):
        d = np.asarray(delta, np.float64) * np.ones(self.T)
        self.delta = self.handle.to_device(d, self.dtype)
        self.sqrt_half_delta = self.handle.to_device(np.sqrt(0.5 * d), self.dtype)

    def to_host(self):
        return self.x.to_host()

    def stats_to_host(self, a):
        return a.to_host()


def sweep_resident(fk, chains, N, backward, key):
    """One Threefry-keyed auxssm_csmc_sweep on resident chains.  Asynchronous."""
    if _random.compat() == "jax":
        raise NotImplementedError('random.set_compat("jax") runs the particle kernels on explicit arrays of the reference\'s draws (T x N x d per chain): pass host '
                                  "states (csmc/_device.py::sweep); resident chains draw inside the kernels from this package's own streams")
    handle, d = chains.handle, chains.dx
    if d != fk.dx:
        raise ValueError(f"state dimension {d} != model dimension {fk.dx}")
    m = fk.struct(handle, chains.dtype, chains.T)
    shd = None
    if fk.proposal == _lib.PROP_AUX_INDEPENDENT:
        if chains.sqrt_half_delta is None:
            raise ValueError("delta is required")
        shd = chains.sqrt_half_delta
    k = _random.as_key(key)
    nz = _lib.CsmcNoise()
    nz.mode, nz.key0, nz.key1 = _lib.NOISE_THREEFRY, int(k[0]), int(k[1])
    _lib.check(handle.lib.auxssm_csmc_sweep(
        handle.h, _lib.dtype_code(chains.dtype), C.byref(m), chains.C, chains.T, N, int(bool(backward)), shd.ptr if shd is not None else None,
        chains.x.ptr, C.byref(nz), chains.ancestors.ptr, None, None, None))


def _fk_struct(fk, handle, dtype, T):
    return fk.struct(handle, dtype, T)


def pit_sweep_resident(fk, chains, N, key):
    """One Threefry-keyed auxssm_csmc_pit_sweep (parallel-in-time cSMC) on resident chains.  Asynchronous."""
    if _random.compat() == "jax":
        raise NotImplementedError('random.set_compat("jax") runs the particle kernels on explicit arrays of the reference\'s draws (T x N x d per chain): pass host '
                                  "states (csmc/_device.py::sweep); resident chains draw inside the kernels from this package's own streams")
    handle = chains.handle
    if chains.dx != fk.dx:
        raise ValueError(f"state dimension {chains.dx} != model dimension {fk.dx}")
    if chains.sqrt_half_delta is None:
        raise ValueError("delta is required")
    m = _fk_struct(fk, handle, chains.dtype, chains.T)
    k = _random.as_key(key)
    nz = _lib.CsmcNoise()
    nz.mode, nz.key0, nz.key1 = _lib.NOISE_THREEFRY, int(k[0]), int(k[1])
    _lib.check(handle.lib.auxssm_csmc_pit_sweep(handle.h, _lib.dtype_code(chains.dtype), C.byref(m), chains.C, chains.T, N,
                                                chains.sqrt_half_delta.ptr, chains.x.ptr, C.byref(nz), chains.ancestors.ptr))


def pit_sweep(fk, x, N, *, key=None, noise=None, delta=None, handle=None):
    """Parallel-in-time cSMC sweep.  x: (T, d) one chain or (C, T, d).  noise: dict(eps_aux (C,T,d), eps_prop (C,T,N,d), u_res (C,T,N)) of
    explicit arrays or None -> Threefry(key).  Returns (x_new, ancestors)."""
    handle = handle or _lib.default_handle()
    x = np.asarray(x)
    single = x.ndim == 2
    xc = x[None] if single else x
    Cn, T, d = xc.shape
    if d != fk.dx:
        raise ValueError(f"state dimension {d} != model dimension {fk.dx}")
    if delta is None:
        raise ValueError("delta is required")
    dtype = np.dtype(np.float32) if xc.dtype == np.float32 else np.dtype(np.float64)
    xd = handle.to_device(xc, dtype)
    anc = handle.zeros((Cn, T), np.int32)
    m = _fk_struct(fk, handle, dtype, T)
    shd = handle.to_device(np.sqrt(0.5 * np.asarray(delta, np.float64)) * np.ones(T), dtype)
    keep = []
    nz = _lib.CsmcNoise()
    if noise is None and _random.compat() == "jax":   # the reference's own draws from this key (random.jax_pit_noise); several chains: `key` (C, 2) or split(key, C)
        kk = np.asarray(key, np.uint32)
        keys = kk if kk.ndim == 2 else (_random.as_key(key)[None] if Cn == 1 else _random.jax_split(_random.as_key(key), Cn))
        if keys.shape[0] != Cn:
            raise ValueError(f"{keys.shape[0]} keys for {Cn} chains")
        per = [_random.jax_pit_noise(k_, T, N, d, dtype, handle) for k_ in keys]
        noise = {name: np.stack([p_[name] for p_ in per]) for name in per[0]}
    if noise is None:
        k = _random.as_key(key)
        nz.mode, nz.key0, nz.key1 = _lib.NOISE_THREEFRY, int(k[0]), int(k[1])
    else:
        nz.mode = _lib.NOISE_EXPLICIT
        for name, shp in dict(eps_prop=(Cn, T, N, d), u_res=(Cn, T, N), eps_aux=(Cn, T, d)).items():
            buf = handle.to_device(np.asarray(noise[name], dtype).reshape(shp))
            keep.append(buf)
            setattr(nz, name, buf.ptr.value)
    _lib.check(handle.lib.auxssm_csmc_pit_sweep(handle.h, _lib.dtype_code(dtype), C.byref(m), Cn, T, N, shd.ptr, xd.ptr, C.byref(nz), anc.ptr))
    xo, ao = xd.to_host(), anc.to_host()
    return (xo[0], ao[0]) if single else (xo, ao)


def sweep(fk, x, N, backward, *, key=None, noise=None, delta=None, handle=None, want_history=False):
    """x: (T, d) one chain or (C, T, d).  noise: dict of explicit arrays (eps_prop, u_res, u_bwd[, eps_aux]) or None -> Threefry(key).
    Returns (x_new, ancestors, history dict or None)."""
    handle = handle or _lib.default_handle()
    x = np.asarray(x)
    single = x.ndim == 2
    xc = x[None] if single else x
    Cn, T, d = xc.shape
    if d != fk.dx:
        raise ValueError(f"state dimension {d} != model dimension {fk.dx}")
    dtype = np.dtype(np.float32) if xc.dtype == np.float32 else np.dtype(np.float64)
    xd = handle.to_device(xc, dtype)
    anc = handle.zeros((Cn, T), np.int32)
    m = fk.struct(handle, dtype, T)
    shd = None
    if fk.proposal == _lib.PROP_AUX_INDEPENDENT:
        if delta is None:
            raise ValueError("delta is required")
        shd_h = np.sqrt(0.5 * np.asarray(delta, np.float64)) * np.ones(T)  # csmc/generic.py:61-63
        shd = handle.to_device(shd_h, dtype)
    keep = []
    nz = _lib.CsmcNoise()
    if noise is None and _random.compat() == "jax":
        # the reference's own draws from this key (random.jax_csmc_noise), as explicit arrays; several chains: one key per chain, `key` (C, 2) or split(key, C)
        # (the plain cSMC kernel: its draws are made by the model's own M0.sample / Mt.sample in the reference -- one normal(key, (N, d)) per call in every model
        # the reference defines, which is what the device proposal kernels apply their Cholesky factors to)
        aux = fk.proposal == _lib.PROP_AUX_INDEPENDENT
        kk = np.asarray(key, np.uint32)
        keys = kk if kk.ndim == 2 else (_random.as_key(key)[None] if Cn == 1 else _random.jax_split(_random.as_key(key), Cn))
        if keys.shape[0] != Cn:
            raise ValueError(f"{keys.shape[0]} keys for {Cn} chains")
        per = [_random.jax_csmc_noise(k_, T, N, d, dtype, bool(backward), handle, auxiliary=aux) for k_ in keys]
        noise = {name: np.stack([p_[name] for p_ in per]) for name in per[0]}
    if noise is None:
        k = _random.as_key(key)
        nz.mode, nz.key0, nz.key1 = _lib.NOISE_THREEFRY, int(k[0]), int(k[1])
    else:
        nz.mode = _lib.NOISE_EXPLICIT
        shapes = dict(eps_prop=(Cn, T, N, d), u_res=(Cn, max(T - 1, 0), N), u_bwd=(Cn, T), eps_aux=(Cn, T, d))
        for name, shp in shapes.items():
            a = noise.get(name)
            if a is None:
                continue
            buf = handle.to_device(np.asarray(a, dtype).reshape(shp))
            keep.append(buf)
            setattr(nz, name, buf.ptr.value)
    hist = None
    xs = lws = As = None
    if want_history:
        xs = handle.empty((Cn, T, N, d), dtype)
        lws = handle.empty((Cn, T, N), dtype)
        As = handle.zeros((Cn, max(T - 1, 1), N), np.int32)
    _lib.check(handle.lib.auxssm_csmc_sweep(
        handle.h, _lib.dtype_code(dtype), C.byref(m), Cn, T, N, int(bool(backward)), shd.ptr if shd is not None else None,
        xd.ptr, C.byref(nz), anc.ptr, xs.ptr if xs else None, lws.ptr if lws else None, As.ptr if As else None))
    xo, ao = xd.to_host(), anc.to_host()
    if want_history:
        hist = dict(xs=xs.to_host(), log_ws=lws.to_host(), As=As.to_host()[:, :T - 1])
        if single:
            hist = {k_: v[0] for k_, v in hist.items()}
    return (xo[0], ao[0], hist) if single else (xo, ao, hist)

"""Built-in device model factories for the auxiliary Kalman sampler.

The reference's factories are JAX closures traced into the sweep; a Python callable cannot run inside a HIP
kernel, so models whose factories are known in closed form are provided as device factories and the whole sweep
(auxssm_kalman_sweep) stays in HBM.  Arbitrary NumPy factories still work through the generic host path of
kalman.get_kernel.
"""
import numpy as np

from .. import _lib
from .._primitives.kalman.base import DeviceLGSSM, _upload_arr


class LGConcatModel:
    """Linear-Gaussian SSM with the auxiliary observations concatenated to the real ones
    (the observation-factory pattern of examples/lorenz/auxiliary_kalman.py:26-35):

        dynamics_factory(x)            -> m0, P0, Fs, Qs, bs                       (independent of x)
        observations_factory(x, u, d)  -> ys=[u; y], Hs=[I; Hobs], Rs=blkdiag(d/2 I, Robs), cs=[0; cobs]
        log_likelihood_fn(x)           -> prior_logpdf(x) + sum_t log N(y_t; Hobs_t x_t + cobs_t, Robs_t)

    Pass the three bound methods to kalman.get_kernel; it recognises them and runs the fused device sweep.
    The same methods are plain NumPy factories, so they also drive the generic host path (and the oracle)."""

    kmodel = _lib.KMODEL_LG_CONCAT
    dense_only = False

    def __init__(self, m0, P0, Fs, Qs, bs, Hobs, Robs, cobs, yobs):
        self.m0, self.P0, self.Fs, self.Qs, self.bs = m0, P0, Fs, Qs, bs
        self.Hobs, self.Robs, self.cobs, self.yobs = Hobs, Robs, cobs, yobs
        self.T = np.shape(yobs)[0]
        self.dx = np.shape(m0)[-1]
        self.p_obs = np.shape(yobs)[-1]
        self._dev = {}

    # ---- NumPy factories (reference call signatures: kalman/generic.py:80-81,89) ----
    def dynamics_factory(self, x):
        return self.m0, self.P0, self.Fs, self.Qs, self.bs

    def observations_factory(self, x, u, delta):
        T, d, po = self.T, self.dx, self.p_obs
        dt = np.asarray(u).dtype
        P = d + po
        ys = np.concatenate([u, np.asarray(self.yobs, dt)], axis=-1)
        Hs = np.concatenate([np.broadcast_to(np.eye(d, dtype=dt), (T, d, d)), np.asarray(self.Hobs, dt)], axis=1)
        Rs = np.zeros((T, P, P), dt)
        Rs[:, :d, :d] = 0.5 * delta * np.eye(d)
        Rs[:, d:, d:] = self.Robs
        cs = np.concatenate([np.zeros((T, d), dt), np.asarray(self.cobs, dt)], axis=-1)
        return ys, Hs, Rs, cs

    def log_likelihood_fn(self, x):
        from .._primitives.kalman.base import joint_logpdf, LGSSM
        return joint_logpdf(self.yobs, x, LGSSM(self.m0, self.P0, self.Fs, self.Qs, self.bs, self.Hobs, self.Robs, self.cobs))

    # ---- device side ----
    def device(self, handle, dtype):
        key = (id(handle), np.dtype(dtype).str)
        dev = self._dev.get(key)
        if dev is None:
            T, d, po = self.T, self.dx, self.p_obs
            lg = (self.m0, self.P0, self.Fs, self.Qs, self.bs, self.Hobs, self.Robs, self.cobs)
            dl = DeviceLGSSM(handle, lg, 1, T, 1, d, po, False, dtype)
            ybuf, yarr = _upload_arr(handle, self.yobs, (po,), 1, T, 1, False, False, np.dtype(dtype), "yobs")
            dev = self._dev[key] = (dl, ybuf, yarr)
        return dev


class SVModel:
    """Multivariate stochastic volatility  y_{t,k} ~ N(0, exp(x_{t,k}))  on linear-Gaussian dynamics
    x_{t+1} = F x_t + b + N(0, Q), with the first- or second-order auxiliary observation factories of
    examples/stochastic_volatility/auxiliary_kalman.py:22-48 (closed-form gradient / Hessian of the potential,
    model.py:56-82, instead of jax.grad):

        dynamics_factory(x)           -> m0, P0, tile(F), tile(Q), tile(b)                        (:22-26)
        observations_factory(x, u, d) -> order 1: ys = u + d/2 grad(x), H = I, R = d/2 I, c = 0     (:28-35)
                                         order 2: Om = (-hess + 2/d I)^-1, ys = Om (2u/d + grad - hess x), H = I, R = Om (:37-46)
        log_likelihood_fn(x)          -> log N(x_0; m0, P0) + sum_t log N(x_t; F x_{t-1} + b, Q) + sum log g   (:48-52)

    Pass the three bound methods to kalman.get_kernel: it runs the device sweep (auxssm_kalman_sweep, model kind SV_FIRST /
    SV_SECOND).  The same methods are NumPy factories for the host path and the oracle."""
    dense_only = True

    def __init__(self, ys, m0, P0, F, Q, b, order=1):
        if order not in (1, 2):
            raise ValueError("order must be 1 or 2")
        self.yobs = np.asarray(ys)
        self.T, self.dx = self.yobs.shape
        self.p_obs = self.dx
        self.m0, self.P0 = np.asarray(m0), np.asarray(P0)
        self.F, self.Q, self.b = np.asarray(F), np.asarray(Q), np.asarray(b)
        self.order = order
        self.kmodel = _lib.KMODEL_SV_FIRST if order == 1 else _lib.KMODEL_SV_SECOND
        n = self.T - 1
        self.Fs = np.broadcast_to(self.F, (n,) + self.F.shape)
        self.Qs = np.broadcast_to(self.Q, (n,) + self.Q.shape)
        self.bs = np.broadcast_to(self.b, (n,) + self.b.shape)
        self._dev = {}

    # potential and its derivatives (model.py:56-82), NaN -> 0 as jnp.nan_to_num
    def _w(self, x):
        return self.yobs.astype(x.dtype) ** 2 * np.exp(-x)

    def log_potential(self, x):
        x = np.asarray(x)
        with np.errstate(all="ignore"):
            val = -0.5 * np.log(2 * np.pi) - 0.5 * x - 0.5 * self._w(x)
        return float(np.sum(np.nan_to_num(val)))

    def dynamics_factory(self, x):
        return self.m0, self.P0, self.Fs, self.Qs, self.bs

    def observations_factory(self, x, u, delta):
        x, u = np.asarray(x), np.asarray(u)
        T, d = self.T, self.dx
        dt = u.dtype
        eyes = np.broadcast_to(np.eye(d, dtype=dt), (T, d, d))
        zeros = np.zeros((T, d), dt)
        with np.errstate(all="ignore"):
            w = self._w(x)
            grad = np.nan_to_num(0.5 * (w - 1.0))
            if self.order == 1:
                return (u + 0.5 * delta * grad).astype(dt), eyes, (0.5 * delta * eyes).astype(dt), zeros
            hess = -0.5 * w
            om = 1.0 / (-hess + 2.0 / delta)
            ys = om * (2.0 * u / delta + grad - hess * x)
        Rs = np.zeros((T, d, d), dt)
        Rs[:, np.arange(d), np.arange(d)] = om
        return ys.astype(dt), eyes, Rs, zeros

    def log_likelihood_fn(self, x):
        from .._primitives.kalman.base import prior_logpdf, LGSSM
        x = np.asarray(x)
        prior = prior_logpdf(x, LGSSM(self.m0, self.P0, self.Fs, self.Qs, self.bs, None, None, None))
        return prior + self.log_potential(x)

    def device(self, handle, dtype):
        key = (id(handle), np.dtype(dtype).str)
        dev = self._dev.get(key)
        if dev is None:
            T, d = self.T, self.dx
            lg = (self.m0, self.P0, self.Fs, self.Qs, self.bs, None, None, None)
            dl = DeviceLGSSM(handle, lg, 1, T, 1, d, d, False, dtype)
            ybuf, yarr = _upload_arr(handle, self.yobs, (d,), 1, T, 1, False, False, np.dtype(dtype), "ys")
            dev = self._dev[key] = (dl, ybuf, yarr)
        return dev

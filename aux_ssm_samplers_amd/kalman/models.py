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
    dense_only = False  # >= 32 chains run chain-minor (lanes over chains), as the LG_CONCAT sweep does

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


class LorenzModel:
    """Stochastic Lorenz-63 with Euler-Maruyama dynamics and sparse linear-Gaussian observations,
    examples/lorenz/auxiliary_kalman.py:14-52 (model.py:10-25; linearisation.py:11-44 with the analytic Jacobian in place of jacfwd):

        dynamics_factory(x)           -> m0, P0, F_t = I + dt J(x_t), tile(Q), b_t = mean(x_t) - F_t x_t       (:27-29)
        observations_factory(x, u, d) -> ys = [u; y], Hs = [I; H], Rs = blkdiag(d/2 I, R), cs = [0; c]        (:31-36)
        log_likelihood_fn(x)          -> log N(x_0; m0, P0) + sum log N(x_{t+1}; mean(x_t), Q) + nansum_t log N(y_t; H_t x_t + c_t, R_t)

    ys (T, po) with NaN rows where nothing is observed; Hs (T, po, 3) may carry NaN rows there too (model.py:43-56).
    Pass the three bound methods to kalman.get_kernel: it runs the device sweep (model kind LORENZ63_EXT)."""
    kmodel = _lib.KMODEL_LORENZ63_EXT
    dense_only = False

    def __init__(self, ys, Hs, Rs, cs, m0, P0, theta, sigma_x, dt):
        self.yobs, self.Hobs, self.Robs, self.cobs = np.asarray(ys), np.asarray(Hs), np.asarray(Rs), np.asarray(cs)
        self.T, self.p_obs = self.yobs.shape
        self.dx = 3
        self.m0, self.P0 = np.asarray(m0, np.float64), np.asarray(P0, np.float64)
        # theta (3,): one parameter vector for all chains; (C, 3): one per chain (the Gibbs sampler over (x, theta), loop.LorenzThetaStep).
        # The NumPy methods below (host path / oracle) use chain 0's.
        self.theta_rows = np.asarray(theta, np.float64).reshape(-1, 3)
        self.theta = self.theta_rows[0].copy()
        self.sigma_x, self.dt = float(sigma_x), float(dt)
        self.Q = self.dt * self.sigma_x ** 2 * np.eye(3)
        self.Qs = np.broadcast_to(self.Q, (self.T - 1, 3, 3))
        self._dev = {}

    def mean(self, x):
        x = np.asarray(x)
        th, dt = self.theta, self.dt
        x1, x2, x3 = x[..., 0], x[..., 1], x[..., 2]
        return x + dt * np.stack([th[0] * (x2 - x1), th[1] * x1 - x2 - x1 * x3, x1 * x2 - th[2] * x3], axis=-1)

    def dynamics_factory(self, x):
        x = np.asarray(x)
        xl = x[:-1]
        th, dt = self.theta, self.dt
        n = xl.shape[0]
        J = np.zeros((n, 3, 3), x.dtype)
        J[:, 0, 0], J[:, 0, 1] = -th[0], th[0]
        J[:, 1, 0], J[:, 1, 1], J[:, 1, 2] = th[1] - xl[:, 2], -1.0, -xl[:, 0]
        J[:, 2, 0], J[:, 2, 1], J[:, 2, 2] = xl[:, 1], xl[:, 0], -th[2]
        Fs = np.eye(3, dtype=x.dtype) + dt * J
        bs = self.mean(xl) - np.einsum("tij,tj->ti", Fs, xl)
        return self.m0.astype(x.dtype), self.P0.astype(x.dtype), Fs, self.Qs.astype(x.dtype), bs

    def observations_factory(self, x, u, delta):
        return LGConcatModel.observations_factory(self, x, u, delta)

    def log_likelihood_fn(self, x):
        """NumPy (host path / oracle): the same nansum-over-steps semantics as the reference (:38-46)."""
        x = np.asarray(x, np.float64)

        def mvn(r, cov):
            L = np.linalg.cholesky(cov)
            z = np.linalg.solve(L, r[..., None])[..., 0]
            return -0.5 * np.sum(z * z, -1) - np.sum(np.log(np.diagonal(L, axis1=-2, axis2=-1)), -1) - 0.5 * r.shape[-1] * np.log(2 * np.pi)

        out = mvn(x[0] - self.m0, self.P0)
        out += np.sum(mvn(x[1:] - self.mean(x[:-1]), self.Q))
        with np.errstate(all="ignore"):
            pred = np.einsum("tij,tj->ti", self.Hobs, x) + self.cobs
            ll = mvn(np.asarray(self.yobs, np.float64) - pred, np.asarray(self.Robs, np.float64))
        return float(out + np.nansum(ll))

    def device(self, handle, dtype):
        key = (id(handle), np.dtype(dtype).str)
        dev = self._dev.get(key)
        if dev is None:
            T, po = self.T, self.p_obs
            # rides in the Fs slot of the C struct (include/auxssm.h, LORENZ63_EXT): rows [theta1, theta2, theta3, dt]
            par = np.concatenate([self.theta_rows, np.full((self.theta_rows.shape[0], 1), self.dt)], axis=1)
            n = T - 1
            lg = (self.m0, self.P0, np.broadcast_to(np.eye(3), (n, 3, 3)), self.Qs, np.broadcast_to(np.zeros(3), (n, 3)),
                  self.Hobs, self.Robs, self.cobs)
            dl = DeviceLGSSM(handle, lg, 1, T, 1, 3, po, False, dtype)
            pbuf = handle.to_device(par, dtype)
            dl.bufs["lorenz_par"] = pbuf
            dl.c.Fs = pbuf.arr(4 if par.shape[0] > 1 else 0, 0, 0)
            ybuf, yarr = _upload_arr(handle, self.yobs, (po,), 1, T, 1, False, False, np.dtype(dtype), "ys")
            dev = self._dev[key] = (dl, ybuf, yarr)
        return dev

    def par_device(self, handle, dtype, C):
        """the device rows [theta, dt] the sweep reads, one per chain (C, 4): a single theta is replicated on first use, so that a theta
        step can then write each chain's own"""
        dl = self.device(handle, dtype)[0]
        pbuf = dl.bufs["lorenz_par"]
        if pbuf.shape[0] != C:
            if pbuf.shape[0] != 1:
                raise ValueError(f"the model holds {pbuf.shape[0]} theta rows, the chains are {C}")
            pbuf = dl.bufs["lorenz_par"] = handle.to_device(np.repeat(pbuf.to_host(), C, axis=0), dtype)
            dl.c.Fs = pbuf.arr(4 if C > 1 else 0, 0, 0)
        return pbuf

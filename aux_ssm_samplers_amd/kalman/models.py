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

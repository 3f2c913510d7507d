"""The closed Feynman-Kac model family the HIP cSMC kernels evaluate in-kernel (include/auxssm.h, auxssm_fk_model).

They subclass the reference's protocol classes so they can be passed wherever the reference takes
(M0, G0, Mt, Gt, Pt); the kernels read their parameters, they never call Python methods on the hot path."""
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np

from .._primitives.csmc.base import Distribution, UnivariatePotential, Dynamics, Potential


@dataclass
class GaussianInit(Distribution, UnivariatePotential):
    """M0 = N(m0, P0) (e.g. test_csmc/common.py:34-49; examples/stochastic_volatility/auxiliary_csmc.py:21-27).
    Used as a potential it is log N(x; m0, P0)."""
    m0: Any
    P0: Any

    def chol(self):
        return np.linalg.cholesky(np.atleast_2d(np.asarray(self.P0, np.float64)))


@dataclass
class LinearGaussianDynamics(Dynamics, Potential):
    """x_{t+1} | x_t ~ N(F x_t + b, Q) (test_csmc/common.py:11-31; SV auxiliary_csmc.py:29-37).  Time-invariant: F (d, d), b (d,), Q (d, d);
    time-varying (the reference scans Mt.params over time, _primitives/csmc/csmc.py:103): F (T-1, d, d), b (T-1, d), Q (T-1, d, d), row t =
    the transition t -> t + 1."""
    F: Any
    b: Any
    Q: Any
    params: Optional[Any] = None

    @property
    def time_varying(self):
        return np.ndim(self.F) == 3

    def chol(self):
        """lower Cholesky factor(s) of Q: (d, d), or (T-1, d, d) when time-varying"""
        Q = np.asarray(self.Q, np.float64)
        return np.linalg.cholesky(Q if Q.ndim == 3 else np.atleast_2d(Q))


@dataclass
class FlatPotential(UnivariatePotential, Potential):
    """G = 0 (test_csmc/common.py:61-75)."""
    params: Optional[Any] = None


@dataclass
class GaussianObsPotential(UnivariatePotential, Potential):
    """log N(y_t; x_t, sig^2 I).  As G0 give y=(d,) [the reference uses GaussianDistribution(mu=y0, sig) there,
    test_csmc.py:88]; as Gt give params = ys[1:] (test_csmc/common.py:52-58)."""
    sig: float = 1.0
    y: Optional[Any] = None
    params: Optional[Any] = None


@dataclass
class SVPotential(UnivariatePotential, Potential):
    """sum_k log N(y_{t,k}; 0, exp(x_{t,k})) (examples/stochastic_volatility/auxiliary_csmc.py:39-46).
    As G0 give y = ys[0]; as Gt give params = ys[1:]."""
    y: Optional[Any] = None
    params: Optional[Any] = None


@dataclass
class Lorenz63Dynamics(Dynamics, Potential):
    """Euler-Maruyama step of the stochastic Lorenz-63 system, examples/lorenz/model.py:10-25:
    x_{t+1} | x_t ~ N(x_t + dt (phi_0(x_t) + theta * phi(x_t)), dt sigma_x^2 I),
    phi_0 = (0, -x2 - x1 x3, x1 x2), phi = (x2 - x1, x1, -x3), theta = (sigma, rho, beta)."""
    theta: Any
    sigma_x: float
    dt: float
    params: Optional[Any] = None

    def chol(self):
        return float(self.sigma_x) * np.sqrt(float(self.dt)) * np.eye(3)

    def mean(self, x):
        x = np.asarray(x)
        th = np.asarray(self.theta, np.float64)
        x1, x2, x3 = x[..., 0], x[..., 1], x[..., 2]
        f = np.stack([th[0] * (x2 - x1), th[1] * x1 - x2 - x1 * x3, x1 * x2 - th[2] * x3], axis=-1)
        return x + self.dt * f


@dataclass
class MaskedGaussianObsPotential(UnivariatePotential, Potential):
    """sum over the FINITE components of y_t of log N(y_{t,k}; x_{t,k}, sig^2): state components observed directly, missing
    components / whole missing steps carry NaN (examples/lorenz/model.py:43-56 observes x2, x3 every 80th step).
    As G0 give y = ys[0]; as Gt give params = ys[1:]."""
    sig: float = 1.0
    y: Optional[Any] = None
    params: Optional[Any] = None

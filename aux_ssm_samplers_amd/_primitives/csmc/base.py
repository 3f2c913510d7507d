"""Boundary types of the cSMC kernels (reference: aux_samplers/_primitives/csmc/base.py:18-71).

The abstract protocol classes stay importable for API compatibility.  A Python `sample` / `__call__` cannot execute
inside a HIP kernel, so the kernels accept the closed model family of `aux_ssm_samplers_amd.csmc.models` (which
subclasses these) and raise NotImplementedError for arbitrary Python components -- there is no CPU fallback."""
from dataclasses import dataclass
from typing import Any, Optional

from ..base import SamplerState


@dataclass
class CSMCState(SamplerState):
    x: Any
    updated: Any


class UnivariatePotential:
    """log-potential of the initial state: __call__(x (N, d)) -> (N,)"""

    def __call__(self, x):
        raise NotImplementedError


class Distribution:
    """initial distribution: sample(key, N) -> (N, d); logpdf(x) -> (N,)"""

    def sample(self, key, N):
        raise NotImplementedError

    def logpdf(self, x):
        raise NotImplementedError


class Potential:
    """log-potential: __call__(x_t_p_1 (N, d), x_t (N, d), params_t) -> (N,); `params` has leading axis T-1"""
    params: Optional[Any] = None

    def __call__(self, x_t_p_1, x_t, params):
        raise NotImplementedError


class Dynamics:
    """transition: sample(key, x_t (N, d), params_t) -> (N, d); logpdf(x_t_p_1, x_t, params_t) -> (N,)"""
    params: Optional[Any] = None

    def sample(self, key, x_t, params):
        raise NotImplementedError

    def logpdf(self, x_t_p_1, x_t, params):
        raise NotImplementedError

"""aux_ssm_samplers_amd -- MI355X (gfx950) native auxiliary-Kalman / conditional-SMC samplers.

Drop-in for the hot path of AdrienCorenflos/aux-ssm-samplers behind its own API:
    aux_ssm_samplers_amd.kalman.get_kernel(dynamics_factory, observations_factory, log_likelihood_fn, parallel)
    aux_ssm_samplers_amd.csmc.get_kernel(factory, N, backward, Pt)
Python host code -> ctypes -> libauxssm.so (hand-written HIP).  No CPU fallback: compute calls raise if the
library or a GPU is missing.
"""
__version__ = "0.1.0"

from ._primitives.base import SamplerState  # noqa: E402,F401  (reference: aux_samplers/__init__.py:1)
from ._primitives.linearisation import extended, gauss_hermite, cubature  # noqa: E402,F401  (reference: aux_samplers/__init__.py:2)
from ._primitives.math import mvn  # noqa: E402,F401  (reference: aux_samplers/__init__.py:3)
from .common import delta_adaptation  # noqa: E402,F401  (reference: aux_samplers/__init__.py:4)

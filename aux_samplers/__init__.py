"""Import-compatibility shim: lets code written against the reference (`from aux_samplers.kalman import get_kernel`,
`from aux_samplers._primitives.kalman import filtering`, ...) run on the MI355X implementation without edits.
It only aliases module names to `aux_ssm_samplers_amd`; there is no code of the reference here."""
import importlib
import sys

import aux_ssm_samplers_amd as _impl

_ALIASES = ["kalman", "csmc", "common", "loop", "random", "parallel", "_primitives", "_primitives.base", "_primitives.kalman",
            "_primitives.kalman.base", "_primitives.kalman.filtering", "_primitives.kalman.sampling", "_primitives.kalman.dnc_sampling",
            "_primitives.linearisation", "diagnostics", "_primitives.csmc",
            "_primitives.csmc.base", "_primitives.csmc.csmc", "_primitives.csmc.resamplings", "_primitives.csmc.pit", "_primitives.math",
            "_primitives.math.utils", "_primitives.math.mvn", "_primitives.math.mvn.base", "csmc.generic", "csmc.independent", "kalman.generic"]
for _name in _ALIASES:
    sys.modules[f"{__name__}.{_name}"] = importlib.import_module(f"aux_ssm_samplers_amd.{_name}")

from aux_ssm_samplers_amd._primitives.base import SamplerState  # noqa: E402,F401  (reference: aux_samplers/__init__.py:1)
from aux_ssm_samplers_amd._primitives.linearisation import extended, gauss_hermite, cubature  # noqa: E402,F401  (reference: aux_samplers/__init__.py:2)
from aux_ssm_samplers_amd._primitives.math import mvn  # noqa: E402,F401  (reference: aux_samplers/__init__.py:3)
from aux_ssm_samplers_amd.common import delta_adaptation  # noqa: E402,F401  (reference: aux_samplers/__init__.py:4)

__version__ = _impl.__version__

from .generic import get_kernel as get_generic_kernel
from .independent import get_kernel as get_independent_kernel
from .generic import get_kernel
from .._primitives.csmc.base import Distribution, UnivariatePotential, Dynamics, Potential, CSMCState
from ._device import CsmcChains
from .models import (GaussianInit, LinearGaussianDynamics, FlatPotential, GaussianObsPotential, SVPotential, Lorenz63Dynamics,
                     MaskedGaussianObsPotential)

__all__ = ["get_kernel", "get_generic_kernel", "get_independent_kernel", "Distribution", "UnivariatePotential", "Dynamics",
           "Potential", "CSMCState", "CsmcChains", "GaussianInit", "LinearGaussianDynamics", "FlatPotential", "GaussianObsPotential",
           "SVPotential", "Lorenz63Dynamics", "MaskedGaussianObsPotential"]

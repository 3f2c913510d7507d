from .csmc import get_kernel
from .base import CSMCState, Distribution, UnivariatePotential, Potential, Dynamics

__all__ = ["get_kernel", "CSMCState", "Distribution", "UnivariatePotential", "Potential", "Dynamics"]

"""Auxiliary particle Gibbs with independent proposals N(u_t, delta_t/2 I) -- Finke & Thiery in the auxiliary
paradigm (reference: aux_samplers/csmc/independent.py).

get_kernel(M0, G0, Mt, Gt, N, backward=False, Pt=None, gradient=False, parallel=False) -> (init, kernel).
Only the classical, non-gradient branch (independent.py:57-75) is on the hot path of this package; the gradient
branch needs autodiff of the user model and the parallel branch is the PIT-cSMC: both raise."""
from .generic import get_kernel as get_base_kernel, IndependentFactory


def get_kernel(M0, G0, Mt, Gt, N, backward=False, Pt=None, gradient=False, parallel=False):
    if parallel:
        raise NotImplementedError("parallel-in-time cSMC (independent.py:78-118) is out of scope of this build")
    if gradient:
        raise NotImplementedError("gradient-informed proposals (independent.py:62-63) need autodiff of the model: out of scope")
    if backward and Pt is None:
        Pt = Mt
    return get_base_kernel(IndependentFactory(M0, G0, Mt, Gt, Pt), N, backward, Pt)

"""Auxiliary particle Gibbs with generic proposals (reference: aux_samplers/csmc/generic.py).

get_kernel(factory, N, backward=False, Pt=None) -> (init, kernel); kernel(key, state, delta) -> CSMCState.
`factory(u, sqrt_half_delta) -> (M0, G0, Mt, Gt)` builds auxiliary model objects in the reference; here the factory
must be one this package can describe for the device: the independent-proposal factory of csmc.independent (an
`IndependentFactory`).  Any other callable raises NotImplementedError (no CPU fallback)."""
import numpy as np

from .._primitives.csmc.base import CSMCState
from . import _device


class IndependentFactory:
    """The classical factory of csmc/independent.py:57-75 in device-describable form."""

    def __init__(self, M0, G0, Mt, Gt, Pt, gradient=0):
        self.fk = _device.describe_independent(M0, G0, Mt, Gt, Pt, gradient)

    def __call__(self, u, scale):
        raise NotImplementedError("the auxiliary model is evaluated inside the HIP kernel; this factory is a descriptor")


def get_kernel(factory, N, backward=False, Pt=None):
    if backward and Pt is None:
        raise ValueError("If backward is True, the true dynamics `Pt` must be provided.")  # generic.py:44-45
    elif backward and not hasattr(Pt, "logpdf"):
        raise ValueError("`Pt` must implement a valid logpdf method.")  # generic.py:46-47
    if not isinstance(factory, IndependentFactory):
        raise NotImplementedError(_device._UNSUPPORTED.format(what=f"factory={factory!r}"))
    fk = factory.fk

    def kernel(key, state, delta, noise=None):
        # generic.py:56-72: u = x + sqrt(delta/2) eps; (m0, g0, mt, gt) = factory(u, sqrt(delta/2)); cSMC sweep
        if isinstance(state.x, _device.CsmcChains):  # resident chains: in place, asynchronous; delta None = the chains' device delta
            if delta is not None:
                state.x.set_delta(delta)
            _device.sweep_resident(fk, state.x, N, backward, key)
            return CSMCState(x=state.x, updated=state.x.ancestors)
        x, anc, extra = _device.sweep(fk, state.x, N, backward, key=key, noise=noise, delta=delta)
        out = CSMCState(x=x, updated=anc != 0)
        out.ancestors = anc
        return out

    def init(x):
        T = np.shape(x)[-2]
        return CSMCState(x=x, updated=np.zeros((T,), bool))  # generic.py:74-77 (ancestors != 0 -> all False)

    return init, kernel

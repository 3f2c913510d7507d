"""Conditional SMC kernel (reference: aux_samplers/_primitives/csmc/csmc.py).

get_kernel(M0, G0, Mt, Gt, N, backward=False, Pt=None) -> (init, kernel), kernel(key, state) -> CSMCState,
same names / order / (init, kernel) return order as the reference (csmc.py:16-66).  The whole sweep -- forward pass
with conditional multinomial resampling, then ancestor tracing or backward sampling -- is one auxssm_csmc_sweep call
(persistent one-workgroup-per-chain HIP kernels)."""
import numpy as np

from ... import _lib
from ...csmc import _device
from .base import CSMCState


def get_kernel(M0, G0, Mt, Gt, N, backward=False, Pt=None):
    if backward and Pt is None:
        Pt = Mt  # csmc.py:47-48
    elif backward and not hasattr(Pt, "logpdf"):
        raise ValueError("When `backward` is True, `Pt` must implement a valid logpdf method.")  # csmc.py:49-50
    fk = _device.describe_bootstrap(M0, G0, Mt, Gt, Pt if backward else None)

    def kernel(key, state, noise=None):
        if isinstance(state.x, _device.CsmcChains):  # resident chains: in place, asynchronous
            _device.sweep_resident(fk, state.x, N, backward, key)
            return CSMCState(x=state.x, updated=state.x.ancestors)
        x, anc, extra = _device.sweep(fk, state.x, N, backward, key=key, noise=noise)
        out = CSMCState(x=x, updated=anc != 0)  # csmc.py:59
        out.ancestors = anc
        out.history = extra
        return out

    def init(x_star):
        T = np.shape(x_star)[-2]
        return CSMCState(x=x_star, updated=np.ones((T,), bool))  # csmc.py:61-64 (ancestors == 0 -> all True)

    return init, kernel

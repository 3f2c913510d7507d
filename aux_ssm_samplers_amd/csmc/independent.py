"""Auxiliary particle Gibbs with independent proposals N(u_t, delta_t/2 I) -- Finke & Thiery in the auxiliary
paradigm (reference: aux_samplers/csmc/independent.py).

get_kernel(M0, G0, Mt, Gt, N, backward=False, Pt=None, gradient=False, parallel=False) -> (init, kernel).
parallel=False: the classical sequential sweep (independent.py:57-75, auxssm_csmc_sweep); parallel=True: the parallel-in-time cSMC
(conditional dSMC, independent.py:78-118 on _primitives/csmc/pit, auxssm_csmc_pit_sweep: log2(T) stitching launches instead of T
sequential steps; `backward` / `Pt` are unused there, as in the reference).  The gradient branch needs autodiff of the user model: raises."""
import numpy as np

from .._primitives.csmc.base import CSMCState
from . import _device
from .generic import get_kernel as get_base_kernel, IndependentFactory


def _get_parallel_kernel(M0, G0, Mt, Gt, N):
    fk = _device.describe_independent(M0, G0, Mt, Gt, None)

    def kernel(key, state, delta, noise=None):
        if isinstance(state.x, _device.CsmcChains):  # resident chains: in place, asynchronous; delta None = the chains' device delta
            if delta is not None:
                state.x.set_delta(delta)
            _device.pit_sweep_resident(fk, state.x, N, key)
            return CSMCState(x=state.x, updated=state.x.ancestors)
        x, anc = _device.pit_sweep(fk, state.x, N, key=key, noise=noise, delta=delta)
        out = CSMCState(x=x, updated=anc != 0)
        out.ancestors = anc
        return out

    def init(x):
        T = np.shape(x)[-2]
        return CSMCState(x=x, updated=np.zeros((T,), bool))  # independent.py:113-116

    return init, kernel


def get_kernel(M0, G0, Mt, Gt, N, backward=False, Pt=None, gradient=False, parallel=False):
    if gradient:
        raise NotImplementedError("gradient-informed proposals (independent.py:62-63) need autodiff of the model: out of scope")
    if parallel:
        return _get_parallel_kernel(M0, G0, Mt, Gt, N)
    if backward and Pt is None:
        Pt = Mt
    return get_base_kernel(IndependentFactory(M0, G0, Mt, Gt, Pt), N, backward, Pt)

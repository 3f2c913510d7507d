"""Auxiliary particle Gibbs with independent proposals N(u_t, delta_t/2 I) -- Finke & Thiery in the auxiliary
paradigm (reference: aux_samplers/csmc/independent.py).

get_kernel(M0, G0, Mt, Gt, N, backward=False, Pt=None, gradient=False, parallel=False) -> (init, kernel).
parallel=False: the classical sequential sweep (independent.py:57-75, auxssm_csmc_sweep); parallel=True: the parallel-in-time cSMC
(conditional dSMC, independent.py:78-118 on _primitives/csmc/pit, auxssm_csmc_pit_sweep: log2(T) stitching launches instead of T
sequential steps; `backward` / `Pt` are unused there, as in the reference).

gradient=True with parallel=True: the proposals are shifted the same way and every leaf carries the importance weight qt.logpdf - mt.logpdf
(independent.py:81-84, pit/csmc.py:83-88), per particle; time-varying dynamics are read row by row in both sweeps.
gradient=True (classical sweep): proposals N(u_t + delta_t/2 grad_t, delta_t/2 I) with grad the gradient at u of the model's joint
log-density (independent.py:121-134), which the reference gets from jax.grad and the device kernel evaluates in closed form for the
model family (csrc/csmc.hip::k_csmc_grad).  gradient=True follows the reference to the letter: the importance correction of the shifted
proposal enters the weights at t = 0 (GradientAuxiliaryG0, :173-190) while for t >= 1 GradientAuxiliaryGt sums it over ALL particles
(jnp.sum without an axis, :265-266), i.e. adds a constant that cancels -- no correction.  gradient="exact" applies the per-particle
correction at every step (the weights the construction intends; AUXSSM_GRAD_EXACT)."""
from dataclasses import dataclass
from typing import Any, Optional

import numpy as np

from .._primitives.csmc.base import CSMCState, Distribution, UnivariatePotential, Potential, Dynamics
from . import _device
from .generic import get_kernel as get_base_kernel, IndependentFactory


# The pieces the reference's parallel kernel hands to _primitives/csmc/pit.get_kernel (independent.py:78-118, classes :164-169, :202-225, :239-248): plain records
# here -- the device kernels evaluate them in closed form, so they carry the model components and the auxiliary variables, not Python densities.
@dataclass
class AuxiliaryMtDistribution(Distribution):
    """proposals N(u_t [+ delta_t / 2 grad_t], delta_t / 2 I): params = (u (T, d), sqrt(delta / 2) scalar or (T,), grad_pi or None)   (independent.py:202-225).
    A non-None third entry switches the gradient-informed proposals on; the device evaluates the gradient of the model's log-density at u itself
    (csrc/csmc.hip::k_csmc_grad), the array's values are not read."""
    params: Any = None


@dataclass
class AuxiliaryG0(UnivariatePotential):
    """G0(x) + M0.logpdf(x)   (independent.py:164-169)"""
    M0: Any = None
    G0: Any = None


@dataclass
class AuxiliaryGt(Potential):
    """Gt(x_t, x_{t-1}) + Mt.logpdf(x_t | x_{t-1})   (independent.py:239-248)"""
    Mt: Any = None
    Gt: Any = None


def _get_parallel_kernel(M0, G0, Mt, Gt, N, gmode=0):
    fk = _device.describe_independent(M0, G0, Mt, Gt, None, gmode)

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
    from .. import _lib
    if gradient not in (False, True, "exact", "reference"):
        raise ValueError("gradient must be False, True (the reference's weights) or 'exact'")
    gmode = _lib.GRAD_NONE if not gradient else (_lib.GRAD_EXACT if gradient == "exact" else _lib.GRAD_REFERENCE)
    if parallel:
        # gradient=True here is independent.py:81-84: proposals mt = N(u + delta/2 grad, delta/2 I) weighted by qt.logpdf - mt.logpdf PER PARTICLE
        # (pit/csmc.py:83-88) -- the parallel kernel has no summed variant, so True and "exact" are the same sampler
        return _get_parallel_kernel(M0, G0, Mt, Gt, N, gmode)
    if backward and Pt is None:
        Pt = Mt
    return get_base_kernel(IndependentFactory(M0, G0, Mt, Gt, Pt, gmode), N, backward, Pt)

"""Parallel-in-time cSMC (reference: aux_samplers/_primitives/csmc/pit/{csmc,operator,dc_map}.py, `get_kernel(Mt, G0, Gt, N, Qt)` :17-66).

The reference's one caller of this kernel is the independent auxiliary kernel with `parallel=True` (csmc/independent.py:78-118), which builds
`mt = AuxiliaryMtDistribution(params=(u, scale, grad))`, `g0 = AuxiliaryG0(M0, G0)`, `gt = AuxiliaryGt(Mt, Gt)` (and `qt` for gradient proposals) and calls
`get_pit_kernel(mt, g0, gt, N, qt)`.  `get_kernel` here takes exactly those records (aux_ssm_samplers_amd.csmc.independent) for the closed Feynman-Kac family of
`aux_ssm_samplers_amd.csmc.models` and runs `auxssm_csmc_pit_sweep` (csrc/pit.hip).  Arbitrary per-time-step proposal objects are Python closures a HIP kernel
cannot evaluate, and there is no CPU fallback: they raise NotImplementedError with a pointer to the closed family."""
import numpy as np

from ..base import CSMCState

_MSG = ("_primitives.csmc.pit.get_kernel runs on the device for the records the reference's own caller builds (csmc/independent.py:78-118): "
        "Mt = AuxiliaryMtDistribution(params=(u, scale, grad)), G0 = AuxiliaryG0(M0, G0), Gt = AuxiliaryGt(Mt, Gt) over the model family of "
        "aux_ssm_samplers_amd.csmc.models.  Arbitrary Python proposal / potential objects cannot be evaluated by the HIP kernels; "
        "aux_ssm_samplers_amd.csmc.get_independent_kernel(M0, G0, Mt, Gt, N, parallel=True) is the usual entry.")


def get_kernel(Mt, G0, Gt, N, Qt=None):
    """-> (init, kernel), kernel(key, state) as pit/csmc.py:52-64.  Gradient proposals: Mt.params[2] not None together with Qt (the plain proposals the weights
    are corrected towards, independent.py:81-84); either one without the other is not a sampler the reference builds and raises."""
    from ....csmc import _device
    from ....csmc.independent import AuxiliaryMtDistribution, AuxiliaryG0, AuxiliaryGt
    from .... import _lib, random as _random
    if not (isinstance(Mt, AuxiliaryMtDistribution) and isinstance(G0, AuxiliaryG0) and isinstance(Gt, AuxiliaryGt)) or \
            not (Qt is None or isinstance(Qt, AuxiliaryMtDistribution)):
        raise NotImplementedError(_MSG)
    u, scale, grad = Mt.params
    if (grad is not None) != (Qt is not None):
        raise NotImplementedError("gradient proposals need both Mt.params[2] and Qt (independent.py:81-84); " + _MSG)
    gmode = _lib.GRAD_EXACT if grad is not None else _lib.GRAD_NONE
    fk = _device.describe_independent(G0.M0, G0.G0, Gt.Mt, Gt.Gt, None, gmode)
    u = np.asarray(u)

    def kernel(key, state):
        x = np.asarray(state.x)
        if x.shape != u.shape or x.ndim != 2:
            raise ValueError(f"state {x.shape} and auxiliary variables {u.shape} must both be (T, d)")
        T, d = x.shape
        dtype = np.dtype(np.float32) if x.dtype == np.float32 else np.dtype(np.float64)
        s = np.broadcast_to(np.asarray(scale, np.float64), (T,))
        handle = _lib.default_handle()
        k_prop, k_res = _random.split(key, 2)
        # the device forms u = x + scale * eps_aux itself: hand it the eps_aux that reproduces the given u (to the rounding of one multiply-add)
        noise = dict(eps_aux=((u.astype(np.float64) - x.astype(np.float64)) / s[:, None])[None],
                     eps_prop=handle.rng_normal(k_prop, 2, (1, T, N, d), dtype).to_host(), u_res=handle.rng_uniform(k_res, 3, (1, T, N), dtype).to_host())
        xo, anc = _device.pit_sweep(fk, x, N, noise=noise, delta=2.0 * s * s, handle=handle)
        out = CSMCState(x=xo, updated=anc != 0)
        out.ancestors = anc
        return out

    def init(x_star):
        T = np.shape(x_star)[0]
        return CSMCState(x=x_star, updated=np.zeros((T,), bool))  # pit/csmc.py:60-63

    return init, kernel

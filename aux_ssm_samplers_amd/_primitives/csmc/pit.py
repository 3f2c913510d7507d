"""Parallel-in-time cSMC (reference: aux_samplers/_primitives/csmc/pit/{csmc,operator,dc_map}.py).

The reference's `get_kernel(Mt, G0, Gt, N, Qt=None)` takes arbitrary per-time-step proposal *objects* (`Mt.sample`, `Gt.__call__` as
Python callables), which cannot execute inside a HIP kernel.  The device path is the one caller the reference has for it,
`aux_samplers.csmc.get_independent_kernel(..., parallel=True)` (csmc/independent.py:78-118): proposals N(u_t, delta_t/2 I), stitching
potential log Mt + Gt from the closed model family -- `aux_ssm_samplers_amd.csmc.get_independent_kernel(M0, G0, Mt, Gt, N, parallel=True)`,
one auxssm_csmc_pit_sweep call per sweep (csrc/pit.hip).  This module keeps the import path and says so."""


def get_kernel(Mt, G0, Gt, N, Qt=None):
    raise NotImplementedError(
        "arbitrary Python proposal / potential objects cannot run inside the HIP kernels; use "
        "aux_ssm_samplers_amd.csmc.get_independent_kernel(M0, G0, Mt, Gt, N, parallel=True) (the reference's only caller of this kernel, "
        "csmc/independent.py:78-118) with model components from aux_ssm_samplers_amd.csmc.models. There is no CPU fallback.")

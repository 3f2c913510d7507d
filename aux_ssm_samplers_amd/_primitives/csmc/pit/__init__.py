"""Parallel-in-time cSMC (reference: aux_samplers/_primitives/csmc/pit/{csmc,operator,dc_map}.py, `get_kernel(Mt, G0, Gt, N, Qt)` :17-66).

The reference's one caller of this kernel is the independent auxiliary kernel with `parallel=True` (csmc/independent.py:78-118), and
that is what runs on the device here: `aux_ssm_samplers_amd.csmc.get_independent_kernel(M0, G0, Mt, Gt, N, parallel=True)` ->
`auxssm_csmc_pit_sweep` (csrc/pit.hip) for the closed Feynman-Kac family of `aux_ssm_samplers_amd.csmc.models`.  Arbitrary per-time-step
proposal objects `Mt` / `Qt` are Python closures a HIP kernel cannot evaluate, and there is no CPU fallback."""

_MSG = ("_primitives.csmc.pit.get_kernel takes arbitrary per-time-step proposal objects (Mt, Qt), which the HIP kernels cannot evaluate. "
        "Use aux_ssm_samplers_amd.csmc.get_independent_kernel(M0, G0, Mt, Gt, N, parallel=True) -- the reference's own caller of this kernel "
        "(csmc/independent.py:78-118) -- which runs the parallel-in-time sweep on the device for the closed model family.")


def get_kernel(Mt, G0, Gt, N, Qt=None):
    raise NotImplementedError(_MSG)

"""reference: aux_samplers/_primitives/kalman/dnc_sampling.py -- the divide-and-conquer LGSSM sampler, which the reference itself
declares "a proof-of-concept (and not efficient) feature kept mostly for pedagogical reasons" and tells callers to replace by
`sampling.sampling(..., parallel=True)` (:38-41).  Kept as an API mode: same signature, same warning, same error for batched input,
and the draw comes from the parallel pathwise sampler it points to (same distribution: test_sampling.py:23-68 checks `dnc` against the
same RTS-smoother moments as the other two modes)."""
import warnings

import numpy as np

from .sampling import sampling as _parallel_sampling


def sampling(key, ms, Ps, lgssm, eps=None, handle=None):
    warnings.warn("`dnc_sampling.sampling` is a proof-of-concept (and not efficient) feature kept mostly for pedagogical reasons."
                  "Use `sampling.sampling` with the argument `parallel=True` instead.", UserWarning)
    if np.ndim(ms) > 2:
        raise ValueError("Batched sampling is not supported for this function. Use `sampling.sampling` instead.")
    return _parallel_sampling(key, ms, Ps, lgssm, True, eps=eps, handle=handle)

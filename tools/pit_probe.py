#!/usr/bin/env python3
"""One configuration of the parallel-in-time cSMC sweep for rocprofv3: python3 tools/pit_probe.py N chains [T] [reps]  (C3's SV model, fp32)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from aux_ssm_samplers_amd import _lib, random as R  # noqa: E402
from aux_ssm_samplers_amd.csmc import CsmcChains, CSMCState, get_independent_kernel, GaussianInit, LinearGaussianDynamics, SVPotential  # noqa: E402

N, chains = int(sys.argv[1]), int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
h = _lib.default_handle()
phi, q, xsv, ysv = bench.sv_data(T, 0)
M0 = GaussianInit(m0=[0.0], P0=[[q]])
Mt = LinearGaussianDynamics(F=[[phi]], b=[0.0], Q=[[q]])
init, k = get_independent_kernel(M0, SVPotential(y=ysv[0]), Mt, SVPotential(params=ysv[1:]), N, parallel=True)
cc = CsmcChains(h, np.repeat(xsv.reshape(1, T, 1), chains, axis=0).astype(np.float32), delta=0.5)
st = CSMCState(x=cc, updated=None)
keys = R.split(R.PRNGKey(3), reps + 1)
k(keys[0], st, None)
h.sync()
h.prof_enable(_lib.K_PIT_STITCH, reps)
t0 = time.perf_counter()
for i in range(reps):
    k(keys[1 + i], st, None)
h.sync()
el = (time.perf_counter() - t0) / reps
n, ms = h.prof_read()
print(f"N={N} chains={chains} T={T}: {el * 1e3:.3f} ms per sweep (stitch levels {ms / max(n, 1):.3f} ms), updated {(cc.ancestors.to_host() != 0).mean():.3f}")

"""Kernel-level probe of the wide-state path at SURVEY config C5 (d = p = 64, fp32, T = 8192): one filter + sampler +
joint log-density per chain through the primitives API; run under rocprofv3 --kernel-trace --stats."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from aux_ssm_samplers_amd.workloads import c5_model
import aux_ssm_samplers_amd._primitives.kalman as P

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = 64
u, lg64, x = c5_model(T, d)
lg = P.LGSSM(*[np.ascontiguousarray(a, np.float32) for a in lg64])
u = u.astype(np.float32)
for rep in range(2):
    t0 = time.time()
    ms, Ps, ell = P.filtering(u, lg, True)
    t1 = time.time()
    xs = P.sampling(None, ms, Ps, lg, True, eps=np.random.default_rng(0).standard_normal((T, d)).astype(np.float32))
    t2 = time.time()
    lp = P.posterior_logpdf(u, xs, ell, lg)
    t3 = time.time()
    print(f"rep {rep}: filter {t1-t0:.3f}s sample {t2-t1:.3f}s logpdf {t3-t2:.3f}s ell={ell:.3f} lp={lp:.3f}", flush=True)

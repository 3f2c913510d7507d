"""Throughput probe of the wide-state path at config C5 (d = p = 64, fp32, T = 8192): B independent sequences in one launch through
the primitives' batch axis (model parameters broadcast, observations per sequence).  Prints kernel-group times from the library's
HIP-event hooks."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from aux_ssm_samplers_amd.workloads import c5_model
from aux_ssm_samplers_amd import _lib
import aux_ssm_samplers_amd._primitives.kalman as P

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = 64
u, lg64, x = c5_model(T, d)
bt = np.broadcast_to
m0, P0, Fs, Qs, bs, Hs, Rs, cs = [np.asarray(a, np.float32) for a in lg64]
F, Q, b_, H, Rm, c_ = Fs[0], Qs[0], bs[0], Hs[0], Rs[0], cs[0]
blg = P.LGSSM(bt(m0, (B, d)), bt(P0, (B, d, d)), bt(F, (T - 1, B, d, d)), bt(Q, (T - 1, B, d, d)), bt(b_, (T - 1, B, d)),
              bt(H, (T, B, d, d)), bt(Rm, (T, B, d, d)), bt(c_, (T, B, d)))
rng = np.random.default_rng(0)
bu = (u[:, None, :] + 0.1 * rng.standard_normal((T, B, d))).astype(np.float32)
h = _lib.default_handle()
tot = 0.0
for kid, name in ((_lib.K_FILTER_INIT, "filter init"), (_lib.K_FILTER_SCAN, "filter scan (incl. log-likelihood)")):
    for rep in range(2):
        h.prof_enable(kid, 4)
        ms, Ps, ell = P.filtering(bu, blg, True)
        n, t = h.prof_read()
        h.prof_disable()
    print(f"B={B} {name}: {t:.2f} ms ({t / B:.2f} ms per sequence)", flush=True)
    tot += t
eps = rng.standard_normal((T, B, d)).astype(np.float32)
for kid, name in ((_lib.K_SAMPLE_INIT, "sampler init"), (_lib.K_SAMPLE_SCAN, "sampler scan")):
    for rep in range(2):
        h.prof_enable(kid, 4)
        xs = P.sampling(None, ms, Ps, blg, True, eps=eps)
        n, t = h.prof_read()
        h.prof_disable()
    print(f"B={B} {name}: {t:.2f} ms ({t / B:.2f} ms per sequence)", flush=True)
    tot += t
print(f"B={B}: filter + sampler kernels {tot:.1f} ms -> {B / tot * 1e3:.1f} sequence-passes/s")

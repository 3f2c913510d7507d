"""Randomised equivalence + timing check of AUXSSM_OPT_OVERLAP_MODEL_STAGE: 80 keyed sweeps with changing step sizes, two resident states taking turns,
data replaced on the device between sweeps, general-path sweeps in between -- overlap on vs off must agree bit for bit.  GPU box: python tools/stress_overlap.py [T]"""
import numpy as np, sys, time
sys.path.insert(0, '.')
from aux_ssm_samplers_amd import _lib, random as R
from aux_ssm_samplers_amd.kalman import get_kernel, DeviceChains, KalmanSampler, LGConcatModel
from aux_ssm_samplers_amd.workloads import lg_model
h = _lib.default_handle()
T, d, C, NS = (int(sys.argv[1]) if len(sys.argv) > 1 else 16384), 2, 256, 80
dtype = np.float64
m = lg_model(T, d, dtype=dtype)
full = lambda a, n: np.ascontiguousarray(np.broadcast_to(a, (n,) + a.shape))
rng = np.random.default_rng(3)
ys = [m["y"]] + [(m["y"] + 0.5 * rng.standard_normal(m["y"].shape)).astype(dtype) for _ in range(3)]
x0 = rng.standard_normal((2, C, T, d)).astype(dtype) * 0.3
plan = [(float(rng.choice([0.2, 0.4, 0.6])), int(rng.integers(0, 2)), int(rng.integers(0, 12))) for _ in range(NS)]
if len(sys.argv) > 2 and sys.argv[2] == 'plain':  # no foreign calls, one state: the steady state the bench times
    plan = [(0.4, 0, 5) for _ in range(NS)]
def run(overlap):
    h.set_option(_lib.OPT_OVERLAP_MODEL_STAGE, overlap)
    model = LGConcatModel(m["m0"], m["P0"], full(m["F"], T - 1), full(m["Q"], T - 1), full(m["b"], T - 1), full(m["Hobs"], T), full(m["Robs"], T), full(m["cobs"], T), m["y"])
    init, kernel = get_kernel(model.dynamics_factory, model.observations_factory, model.log_likelihood_fn, True)
    chains = [DeviceChains(h, x0[0], chain_minor=True), DeviceChains(h, x0[1], chain_minor=True)]
    _, ybuf, _ = model.device(h, dtype)
    ydev = []
    for y in ys:
        a = _lib.DeviceArray(h, ybuf.shape, ybuf.dtype); a.copy_from_host(np.ascontiguousarray(y, dtype=dtype).reshape(ybuf.shape)); ydev.append(a)
    junk = h.empty((1 << 26,), np.float32) if hasattr(h, 'empty') else None
    t0 = time.time()
    for i, (dl, which, ev) in enumerate(plan):
        if ev == 0:   # new data behind other queued work
            ybuf.copy_from(ydev[i % 4])
        elif ev == 1:
            h.set_option(_lib.OPT_SHARE_MODEL, 0)
        kernel(R.PRNGKey(5000 + i), KalmanSampler(x=chains[which], updated=None), dl)
        if ev == 1:
            h.set_option(_lib.OPT_SHARE_MODEL, 1)
    h.sync() if hasattr(h, 'sync') else None
    el = time.time() - t0
    out = (chains[0].to_host(), chains[1].to_host(), chains[0].logs.to_host(), chains[1].logs.to_host())
    return out, el
run(1); run(0)  # (first runs: slabs, workspaces)
a, ta = run(1)
b, tb = run(0)
h.set_option(_lib.OPT_OVERLAP_MODEL_STAGE, 1)
ok = all(np.array_equal(p, q) for p, q in zip(a, b))
print("bitwise equal:", ok, " overlap %.3f s, single stream %.3f s for %d sweeps" % (ta, tb, NS))
assert ok

#!/usr/bin/env python3
"""Where does the acceptance rate of the C4 Lorenz auxiliary Kalman sweep come from (bench.py prints 0.078 at delta = 1e-4, T = 16384, fp32)?
Same explicit noise through the fp32 and the fp64 device sweep, chain by chain and sweep by sweep (the fp32 state is reset to the fp64 chain's before every
sweep, so one flipped decision does not fork the comparison), over a scan of step sizes: acceptance and the five log terms of both precisions.
GPU box: python tools/c4_accept_probe.py [T] [chains] [sweeps]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aux_ssm_samplers_amd import _lib  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import _get_device_kernel  # noqa: E402
from aux_ssm_samplers_amd.kalman.generic import DeviceChains, KalmanSampler  # noqa: E402
from aux_ssm_samplers_amd.workloads import lorenz_kalman_setup  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    h = _lib.default_handle()
    model, xtrue = lorenz_kalman_setup(T, every=80, dt=1.25e-4)
    for pol, delta in [(p, d) for p in ("reference", "masked") for d in (1e-5, 1e-4, 1e-3)]:
        init, kernel = _get_device_kernel(model, True, nan_policy=pol)
        rng = np.random.default_rng(11)
        x0 = np.repeat(xtrue[None], Cn, axis=0)
        c64 = DeviceChains(h, x0.astype(np.float64))
        c32 = DeviceChains(h, x0.astype(np.float32))
        acc64 = acc32 = agree = 0
        dmax = 0.0
        la = []
        dcol = np.zeros(5)
        for s in range(S):
            nz = dict(eps_aux=rng.standard_normal((Cn, T, 3)), eps_samp=rng.standard_normal((Cn, T, 3)), u_accept=rng.random(Cn))
            xs = c64.to_host()
            c32.x.copy_from_host(c32._to_layout(xs.astype(np.float32)))
            kernel(None, KalmanSampler(x=c64, updated=None), delta, noise=nz)
            kernel(None, KalmanSampler(x=c32, updated=None), delta, noise=nz)
            a64, a32 = c64.accepted.to_host(), c32.accepted.to_host()
            l64, l32 = c64.logs.to_host(), c32.logs.to_host()
            acc64 += int(a64.sum())
            acc32 += int(a32.sum())
            agree += int((a64 == a32).sum())
            dmax = max(dmax, float(np.nanmax(np.abs(l64[:, 0] - l32[:, 0]))))
            dcol = np.maximum(dcol, np.nanmax(np.abs(l64 - l32), axis=0))
            la.append(l64[:, 0])
        la = np.concatenate(la)
        print(json.dumps(dict(nan_policy=pol, max_abs_diff_la_lpprop_lprev_ltprop_ltrev=[round(float(v), 4) for v in dcol], T=T, chains=Cn, sweeps=S, delta=delta, accept_fp64=round(acc64 / (S * Cn), 3), accept_fp32=round(acc32 / (S * Cn), 3),
                              decisions_agree=round(agree / (S * Cn), 3), max_abs_dlogalpha=round(dmax, 4), log_alpha_fp64_mean=round(float(np.mean(la)), 2),
                              log_alpha_fp64_sd=round(float(np.std(la)), 2), terms_scale=float(np.abs(l64[:, 1:]).max()))), flush=True)


if __name__ == "__main__":
    main()

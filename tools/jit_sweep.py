#!/usr/bin/env python
"""Saturating sweep of the compiled-callback kernels (round 6): the notebook funnel at 2^10 ... 2^20 chains, kernel time from the library's
HIP events, chain-steps/s and the share of the fp32 vector peak by the compiled graph's own operation count.  python tools/jit_sweep.py [rmhmc]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hamiltorch_amd as ht  # noqa: E402
from hamiltorch_amd import _abi, jit  # noqa: E402
from benchlib.workloads import FunnelHMC, FunnelRMHMC, funnel_ll_device  # noqa: E402

PEAK_TF = 157.3
rm = len(sys.argv) > 1 and sys.argv[1] == "rmhmc"
W = FunnelRMHMC if rm else FunnelHMC
for lg in ((10, 12, 14, 16, 17, 18) if rm else (10, 12, 14, 16, 18, 20)):
    C = 1 << lg
    T = 2 if rm else max(4, min(200, (1 << 24) // C))
    w = W(torch.device("cuda"), C, T, 0)
    w._steps_done = 1
    w.step(0); torch.cuda.synchronize()
    _abi.set_tuning("profile", 1)
    for k in range(3):
        w.step(1 + k)
    ms, n = _abi.profile_collect()
    _abi.set_tuning("profile", 0)
    ms /= 3                               # per step (a step may be several launches: the pre-drawn records are capped per launch)
    rate = C * T * w.L / (ms * 1e-3)
    flops = w._graph_flops()
    print("%-12s C=%8d T=%4d  kernel %9.3f ms  %.3e chain-steps/s  %6.0f flop per step -> %.4f of the fp32 vector peak  waves/SIMD %.2f  %s"
          % (w.key, C, T, ms, rate, flops, flops * rate / 1e12 / PEAK_TF, (C / 64) / 1024, _abi.last_route()), flush=True)
    del w
    torch.cuda.empty_cache()

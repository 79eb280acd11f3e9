#!/usr/bin/env python
"""Rate of the compiled-callback HMC route on the notebook funnel (round 6): chain-steps/s through hamiltorch_amd.sample() at a
list of chain counts, compiled (one launch per call) against the torch-evaluated callback path, with the one-off trace / compile
times.  Usage: python tools/jit_rate.py [chains ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hamiltorch_amd as ht  # noqa: E402
from hamiltorch_amd import _abi, jit  # noqa: E402
from benchlib.workloads import funnel_ll_device, funnel_ll_notebook  # noqa: E402


def rate(fn, C, T, L=25, eps=0.2, reps=3):
    th0 = torch.ones(C, 11, device="cuda"); th0[:, 0] = 0.0
    kw = dict(num_samples=T, num_steps_per_sample=L, step_size=eps, burn=-1, verbose=False)
    t0 = time.perf_counter()
    ht.sample(fn, th0, seed=1, **kw); torch.cuda.synchronize()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for r in range(reps):
        ht.sample(fn, th0, seed=2 + r, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return C * T * L / dt, dt, first, _abi.last_route()


def main():
    chains = [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384, 65536, 262144]
    for fn in (funnel_ll_device, funnel_ll_notebook):
        for C in chains:
            T = 50 if C <= 65536 else 10
            v, dt, first, rt = rate(fn, C, T)
            print("%-20s C=%7d T=%3d  %.3e chain-steps/s  %.3f ms/call  first call %.2f s  %s" % (fn.__name__, C, T, v, dt * 1e3, first, rt))
    print("jit stats", jit.stats, jit.runtime.stats)
    os.environ["HAMILTORCH_AMD_JIT"] = "0"
    v, dt, first, rt = rate(funnel_ll_device, 1024, 50)
    print("callback path        C=   1024 T= 50  %.3e chain-steps/s  %.3f ms/call  %s" % (v, dt * 1e3, rt))


if __name__ == "__main__":
    main()

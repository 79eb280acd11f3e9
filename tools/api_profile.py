#!/usr/bin/env python
"""Host-side cost of hamiltorch_amd.sample() at BASELINE config 2 (1024 chains, 1000 trajectories, L = 25): wall time per call
pipelined (no synchronisation between calls, as bench.py's headline bracket) and synchronised, and a cProfile of the
Python side.  Usage: python tools/api_profile.py [calls]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hamiltorch_amd as ht  # noqa: E402

SIGMA3 = [[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    tgt = ht.GaussianTarget(torch.zeros(3, device=dev), covariance=torch.tensor(SIGMA3, device=dev))
    th0 = 0.1 * torch.randn(1024, 3, device=dev)

    def call(k):
        return ht.sample(tgt, th0, num_samples=1000, num_steps_per_sample=25, step_size=0.3, burn=-1, verbose=False, seed=k)
    for k in range(5):
        call(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        out = call(k)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("pipelined: %.1f us host per call, %.1f us per call incl. the final synchronize" % (t_host / n * 1e6, t_all / n * 1e6))
    ts = []
    for k in range(20):
        t0 = time.perf_counter(); out = call(k); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("synchronised single call: median %.1f us, min %.1f us" % (ts[len(ts) // 2] * 1e6, ts[0] * 1e6))
    t0 = time.perf_counter(); s = torch.stack(out); torch.cuda.synchronize(); print("torch.stack(out): %.1f us" % ((time.perf_counter() - t0) * 1e6))
    t0 = time.perf_counter(); rows = list(out); print("materialise: %.1f us (%d rows)" % ((time.perf_counter() - t0) * 1e6, len(rows)))
    pr = cProfile.Profile()
    pr.enable()
    for k in range(n):
        out = call(k)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()

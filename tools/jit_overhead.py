#!/usr/bin/env python
"""Where a compiled-callback sample() call spends its time (round 6): the kernel alone (HIP events inside the library), the call
with / without the Gaussian probe and the run-time check, and a cProfile of the host side.  python tools/jit_overhead.py [chains]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hamiltorch_amd as ht  # noqa: E402
from hamiltorch_amd import _abi  # noqa: E402
from benchlib.workloads import funnel_ll_device  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T, L = 50, 25
th0 = torch.ones(C, 11, device="cuda"); th0[:, 0] = 0.0
kw = dict(num_samples=T, num_steps_per_sample=L, step_size=0.2, burn=-1, verbose=False)


def wall(reps=10):
    ht.sample(funnel_ll_device, th0, seed=1, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        ht.sample(funnel_ll_device, th0, seed=2 + r, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("chains", C, "trajectories", T, "L", L)
print("sample() wall per call: %.3f ms" % wall())
_abi.set_tuning("profile", 1)
for r in range(5):
    ht.sample(funnel_ll_device, th0, seed=20 + r, **kw)
ms, n = _abi.profile_collect()
_abi.set_tuning("profile", 0)
print("kernel alone: %.4f ms per launch (%d launches) = %.3e chain-steps/s; %.0f cycles per leapfrog step at 2.4 GHz"
      % (ms / n, n, C * T * L / (ms / n * 1e-3), ms / n * 1e-3 * 2.4e9 / (T * L)))
for env in ({"HAMILTORCH_AMD_JIT_VERIFY": "0"}, {"HAMILTORCH_AMD_PROBE": "0"}, {"HAMILTORCH_AMD_JIT_VERIFY": "0", "HAMILTORCH_AMD_PROBE": "0"}):
    os.environ.update(env)
    print(env, "%.3f ms" % wall())
    for k in env:
        del os.environ[k]
pr = cProfile.Profile()
pr.enable()
for r in range(10):
    ht.sample(funnel_ll_device, th0, seed=40 + r, **kw)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

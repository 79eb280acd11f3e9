"""A/B of the explicit-RMHMC trajectory kernels on cfg3's target (D = 100, L = 10, jitter 1e-3) in ONE process: every argument is
`<chains>:<key=value,...or ->`; prints steps/s (whole call, HIP events on the current stream), the kernel-only time of the
profiled launches, the route the library reports and the acceptance rate.  Run through gpurun:

    python tools/ab_rmhmc.py 256:- 256:rmhmc_uv_co=1 1024:rmhmc_uv_co=1,rmhmc_uv=2
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hamiltorch_amd as ht            # noqa: E402
from hamiltorch_amd import _abi        # noqa: E402
if os.environ.get("HTA_LIB"):          # a developer build of the library (tools/uv_ablate.sh)
    _abi.LIB_PATH = os.path.join(ROOT, os.environ["HTA_LIB"])


def main():
    dev = torch.device("cuda:0")
    D, L, eps, omega, alpha, jitter = 100, int(os.environ.get("AB_L", "10")), 0.1, 10.0, 1e6, 1e-3
    g = torch.Generator().manual_seed(0)
    Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
    P = (Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T
    P = 0.5 * (P + P.T)
    tgt = ht.GaussianTarget(torch.zeros(D, device=dev), precision=P.float().to(dev), normalized=False)
    reps = int(os.environ.get("AB_REPS", "3"))
    ref = {}
    for combo in sys.argv[1:]:
        cs, tun = combo.split(":", 1)
        C = int(cs)
        T = 200 if C <= 512 else (100 if C <= 2048 else 50)
        _abi.reset_tuning()
        if tun != "-":
            for kv in tun.split(","):
                k, v = kv.split("=")
                _abi.set_tuning(k, int(v))
        th0 = (0.1 * torch.randn(C, D, generator=torch.Generator().manual_seed(C))).to(dev)
        cur = th0.clone()
        rej = torch.zeros(C, dtype=torch.int32, device=dev)
        samples = torch.empty(T + 1, C, D, device=dev)
        ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev)
        _abi.rmhmc_gaussian_prepare(cur, tgt.precision, tgt.mean, _abi.METRIC_SOFTABS, alpha, jitter, C, ws)

        def call(k):
            _abi.rmhmc_gaussian_sample(cur, th0, tgt.precision, tgt.mean, tgt.log_norm, _abi.METRIC_SOFTABS, alpha, jitter, L, eps, omega,
                                       T, 0, -1, 100 + k, 0, samples, rej, ws)
        call(0)
        torch.cuda.synchronize()
        first = samples[1:3].cpu().numpy().copy()
        route = _abi.last_route()
        best, kbest = 1e9, 1e9
        for r in range(reps):
            rej.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _abi.set_tuning("profile", 1)
            e0.record()
            call(1 + r)
            e1.record()
            torch.cuda.synchronize()
            kms, n = _abi.profile_collect()
            _abi.set_tuning("profile", 0)
            best = min(best, e0.elapsed_time(e1))
            kbest = min(kbest, kms)
        acc = 1.0 - float(rej.double().mean()) / T
        key = C
        dev_note = ""
        if key in ref:
            dev_note = "  max|d| vs first variant %.2e" % float(np.abs(first - ref[key]).max())
        else:
            ref[key] = first
        print("chains %5d %-44s %.3e steps/s (call %.3f ms, kernels %.3f ms) acc %.4f  %s%s" % (
            C, tun, C * T * L / (best * 1e-3), best, kbest, acc, route, dev_note), flush=True)
    _abi.reset_tuning()


if __name__ == "__main__":
    main()

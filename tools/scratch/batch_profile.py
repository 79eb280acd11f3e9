"""Developer check: cfg3-shaped run at 4096 chains, kernel times (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hamiltorch_amd as ht
from hamiltorch_amd import _abi
dev = torch.device("cuda:0")
D, L, T, C = 100, 10, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator().manual_seed(0)
Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
P = (Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T
tgt = ht.GaussianTarget(torch.zeros(D, device=dev), precision=(0.5 * (P + P.T)).float().to(dev), normalized=False)
th0 = (0.1 * torch.randn(C, D, generator=g)).to(dev)
ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev)
for ov in (1, 0):
    _abi.set_tuning("rmhmc_overlap", ov)
    cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev)
    def run():
        _abi.rmhmc_gaussian_sample(cur, th0, tgt.precision, tgt.mean, tgt.log_norm, _abi.METRIC_SOFTABS, 1e6, 1e-3, L, 0.1, 10.0,
                                   T, 0, -1, 1, 0, None, rej, ws)
    run(); torch.cuda.synchronize(); t0 = time.time(); run(); torch.cuda.synchronize(); dt = time.time() - t0
    print("C=%d overlap=%d: %.1f ms  %.3g steps/s" % (C, ov, dt * 1e3, C * T * L / dt))

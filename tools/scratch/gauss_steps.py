"""Developer check: time per launch of the cfg2 kernel vs L, for the eigenbasis and the direct route -> ns per step and per-trajectory overhead."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hamiltorch_amd as ht
from hamiltorch_amd import _abi
dev = torch.device("cuda:0")
C, T = 1024, 1000
cov = torch.tensor([[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]], device=dev)
tgt = ht.GaussianTarget(torch.zeros(3, device=dev), covariance=cov)
th0 = 0.1 * torch.randn(C, 3, device=dev); cur = th0.clone()
samples = torch.empty(T + 1, C, 3, device=dev); rej = torch.zeros(C, dtype=torch.int32, device=dev)
ws = torch.empty(_abi.gaussian_workspace_bytes(C, 3, T, 4), dtype=torch.uint8, device=dev)
for mode in (1, 0):
    _abi.set_tuning("gauss_eig", mode)
    res = {}
    for L in (0, 5, 25, 50, 100):
        def run():
            _abi.hmc_gaussian_sample(cur, th0, tgt.precision, tgt.mean, tgt.log_norm, 0, None, None, L, 0.05, T, 0, -1, 1, 0,
                                     samples, rej, workspace=ws)
        for _ in range(3): run()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        res[L] = e0.elapsed_time(e1) / 10 * 1e6 / T          # ns per trajectory
    print("gauss_eig=%d  ns/trajectory:" % mode, {k: round(v, 1) for k, v in res.items()},
          " ns/step (L 50->100): %.2f   overhead(L=0): %.1f" % ((res[100] - res[50]) / 50, res[0]))

"""Developer check: fused RMHMC, one vs two chains per workgroup, at several chain counts (cfg3 shape)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hamiltorch_amd as ht
from hamiltorch_amd import _abi
dev = torch.device("cuda:0")
D, L, T = 100, 10, 100
g = torch.Generator().manual_seed(0)
Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
P = (Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T
tgt = ht.GaussianTarget(torch.zeros(D, device=dev), precision=(0.5 * (P + P.T)).float().to(dev), normalized=False)
for C in (256, 512, 1024, 2048, 4096):
    th0 = (0.1 * torch.randn(C, D, generator=g)).to(dev)
    ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev)
    out = []
    for mode in (1, 3):
        _abi.set_tuning("rmhmc_fused", mode)
        cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev)
        def run():
            _abi.rmhmc_gaussian_sample(cur, th0, tgt.precision, tgt.mean, tgt.log_norm, _abi.METRIC_SOFTABS, 1e6, 1e-3, L, 0.1, 10.0,
                                       T, 0, -1, 1, 0, None, rej, ws)
        run(); torch.cuda.synchronize(); t0 = time.time(); run(); torch.cuda.synchronize(); dt = time.time() - t0
        out.append("%s %.1f ms %.3g steps/s" % ("one" if mode == 1 else "pair", dt * 1e3, C * T * L / dt))
    _abi.set_tuning("rmhmc_fused", 1)
    print("C=%5d: " % C + "   ".join(out))

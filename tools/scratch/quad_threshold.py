"""Developer check: chain-per-quad vs chain-per-lane eigenbasis kernel across chain counts (sets the dispatcher's threshold)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hamiltorch_amd as ht
from hamiltorch_amd import _abi
dev = torch.device("cuda:0")
cov = torch.tensor([[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]], device=dev)
tgt = ht.GaussianTarget(torch.zeros(3, device=dev), covariance=cov)
for C, T in ((1024, 1000), (4096, 1000), (16384, 500), (32768, 256), (65536, 256), (131072, 128), (262144, 64), (1048576, 16)):
    th0 = 0.1 * torch.randn(C, 3, device=dev); cur = th0.clone()
    samples = torch.empty(T + 1, C, 3, device=dev); rej = torch.zeros(C, dtype=torch.int32, device=dev)
    ws = torch.empty(_abi.gaussian_workspace_bytes(C, 3, T, 4), dtype=torch.uint8, device=dev)
    out = []
    for name, qmax in (("quad", 1 << 30), ("lane", 0)):
        _abi.set_tuning("quad_max_chains", qmax)
        def run():
            _abi.hmc_gaussian_sample(cur, th0, tgt.precision, tgt.mean, tgt.log_norm, 0, None, None, 25, 0.3, T, 0, -1, 1, 0,
                                     samples, rej, workspace=ws)
        for _ in range(2): run()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out.append("%s %.3f ms %.3g steps/s" % (name, ms, C * T * 25 / ms * 1e3))
    print("C=%8d T=%5d  " % (C, T) + "   ".join(out))

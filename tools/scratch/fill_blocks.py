"""Developer check: cfg2 call time vs grid cap of the pre-draw pass."""
import os, sys, torch
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
for blocks in (512, 1024, 2048, 4096, 8192, 16384):
    _abi.set_tuning("fill_blocks", blocks)
    def run():
        _abi.hmc_gaussian_sample(cur, th0, tgt.precision, tgt.mean, tgt.log_norm, 0, None, None, 25, 0.3, T, 0, -1, 1, 0,
                                 samples, rej, workspace=ws)
    for _ in range(5): run()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("fill_blocks %5d: %.1f us per call" % (blocks, e0.elapsed_time(e1) / 20 * 1e3))

"""Developer check: wave-per-chain Gaussian HMC, eigenbasis route vs direct kernel."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hamiltorch_amd as ht
from hamiltorch_amd import _abi
dev = torch.device("cuda:0")
for D, C, T, L in ((20, 1024, 200, 25), (100, 1024, 200, 25), (100, 4096, 100, 25), (128, 1024, 100, 10)):
    g = torch.Generator().manual_seed(0)
    Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
    P = (Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T
    tgt = ht.GaussianTarget(torch.zeros(D, device=dev), precision=(0.5 * (P + P.T)).float().to(dev), normalized=False)
    th0 = (0.1 * torch.randn(C, D, generator=g)).to(dev)
    res = []
    for mode in (1, 0):
        _abi.set_tuning("gauss_eig", mode)
        cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev)
        samples = torch.empty(T + 1, C, D, device=dev)
        ws = torch.empty(_abi.gaussian_workspace_bytes(C, D, T, 4), dtype=torch.uint8, device=dev)
        def run():
            _abi.hmc_gaussian_sample(cur, th0, tgt.precision, tgt.mean, tgt.log_norm, 0, None, None, L, 0.1, T, 0, -1, 1, 0,
                                     samples, rej, workspace=ws)
        run(); torch.cuda.synchronize(); t0 = time.time(); run(); torch.cuda.synchronize(); dt = time.time() - t0
        res.append("%s %.2f ms %.3g steps/s" % ("eig" if mode else "direct", dt * 1e3, C * T * L / dt))
    _abi.set_tuning("gauss_eig", 1)
    print("D=%d C=%d T=%d L=%d: " % (D, C, T, L) + "   ".join(res))

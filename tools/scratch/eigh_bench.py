"""Timing breakdown of metric_eval_kernel at cfg3 shape (256 systems, D=100)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hamiltorch_amd import _abi
if os.environ.get("HTA_LIB"):
    _abi.LIB_PATH = os.environ["HTA_LIB"]
dev = torch.device("cuda:0")
D, B = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator().manual_seed(0)
Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
P = (Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T
P = (0.5 * (P + P.T)).float().to(dev)
m = torch.randn(B, D, device=dev); x = torch.empty(B, D, device=dev)
V0 = torch.empty(1, D, D, device=dev); lam0 = torch.empty(1, D, device=dev)
_abi.metric_eval(P, 1, D, 1, P, 0, 1e6, V_out=V0, lamraw_out=lam0)
def timeit(label, **kw):
    for _ in range(2): _abi.metric_eval(P, B, D, 1, P, 0, 1e6, m=m, x_out=x, seed=1, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(5): _abi.metric_eval(P, B, D, 1, P, 0, 1e6, m=m, x_out=x, seed=1, draw=k, **kw)
    e.record(); torch.cuda.synchronize()
    print("%-44s %8.3f ms" % (label, s.elapsed_time(e) / 5))
for sw in (1, 2, 3, 4, 8, 16):
    timeit("cold  jitter max_sweeps=%d" % sw, jitter=1e-3, max_sweeps=sw)
for sw in (1, 2, 3):
    timeit("warm  jitter max_sweeps=%d" % sw, jitter=1e-3, max_sweeps=sw, V0=V0[0], lam0=lam0[0])
timeit("warm  jitter default", jitter=1e-3, V0=V0[0], lam0=lam0[0])
timeit("warm  no jitter (diagonal)", V0=V0[0], lam0=lam0[0])

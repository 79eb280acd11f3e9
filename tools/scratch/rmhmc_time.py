"""Developer timing of the fused explicit-RMHMC kernel (cfg3 shape): kernel ms per trajectory for a few (L, jitter)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hamiltorch_amd as ht
from hamiltorch_amd import _abi
if os.environ.get("TIMING_LIB"):
    _abi.LIB_PATH = os.environ["TIMING_LIB"]
for kv in filter(None, os.environ.get("HTA_TUNING", "").split(",")):
    k_, v_ = kv.split("="); _abi.set_tuning(k_, int(v_))
dev = torch.device("cuda:0")
_abi.set_tuning("rmhmc_fused", int(os.environ.get("FUSED", "1")))
D, C, T = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = torch.Generator().manual_seed(0)
Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
P = (Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T
P = 0.5 * (P + P.T)
tgt = ht.GaussianTarget(torch.zeros(D, device=dev), precision=P.float().to(dev), normalized=False)
th0 = (0.1 * torch.randn(C, D, generator=g)).to(dev)
ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev)
for L, jit in ((10, 1e-3),):
    cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(3):
        _abi.set_tuning("profile", 1)
        _abi.rmhmc_gaussian_sample(cur, th0, tgt.precision, tgt.mean, tgt.log_norm, _abi.METRIC_SOFTABS, 1e6, jit, L, 0.1, 10.0,
                                   T, 0, -1, rep, 0, None, rej, ws)
        torch.cuda.synchronize()
        ms, n = _abi.profile_collect(); _abi.set_tuning("profile", 0)
        best = min(best, ms)
    print("C=%d L=%2d jitter=%s: %.3f ms per trajectory (%d launches)" % (C, L, jit, best / T, n))

import ctypes
lib = ctypes.CDLL(_abi.LIB_PATH)
if hasattr(lib, "hta_rm_dbg_read"):
    buf = (ctypes.c_ulonglong * 16)()
    lib.hta_rm_dbg_read(buf)
    names = ["everything else", "-", "-", "refinements + tail | tracked: refresh products", "prologue (publish)", "barrier", "products", "element-wise after products"]
    tot = sum(buf)
    for k in range(8):
        print("   %-20s %12d cycles  %5.1f %%" % (names[k], buf[k], 100.0 * buf[k] / max(1, tot)))

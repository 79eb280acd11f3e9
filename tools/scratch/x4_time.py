"""Developer cycle attribution of rmhmc_mfma4x4_kernel (wave 0 of workgroup 0): needs the timing build
   (tools/scratch/x4_time.sh builds tools/scratch/_abl/libhta_timing.so with -DHTA_RM_TIMING=1)."""
import sys, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hamiltorch_amd import _abi
_abi.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "_abl", "libhta_timing.so")
import hamiltorch_amd as ht
dev = torch.device("cuda:0")
D, C, T, L = 100, int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 8, 10
for kv in filter(None, os.environ.get("HTA_TUNING", "").split(",")):
    k, v = kv.split("="); _abi.set_tuning(k, int(v))
g = torch.Generator().manual_seed(0)
Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
P = (Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T
P = 0.5 * (P + P.T)
tgt = ht.GaussianTarget(torch.zeros(D, device=dev), precision=P.float().to(dev), normalized=False)
th0 = (0.1 * torch.randn(C, D, generator=g)).to(dev)
ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev)
lib = ctypes.CDLL(_abi.LIB_PATH)
names = ["jitter (Philox pair)", "barrier wait", "products (MFMA loops + combine)", "element-wise + LDS publish", "rotation + publish",
         "hamiltonian (whole)", "accept / bookkeeping", "momentum load"]
for overlap in (1, 0):
    _abi.set_tuning("rmhmc_overlap", overlap)
    cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(3):
        _abi.set_tuning("profile", 1)
        _abi.rmhmc_gaussian_sample(cur, th0, tgt.precision, tgt.mean, tgt.log_norm, _abi.METRIC_SOFTABS, 1e6, 1e-3, L, 0.1, 10.0,
                                   T, 0, -1, rep, 0, None, rej, ws)
        torch.cuda.synchronize()
        ms, n = _abi.profile_collect(); _abi.set_tuning("profile", 0)
        best = min(best, ms)
    print("C=%d overlap=%d: %.3f ms per trajectory (trajectory kernel only, %d launches)" % (C, overlap, best / T, n))
    buf = (ctypes.c_ulonglong * 16)()
    lib.hta_rm_dbg_read(buf)
    tot = sum(buf)
    halfpairs = T * L * 2
    for k in range(8):
        print("   %-34s %12d ticks  %5.1f %%   %8.1f per half-step pair" % (names[k], buf[k], 100.0 * buf[k] / max(1, tot), buf[k] / halfpairs))
    print("   total %d ticks, %.1f per half-step pair" % (tot, tot / halfpairs))

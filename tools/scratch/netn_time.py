"""Developer cycle attribution of netn_hmc_kernel (workgroup 0 = chain 0): needs tools/scratch/netn_time.sh's build."""
import sys, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hamiltorch_amd import _abi
_abi.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "_abl", "libhta_netn_timing.so")
dev = torch.device("cuda:0")
names = ["publish", "inputs (global loads)", "forward layers", "loss", "gradient reductions", "delta propagation", "between passes", "-"]
lib = None
for dims, loss, N, M in (([1, 10, 10, 1], "regression", 400, 1), ([4, 3], "multi_class_linear_output", 150, 1), ([1, 10, 10, 1], "regression", 400, 4)):
    D = sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))
    C, T, L = 1024, 8, 10
    X = torch.randn(N, dims[0], device=dev)
    Y = (torch.randint(0, dims[-1], (N,), device=dev).float() if loss != "regression" else torch.randn(N, 1, device=dev)).contiguous()
    th = 0.1 * torch.randn(C, D, device=dev); th0 = th.clone()
    rej = torch.zeros(C, dtype=torch.int32, device=dev)
    taus = [1.0] * (2 * (len(dims) - 1))
    for rep in range(2):
        _abi.set_tuning("profile", 1)
        _abi.netn_hmc_sample(th, th0, dims, "relu", X, Y, M, N // M, taus, 1.0, float(M), _abi.MASS_NONE, None, None, L, 1e-3, T, 0, -1, rep, 0, None, rej, loss=loss)
        torch.cuda.synchronize()
        ms, n = _abi.profile_collect(); _abi.set_tuning("profile", 0)
    lib = lib or ctypes.CDLL(_abi.LIB_PATH)
    buf = (ctypes.c_ulonglong * 8)()
    lib.hta_netn_dbg_read(buf)
    tot = sum(buf)
    print("dims=%s loss=%s N=%d M=%d: %.3f ms per trajectory (kernel), %d ticks per trajectory in chain 0" % (dims, loss, N, M, ms / T, tot // T))
    for k in range(7):
        print("   %-24s %12d ticks  %5.1f %%" % (names[k], buf[k], 100.0 * buf[k] / max(1, tot)))

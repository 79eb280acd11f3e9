"""Developer cycle attribution of mlp3_mfma_kernel (wave 0 of workgroup 0 = chain 0): needs tools/scratch/m3_time.sh's build."""
import sys, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hamiltorch_amd import _abi
_abi.LIB_PATH = os.path.join(ROOT, "tools", "scratch", "_abl", "libhta_m3_timing%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "0"))
print("library:", _abi.LIB_PATH)
dev = torch.device("cuda:0")
names = ["(outside) -> top of chunk", "top barrier wait", "data staging (+ barrier)", "layer 1 (a1, both layouts)", "barrier A1 ready", "GEMM1 forward",
         "activations + f partials", "barrier f partials", "residuals + W2 staging", "barrier residuals", "delta2 + thin last layer", "GEMM3 dW2",
         "barrier delta2", "GEMM2 delta1", "thin first layer grads", "kick / block_sum", "between passes (drift, MH, draws)"]
for M, Nb, C in ((4, 100, 1024), (1, 400, 1024)):
    dims = [1, 100, 100, 1]
    D = 10401
    T, L = 1, 30
    X = torch.randn(400, 1, device=dev); Y = torch.randn(400, device=dev)
    th = (0.1 * torch.randn(C, D, device=dev)).contiguous(); th0 = th.clone()
    rej = torch.zeros(C, dtype=torch.int32, device=dev)
    for rep in range(2):
        _abi.set_tuning("profile", 1)
        _abi.netn_hmc_sample(th, th0, dims, "relu", X, Y, M, Nb, [1.0] * 6, 110.44, float(M), _abi.MASS_NONE, None, None, L, 5e-4, T, 0, -1, rep, 0,
                             None, rej, integrator=_abi.SPLIT_SYMMETRIC if M > 1 else 0)
        torch.cuda.synchronize()
        ms, n = _abi.profile_collect(); _abi.set_tuning("profile", 0)
    lib = ctypes.CDLL(_abi.LIB_PATH)
    buf = (ctypes.c_ulonglong * 20)()
    lib.hta_m3_dbg_read(buf)
    tot = sum(buf[:18])
    npass = (L * 2 * M if M > 1 else L + 1) + 2 * (4 if M == 1 else 4)
    print("M=%d Nb=%d C=%d: %.3f ms per launch, %d ticks in the last chain of workgroup 0 (%s), ~%d ticks per gradient pass" % (M, Nb, C, ms, tot, _abi.last_route(), tot // max(1, (L * 2 * M if M > 1 else (L + 1) * 4))))
    for k in range(17):
        print("   %-36s %12d ticks  %5.1f %%" % (names[k], buf[k], 100.0 * buf[k] / max(1, tot)))

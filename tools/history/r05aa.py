"""r05aa: hta_metric_eval on the run-time work-list instance at D = 512 / 1024: time per evaluation and the error against numpy's eigh."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from hamiltorch_amd import _abi
dev = torch.device("cuda", 0)
for dtype, D, B in ((torch.float32, 512, 2), (torch.float64, 512, 2), (torch.float32, 1024, 1)):
    rng = np.random.default_rng(D)
    M = rng.standard_normal((B, D, D)); Hs = ((M + M.transpose(0, 2, 1)) / 2 / np.sqrt(D)).astype(np.float32 if dtype == torch.float32 else np.float64)
    m = rng.standard_normal((B, D)).astype(Hs.dtype)
    x = torch.empty(B, D, device=dev, dtype=dtype); lam = torch.empty_like(x); ld = torch.empty(B, device=dev, dtype=dtype)
    H = torch.tensor(Hs, device=dev); mm = torch.tensor(m, device=dev)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _abi.metric_eval(x, B, D, _abi.METRIC_SOFTABS, H, D * D, 1.3, None, 0, 0, 0, 0, m=mm, x_out=x, lam_out=lam, logdet_out=ld)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    w, Q = np.linalg.eigh(Hs.astype(np.float64))
    lt = w / np.tanh(1.3 * w)
    want = np.einsum("bij,bj->bi", Q, np.einsum("bji,bj->bi", Q, m.astype(np.float64)) / lt)
    print(dtype, "D", D, "B", B, _abi.last_route(), "%.1f ms per call" % (dt * 1e3), "max |x - x_ref| %.2e" % np.abs(x.cpu().numpy() - want).max(),
          "max |lam~ - ref| %.2e" % np.abs(np.sort(lam.cpu().numpy(), 1) - np.sort(lt, 1)).max())

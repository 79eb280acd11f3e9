"""round 5, call c: why the vglobal instance of hta_metric_eval fails on a caller-provided slab (r05b) - zeroed / NaN-filled /
oversized workspaces, fp64 D = 100, 110 and fp32 D = 141."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hmc_oracle as O
from hamiltorch_amd import _abi
dev = torch.device("cuda:0")
for dtype, D in ((torch.float64, 100), (torch.float64, 110), (torch.float32, 141), (torch.float64, 128)):
    npd = np.float64 if dtype == torch.float64 else np.float32
    rng = np.random.default_rng(D)
    B = 5
    Hs = []
    for b in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((D, D))); lam = rng.uniform(0.5, 2.0, D); A = (Q * lam) @ Q.T; Hs.append(0.5 * (A + A.T))
    Hs = np.stack(Hs).astype(npd)
    G, lam, _ = O.softabs_metric(Hs.astype(np.float64), 1e6)
    t = torch.tensor(Hs, device=dev)
    need = _abi.metric_eval_workspace_bytes(B, D, t.element_size())
    for name, ws in (("auto", None), ("zeros", torch.zeros(need, dtype=torch.uint8, device=dev)),
                     ("nan", torch.full((need // 4,), float("nan"), device=dev).view(torch.uint8)),
                     ("big-zeros", torch.zeros(4 * need + 4096, dtype=torch.uint8, device=dev)),
                     ("offset", torch.zeros(need + 4096, dtype=torch.uint8, device=dev)[2048:])):
        lamd = torch.empty(B, D, dtype=dtype, device=dev); Gd = torch.empty(B, D, D, dtype=dtype, device=dev)
        res = []
        for rep in range(2):
            _abi.metric_eval(t, B, D, _abi.METRIC_SOFTABS, t, D * D, 1e6, lam_out=lamd, G_out=Gd, workspace=ws)
            torch.cuda.synchronize()
            el = np.abs(np.sort(lamd.cpu().numpy(), 1) - np.sort(lam, 1)).max(); eg = np.abs(Gd.cpu().numpy() - G).max()
            res.append("lam %.2e G %.2e" % (el, eg))
        print(dtype, D, "need", need, name, _abi.last_route(), res, flush=True)

"""round 5, call d: which outputs the fp64 vglobal instance leaves unwritten (r05c: lam_out right, G_out untouched)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hmc_oracle as O
from hamiltorch_amd import _abi
dev = torch.device("cuda:0")
CASES = ((torch.float64, 98), (torch.float64, 100), (torch.float32, 141))
if os.environ.get("R05D_ONLY100"):
    CASES = ((torch.float64, 100),)
for dtype, D in CASES:
    npd = np.float64 if dtype == torch.float64 else np.float32
    rng = np.random.default_rng(D)
    B = 2
    Hs = []
    for b in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((D, D))); lam = rng.uniform(0.5, 2.0, D); A = (Q * lam) @ Q.T; Hs.append(0.5 * (A + A.T))
    Hs = np.stack(Hs).astype(npd)
    m = rng.standard_normal((B, D)).astype(npd)
    G, lam, _ = O.softabs_metric(Hs.astype(np.float64), 1e6)
    x64 = np.linalg.solve(G, m.astype(np.float64)[..., None])[..., 0]
    t = torch.tensor(Hs, device=dev); mt = torch.tensor(m, device=dev)
    def S(*shape):
        return torch.full(shape, 7.0, dtype=dtype, device=dev)
    for names in (("lam_out",), ("G_out",), ("V_out",), ("lam_out", "G_out"), ("x_out",), ("L_out",), ("logdet_out", "quad_out"), ("G_out", "V_out", "L_out", "x_out", "lam_out")):
        outs = {"lam_out": S(B, D), "G_out": S(B, D, D), "V_out": S(B, D, D), "x_out": S(B, D), "L_out": S(B, D, D), "logdet_out": S(B), "quad_out": S(B)}
        kw = {k: outs[k] for k in names}
        if "x_out" in names or "quad_out" in names:
            kw["m"] = mt
        _abi.metric_eval(t, B, D, _abi.METRIC_SOFTABS, t, D * D, 1e6, **kw)
        torch.cuda.synchronize()
        rep = []
        for k in names:
            v = outs[k].cpu().numpy()
            untouched = float((v == 7.0).mean())
            if k == "G_out": err = np.abs(v - G).max()
            elif k == "lam_out": err = np.abs(np.sort(v, 1) - np.sort(lam, 1)).max()
            elif k == "x_out": err = np.abs(v - x64).max()
            elif k == "V_out": err = np.abs(np.einsum("bij,bik->bjk", v, v) - np.eye(D)).max()
            elif k == "L_out": err = np.abs(v @ np.swapaxes(v, 1, 2) - G).max()
            elif k == "logdet_out": err = np.abs(v - np.log(lam).sum(1)).max()
            else: err = np.abs(v - (m * x64).sum(1)).max()
            rep.append("%s err %.1e untouched %.2f" % (k, err, untouched))
        print(dtype, D, _abi.last_route(), " | ".join(rep), flush=True)

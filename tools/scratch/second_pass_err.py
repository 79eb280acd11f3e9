"""Developer check: errors of the matrix-core metric evaluation against the float64 oracle with the closed-form second pass (1) and
the three-product pass (0), and of the Jacobi kernel.   python tools/scratch/second_pass_err.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hamiltorch_amd as ht
from hamiltorch_amd import _abi
import test_gpu_rmhmc as T
O = T.O
for D, alpha, jitter in [(100, 1e6, 1e-3), (100, 1.3, 1e-3), (64, 2.0, 5e-4), (112, 1e6, 1e-3)]:
    rng = np.random.default_rng(D + 1)
    P = T.cfg3_target(ht, D, torch.float32, seed=7)[1].P.astype(np.float64)
    B, seed = 41, 123
    X = (0.3 * rng.standard_normal((B, D))).astype(np.float32)
    m = rng.standard_normal((B, D)).astype(np.float32)
    res = {}
    for name, second, mode in (("closed", 1, 1), ("full", 0, 1), ("jacobi", 1, 0)):
        _abi.set_tuning("metric_second", second)
        res[name] = T._warm_eval(ht, P, X, m, alpha, jitter, seed, mode)
    _abi.set_tuning("metric_second", 1)
    Hs = np.broadcast_to(P, (B, D, D)).astype(np.float64).copy()
    ju = O.philox_uniforms(seed, 3 + np.arange(B), 7, D, O.PURPOSE_JITTER, 2, dtype=np.float64)
    G, lam, _ = O.softabs_metric(Hs, alpha, jitter, ju)
    x64 = 0.5 * np.linalg.solve(G, m.astype(np.float64)[..., None])[..., 0]
    sx = np.abs(x64).max()
    print("D=%d alpha=%g jitter=%g  max|x|=%.3g" % (D, alpha, jitter, sx))
    for name in res:
        r = res[name]
        print("  %-7s x err max %.2e rms %.2e | lam err %.2e | logdet err %.2e | quad rel err %.2e" % (
            name, np.abs(r["x"] - x64).max() / sx, np.sqrt(((r["x"] - x64) ** 2).mean()) / sx,
            np.abs(np.sort(r["lam"], 1) - np.sort(lam, 1)).max(), np.abs(r["ld"] - np.log(lam).sum(1)).max(),
            (np.abs(r["q"] - 2 * (m * x64).sum(1)) / np.abs(2 * (m * x64).sum(1))).max()))
    print("  closed vs full: x %.2e" % (np.abs(res["closed"]["x"] - res["full"]["x"]).max() / sx))

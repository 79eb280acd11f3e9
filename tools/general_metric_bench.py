"""General-target soft-abs metric evaluations (per-chain Hessians): the cold Jacobi kernel against the per-chain warm start on the
matrix cores (HtaMetricArgs.v0_stride, ABI 7).  Target: log p = -1/2 w^T P w - sum_i log cosh(a_i . w) (log-concave, curvature
depends on w), D = 100, 256 chains.  (1) hta_metric_eval alone along a short explicit-RMHMC path: the Hessians at consecutive
evaluation points, each evaluation warm-started from the previous one's basis; (2) hamiltorch_amd.sample(RMHMC, EXPLICIT) end to
end with HAMILTORCH_AMD_WARM_METRIC = 0 / 1 (the callback's torch.func derivatives are the same in both)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hamiltorch_amd as ht
from hamiltorch_amd import _abi

dev = torch.device("cuda:0")
D, C = int(os.environ.get("GM_D", 100)), int(os.environ.get("GM_C", 256))
g = torch.Generator().manual_seed(0)
Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
P = ((Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T).float().to(dev)
A = (0.6 * torch.randn(2 * D, D, generator=g) / D ** 0.5).to(dev)


def logp(w):
    return -0.5 * (w @ (P @ w)) - torch.log(torch.cosh(A @ w)).sum()


def neg_hessian(th):          # closed form: P + A^T diag(sech^2(A w)) A
    s = 1.0 / torch.cosh(th @ A.T) ** 2
    return P[None] + torch.einsum("ki,ck,kj->cij", A, s, A)


th = 0.3 * torch.randn(C, D, generator=g).to(dev)
steps = [th + 0.05 * k * torch.randn(C, D, generator=g).to(dev) for k in range(9)]       # a path: consecutive evaluation points
Hs = [neg_hessian(t).contiguous() for t in steps]
m = torch.randn(C, D, generator=g).to(dev)
x = torch.empty(C, D, device=dev); M = torch.empty(C, D, D, device=dev)


def run(general, warm):
    _abi.set_tuning("metric_general", general)
    V = torch.eye(D, device=dev).repeat(C, 1, 1).contiguous()
    kw = dict(V0=V, v0_stride=D * D, V_out=V) if warm else {}
    ts = []
    for k, H in enumerate(Hs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _abi.metric_eval(th, C, D, _abi.METRIC_SOFTABS, H, D * D, 1e6, 1e-3, 1, 0, 0, k, m=m, x_out=x, dmetric_out=(M if k % 2 else None), **kw)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return ts, _abi.last_route(), x.clone()


cold, rc, xc = run(0, False)
warm, rw, xw = run(1, True)
print("D=%d, %d chains, soft-abs metric evaluations along a path of 9 points (odd ones with the derivative matrix)" % (D, C))
print("  cold  (%s): %s ms" % (rc, " ".join("%.3f" % t for t in cold)))
print("  warm  (%s): %s ms   (first call starts from the identity)" % (rw, " ".join("%.3f" % t for t in warm)))
print("  mean of calls 2..9: cold %.3f ms, warm %.3f ms (%.1fx);  max |x_cold - x_warm| = %.2e"
      % (sum(cold[1:]) / 8, sum(warm[1:]) / 8, sum(cold[1:]) / sum(warm[1:]), float((xc - xw).abs().max())))
_abi.reset_tuning()

for flag in ("0", "1"):
    os.environ["HAMILTORCH_AMD_WARM_METRIC"] = flag
    kw = dict(num_samples=3, num_steps_per_sample=3, step_size=0.05, jitter=1e-3, softabs_const=1e6, explicit_binding_const=10.0,
              sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, verbose=False, seed=3)
    ht.sample(logp, th.clone(), **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = ht.sample(logp, th.clone(), **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("  sample(RMHMC, EXPLICIT) 2 trajectories x 3 steps, HAMILTORCH_AMD_WARM_METRIC=%s: %.1f ms  (%.3e chain-steps/s)"
          % (flag, dt * 1e3, C * 2 * 3 / dt))

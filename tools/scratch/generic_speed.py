"""Developer check: generic-callback HMC / explicit RMHMC throughput with and without HIP-graph replay of the callback."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hamiltorch_amd as ht
dev = torch.device("cuda:0")
D, C = 11, 256


def funnel(w):          # device-only arithmetic (capturable)
    v, x = w[0], w[1:]
    return -v * v / 18.0 + (-0.5 * torch.exp(v) * (x * x).sum() + 0.5 * x.numel() * v)


th0 = torch.ones(C, D, device=dev); th0[:, 0] = 0
for name, kw in (("HMC L=25", dict(num_samples=100, step_size=0.2, num_steps_per_sample=25)),
                 ("explicit RMHMC L=10", dict(num_samples=10, step_size=0.14, num_steps_per_sample=10, sampler=ht.Sampler.RMHMC,
                                              integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, softabs_const=1e6,
                                              explicit_binding_const=10, jitter=0.001))):
    ht.sample(funnel, th0, verbose=False, seed=1, **{**kw, "num_samples": 3})      # warm-up / capture
    torch.cuda.synchronize(); t0 = time.time()
    ht.sample(funnel, th0, verbose=False, seed=1, **kw)
    torch.cuda.synchronize(); dt = time.time() - t0
    steps = kw["num_samples"] * kw["num_steps_per_sample"] * C
    print("%-22s graphs=%s: %.2f s  -> %.3g chain-steps/s" % (name, os.environ.get("HAMILTORCH_AMD_GRAPHS", "1"), dt, steps / dt))

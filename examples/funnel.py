"""The funnel of the reference's notebook (notebooks/hamiltorch_log_prob_examples.ipynb) on the MI355X engine:
HMC, HMC with step-size adaptation, implicit and explicit RMHMC - many chains at once instead of one.

    python examples/funnel.py            (needs a GPU; `import hamiltorch_amd as hamiltorch` is the only change)
"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import hamiltorch_amd as hamiltorch  # noqa: E402

D = 10 + 1


def funnel_ll(w):
    """v = w[0] ~ N(0, 3^2), x = w[1:] ~ N(0, exp(-v)).  Written with device-side arithmetic only, so the engine can replay
    it as a HIP graph; the notebook's torch.distributions.Normal(0, 3) version also works (it mixes host scalars in, is not
    capturable and is evaluated eagerly: ~8x slower)."""
    v, x = w[0], w[1:]
    hl2p = 0.9189385332046727
    ll_v = -v * v / 18.0 - 1.0986122886681098 - hl2p
    ll_x = -0.5 * torch.exp(v) * (x * x).sum() + 0.5 * x.numel() * v - x.numel() * hl2p
    return ll_v + ll_x


def main():
    dev = torch.device("cuda:0")
    hamiltorch.set_random_seed(123)
    chains = 256
    params_init = torch.ones(chains, D, device=dev)
    params_init[:, 0] = 0.0
    runs = [
        ("HMC", dict(num_samples=400, step_size=0.2, num_steps_per_sample=25)),
        ("HMC_NUTS (dual averaging)", dict(num_samples=400, step_size=0.2, num_steps_per_sample=25, burn=100,
                                           sampler=hamiltorch.Sampler.HMC_NUTS, desired_accept_rate=0.75)),
        ("implicit RMHMC", dict(num_samples=60, step_size=0.14, num_steps_per_sample=10, sampler=hamiltorch.Sampler.RMHMC,
                                integrator=hamiltorch.Integrator.IMPLICIT, metric=hamiltorch.Metric.SOFTABS, softabs_const=1e6,
                                fixed_point_threshold=1e-3, fixed_point_max_iterations=20, jitter=0.001)),
        ("explicit RMHMC", dict(num_samples=60, step_size=0.14, num_steps_per_sample=10, sampler=hamiltorch.Sampler.RMHMC,
                                integrator=hamiltorch.Integrator.EXPLICIT, metric=hamiltorch.Metric.SOFTABS, softabs_const=1e6,
                                explicit_binding_const=10, jitter=0.001)),
    ]
    for name, kw in runs:
        t0 = time.time()
        out = hamiltorch.sample(log_prob_func=funnel_ll, params_init=params_init, verbose=False, debug=2, **kw)
        torch.cuda.synchronize()
        samples, extra = out
        s = torch.stack(samples[len(samples) // 2:])              # [S, chains, D]
        v = s[..., 0]
        acc = extra if torch.is_tensor(extra) else torch.tensor(float(extra))
        print("%-28s %d chains x %d samples in %.1f s | v: mean %+.2f sd %.2f (target 0, 3) | acceptance / step %.2f"
              % (name, chains, len(samples), time.time() - t0, float(v.mean()), float(v.std()), float(acc.float().mean())))


if __name__ == "__main__":
    main()

"""Host-resident inputs (hamiltorch_amd/host.py): the reference's callers hand CPU tensors to sample / leapfrog (S:850, S:925;
every notebook; tests/test_util.py:97-110) and get CPU tensors back.  The engine still runs on the GPU."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hamiltorch():
    import hamiltorch_amd
    return hamiltorch_amd


def test_reference_test_vanilla_HMC_reversible_body_runs_unmodified(hamiltorch):
    """/root/reference/tests/test_util.py:97-110, the only test of the reference that pins `leapfrog` - its body VERBATIM
    (CPU tensors, a closure over CPU tensors, bit-exact reversibility asserted with torch.eq), `hamiltorch` being this
    package."""
    def log_prob(omega):
        mean = torch.zeros(2)
        var = torch.tensor([.10,.10])
        return torch.distributions.MultivariateNormal(mean, torch.diag(var)).log_prob(omega).sum()

    params_init = torch.tensor([1.,1.])
    momentum_init = torch.tensor([1.,1.])
    inv_mass = torch.tensor([1.,1.])
    p,m = hamiltorch.samplers.leapfrog(params_init, momentum_init, log_prob, steps=100, step_size=0.1, jitter=None, normalizing_const=1., softabs_const=1e6, explicit_binding_const=100, fixed_point_threshold=1e-20, fixed_point_max_iterations=6, jitter_max_tries=10, inv_mass=inv_mass, ham_func=None, sampler=hamiltorch.Sampler.HMC, integrator=hamiltorch.Integrator.EXPLICIT, metric=hamiltorch.Metric.HESSIAN, debug=False)
    momentum_reversed = - m[-1].clone()
    p,m = hamiltorch.samplers.leapfrog(p[-1], momentum_reversed, log_prob, steps=100, step_size=0.1, jitter=None, normalizing_const=1., softabs_const=1e6, explicit_binding_const=100, fixed_point_threshold=1e-20, fixed_point_max_iterations=6, jitter_max_tries=10, inv_mass=inv_mass, ham_func=None, sampler=hamiltorch.Sampler.HMC, integrator=hamiltorch.Integrator.EXPLICIT, metric=hamiltorch.Metric.HESSIAN, debug=False)

    assert torch.all(torch.eq(p[-1], params_init))
    assert not p[-1].is_cuda and not m[-1].is_cuda and len(p) == 100


def test_leapfrog_kat_with_host_tensors(hamiltorch):
    """KAT1 (SURVEY 8c, T:97-106 set-up) through host tensors: the reference's digits."""
    def log_prob(omega):
        return torch.distributions.MultivariateNormal(torch.zeros(2), torch.diag(torch.tensor([.1, .1]))).log_prob(omega).sum()
    p, m = hamiltorch.samplers.leapfrog(torch.tensor([1., 1.]), torch.tensor([1., 1.]), log_prob, steps=3, step_size=0.1,
                                        inv_mass=torch.tensor([1., 1.]), sampler=hamiltorch.Sampler.HMC,
                                        integrator=hamiltorch.Integrator.EXPLICIT)
    np.testing.assert_allclose(p[-1].numpy(), [0.84049994] * 2, rtol=2e-6)
    np.testing.assert_allclose(m[-1].numpy(), [-1.96525002] * 2, rtol=2e-6)


def test_notebook_cell_with_host_tensors_runs_on_the_fused_kernel(hamiltorch):
    """notebooks/hamiltorch_log_prob_examples.ipynb cells 5-9 as written (a closure over CPU mean / covariance, params_init =
    torch.zeros(3)): recognised as a quadratic form -> one fused launch on the GPU, CPU tensors back; posterior moments and
    the acceptance rate of the reference (0.99)."""
    from hamiltorch_amd import _abi
    mean = torch.tensor([0., 0., 0.])
    stddev = torch.tensor([.5, 1., 2.])

    def log_prob(omega):
        return torch.distributions.MultivariateNormal(mean, torch.diag(stddev ** 2)).log_prob(omega).sum()
    hamiltorch.set_random_seed(123)
    params_init = torch.zeros(3)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out, acc = hamiltorch.sample(log_prob_func=log_prob, params_init=params_init, num_samples=4000, step_size=.3,
                                     num_steps_per_sample=5, debug=2, verbose=False)
    assert any("works on host tensors" in str(w.message) for w in caught)           # announced, by name
    assert _abi.last_route().startswith("hmc_gauss")                                 # the fused Gaussian kernels, not callbacks
    assert isinstance(out, list) and len(out) == 4000 and all(not t.is_cuda and t.shape == (3,) for t in out[:5])
    s = torch.stack(out[500:]).double()
    assert abs(float(acc) - 0.99) < 0.02
    np.testing.assert_allclose(s.std(0).numpy(), stddev.numpy(), rtol=0.15)
    assert (s.mean(0).abs() < 4 * stddev.double() / np.sqrt(100)).all()


def test_opaque_host_closure_is_evaluated_on_the_host_and_matches_the_device_run(hamiltorch):
    """A non-Gaussian closure over HOST tensors (a scaled funnel): the callback path with host evaluations gives the same
    samples as the same function over device tensors (same Philox streams; the callback's arithmetic runs on another
    processor: equal to rounding, chain by chain while no Metropolis decision flips)."""
    scales_host = torch.tensor([1.0, 2.0, 0.5])

    def make(scales):
        def f(w):
            v, x = w[0], w[1:]
            return -v * v / 18.0 - 0.5 * torch.exp(v) * (x * x / scales).sum() + 0.5 * x.numel() * v
        return f
    init = torch.zeros(8, 4); init[:, 1:] = 0.5
    kw = dict(num_samples=12, step_size=0.1, num_steps_per_sample=6, verbose=False, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        host = torch.stack(hamiltorch.sample(make(scales_host), init, **kw))
    dev = torch.stack(hamiltorch.sample(make(scales_host.cuda()), init.cuda(), **kw)).cpu()
    assert not host.is_cuda and host.shape == (12, 8, 4)
    err = (host - dev).abs().amax(dim=(0, 2))
    assert int((err > 1e-4).sum()) <= 1, err


def test_sample_model_with_host_module_and_data(hamiltorch):
    """sample_model on a CPU module / data / params_init (every BNN notebook): a deep copy is staged, the native MLP kernel
    runs, the list comes back on the host; same samples as the device-resident call."""
    from hamiltorch_amd import _abi
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.ReLU(), torch.nn.Linear(16, 1))
    X = torch.randn(50, 3); Y = torch.sin(X.sum(1, keepdim=True))
    init = hamiltorch.util.flatten(net).clone()
    kw = dict(model_loss="regression", num_samples=20, step_size=1e-3, num_steps_per_sample=5, tau_out=10.0, verbose=False, seed=11)
    host = hamiltorch.sample_model(net, X, Y, init, **kw)
    route = _abi.last_route()
    dev = hamiltorch.sample_model(net.cuda(), X.cuda(), Y.cuda(), init.cuda(), **kw)
    assert all(not t.is_cuda for t in host) and len(host) == len(dev) == 20
    assert route == _abi.last_route() and not route.startswith("hmc_pieces")
    assert torch.equal(torch.stack(host), torch.stack(list(dev)).cpu())
    assert next(net.parameters()).is_cuda                     # (net.cuda() above moved the caller's module; the host call had not)

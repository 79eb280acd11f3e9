"""GPU parity for the small-network kernel (csrc/netn_hmc.hip: 1 .. 4 Linear layers, several outputs, softmax / Bernoulli /
Gaussian likelihoods): the shapes of the reference's notebooks that the one-hidden-layer kernels do not cover.  Against the
values recorded from the unmodified reference (tests/golden/deepnet.npz, losses.npz) through the C ABI, against the oracle on
batches of chains, and end to end through sample_model / sample_split_model with notebook-style ``Net`` classes."""
import numpy as np
import pytest
import torch

import hmc_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ht():
    import hamiltorch_amd
    assert torch.cuda.is_available()
    return hamiltorch_amd


def dev():
    return torch.device("cuda:0")


class Net(torch.nn.Module):
    """The notebooks' model class: layers as attributes l1, l2, ..., activations as function calls in forward()."""

    def __init__(self, layer_sizes, act="relu"):
        super().__init__()
        self.act, self.n = act, len(layer_sizes) - 1
        self.l1 = torch.nn.Linear(layer_sizes[0], layer_sizes[1])
        if self.n > 1:
            self.l2 = torch.nn.Linear(layer_sizes[1], layer_sizes[2])
        if self.n > 2:
            self.l3 = torch.nn.Linear(layer_sizes[2], layer_sizes[3])
        if self.n > 3:
            self.l4 = torch.nn.Linear(layer_sizes[3], layer_sizes[4])

    def forward(self, x):
        f = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[self.act]
        x = self.l1(x)
        if self.n > 1:
            x = self.l2(f(x))
        if self.n > 2:
            x = self.l3(f(x))
        if self.n > 3:
            x = self.l4(f(x))
        return x


def n_params(dims):
    return sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))


def make_data(dims, loss, N, seed=1):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, dims[0], generator=g)
    if loss == "multi_class_linear_output":
        Y = torch.randint(0, dims[-1], (N, 1), generator=g).float()
    elif loss == "binary_class_linear_output":
        Y = torch.randint(0, 2, (N, dims[-1]), generator=g).float()
    else:
        Y = torch.sin(X.sum(1, keepdim=True)) + 0.1 * torch.randn(N, 1, generator=g)
    return X, Y


def abi_y(Y, loss, dtype):
    return (Y.reshape(-1) if loss == "multi_class_linear_output" else Y).to(dev(), dtype).contiguous()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float64, 1e-9)])
def test_netn_logp_grad_vs_reference_fixture(ht, golden, dtype, tol):
    """Every closure of the reference's define_split_model_log_prob on the notebook shapes (Net([1, 10, 10, 1]) regression,
    Linear(4, 3) softmax, a [3, 5, 4, 2] tanh net with two Bernoulli outputs): value and gradient through hta_netn_logp_grad."""
    from hamiltorch_amd import _abi
    g = golden("deepnet")
    for name in ("deepreg", "softmaxlin", "bin2"):
        dims, act, loss = [int(d) for d in g[name + "_dims"]], str(g[name + "_act"]), str(g[name + "_loss"])
        M = int(g[name + "_M"]); X, Y = torch.tensor(g[name + "_X"]), torch.tensor(g[name + "_Y"]); Nb = X.shape[0] // M
        th = torch.tensor(g[name + "_theta"][None], dtype=dtype, device=dev())
        for m in range(M):
            grad = torch.empty_like(th); lp = torch.empty(1, dtype=dtype, device=dev())
            _abi.netn_logp_grad(th, dims, act, X.to(dev(), dtype).contiguous(), abi_y(Y, loss, dtype), M, Nb, m, list(g[name + "_tau_list"]),
                                float(g[name + "_tau_out"]), float(M), grad, lp, loss=loss)
            np.testing.assert_allclose(lp.cpu().numpy()[0], g[name + "_logp"][m], rtol=max(tol, 3e-6), atol=tol)
            np.testing.assert_allclose(grad.cpu().numpy()[0], g[name + "_grad"][m], rtol=max(tol, 3e-5), atol=max(tol, 3e-6) * 5)


def test_netn_softmax_matches_reference_losses_fixture(ht, golden):
    """losses.npz 'multi': Linear(4,6)-Tanh-Linear(6,3) with model_loss='multi_class_linear_output', the reference's own value and gradient."""
    from hamiltorch_amd import _abi
    g = golden("losses")
    th = torch.tensor(g["multi_theta"][None].astype(np.float64), device=dev())
    grad = torch.empty_like(th); lp = torch.empty(1, dtype=torch.float64, device=dev())
    _abi.netn_logp_grad(th, [4, 6, 3], "tanh", torch.tensor(g["multi_X"], dtype=torch.float64, device=dev()),
                        torch.tensor(g["multi_Y"].reshape(-1), dtype=torch.float64, device=dev()), 1, 10, 0, list(g["multi_tau_list"]), 2.0, 1.0,
                        grad, lp, loss="multi_class_linear_output")
    np.testing.assert_allclose(lp.cpu().numpy(), g["multi_logp"], rtol=2e-6)
    np.testing.assert_allclose(grad.cpu().numpy()[0], g["multi_grad"], rtol=2e-5, atol=2e-6)


SHAPES = [([4, 3], "relu", "multi_class_linear_output", 150, 3), ([1, 10, 10, 1], "relu", "regression", 400, 4),
          ([2, 4, 3, 1], "tanh", "regression", 24, 3), ([3, 8, 8, 8, 2], "sigmoid", "binary_class_linear_output", 70, 2),
          ([4, 64, 2], "tanh", "multi_class_linear_output", 130, 1), ([64, 5, 1], "relu", "regression", 65, 1),
          ([6, 1], "relu", "binary_class_linear_output", 64, 2), ([2, 3, 10], "tanh", "multi_class_linear_output", 33, 1),
          ([7, 9, 1], "relu", "regression", 129, 1)]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.float64, 1e-10)])
@pytest.mark.parametrize("dims,act,loss,N,M", SHAPES)
def test_netn_logp_grad_vs_oracle(ht, dtype, tol, dims, act, loss, N, M):
    """Value and gradient of every split closure for a batch of chains: no / one / two / three hidden layers, 1 .. 10 outputs,
    every likelihood, point counts on both sides of a wave's 64 lanes, the widest layer (64)."""
    from hamiltorch_amd import _abi
    npdt = np.float32 if dtype == torch.float32 else np.float64
    X, Y = make_data(dims, loss, N)
    D, C, Nb = n_params(dims), 7, N // M
    theta = (0.4 * np.random.default_rng(D).standard_normal((C, D))).astype(npdt)
    taus = [1.0 + 0.25 * k for k in range(2 * (len(dims) - 1))]
    tau_out, ps = 3.0, float(M)
    th = torch.tensor(theta, dtype=dtype, device=dev())
    Xd, Yd = X.to(dev(), dtype).contiguous(), abi_y(Y, loss, dtype)
    for m in range(M):
        g = torch.empty_like(th); lp = torch.empty(C, dtype=dtype, device=dev())
        _abi.netn_logp_grad(th, dims, act, Xd, Yd, M, Nb, m, taus, tau_out, ps, g, lp, loss=loss)
        o = O.MLPRegressionTarget(dims, X.numpy()[m * Nb:(m + 1) * Nb], Y.numpy()[m * Nb:(m + 1) * Nb], taus, tau_out, ps, act, loss=loss)
        wl, wg = o.logp_and_grad(theta.astype(np.float64))
        np.testing.assert_allclose(lp.cpu().numpy(), wl, rtol=tol, atol=tol * max(1.0, np.abs(wl).max()))
        np.testing.assert_allclose(g.cpu().numpy(), wg, rtol=tol, atol=tol * max(1.0, np.abs(wg).max()))


def _cmp(out, ref, tol, max_bad=0.1):
    got = np.stack([o.cpu().numpy() for o in out]); want = np.stack(ref)
    assert got.shape == want.shape
    bad = np.abs(got - want).max(axis=(0, 2)) > tol
    assert bad.mean() <= max_bad, "%d of %d chains differ, max err %.3g" % (bad.sum(), bad.size, np.abs(got - want).max())


def _launch_count(fn):
    from hamiltorch_amd import _abi
    _abi.set_tuning("profile", 1)
    try:
        out = fn()
        torch.cuda.synchronize()
        return out, _abi.profile_collect()[1]
    finally:
        _abi.set_tuning("profile", 0)


@pytest.mark.parametrize("dims,act,loss,eps", [([1, 10, 10, 1], "relu", "regression", 2e-3), ([4, 3], "relu", "multi_class_linear_output", 2e-2),
                                                ([3, 6, 4], "tanh", "multi_class_linear_output", 1e-2)])
def test_sample_model_notebook_nets_vs_oracle(ht, dims, act, loss, eps):
    """sample_model on the notebooks' model class (layers as attributes, torch.relu in forward): recognised by tracing, the whole run
    is ONE native launch and agrees chain by chain with the oracle on the same Philox draws."""
    N, tau_out, L, C, NS, seed = 100, 4.0, 5, 24, 7, 5
    torch.manual_seed(3)
    net = Net(dims, act).to(dev())
    X, Y = make_data(dims, loss, N)
    D = n_params(dims)
    tau_list = torch.tensor([1.0 + 0.5 * k for k in range(2 * (len(dims) - 1))])
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    o = O.MLPRegressionTarget(dims, X.numpy(), Y.numpy(), tau_list.numpy(), tau_out, 1.0, act, loss=loss)
    (out, acc), launches = _launch_count(lambda: ht.sample_model(net, X, Y, torch.tensor(th0, device=dev()), model_loss=loss, num_samples=NS,
                                                                 num_steps_per_sample=L, step_size=eps, tau_out=tau_out, tau_list=tau_list,
                                                                 burn=1, debug=2, verbose=False, seed=seed))
    assert launches == 1, launches
    ref, info = O.sample_hmc(o, th0, NS, L, eps, 1, None, O.PhiloxDraws(seed, np.arange(C)))
    _cmp(out, ref, 5e-4)
    out_g, _ = ht.sample_model(net, X, Y, torch.tensor(th0, device=dev()), model_loss=loss, num_samples=NS, num_steps_per_sample=L,
                               step_size=eps, tau_out=tau_out, tau_list=tau_list, burn=1, debug=2, verbose=False, seed=seed, native=False)
    _cmp(out_g, ref, 5e-4)


@pytest.mark.parametrize("integrator", ["SPLITTING", "SPLITTING_RAND", "SPLITTING_KMID"])
@pytest.mark.parametrize("mass", ["none", "diag"])
def test_sample_split_model_deep_net_vs_oracle(ht, integrator, mass):
    """sample_split_model on Net([2, 6, 5, 1]) with every split integrator and a diagonal mass: native, against the oracle."""
    dims, act, N, M, tau_out, eps, L, C, NS, seed = [2, 6, 5, 1], "tanh", 36, 3, 6.0, 4e-3, 3, 20, 6, 11
    torch.manual_seed(2)
    net = Net(dims, act).to(dev())
    X, Y = make_data(dims, "regression", N)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=N // M, shuffle=False)
    D = n_params(dims)
    tau_list = torch.tensor([1.0 + 0.5 * k for k in range(6)])
    nb = N // M
    otg = [O.MLPRegressionTarget(dims, X.numpy()[m * nb:(m + 1) * nb], Y.numpy()[m * nb:(m + 1) * nb], tau_list.numpy(), tau_out, M, act)
           for m in range(M)]
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    imv = (0.5 + np.random.default_rng(1).random(D)).astype(np.float32)
    im = torch.tensor(imv, device=dev()) if mass == "diag" else None
    integ = getattr(ht.Integrator, integrator)
    kind = {"SPLITTING": "symmetric", "SPLITTING_RAND": "rand", "SPLITTING_KMID": "kmid"}[integrator]
    kw = dict(model_loss="regression", num_samples=NS, num_steps_per_sample=L, step_size=eps, burn=1, inv_mass=im, tau_out=tau_out,
              tau_list=tau_list, verbose=False, seed=seed, integrator=integ)
    out, launches = _launch_count(lambda: ht.sample_split_model(net, loader, torch.tensor(th0, device=dev()), M, **kw))
    assert launches == 1, launches
    ref, _ = O.sample_hmc(None, th0, NS, L, eps, 1, None if im is None else imv, O.PhiloxDraws(seed, np.arange(C)),
                          grad_fns=[t.grad for t in otg], logp_fns=[t.logp for t in otg], split_kind=kind)
    _cmp(out, ref, 5e-4)


def test_netn_limits_fall_back_to_the_callback_path(ht):
    """Shapes beyond the kernel's limits (five Linear layers; more than 512 parameters) are not claimed: the callback path runs them."""
    from hamiltorch_amd import bnn, mlp
    for dims in ([2, 3, 3, 3, 3, 1], [8, 60, 40]):
        net = torch.nn.Sequential(*[m for i in range(len(dims) - 1)
                                    for m in ([torch.nn.Linear(dims[i], dims[i + 1])] + ([torch.nn.Tanh()] if i < len(dims) - 2 else []))]).to(dev())
        loss = "regression" if dims[-1] == 1 else "multi_class_linear_output"
        X, Y = make_data(dims, loss, 20)
        sizes = [w.nelement() for w in net.parameters()]; shapes = [w.shape for w in net.parameters()]
        f = bnn.define_model_log_prob(net, loss, X, Y, sizes, shapes, [1.0] * len(sizes), 1.0, device=dev())
        assert mlp.hmc_engine(f, torch.zeros(1, n_params(dims), device=dev())) is None


@pytest.mark.parametrize("waves", [2, 4])
def test_netn_several_waves_per_chain(ht, waves):
    """The tuning key "netn_waves": a chain's sweeps over 2 / 4 waves of one workgroup (shared parameter copy and gradient vector,
    likelihood sums exchanged through LDS).  Slower than one wave per chain at 1024 chains, kept as a measured variant: same
    run as with one wave, chain by chain to rounding - gradients, a split run with SPLITTING_RAND's shared subset order."""
    from hamiltorch_amd import _abi
    dims, act, loss, N, M = [3, 7, 5, 2], "tanh", "multi_class_linear_output", 330, 3
    X, Y = make_data(dims, loss, N)
    D, C, Nb = n_params(dims), 9, N // M
    taus = [1.0 + 0.25 * k for k in range(2 * (len(dims) - 1))]
    th0 = torch.tensor((0.3 * np.random.default_rng(4).standard_normal((C, D))).astype(np.float32), device=dev())
    Xd, Yd = X.to(dev()).contiguous(), abi_y(Y, loss, torch.float32)
    outs = []
    try:
        for wv in (1, waves):
            _abi.set_tuning("netn_waves", wv)
            g = torch.empty_like(th0); lp = torch.empty(C, device=dev())
            _abi.netn_logp_grad(th0, dims, act, Xd, Yd, M, Nb, 1, taus, 2.0, float(M), g, lp, loss=loss)
            cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev()); T = 6
            samples = torch.zeros(T + 1, C, D, device=dev())
            _abi.netn_hmc_sample(cur, th0, dims, act, Xd, Yd, M, Nb, taus, 2.0, float(M), _abi.MASS_NONE, None, None, 4, 5e-3, T, 0, 0, 7, 0,
                                 samples, rej, integrator=_abi.SPLIT_RAND, loss=loss)
            torch.cuda.synchronize()
            outs.append((g.cpu().numpy(), lp.cpu().numpy(), samples.cpu().numpy(), rej.cpu().numpy()))
    finally:
        _abi.set_tuning("netn_waves", 1)
    np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=2e-5, atol=2e-5 * np.abs(outs[0][0]).max())
    np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=2e-6)
    assert np.abs(outs[0][2][1:]).max() > 0
    err = np.abs(outs[1][2] - outs[0][2]).max(axis=(0, 2))
    assert (err > 2e-4).mean() <= 0.12, "max err %.3g" % err.max()
    good = err <= 2e-4
    assert np.array_equal(outs[1][3][good], outs[0][3][good])


def test_log_softmax_models_run_on_the_softmax_kernel(ht, golden):
    """'multi_class_log_softmax_output' (nll_loss with its MEAN reduction, S:1180) on a model that ends in LogSoftmax(dim=1) is
    -(tau_out / N) x the cross-entropy of the logits: the softmax kernel with a scaled precision.  Pinned to the reference's own
    value and gradient (losses.npz 'logsoftmax': Linear(4,6)-Tanh-Linear(6,3)-LogSoftmax, tau_out = 2, N = 10), then
    sample_model natively (one launch) against the callback path on the same draws."""
    from hamiltorch_amd import _abi
    g = golden("losses")
    th = torch.tensor(g["logsoftmax_theta"][None].astype(np.float64), device=dev())
    grad = torch.empty_like(th); lp = torch.empty(1, dtype=torch.float64, device=dev())
    _abi.netn_logp_grad(th, [4, 6, 3], "tanh", torch.tensor(g["logsoftmax_X"], dtype=torch.float64, device=dev()),
                        torch.tensor(g["logsoftmax_Y"].reshape(-1), dtype=torch.float64, device=dev()), 1, 10, 0, list(g["logsoftmax_tau_list"]),
                        2.0 / 10, 1.0, grad, lp, loss="multi_class_linear_output")
    np.testing.assert_allclose(lp.cpu().numpy(), g["logsoftmax_logp"], rtol=2e-6)
    np.testing.assert_allclose(grad.cpu().numpy()[0], g["logsoftmax_grad"], rtol=2e-5, atol=2e-6)
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(4, 6), torch.nn.Tanh(), torch.nn.Linear(6, 3), torch.nn.LogSoftmax(dim=1)).to(dev())
    X, Y = make_data([4, 6, 3], "multi_class_linear_output", 90)
    C, NS, L, seed = 16, 6, 4, 3
    th0 = torch.tensor((0.3 * O.philox_normals(seed, np.arange(C), 0, 51, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), device=dev())
    kw = dict(model_loss="multi_class_log_softmax_output", num_samples=NS, num_steps_per_sample=L, step_size=0.05, tau_out=40.0, verbose=False, seed=seed)
    out, launches = _launch_count(lambda: ht.sample_model(net, X, Y, th0, **kw))
    assert launches == 1, launches
    ref = ht.sample_model(net, X, Y, th0, native=False, **kw)
    _cmp(out, [r.cpu().numpy() for r in ref], 5e-4)

"""GPU parity for csrc/mlp3_mfma.hip: Bayesian MLPs with two wide hidden layers on the matrix cores - the reference's published
split-HMC model, notebooks/hamiltorch_split_HMC_BNN_example.ipynb cell 9: Linear(1,100)-ReLU-Linear(100,100)-ReLU-Linear(100,1),
D = 10401 (13.47 samples/s full HMC, 1.83 samples/s split HMC on its author's RTX 2080 Max-Q).  Against the values recorded
from the unmodified reference at that size (tests/golden/nbmlp.npz) through the C ABI, against the oracle on batches of chains
for other widths / activations / input widths / ragged and multi-chunk splits, and end to end through sample_model /
sample_split_model with the notebook's ``Net`` class, every split integrator and a diagonal mass.

Tolerances: SURVEY 8c's - 1e-4 relative on log-prob / gradient (fp32, sums over 100 x 100 products), 5e-4 on states end to
end; a Metropolis decision within rounding of its threshold may flip a chain (<= 10 % of the compared chains)."""
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


class NotebookNet(torch.nn.Module):
    """notebooks/hamiltorch_split_HMC_BNN_example.ipynb cell 9 (fc1, fc2, fc3; F.relu in forward); widths as arguments."""

    def __init__(self, n_in=1, h1=100, h2=100, act="relu"):
        super().__init__()
        self.act = act
        self.fc1 = torch.nn.Linear(n_in, h1)
        self.fc2 = torch.nn.Linear(h1, h2)
        self.fc3 = torch.nn.Linear(h2, 1)

    def forward(self, x):
        f = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[self.act]
        x = f(self.fc1(x))
        x = f(self.fc2(x))
        return self.fc3(x)


def n_params(dims):
    return sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))


def make_data(n_in, N, seed=1):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, n_in, generator=g)
    Y = torch.sin(1.5 * X.sum(1, keepdim=True)) + 0.1 * torch.randn(N, 1, generator=g)
    return X, Y


def init_theta(dims, C, seed, scale=1.0):
    """torch-style initialisation scale per layer (uniform +- 1/sqrt(fan_in)) drawn from the oracle's Philox init stream."""
    D = n_params(dims)
    z = O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)
    out, off = np.empty((C, D)), 0
    for i in range(len(dims) - 1):
        n = dims[i] * dims[i + 1] + dims[i + 1]
        out[:, off:off + n] = scale * z[:, off:off + n] / np.sqrt(3.0 * dims[i])
        off += n
    return out.astype(np.float32)


def test_mlp3_reference_fixture_through_c_abi(ht, golden):
    """tests/golden/nbmlp.npz (the unmodified reference at D = 10401): log-prob + gradient of the full-data closure (4 chunks of
    points) and of the split closures from hta_netn_logp_grad, dispatched to mlp3_mfma_kernel<0>."""
    from hamiltorch_amd import _abi
    g = golden("nbmlp")
    n = "nbmlp"
    M, tau_out = int(g[f"{n}_cfg"][0]), float(g[f"{n}_cfg"][1])
    dims = [int(d) for d in g[f"{n}_dims"]]
    X, Y = torch.tensor(g[f"{n}_X"], device=dev()), torch.tensor(g[f"{n}_Y"], device=dev())
    th = torch.tensor(g[f"{n}_theta"][None], device=dev())
    taus = [float(t) for t in g[f"{n}_tau_list"]]
    grad = torch.empty_like(th); lp = torch.empty(1, device=dev())
    _abi.netn_logp_grad(th, dims, "relu", X, Y, 1, 400, 0, taus, tau_out, 1.0, grad, lp)
    assert _abi.last_route() == "mlp3_mfma_kernel<0>", _abi.last_route()
    gs = np.abs(g[f"{n}_grad"]).max()
    np.testing.assert_allclose(lp.cpu().numpy(), g[f"{n}_logp"], rtol=1e-4)
    np.testing.assert_allclose(grad.cpu().numpy()[0], g[f"{n}_grad"], rtol=1e-3, atol=1e-4 * gs)
    lps = []
    for m in range(M):
        _abi.netn_logp_grad(th, dims, "relu", X, Y, M, 100, m, taus, tau_out, float(M), grad, lp)
        lps.append(float(lp))
        if m == 1:
            np.testing.assert_allclose(grad.cpu().numpy()[0], g[f"{n}_split1_grad"], rtol=1e-3, atol=1e-4 * gs)
    np.testing.assert_allclose(lps, g[f"{n}_split_logp"], rtol=1e-4)


def test_mlp3_reference_paths_replayed_through_c_abi(ht, golden):
    """The reference's 3-step SPLITTING path and 4-step leapfrog path at D = 10401 replayed from its own (theta, p0): every half
    kick's gradient from hta_netn_logp_grad (S:499-540, S:281-302), the updates by hta_kick_drift."""
    from hamiltorch_amd import _abi
    g = golden("nbmlp")
    n = "nbmlp"
    M, tau_out, eps, Ls, Lf = g[f"{n}_cfg"]
    M, Ls, Lf, eps, tau_out = int(M), int(Ls), int(Lf), float(eps), float(tau_out)
    dims = [int(d) for d in g[f"{n}_dims"]]
    X, Y = torch.tensor(g[f"{n}_X"], device=dev()), torch.tensor(g[f"{n}_Y"], device=dev())
    taus = [float(t) for t in g[f"{n}_tau_list"]]
    D = 10401
    gr = torch.empty(1, D, device=dev()); lp = torch.empty(1, device=dev())

    def grad(th, m, Msp, nb, ps):
        _abi.netn_logp_grad(th, dims, "relu", X, Y, Msp, nb, m, taus, tau_out, ps, gr, lp)
        return gr
    # SPLITTING (S:499-540)
    th = torch.tensor(g[f"{n}_theta"][None], device=dev()); p = torch.tensor(g[f"{n}_p0"][None], device=dev())
    dq = eps / ((M - 1) * 2)
    for _ in range(Ls):
        for m in range(M):
            _abi.kick_drift(th, p, grad(th, m, M, 100, float(M)), 0.5 * eps, dq if m < M - 1 else 0.0, _abi.MASS_NONE, None)
        for m in reversed(range(M)):
            _abi.kick_drift(th, p, grad(th, m, M, 100, float(M)), 0.5 * eps, dq if m > 0 else 0.0, _abi.MASS_NONE, None)
    np.testing.assert_allclose(th.cpu().numpy()[0], g[f"{n}_lf_theta"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(p.cpu().numpy()[0], g[f"{n}_lf_p"], rtol=1e-3, atol=3e-3)
    # plain leapfrog on the full-data closure (S:281-302)
    th = torch.tensor(g[f"{n}_theta"][None], device=dev()); p = torch.tensor(g[f"{n}_p0"][None], device=dev())
    _abi.kick_drift(th, p, grad(th, 0, 1, 400, 1.0), 0.5 * eps, eps, _abi.MASS_NONE, None)
    for k in range(Lf):
        last = k == Lf - 1
        _abi.kick_drift(th, p, grad(th, 0, 1, 400, 1.0), eps, 0.0 if last else eps, _abi.MASS_NONE, None)
    _abi.kick_drift(th, p, gr, -0.5 * eps, 0.0, _abi.MASS_NONE, None)
    np.testing.assert_allclose(th.cpu().numpy()[0], g[f"{n}_full_lf_theta"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(p.cpu().numpy()[0], g[f"{n}_full_lf_p"], rtol=1e-3, atol=3e-3)


SHAPES = [([1, 100, 100, 1], "relu", 400, 4),       # the notebook model: 4 splits of 100 points (one chunk of 7 tiles each)
          ([1, 100, 100, 1], "relu", 400, 1),       # the same as full HMC: 4 chunks, the last one ragged (64 points)
          ([2, 72, 100, 1], "tanh", 150, 3),        # H1 < 96 (empty remainder block), two inputs, 50-point splits
          ([4, 104, 97, 1], "sigmoid", 90, 2),      # the widest first layer (two remainder steps), odd H2, four inputs
          ([3, 65, 33, 1], "relu", 230, 1),         # just beyond the small-net kernel, 3 chunks
          ([1, 100, 100, 1], "tanh", 37, 1)]        # fewer points than a chunk


@pytest.mark.parametrize("dims,act,N,M", SHAPES)
def test_mlp3_logp_grad_vs_oracle(ht, dims, act, N, M):
    """Value and gradient of every split closure for a batch of chains (more chains than one wave of workgroups would hold of a
    single alignment class: D is odd for the notebook model, so rows of theta are 16-byte aligned for every fourth chain only)."""
    from hamiltorch_amd import _abi
    X, Y = make_data(dims[0], N)
    D, C, Nb = n_params(dims), 9, N // M
    theta = init_theta(dims, C, 5, scale=1.5)
    taus = [1.0 + 0.25 * k for k in range(6)]
    tau_out, ps = 7.0, float(M)
    th = torch.tensor(theta, device=dev())
    Xd, Yd = X.to(dev()).contiguous(), Y.to(dev()).contiguous()
    for m in range(M):
        g = torch.empty_like(th); lp = torch.empty(C, device=dev())
        _abi.netn_logp_grad(th, dims, act, Xd, Yd, M, Nb, m, taus, tau_out, ps, g, lp)
        assert _abi.last_route() == "mlp3_mfma_kernel<%d>" % {"relu": 0, "tanh": 1, "sigmoid": 2}[act]
        o = O.MLPRegressionTarget(dims, X.numpy()[m * Nb:(m + 1) * Nb], Y.numpy()[m * Nb:(m + 1) * Nb], taus, tau_out, ps, act)
        wl, wg = o.logp_and_grad(theta.astype(np.float64))
        np.testing.assert_allclose(lp.cpu().numpy(), wl, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(wl).max()))
        np.testing.assert_allclose(g.cpu().numpy(), wg, rtol=1e-3, atol=1e-4 * max(1.0, np.abs(wg).max()))
    # a split index out of range is refused before any launch
    with pytest.raises(_abi.InvalidArguments):
        _abi.netn_logp_grad(th, dims, act, Xd, Yd, M, Nb, M, taus, tau_out, ps, g, lp)


def _cmp(out, ref, tol, max_bad=0.1):
    got = np.stack([o.cpu().numpy() for o in out]); want = np.stack(ref)
    assert got.shape == want.shape
    bad = np.abs(got - want).max(axis=(0, 2)) > tol
    assert bad.mean() <= max_bad, "%d of %d chains differ, max err %.3g" % (bad.sum(), bad.size, np.abs(got - want).max())
    return bad


def _launch_count(fn):
    from hamiltorch_amd import _abi
    _abi.set_tuning("profile", 1)
    try:
        out = fn()
        torch.cuda.synchronize()
        return out, _abi.profile_collect()[1]
    finally:
        _abi.set_tuning("profile", 0)


@pytest.mark.parametrize("integrator", ["SPLITTING", "SPLITTING_RAND", "SPLITTING_KMID"])
def test_sample_split_model_notebook_net_vs_oracle(ht, integrator):
    """sample_split_model on the notebook's model at its size (D = 10401, 400 points, M = 4, tau_out = 110.44, inv_mass = ones,
    eps = 5e-4) with every split integrator: recognised by tracing the class, ONE native launch of mlp3_mfma_kernel<0>, chain by
    chain against the oracle on the same Philox draws - 12 chains x 3 trajectories x L = 3, burn-in and the Q2 reset included."""
    from hamiltorch_amd import _abi
    dims, N, M, tau_out, eps, L, C, NS, seed = [1, 100, 100, 1], 400, 4, 110.4439498986428, 5e-4, 3, 12, 4, 11
    torch.manual_seed(2)
    net = NotebookNet().to(dev())
    X, Y = make_data(1, N)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=N // M, shuffle=False)
    D = n_params(dims)
    assert D == 10401
    tau_list = torch.ones(6)
    nb = N // M
    otg = [O.MLPRegressionTarget(dims, X.numpy()[m * nb:(m + 1) * nb], Y.numpy()[m * nb:(m + 1) * nb], tau_list.numpy(), tau_out, M, "relu")
           for m in range(M)]
    th0 = init_theta(dims, C, seed)
    integ = getattr(ht.Integrator, integrator)
    kind = {"SPLITTING": "symmetric", "SPLITTING_RAND": "rand", "SPLITTING_KMID": "kmid"}[integrator]
    kw = dict(model_loss="regression", num_samples=NS, num_steps_per_sample=L, step_size=eps, burn=1, inv_mass=torch.ones(D, device=dev()),
              tau_out=tau_out, tau_list=tau_list, verbose=False, seed=seed, integrator=integ, debug=2)
    (out, acc), launches = _launch_count(lambda: ht.sample_split_model(net, loader, torch.tensor(th0, device=dev()), M, **kw))
    assert launches == 1, launches
    assert _abi.last_route() == "mlp3_mfma_kernel<0>", _abi.last_route()
    ref, info = O.sample_hmc(None, th0, NS, L, eps, 1, np.ones(D, np.float32), O.PhiloxDraws(seed, np.arange(C)),
                             grad_fns=[t.grad for t in otg], logp_fns=[t.logp for t in otg], split_kind=kind)
    bad = _cmp(out, ref, 5e-4)
    np.testing.assert_allclose(acc.cpu().numpy()[~bad], info["acc_rate"][~bad], atol=1e-12)


@pytest.mark.parametrize("mass", ["none", "diag"])
def test_sample_model_full_hmc_notebook_net_vs_oracle(ht, mass):
    """sample_model (plain leapfrog, S:281-302, on all 400 points: four chunks per gradient) with a non-trivial diagonal mass."""
    from hamiltorch_amd import _abi
    dims, N, tau_out, eps, L, C, NS, seed = [1, 100, 100, 1], 400, 110.4439498986428, 2e-4, 4, 10, 4, 3
    torch.manual_seed(4)
    net = NotebookNet().to(dev())
    X, Y = make_data(1, N)
    D = n_params(dims)
    tau_list = torch.tensor([1.0, 2.0, 1.0, 0.5, 1.5, 1.0])
    o = O.MLPRegressionTarget(dims, X.numpy(), Y.numpy(), tau_list.numpy(), tau_out, 1.0, "relu")
    th0 = init_theta(dims, C, seed)
    imv = (0.5 + np.random.default_rng(1).random(D)).astype(np.float32)
    im = torch.tensor(imv, device=dev()) if mass == "diag" else None
    (out, acc), launches = _launch_count(lambda: ht.sample_model(net, X.to(dev()), Y.to(dev()), torch.tensor(th0, device=dev()), model_loss="regression",
                                                                 num_samples=NS, num_steps_per_sample=L, step_size=eps, tau_out=tau_out, tau_list=tau_list,
                                                                 inv_mass=im, burn=-1, debug=2, verbose=False, seed=seed))
    assert launches == 1 and _abi.last_route() == "mlp3_mfma_kernel<0>", (launches, _abi.last_route())
    ref, info = O.sample_hmc(o, th0, NS, L, eps, -1, None if im is None else imv, O.PhiloxDraws(seed, np.arange(C)))
    bad = _cmp(out, ref, 5e-4)
    np.testing.assert_allclose(acc.cpu().numpy()[~bad], info["acc_rate"][~bad], atol=1e-12)


def test_other_widths_end_to_end_and_callback_agreement(ht):
    """A tanh net with two inputs and unequal widths (2-72-100-1), ragged splits of 50 points: the native run against the oracle
    AND against the generic-callback path of the same call (native=False)."""
    dims, act, N, M, tau_out, eps, L, C, NS, seed = [2, 72, 100, 1], "tanh", 150, 3, 9.0, 1e-3, 3, 8, 3, 17
    torch.manual_seed(1)
    net = NotebookNet(2, 72, 100, act).to(dev())
    X, Y = make_data(2, N)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=N // M, shuffle=False)
    D = n_params(dims)
    tau_list = torch.tensor([1.0 + 0.5 * k for k in range(6)])
    nb = N // M
    otg = [O.MLPRegressionTarget(dims, X.numpy()[m * nb:(m + 1) * nb], Y.numpy()[m * nb:(m + 1) * nb], tau_list.numpy(), tau_out, M, act)
           for m in range(M)]
    th0 = init_theta(dims, C, seed)
    kw = dict(model_loss="regression", num_samples=NS, num_steps_per_sample=L, step_size=eps, burn=0, tau_out=tau_out, tau_list=tau_list,
              verbose=False, seed=seed, inv_mass=torch.ones(D, device=dev()))
    out = ht.sample_split_model(net, loader, torch.tensor(th0, device=dev()), M, **kw)
    from hamiltorch_amd import _abi
    assert _abi.last_route() == "mlp3_mfma_kernel<1>"
    ref, _ = O.sample_hmc(None, th0, NS, L, eps, 0, np.ones(D, np.float32), O.PhiloxDraws(seed, np.arange(C)),
                          grad_fns=[t.grad for t in otg], logp_fns=[t.logp for t in otg])
    _cmp(out, ref, 5e-4)
    out_g = ht.sample_split_model(net, loader, torch.tensor(th0, device=dev()), M, native=False, **kw)
    _cmp(out_g, ref, 5e-4)


def test_fp64_and_route_key_fall_back_to_the_callback_path(ht):
    """The kernel is fp32 only: a float64 run of the same model, and hta_set_tuning('mlp3_route', 0), take the generic-callback
    path (no native launch) and still agree with the oracle."""
    from hamiltorch_amd import _abi
    dims, N, tau_out, eps, L, C, NS, seed = [1, 100, 100, 1], 60, 20.0, 5e-4, 2, 3, 3, 9
    torch.manual_seed(4)
    X, Y = make_data(1, N)
    tau_list = torch.ones(6)
    th0 = init_theta(dims, C, seed)
    for dtype, key in ((torch.float64, 1), (torch.float32, 0)):
        net = NotebookNet().to(dev(), dtype)
        _abi.set_tuning("mlp3_route", key)
        npdt = np.float64 if dtype == torch.float64 else np.float32
        o = O.MLPRegressionTarget(dims, X.numpy().astype(npdt), Y.numpy().astype(npdt), tau_list.numpy(), tau_out, 1.0, "relu")
        out, launches = _launch_count(lambda: ht.sample_model(net, X.to(dev(), dtype), Y.to(dev(), dtype), torch.tensor(th0, dtype=dtype, device=dev()),
                                                              model_loss="regression", num_samples=NS, num_steps_per_sample=L, step_size=eps,
                                                              tau_out=tau_out, tau_list=tau_list, burn=-1, verbose=False, seed=seed))
        assert launches == 0, launches
        ref, _ = O.sample_hmc(o, th0.astype(npdt), NS, L, eps, -1, None, O.PhiloxDraws(seed, np.arange(C), npdt))
        _cmp(out, ref, 5e-4 if dtype == torch.float32 else 1e-8)

"""GPU parity for the Bayesian-MLP path (BASELINE config 4): define_model_log_prob closures, the
SPLITTING integrator and sample_model / sample_split_model, native kernel and generic-callback route,
against the oracle (same Philox draws) and the values recorded from the reference."""
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


def make_net(dims, act, seed=0):
    torch.manual_seed(seed)
    layers = []
    for i in range(len(dims) - 1):
        layers.append(torch.nn.Linear(int(dims[i]), int(dims[i + 1])))
        if i < len(dims) - 2:
            layers.append({"relu": torch.nn.ReLU, "tanh": torch.nn.Tanh, "sigmoid": torch.nn.Sigmoid}[act]())
    return torch.nn.Sequential(*layers).to(dev())


def make_data(N, n_in, seed=1):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, n_in, generator=g)
    Y = torch.sin(X.sum(1, keepdim=True)) + 0.1 * torch.randn(N, 1, generator=g)
    return X, Y


@pytest.fixture
def valu_kernel():
    """Keeps the fp32 sampler on the VALU kernel (mlp_hmc.hip) instead of the MFMA one (mlp_mfma.hip)."""
    from hamiltorch_amd import _abi
    _abi.set_tuning("mlp_valu", 1)
    yield
    _abi.set_tuning("mlp_valu", 0)


SHAPES = [(3, 5, "relu", 12, 3), (8, 100, "relu", 400, 4), (1, 17, "tanh", 30, 2), (5, 64, "sigmoid", 64, 1),
          (16, 130, "tanh", 50, 5), (2, 300, "relu", 40, 2),
          # MFMA kernel corners: 3 input blocks + ragged 128-point chunks; 4 input blocks (no free ones-row); 17 tiles
          (12, 40, "relu", 300, 1), (16, 100, "sigmoid", 200, 1), (9, 256, "tanh", 36, 2)]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.float64, 1e-10)])
@pytest.mark.parametrize("n_in,H,act,N,M", SHAPES)
def test_native_logp_grad_vs_oracle(ht, dtype, tol, n_in, H, act, N, M):
    """Value and gradient of every split closure (S:1145-1199) for a batch of chains (fp32: the MFMA kernel where it applies)."""
    _logp_grad_case(dtype, tol, n_in, H, act, N, M)


@pytest.mark.parametrize("n_in,H,act,N,M", SHAPES)
def test_native_logp_grad_vs_oracle_valu_kernel(ht, valu_kernel, n_in, H, act, N, M):
    """The same closures with the fp32 VALU kernel forced."""
    _logp_grad_case(torch.float32, 3e-4, n_in, H, act, N, M)


def _logp_grad_case(dtype, tol, n_in, H, act, N, M):
    from hamiltorch_amd import _abi
    npdt = np.float32 if dtype == torch.float32 else np.float64
    X, Y = make_data(N, n_in)
    D = H * n_in + 2 * H + 1
    C, Nb = 7, N // M
    rng = np.random.default_rng(H)
    theta = (0.4 * rng.standard_normal((C, D))).astype(npdt)
    tau = [1.0, 1.5, 2.0, 2.5]
    tau_out, ps = 7.0, float(M)
    th = torch.tensor(theta, dtype=dtype, device=dev())
    Xd, Yd = X.to(dev(), dtype).contiguous(), Y.reshape(-1).to(dev(), dtype).contiguous()
    for m in range(M):
        g = torch.empty_like(th); lp = torch.empty(C, dtype=dtype, device=dev())
        _abi.mlp_logp_grad(th, n_in, H, act, Xd, Yd, M, Nb, m, tau, tau_out, ps, g, lp)
        o = O.MLPRegressionTarget([n_in, H, 1], X.numpy()[m * Nb:(m + 1) * Nb], Y.numpy()[m * Nb:(m + 1) * Nb], tau, tau_out, ps, act)
        wl, wg = o.logp_and_grad(theta.astype(np.float64))
        np.testing.assert_allclose(lp.cpu().numpy(), wl, rtol=tol, atol=tol * max(1.0, np.abs(wl).max()))
        np.testing.assert_allclose(g.cpu().numpy(), wg, rtol=tol, atol=tol * max(1.0, np.abs(wg).max()))


def test_native_matches_reference_fixture(ht, golden):
    g = golden("mlp")
    from hamiltorch_amd import _abi
    M, tau_out, eps, L = g["relu2_cfg"]
    X = torch.tensor(g["relu2_X"], device=dev()); Y = torch.tensor(g["relu2_Y"].reshape(-1), device=dev())
    th = torch.tensor(g["relu2_theta"][None], device=dev())
    grad = torch.empty_like(th); lp = torch.empty(1, device=dev())
    _abi.mlp_logp_grad(th, 3, 5, "relu", X, Y, 1, 12, 0, list(g["relu2_tau_list"]), float(tau_out), 1.0, grad, lp)
    np.testing.assert_allclose(lp.cpu().numpy(), g["relu2_logp"], rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(grad.cpu().numpy()[0], g["relu2_grad"], rtol=2e-4, atol=2e-4)


def _split_setup(ht, dims, act, N, M, tau_out, dtype=torch.float32):
    net = make_net(dims, act)
    X, Y = make_data(N, dims[0])
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=N // M, shuffle=False)
    ntens = len(list(net.parameters()))
    tau_list = torch.tensor([1.0 + 0.5 * k for k in range(ntens)])
    nb = N // M
    otg = [O.MLPRegressionTarget(dims, X.numpy()[m * nb:(m + 1) * nb], Y.numpy()[m * nb:(m + 1) * nb], tau_list.numpy(), tau_out, M, act)
           for m in range(M)]
    return net, X, Y, loader, tau_list, otg


def _cmp(out, ref, tol, max_bad=0.07):
    got = np.stack([o.cpu().numpy() for o in out]); want = np.stack(ref)
    assert got.shape == want.shape
    bad = np.abs(got - want).max(axis=(0, 2)) > tol
    assert bad.mean() <= max_bad, "%d of %d chains differ, max err %.3g" % (bad.sum(), bad.size, np.abs(got - want).max())


@pytest.mark.parametrize("dims,act,native_expected", [([3, 5, 1], "relu", True), ([4, 33, 1], "tanh", True), ([2, 4, 3, 1], "tanh", True)])       # (two hidden layers: csrc/netn_hmc.hip)
@pytest.mark.parametrize("mass", ["ones", "none"])
def test_sample_split_model_vs_oracle(ht, dims, act, native_expected, mass):
    """sample_split_model end to end: closures per DataLoader batch, SPLITTING integrator, MH, bookkeeping."""
    N, M, tau_out, eps, L, C, NS, seed = 24, 3, 6.0, 4e-3, 3, 20, 7, 11
    net, X, Y, loader, tau_list, otg = _split_setup(ht, dims, act, N, M, tau_out)
    D = sum(p.numel() for p in net.parameters())
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    im = torch.ones(D, device=dev()) if mass == "ones" else None
    kw = dict(model_loss="regression", num_samples=NS, num_steps_per_sample=L, step_size=eps, burn=1, inv_mass=im,
              tau_out=tau_out, tau_list=tau_list, verbose=False, seed=seed)
    out = ht.sample_split_model(net, loader, torch.tensor(th0, device=dev()), M, **kw)
    ref, _ = O.sample_hmc(None, th0, NS, L, eps, 1, None if im is None else np.ones(D, np.float32), O.PhiloxDraws(seed, np.arange(C)),
                          grad_fns=[t.grad for t in otg], logp_fns=[t.logp for t in otg])
    _cmp(out, ref, 5e-4)
    out_g = ht.sample_split_model(net, loader, torch.tensor(th0, device=dev()), M, native=False, **kw)
    _cmp(out_g, ref, 5e-4)
    from hamiltorch_amd import bnn, mlp
    sizes = [w.nelement() for w in net.parameters()]; shapes = [w.shape for w in net.parameters()]
    fl = bnn.define_split_model_log_prob(net, "regression", loader, M, sizes, shapes, tau_list, tau_out, device=dev(), verbose=False)
    assert (mlp.split_engine(fl, torch.tensor(th0, device=dev())) is not None) == native_expected


def test_sample_model_full_data_vs_oracle_valu_kernel(ht, valu_kernel):
    test_sample_model_full_data_vs_oracle(ht)


def test_sample_split_model_vs_oracle_valu_kernel(ht, valu_kernel):
    test_sample_split_model_vs_oracle(ht, [4, 33, 1], "tanh", True, "ones")


def test_sample_model_full_data_vs_oracle(ht):
    """sample_model: one closure over all data, plain leapfrog (S:281-302) in the native kernel."""
    dims, act, N, tau_out, eps, L, C, NS, seed = [3, 9, 1], "relu", 20, 5.0, 3e-3, 4, 24, 6, 5
    net = make_net(dims, act)
    X, Y = make_data(N, 3)
    tau_list = torch.tensor([1.0, 1.5, 2.0, 2.5])
    D = sum(p.numel() for p in net.parameters())
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    o = O.MLPRegressionTarget(dims, X.numpy(), Y.numpy(), tau_list.numpy(), tau_out, 1.0, act)
    out, acc = ht.sample_model(net, X, Y, torch.tensor(th0, device=dev()), model_loss="regression", num_samples=NS,
                               num_steps_per_sample=L, step_size=eps, tau_out=tau_out, tau_list=tau_list, debug=2,
                               verbose=False, seed=seed)
    ref, info = O.sample_hmc(o, th0, NS, L, eps, 0, None, O.PhiloxDraws(seed, np.arange(C)))
    _cmp(out, ref, 5e-4)
    one = ht.sample_model(net, X, Y, torch.tensor(th0[0], device=dev()), model_loss="regression", num_samples=NS,
                          num_steps_per_sample=L, step_size=eps, tau_out=tau_out, tau_list=tau_list, verbose=False, seed=seed)
    assert len(one) == NS and one[0].shape == (D,)
    np.testing.assert_allclose(torch.stack(one).cpu().numpy(), np.stack(ref)[:, 0], atol=5e-4)


def test_cfg4_shape(ht):
    """BASELINE config 4 shape: Linear(8,100)-ReLU-Linear(100,1), 400 points, M=4, eps=5e-4, L=10, 512 chains.
    The reference accepts 0.86 here; a short run must be finite and accept at a similar rate."""
    net = make_net([8, 100, 1], "relu")
    g = torch.Generator().manual_seed(0)
    X = torch.randn(400, 8, generator=g); w = torch.randn(8, 1, generator=g)
    Y = torch.sin(X @ w) + 0.1 * torch.randn(400, 1, generator=g)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=100, shuffle=False)
    D = 1001
    theta0 = ht.util.flatten(net).detach().repeat(512, 1).contiguous()
    out, acc = ht.sample_split_model(net, loader, theta0, 4, model_loss="regression", num_samples=12, num_steps_per_sample=10,
                                     step_size=5e-4, inv_mass=torch.ones(D, device=dev()), tau_out=100.0,
                                     tau_list=torch.ones(4), debug=2, verbose=False, seed=3)
    s = torch.stack(out)
    assert s.shape == (12, 512, D) and torch.isfinite(s).all()
    assert 0.6 < float(acc.mean()) <= 1.0


def test_predict_model_on_device(ht):
    """predict_model (S:1468-1562): predictions of every sample, stacked [S, N, O], and the per-sample log-probs."""
    net = make_net([3, 9, 1], "relu")
    X, Y = make_data(20, 3)
    Xd, Yd = X.to(dev()), Y.to(dev())
    theta0 = ht.util.flatten(net).detach().clone()
    samples = ht.sample_model(net, Xd, Yd, theta0, model_loss="regression", num_samples=6, num_steps_per_sample=3,
                              step_size=1e-3, tau_out=5.0, verbose=False, seed=2)
    pred, lps = ht.predict_model(net, samples, x=Xd, y=Yd, model_loss="regression", tau_out=5.0)
    assert pred.shape == (6, 20, 1) and len(lps) == 6
    o = O.MLPRegressionTarget([3, 9, 1], X.numpy(), Y.numpy(), [1.0] * 4, 5.0, 1.0, "relu")
    th = torch.stack(samples).cpu().numpy()
    np.testing.assert_allclose(pred.cpu().numpy(), o.predict(th), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose([float(v.sum()) for v in lps], o.logp(th), rtol=1e-4, atol=1e-3)
    with pytest.raises(RuntimeError):
        ht.predict_model(net, samples)


# ---- SPLITTING_RAND / SPLITTING_KMID --------------------------------------------------------
@pytest.mark.parametrize("kind", ["rand", "kmid"])
@pytest.mark.parametrize("route", ["mfma", "valu", "generic"])
def test_split_kinds_sample_vs_oracle(ht, kind, route):
    """sample_split_model with Integrator.SPLITTING_RAND / SPLITTING_KMID: the MFMA kernel, the VALU kernel and the
    generic-callback route against the oracle; the subset order of RAND is the Philox permutation of (seed, trajectory)."""
    from hamiltorch_amd import _abi
    dims, act, N, M, tau_out, eps, L, C, NS, seed = [4, 33, 1], "tanh", 40, 4, 6.0, 4e-3, 3, 20, 7, 17
    net, X, Y, loader, tau_list, otg = _split_setup(ht, dims, act, N, M, tau_out)
    D = sum(p.numel() for p in net.parameters())
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    integ = ht.Integrator.SPLITTING_RAND if kind == "rand" else ht.Integrator.SPLITTING_KMID
    kw = dict(model_loss="regression", num_samples=NS, num_steps_per_sample=L, step_size=eps, burn=1,
              inv_mass=torch.ones(D, device=dev()), tau_out=tau_out, tau_list=tau_list, integrator=integ, verbose=False, seed=seed)
    _abi.set_tuning("mlp_valu", 1 if route == "valu" else 0)
    try:
        out = ht.sample_split_model(net, loader, torch.tensor(th0, device=dev()), M, native=(route != "generic"), **kw)
    finally:
        _abi.set_tuning("mlp_valu", 0)
    ref, _ = O.sample_hmc(None, th0, NS, L, eps, 1, np.ones(D, np.float32), O.PhiloxDraws(seed, np.arange(C)),
                          grad_fns=[t.grad for t in otg], logp_fns=[t.logp for t in otg], split_kind=kind)
    _cmp(out, ref, 5e-4)


@pytest.mark.parametrize("kind", ["rand", "kmid"])
def test_split_kinds_leapfrog_api_vs_reference_fixture(ht, golden, kind):
    """samplers.leapfrog(integrator=SPLITTING_KMID) reproduces the reference's 3-step path; for SPLITTING_RAND the
    order is our own Philox permutation, so the path is checked against the oracle run with that order."""
    g = golden("splitkinds")
    M, tau_out, eps, L = g["cfg"]; M = int(M)
    net = make_net(list(g["dims"]), "relu")
    with torch.no_grad():
        torch.nn.utils.vector_to_parameters(torch.tensor(g["theta"], device=dev()), net.parameters())
    X, Y = torch.tensor(g["X"]), torch.tensor(g["Y"])
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=X.shape[0] // M, shuffle=False)
    from hamiltorch_amd import bnn, util
    sizes = [w.nelement() for w in net.parameters()]; shapes = [w.shape for w in net.parameters()]
    fl = bnn.define_split_model_log_prob(net, "regression", loader, M, sizes, shapes, torch.tensor(g["tau_list"]), float(tau_out),
                                         device=dev(), verbose=False)
    th, p0 = torch.tensor(g["theta"], device=dev()), torch.tensor(g["p0"], device=dev())
    integ = ht.Integrator.SPLITTING_RAND if kind == "rand" else ht.Integrator.SPLITTING_KMID
    util.set_random_seed(5)
    lp_, lm_ = ht.samplers.leapfrog(th, p0, fl, steps=3, step_size=float(eps), inv_mass=torch.ones_like(th),
                                    sampler=ht.Sampler.HMC, integrator=integ)
    got_t, got_p = torch.stack(lp_).cpu().numpy(), torch.stack(lm_).cpu().numpy()
    if kind == "kmid":
        np.testing.assert_allclose(got_t, g["kmid_lf_theta"], rtol=3e-5, atol=3e-5)
        np.testing.assert_allclose(got_p, g["kmid_lf_p"], rtol=3e-4, atol=3e-4)
    else:
        nb = X.shape[0] // M
        otg = [O.MLPRegressionTarget(list(g["dims"]), g["X"][m * nb:(m + 1) * nb], g["Y"][m * nb:(m + 1) * nb], g["tau_list"],
                                     float(tau_out), M, "relu") for m in range(M)]
        util.set_random_seed(5)
        perm = util.split_permutation(util.next_stream_seed(), 0, M)
        a, b = O.split_leapfrog(g["theta"][None], g["p0"][None], [t.grad for t in otg], 3, float(eps), np.ones_like(g["theta"]), "rand", perm)
        np.testing.assert_allclose(got_t[-1], a[0], rtol=3e-5, atol=3e-5)
        np.testing.assert_allclose(got_p[-1], b[0], rtol=3e-4, atol=3e-4)


# ---- Bernoulli-with-logits likelihood in the native kernels (model_loss='binary_class_linear_output', S:1172) -------------
@pytest.mark.parametrize("route", ["mfma", "valu", "f64"])
def test_native_binary_logits_likelihood(ht, golden, route):
    """hta_mlp_logp_grad with HTA_LOSS_BINARY_LOGITS: the reference's fixture (Linear(4,6)-Tanh-Linear(6,1), tau_out = 2) and
    the oracle on random batches (several shapes, splits, large logits) - the VALU kernel in fp32 (default route and forced:
    the MFMA kernel serves the Gaussian likelihood only) and fp64."""
    from hamiltorch_amd import _abi
    dtype = torch.float64 if route == "f64" else torch.float32
    npdt = np.float64 if route == "f64" else np.float32
    tol = 1e-10 if route == "f64" else 3e-4
    _abi.set_tuning("mlp_valu", 1 if route == "valu" else 0)
    try:
        g = golden("losses")
        X = torch.tensor(g["binary_X"], dtype=dtype, device=dev()); Y = torch.tensor(g["binary_Y"].reshape(-1), dtype=dtype, device=dev())
        th = torch.tensor(g["binary_theta"][None], dtype=dtype, device=dev())
        grad = torch.empty_like(th); lp = torch.empty(1, dtype=dtype, device=dev())
        _abi.mlp_logp_grad(th, 4, 6, "tanh", X, Y, 1, 10, 0, list(g["binary_tau_list"]), 2.0, 1.0, grad, lp, loss="binary_class_linear_output")
        np.testing.assert_allclose(lp.cpu().numpy(), g["binary_logp"], rtol=2e-5)
        np.testing.assert_allclose(grad[0].cpu().numpy(), g["binary_grad"], rtol=3e-4, atol=3e-5)
        for n_in, H, act, N, M, scale in [(3, 5, "relu", 12, 3, 0.4), (8, 100, "relu", 400, 4, 0.4), (5, 64, "sigmoid", 64, 1, 0.4), (2, 17, "tanh", 30, 2, 30.0)]:
            rng = np.random.default_rng(H)
            Xn = rng.standard_normal((N, n_in)).astype(npdt); Yn = (rng.uniform(size=(N, 1)) < 0.5).astype(npdt)
            D = H * n_in + 2 * H + 1
            C, Nb = 6, N // M
            theta = (scale * rng.standard_normal((C, D))).astype(npdt)          # scale 30: logits of order +-100 (no overflow)
            tau = [1.0, 1.5, 2.0, 2.5]
            thd = torch.tensor(theta, device=dev()); Xd = torch.tensor(Xn, device=dev()); Yd = torch.tensor(Yn.reshape(-1), device=dev())
            for m in range(M):
                gd = torch.empty_like(thd); lpd = torch.empty(C, dtype=dtype, device=dev())
                _abi.mlp_logp_grad(thd, n_in, H, act, Xd, Yd, M, Nb, m, tau, 3.0, float(M), gd, lpd, loss="binary_class_linear_output")
                o = O.MLPRegressionTarget([n_in, H, 1], Xn[m * Nb:(m + 1) * Nb], Yn[m * Nb:(m + 1) * Nb], tau, 3.0, float(M), act,
                                          loss="binary_class_linear_output")
                wl, wg = o.logp_and_grad(theta.astype(np.float64))
                np.testing.assert_allclose(lpd.cpu().numpy(), wl, rtol=tol, atol=tol * max(1.0, np.abs(wl).max()))
                np.testing.assert_allclose(gd.cpu().numpy(), wg, rtol=tol, atol=tol * max(1.0, np.abs(wg).max()))
    finally:
        _abi.set_tuning("mlp_valu", 0)


def test_sample_model_binary_classifier_native_vs_oracle_and_generic(ht):
    """sample_model(model_loss='binary_class_linear_output') on a one-hidden-layer classifier runs in the native kernel
    (one launch, no callbacks) and reproduces the oracle chain by chain; native=False (torch callbacks) agrees."""
    from hamiltorch_amd import _abi
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 9), torch.nn.Tanh(), torch.nn.Linear(9, 1)).to(dev())
    g = torch.Generator().manual_seed(1)
    X = torch.randn(40, 3, generator=g); Y = (X.sum(1, keepdim=True) + 0.3 * torch.randn(40, 1, generator=g) > 0).float()
    tau_list = torch.tensor([1.0, 1.5, 2.0, 2.5])
    D = sum(p.numel() for p in net.parameters())
    C, NS, L, eps, seed = 32, 8, 5, 2e-2, 21
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    kw = dict(model_loss="binary_class_linear_output", num_samples=NS, num_steps_per_sample=L, step_size=eps, tau_out=1.0,
              tau_list=tau_list, verbose=False, seed=seed)
    _abi.set_tuning("profile", 1)
    out = ht.sample_model(net, X, Y, torch.tensor(th0, device=dev()), **kw)
    torch.cuda.synchronize()
    _, launches = _abi.profile_collect()
    _abi.set_tuning("profile", 0)
    assert launches == 1, launches
    o = O.MLPRegressionTarget([3, 9, 1], X.numpy(), Y.numpy(), tau_list.numpy(), 1.0, 1.0, "tanh", loss="binary_class_linear_output")
    ref, info = O.sample_hmc(o, th0, NS, L, eps, 0, None, O.PhiloxDraws(seed, np.arange(C)))
    _cmp(out, ref, 5e-4)
    out_g = ht.sample_model(net, X, Y, torch.tensor(th0, device=dev()), native=False, **kw)
    _cmp(out_g, ref, 5e-4)
    assert 0.3 < info["acc_rate"].mean() <= 1.0

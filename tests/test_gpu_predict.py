"""GPU parity of the NATIVE predict_model (csrc/net_forward.hip through hta_net_forward + batched element-wise
log-probs, hamiltorch_amd/bnn.py) against the values the unmodified reference recorded (tests/golden/losses.npz: tensor and
DataLoader form, binary / multi-class likelihoods), against the oracle's forward pass, and against the torch path it replaces."""
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


def make_net(dims, act, seed=0, tail=None):
    torch.manual_seed(seed)
    layers = []
    for i in range(len(dims) - 1):
        layers.append(torch.nn.Linear(int(dims[i]), int(dims[i + 1])))
        if i < len(dims) - 2:
            layers.append({"relu": torch.nn.ReLU, "tanh": torch.nn.Tanh, "sigmoid": torch.nn.Sigmoid}[act]())
    if tail:
        layers.append(tail)
    return torch.nn.Sequential(*layers).to(dev())


@pytest.mark.parametrize("name,loss,dims,native", [("binary", "binary_class_linear_output", [4, 6, 1], True),
                                                  ("multi", "multi_class_linear_output", [4, 6, 3], True),
                                                  ("logsoftmax", "multi_class_log_softmax_output", [4, 6, 3], False)])
def test_predict_model_matches_the_reference_fixture(ht, golden, name, loss, dims, native):
    """losses.npz: hamiltorch.predict_model of the unmodified reference on three samples of a Linear-Tanh-Linear classifier,
    x / y form and DataLoader form (two batches: the prior enters each closure divided by the batch count, S:1520-1541).
    The recognised pairings run natively - asserted, not assumed - the log-softmax model stays on the torch path."""
    from hamiltorch_amd import bnn
    g = golden("losses")
    net = make_net(dims, "tanh", tail=torch.nn.LogSoftmax(dim=1) if name == "logsoftmax" else None)
    X, Y = torch.tensor(g[f"{name}_X"], device=dev()), torch.tensor(g[f"{name}_Y"], device=dev())
    samples = [torch.tensor(r, device=dev()) for r in g[f"{name}_samples"]]
    tau_list = torch.tensor(g[f"{name}_tau_list"])
    pred, lps = ht.predict_model(net, samples, x=X, y=Y, model_loss=loss, tau_out=2.0, tau_list=tau_list)
    assert bnn.predict_route["last"] == ("native" if native else "torch")
    np.testing.assert_allclose(pred.cpu().numpy(), g[f"{name}_pred"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(np.stack([v.reshape(-1).cpu().numpy() for v in lps]), g[f"{name}_pred_lp"], rtol=2e-5, atol=1e-4)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X.cpu(), Y.cpu()), batch_size=5, shuffle=False)
    pred, lps = ht.predict_model(net, samples, test_loader=loader, model_loss=loss, tau_out=2.0, tau_list=tau_list)
    assert bnn.predict_route["last"] == ("native" if native else "torch")
    np.testing.assert_allclose(pred.cpu().numpy(), g[f"{name}_pred_loader"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(np.stack([v.reshape(-1).numpy() for v in lps]), g[f"{name}_pred_loader_lp"], rtol=2e-5, atol=1e-4)
    assert all(not v.is_cuda for v in lps)                   # S:1536: the loader form's values live on the host


@pytest.mark.parametrize("dims,act,N,S,dtype", [([3, 9, 1], "relu", 20, 6, torch.float32), ([1, 100, 100, 1], "relu", 400, 70, torch.float32),
                                                ([8, 100, 1], "relu", 400, 33, torch.float32), ([4, 104, 97, 1], "sigmoid", 130, 9, torch.float32),
                                                ([3, 8, 2], "tanh", 65, 5, torch.float32), ([2, 1000, 1], "tanh", 64, 4, torch.float32),
                                                ([5, 16, 16, 16, 3], "relu", 31, 7, torch.float32), ([3, 9, 1], "tanh", 20, 6, torch.float64)])
def test_native_regression_predict_vs_oracle_and_torch_path(ht, dims, act, N, S, dtype, monkeypatch):
    """Regression likelihood: predictions [S, N, O] and per-sample log-probs (one value per output, S:1184) of the native path
    against the oracle's forward pass and log-density, and against the torch path on the same inputs (vmap of the closure) -
    the reference's notebook model (1-100-100-1, 400 points), BASELINE config 4's model (8-100-1), wide / deep / multi-output
    nets, a hidden layer beyond any LDS staging (1000 units, streamed), ragged point counts, fp64."""
    from hamiltorch_amd import bnn
    npdt = np.float32 if dtype == torch.float32 else np.float64
    net = make_net(dims, act).to(dtype)
    rng = np.random.default_rng(len(dims) * 100 + N)
    X = rng.standard_normal((N, dims[0])).astype(npdt); Y = rng.standard_normal((N, dims[-1])).astype(npdt)
    D = sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))
    th = (0.4 * rng.standard_normal((S, D))).astype(npdt)
    nt = 2 * (len(dims) - 1)
    tau_list = [1.0 + 0.25 * k for k in range(nt)]
    Xd, Yd = torch.tensor(X, device=dev()), torch.tensor(Y, device=dev())
    samples = [torch.tensor(r, device=dev()) for r in th]
    pred, lps = ht.predict_model(net, samples, x=Xd, y=Yd, model_loss="regression", tau_out=3.0, tau_list=torch.tensor(tau_list))
    assert bnn.predict_route["last"] == "native"
    assert pred.shape == (S, N, dims[-1]) and len(lps) == S and tuple(lps[0].shape) == (dims[-1],)
    tol = 2e-4 if dtype == torch.float32 else 1e-10
    if dims[-1] == 1:
        o = O.MLPRegressionTarget(dims, X, Y, tau_list, 3.0, 1.0, act)
        want = o.predict(th.astype(np.float64))
        np.testing.assert_allclose(pred.cpu().numpy(), want, rtol=tol, atol=tol * max(1.0, np.abs(want).max()))
        wl = o.logp(th.astype(np.float64))
        np.testing.assert_allclose([float(v.sum()) for v in lps], wl, rtol=tol, atol=tol * max(1.0, np.abs(wl).max()))
    monkeypatch.setattr(bnn, "_native_forward_ok", lambda *a, **k: None)
    pred_t, lps_t = ht.predict_model(net, samples, x=Xd, y=Yd, model_loss="regression", tau_out=3.0, tau_list=torch.tensor(tau_list))
    assert bnn.predict_route["last"] == "torch"
    np.testing.assert_allclose(pred.cpu().numpy(), pred_t.cpu().numpy(), rtol=tol, atol=tol * max(1.0, float(pred_t.abs().max())))
    a, b = torch.stack(lps).cpu().numpy(), torch.stack([v.reshape(-1) for v in lps_t]).cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol * max(1.0, np.abs(b).max()))


def test_predict_model_reads_sample_models_lazy_list_natively(ht):
    """sample_model -> predict_model, the notebooks' sequence: the lazy list's backing tensor goes to the kernel as it is
    (no per-row objects), the result equals the per-row torch evaluation."""
    from hamiltorch_amd import bnn
    from hamiltorch_amd.samplelist import SampleList
    net = make_net([3, 9, 1], "relu")
    g = torch.Generator().manual_seed(1)
    X = torch.randn(20, 3, generator=g).to(dev()); Y = torch.sin(X.sum(1, keepdim=True))
    theta0 = ht.util.flatten(net).detach().clone()
    samples = ht.sample_model(net, X, Y, theta0, model_loss="regression", num_samples=80, num_steps_per_sample=3, step_size=1e-3,
                              tau_out=5.0, verbose=False, seed=2)
    assert isinstance(samples, SampleList) and not samples._done
    pred, lps = ht.predict_model(net, samples, x=X, y=Y, model_loss="regression", tau_out=5.0)
    assert bnn.predict_route["last"] == "native" and not samples._done and pred.shape == (80, 20, 1)
    rows = [r.clone() for r in list(samples)]
    with torch.no_grad():
        want = torch.stack([torch.func.functional_call(net, dict(zip([n for n, _ in net.named_parameters()],
                                                                    ht.util.unflatten(net, r))), (X,)) for r in rows])
    np.testing.assert_allclose(pred.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dims,dtype,loss", [([784, 100, 10], torch.float32, "multi_class_linear_output"),
                                            ([784, 512, 10], torch.float32, "multi_class_linear_output"),
                                            ([1, 200, 200, 1], torch.float64, "regression")])
def test_models_outside_the_forward_kernels_limits_take_the_torch_path(ht, dims, dtype, loss):
    """a flattened-MNIST MLP (784 inputs: 401 KB of LDS staging) and 200-wide float64 nets passed the native gate
    and predict_model RAISED (InvalidArguments / a failed launch); before the native route they ran under vmap.  The gate mirrors
    the kernel's limits now (bnn.native_forward_fits) and a refusal by the kernel falls back as well: these models predict on the torch
    path, values equal to a plain forward pass."""
    from hamiltorch_amd import bnn
    net = make_net(dims, "relu").to(dtype)
    D = sum(p.numel() for p in net.parameters())
    g = torch.Generator().manual_seed(1)
    S, N = 3, 17
    samples = [0.05 * torch.randn(D, generator=g, dtype=dtype).to(dev()) for _ in range(S)]
    X = torch.randn(N, dims[0], generator=g, dtype=dtype).to(dev())
    Y = torch.randint(0, dims[-1], (N,), generator=g).to(dev()) if loss != "regression" else torch.randn(N, 1, generator=g, dtype=dtype).to(dev())
    pred, lps = ht.predict_model(net, samples, x=X, y=Y, model_loss=loss, tau_out=1.0, tau_list=torch.ones(2 * (len(dims) - 1)))
    assert bnn.predict_route["last"] == "torch"
    assert tuple(pred.shape) == (S, N, dims[-1]) and len(lps) == S
    for s_ in range(S):
        torch.nn.utils.vector_to_parameters(samples[s_], net.parameters())
        np.testing.assert_allclose(pred[s_].cpu().numpy(), net(X).detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    # and if the kernel itself refuses (a limit the gate does not know), the call still answers
    keep = bnn.native_forward_fits
    bnn.native_forward_fits = lambda *a, **k: True
    try:
        pred2, _ = ht.predict_model(net, samples, x=X, y=Y, model_loss=loss, tau_out=1.0, tau_list=torch.ones(2 * (len(dims) - 1)))
    finally:
        bnn.native_forward_fits = keep
    assert bnn.predict_route["last"] == "torch" and torch.allclose(pred2, pred)

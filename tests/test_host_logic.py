"""CPU: host-side logic of the API mirror (argument checks, list lengths, closures) -- no kernels run."""
import os

import numpy as np
import pytest
import torch

import hamiltorch_amd as ht
from hamiltorch_amd import bnn, samplers, util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_reference_init():
    # hamiltorch/__init__.py:3-4
    for name in ("sample", "sample_model", "predict_model", "sample_split_model", "Sampler", "Integrator", "Metric",
                 "set_random_seed"):
        assert hasattr(ht, name)
    assert [e.value for e in ht.Sampler] == [1, 2, 3]
    assert [e.name for e in ht.Integrator] == ["EXPLICIT", "IMPLICIT", "S3", "SPLITTING", "SPLITTING_RAND", "SPLITTING_KMID"]
    assert [e.name for e in ht.Metric] == ["HESSIAN", "SOFTABS", "JACOBIAN_DIAG"]
    for name in ("leapfrog", "hamiltonian", "rm_hamiltonian", "fisher", "gibbs", "cholesky_inverse", "acceptance",
                 "collect_gradients", "define_model_log_prob", "define_split_model_log_prob"):
        assert hasattr(samplers, name)
    for name in ("LogProbError", "flatten", "unflatten", "setup_chain", "multi_chain", "set_random_seed", "has_nan_or_inf"):
        assert hasattr(util, name)
    # hamiltorch/__init__.py:1 - a script that gates on hamiltorch.__version__ sees the API level this package mirrors
    sig = __import__("json").load(open(os.path.join(ROOT, "tests", "golden", "signatures.json")))
    assert ht.__version__ == "0.4.1" == sig.get("__version__", "0.4.1") and ht.__amd_version__ != ht.__version__


def test_argument_errors_match_reference():
    f = lambda w: -(w * w).sum()  # noqa: E731
    with pytest.raises(RuntimeError, match="burn must be less than num_samples"):      # S:928-929
        ht.sample(f, torch.zeros(3), num_samples=5, burn=5)
    with pytest.raises(RuntimeError, match="burn must be greater than 0 for NUTS"):    # S:933-934
        ht.sample(f, torch.zeros(3), num_samples=5, sampler=ht.Sampler.HMC_NUTS)
    with pytest.raises(RuntimeError, match="params_init must be a 1d tensor"):         # S:925-926
        ht.sample(f, torch.zeros(2, 2, 2), num_samples=5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ht.sample(f, torch.zeros(3), num_samples=5)
    with pytest.raises(ValueError):                                                    # U:126-127
        util.unflatten(torch.nn.Linear(2, 2), torch.zeros(2, 3))


def test_row_count_matches_reference_list_length():
    # measured on the reference: burn=0 -> N, burn=10 -> N-10, burn=-1 -> N+1
    assert samplers._num_rows(400, 0) == 400
    assert samplers._num_rows(60, 10) == 50
    assert samplers._num_rows(40, -1) == 41


def test_has_nan_or_inf_and_flatten_roundtrip():
    assert util.has_nan_or_inf(torch.tensor([1.0, float("inf")]))
    assert util.has_nan_or_inf(torch.tensor([float("inf"), -float("inf")]))    # sum is NaN (U:94)
    assert not util.has_nan_or_inf(torch.tensor([1.0, 2.0]))
    m = torch.nn.Linear(4, 4)
    flat = util.flatten(m)
    parts = util.unflatten(m, flat)
    util.update_model_params_in_place(m, parts)
    assert torch.equal(util.flatten(m), flat)                                  # tests/test_util.py:12-24


def test_stream_seed_is_reproducible():
    ht.set_random_seed(5)
    a = [util.next_stream_seed() for _ in range(3)]
    ht.set_random_seed(5)
    b = [util.next_stream_seed() for _ in range(3)]
    assert a == b and len(set(a)) == 3


def _net(dims, act):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(torch.nn.Linear(int(dims[i]), int(dims[i + 1])))
        if i < len(dims) - 2:
            layers.append({"relu": torch.nn.ReLU, "tanh": torch.nn.Tanh}[act]())
    return torch.nn.Sequential(*layers)


@pytest.mark.parametrize("name,act", [("relu2", "relu"), ("tanh3", "tanh")])
def test_bnn_closures_match_reference_fixture(golden, name, act):
    """define_model_log_prob / define_split_model_log_prob (torch, CPU) vs values recorded from the reference."""
    g = golden("mlp")
    M, tau_out, eps, L = g[f"{name}_cfg"]
    net = _net(g[f"{name}_dims"], act)
    theta = torch.tensor(g[f"{name}_theta"])
    X, Y = torch.tensor(g[f"{name}_X"]), torch.tensor(g[f"{name}_Y"])
    tau_list = torch.tensor(g[f"{name}_tau_list"])
    sizes = [w.nelement() for w in net.parameters()]
    shapes = [w.shape for w in net.parameters()]
    f = bnn.define_model_log_prob(net, "regression", X, Y, sizes, shapes, tau_list, float(tau_out))
    th = theta.clone().requires_grad_()
    v = f(th)
    np.testing.assert_allclose(v.detach().numpy().reshape(-1), g[f"{name}_logp"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(torch.autograd.grad(v.sum(), th)[0].numpy(), g[f"{name}_grad"], rtol=1e-4, atol=1e-4)
    assert f._hta_spec["dims"] == list(g[f"{name}_dims"]) and f._hta_spec["act"] == act
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=X.shape[0] // int(M), shuffle=False)
    fl = bnn.define_split_model_log_prob(net, "regression", loader, int(M), sizes, shapes, tau_list, float(tau_out), verbose=False)
    assert len(fl) == int(M)
    np.testing.assert_allclose([float(fm(theta).sum()) for fm in fl], g[f"{name}_split_logp"], rtol=1e-5, atol=1e-4)
    # batched evaluation used by the generic path
    cb = samplers._BatchedCallback(f)
    gb, lb = cb.grad(theta.repeat(3, 1))
    np.testing.assert_allclose(gb[1].numpy(), g[f"{name}_grad"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lb.numpy(), np.repeat(g[f"{name}_logp"], 3), rtol=1e-5, atol=1e-4)


def test_batched_callback_loop_fallback_and_pass_grad():
    def f(w):                      # .item() defeats vmap -> per-chain loop
        return -(w * w).sum() * float(torch.tensor(1.0).item()) + 0 * float(w[0].item())
    cb = samplers._BatchedCallback(f)
    th = torch.tensor([[1.0, 2.0], [3.0, 4.0]])
    with pytest.warns(UserWarning):
        g, v = cb.grad(th)
    np.testing.assert_allclose(g.numpy(), -2 * th.numpy())
    np.testing.assert_allclose(v.numpy(), [-5.0, -25.0])
    cb2 = samplers._BatchedCallback(lambda w: -(w * w).sum(), pass_grad=lambda w: -2 * w)
    g2, _ = cb2.grad(th)
    np.testing.assert_allclose(g2.numpy(), -2 * th.numpy())


def test_gaussian_target_is_a_valid_callback():
    sigma = torch.tensor([[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]])
    t = ht.GaussianTarget(torch.zeros(3), covariance=sigma)
    w = torch.tensor([0.3, -0.2, 0.5])
    ref = torch.distributions.MultivariateNormal(torch.zeros(3), sigma).log_prob(w)
    assert abs(float(t(w)) - float(ref)) < 1e-5                    # KAT2 logp = -2.97352004
    assert abs(float(t(w)) + 2.97352004) < 1e-5
    from hamiltorch_amd.models import as_gaussian
    d = torch.distributions.MultivariateNormal(torch.zeros(3), sigma)
    t2 = as_gaussian(d.log_prob)
    assert t2 is not None and torch.allclose(t2.precision, t.precision, atol=1e-5)
    assert as_gaussian(lambda w: w.sum()) is None


def test_ess_matches_oracle_definition():
    import hmc_oracle as O
    from hamiltorch_amd.ess import ess_bulk, ess_min
    g = torch.Generator().manual_seed(0)
    S, C = 400, 6
    x = torch.zeros(S, C, dtype=torch.float64)
    e = torch.randn(S, C, generator=g, dtype=torch.float64)
    for t in range(1, S):
        x[t] = 0.8 * x[t - 1] + e[t]                    # AR(1): ESS ~ S*C*(1-.8)/(1+.8)
    a, b = ess_bulk(x), O.ess_bulk(x.numpy())
    assert abs(a - b) / b < 1e-9
    assert 0.4 * S * C / 9 < a < 2.5 * S * C / 9
    iid = torch.randn(S, C, 2, generator=g, dtype=torch.float64)
    assert 0.7 * S * C < ess_min(iid) < 1.4 * S * C


def _custom_loss(out, y):                    # the user log-likelihood of oracle/gen_golden.py:custom_loss
    return 1.5 * ((out - y) ** 2).sum(1)


LOSS_CASES = {"binary": ("binary_class_linear_output", [4, 6, 1], False), "multi": ("multi_class_linear_output", [4, 6, 3], False),
              "logsoftmax": ("multi_class_log_softmax_output", [4, 6, 3], True), "custom": (_custom_loss, [4, 6, 2], False)}


@pytest.mark.parametrize("name", sorted(LOSS_CASES))
def test_model_loss_kinds_and_predict_model_match_reference_fixture(golden, name):
    """Every model_loss kind besides 'regression' (S:1170-1190), the prior-only closure (S:1160-1162) and predict_model in its
    tensor and DataLoader forms (S:1468-1562; the loader form adds the prior once per batch, as the reference does) against
    values recorded from the unmodified reference (tests/golden/losses.npz)."""
    g = golden("losses")
    loss, dims, log_softmax = LOSS_CASES[name]
    net = _net(dims, "tanh")
    if log_softmax:
        net = torch.nn.Sequential(*list(net.children()), torch.nn.LogSoftmax(dim=1))
    X, Y, theta = torch.tensor(g[f"{name}_X"]), torch.tensor(g[f"{name}_Y"]), torch.tensor(g[f"{name}_theta"])
    tau_list = torch.tensor(g[f"{name}_tau_list"])
    sizes = [w.nelement() for w in net.parameters()]
    shapes = [w.shape for w in net.parameters()]
    f = bnn.define_model_log_prob(net, loss, X, Y, sizes, shapes, tau_list, 2.0)
    th = theta.clone().requires_grad_()
    v = f(th)
    np.testing.assert_allclose(v.detach().numpy().reshape(-1), g[f"{name}_logp"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(torch.autograd.grad(v.sum(), th)[0].numpy(), g[f"{name}_grad"], rtol=1e-4, atol=1e-4)
    f0 = bnn.define_model_log_prob(net, loss, None, None, sizes, shapes, tau_list, 2.0, prior_scale=3.0)
    np.testing.assert_allclose(f0(theta).detach().numpy().reshape(-1), g[f"{name}_prior_only"], rtol=1e-5, atol=1e-5)
    samples = [torch.tensor(s) for s in g[f"{name}_samples"]]
    pred, lps = ht.predict_model(net, samples, x=X, y=Y, model_loss=loss, tau_out=2.0, tau_list=tau_list)
    np.testing.assert_allclose(pred.numpy(), g[f"{name}_pred"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(np.stack([t.numpy().reshape(-1) for t in lps]), g[f"{name}_pred_lp"], rtol=1e-5, atol=1e-4)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=5, shuffle=False)
    pred, lps = ht.predict_model(net, samples, test_loader=loader, model_loss=loss, tau_out=2.0, tau_list=tau_list)
    np.testing.assert_allclose(pred.numpy(), g[f"{name}_pred_loader"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(np.stack([t.numpy().reshape(-1) for t in lps]), g[f"{name}_pred_loader_lp"], rtol=1e-5, atol=1e-4)
    before = dict(bnn.predict_stats)
    ht.predict_model(net, samples, x=X, y=Y, model_loss=loss, tau_out=2.0, tau_list=tau_list)
    assert bnn.predict_stats["batched"] == before["batched"] + 1 and bnn.predict_stats["looped"] == before["looped"], \
        "predict_model fell back to the per-sample loop for model_loss=%r" % (loss,)
    with pytest.raises(RuntimeError):                                            # S:1557: no data at all
        ht.predict_model(net, samples)
    with pytest.raises(NotImplementedError):                                     # S:1190: unknown loss
        bnn.define_model_log_prob(net, "no_such_loss", X, Y, sizes, shapes, tau_list, 2.0)(theta)


def test_tuple_protocol_callback():
    """log_prob_func may return (log_prob, iterable_of_params) whose .grad's are concatenated (S:54-58): the batched
    evaluator falls back to the per-chain loop and collect_gradients follows the reference."""
    def f(w):
        a = w[:2].detach().requires_grad_()
        b = w[2:].detach().requires_grad_()
        return -(a * a).sum() - 2.0 * (b * b).sum(), [a, b]
    th = torch.tensor([[1.0, 2.0, 3.0], [-1.0, 0.5, 0.25]])
    cb = samplers._BatchedCallback(f)
    with pytest.warns(UserWarning):
        g, v = cb.grad(th)
    np.testing.assert_allclose(g.numpy(), np.concatenate([-2 * th[:, :2].numpy(), -4 * th[:, 2:].numpy()], axis=1))
    np.testing.assert_allclose(v.numpy(), [-(1 + 4) - 2 * 9, -(1 + 0.25) - 2 * 0.0625])
    np.testing.assert_allclose(cb.logp(th).numpy(), v.numpy())
    p = torch.tensor([1.0, 2.0, 3.0], requires_grad=True)
    out = samplers.collect_gradients(f(p), p)
    np.testing.assert_allclose(out.grad.numpy(), [-2.0, -4.0, -12.0])
    # a fixed gradient tensor and a callable (S:59-63)
    assert torch.equal(samplers.collect_gradients(None, p.detach(), torch.ones(3)).grad, torch.ones(3))
    assert torch.equal(samplers.collect_gradients(None, p.detach(), lambda w: 2 * w).grad, 2 * p.detach())


def test_public_signatures_match_reference():
    """Every public function of the path keeps the reference's parameter names, order and defaults (extra keyword
    parameters - seed, chain_offset, ... - may follow), and the enums keep their members and values
    (tests/golden/signatures.json, recorded from the unmodified reference by oracle/gen_golden.py)."""
    import inspect
    import json
    import os
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "signatures.json")))
    enums = ref.pop("enums")
    for ename, members in enums.items():
        mine = {m.name: m.value for m in getattr(ht, ename)}
        assert mine == members, ename
    for name, params in ref.items():
        obj = ht
        for part in name.split("."):
            obj = getattr(obj, part)
        mine = inspect.signature(obj).parameters
        order = [k for k in mine if k in {p[0] for p in params}]
        assert order == [p[0] for p in params], name
        for k, default in params:
            got = mine[k].default
            assert (default is None and got is inspect._empty) or repr(got) == default, (name, k, default, got)


# ---- recognition of the reference's own example closures (models.probe_gaussian) -------------------------------------
def _ref_closure():                       # tests/test_util.py:98-101 of the reference, verbatim in structure
    def log_prob(omega):
        mean = torch.zeros(2)
        var = torch.tensor([.10, .10])
        return torch.distributions.MultivariateNormal(mean, torch.diag(var)).log_prob(omega).sum()
    return log_prob


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 1e-12)])
def test_probe_recognises_reference_closures(dtype, tol):
    from hamiltorch_amd.models import GaussianTarget, probe_gaussian, verify_gaussian
    t = probe_gaussian(_ref_closure(), torch.ones(2)) if dtype == torch.float32 else None
    if t is not None:
        np.testing.assert_allclose(t.precision.numpy(), np.diag([10.0, 10.0]), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(t.mean.numpy(), 0.0, atol=1e-5)
    # the notebooks' idiom: a correlated MVN with a mean, batched initial state
    mean = torch.tensor([0.3, -1.0, 2.0], dtype=dtype)
    cov = torch.tensor([[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]], dtype=dtype)
    f = lambda w: torch.distributions.MultivariateNormal(mean, cov).log_prob(w).sum()  # noqa: E731
    t = probe_gaussian(f, torch.zeros(16, 3, dtype=dtype))
    ref = GaussianTarget(mean, covariance=cov)
    assert t is not None and t.mean.dtype == dtype
    np.testing.assert_allclose(t.precision.numpy(), ref.precision.numpy(), rtol=tol * 10, atol=tol * 10)
    np.testing.assert_allclose(t.mean.numpy(), mean.numpy(), rtol=tol * 10, atol=tol * 10)
    assert abs(t.log_norm - ref.log_norm) < tol * 100
    x = torch.randn(7, 5, 3, dtype=dtype)
    assert verify_gaussian(t, f, x)
    # bare quadratic form, D = 100 (BASELINE config 3's log_prob as the survey writes it)
    g = torch.Generator().manual_seed(0)
    Q = torch.linalg.qr(torch.randn(100, 100, generator=g, dtype=torch.float64))[0]
    P = ((Q * torch.linspace(0.5, 2.0, 100, dtype=torch.float64)) @ Q.T).to(dtype)
    t = probe_gaussian(lambda w: -0.5 * torch.dot(w, torch.mv(P, w)), torch.zeros(4, 100, dtype=dtype))
    assert t is not None
    np.testing.assert_allclose(t.precision.numpy(), (0.5 * (P + P.T)).numpy(), rtol=0, atol=tol * 10)
    assert float(t.mean.abs().max()) < tol * 1e3 and abs(t.log_norm) < tol * 1e3


def test_probe_rejects_everything_else():
    from hamiltorch_amd.models import probe_gaussian, verify_gaussian
    th = torch.zeros(4, 5)
    assert probe_gaussian(lambda w: -(w ** 4).sum(), th) is None                                     # not quadratic
    assert probe_gaussian(lambda w: -0.5 * (w * w).sum() - 0.1 * torch.cos(w).sum(), th) is None      # curvature varies
    assert probe_gaussian(lambda w: w.sum(), th) is None                                             # linear: no curvature
    assert probe_gaussian(lambda w: -0.5 * w[0] * w[0] / 9 - 0.5 * (w[1:] ** 2).sum() * torch.exp(-w[0]), th) is None   # funnel
    assert probe_gaussian(lambda w: -0.5 * float((w * w).sum()) * torch.ones(()), th) is None         # not differentiable by torch.func
    assert probe_gaussian(lambda w: (-(w * w).sum(), [w]), th) is None                               # tuple protocol
    assert probe_gaussian([lambda w: -(w * w).sum()], th) is None                                    # split list
    assert probe_gaussian(lambda w: -(w * w).sum(), torch.zeros(2, 2000)) is None                    # beyond the fused kernels
    # piecewise quadratic with every probe inside one piece: passes the probe, caught on the samples
    hub = lambda w: -torch.nn.functional.huber_loss(w, torch.zeros_like(w), reduction="sum", delta=5000.0)  # noqa: E731
    t = probe_gaussian(hub, th)
    assert t is not None
    assert verify_gaussian(t, hub, torch.randn(6, 4, 5))
    assert not verify_gaussian(t, hub, 1e4 * torch.randn(6, 4, 5))


def test_graphed_callable_is_transparent_on_cpu_and_logs_refusals():
    g = util.GraphedCallable(lambda w: (w * w).sum(-1))
    x = torch.randn(3, 4)
    assert torch.equal(g(x), (x * x).sum(-1)) and g.cache == {}          # host tensors: plain call, nothing captured
    assert isinstance(util.graph_log, list)


def test_model_structure_recognition_of_notebook_style_modules():
    """bnn._mlp_structure: Sequential chains and plain nn.Module classes whose forward() is Linear / activation calls in order
    (the reference notebooks' Net) are recognised by tracing; residual connections, parameters flattened in another order than
    they are used, mixed activations, convolutions are not (they keep the callback path)."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from hamiltorch_amd import bnn

    class Net(nn.Module):
        def __init__(self, ls):
            super().__init__()
            self.l1 = nn.Linear(ls[0], ls[1]); self.l2 = nn.Linear(ls[1], ls[2]); self.l3 = nn.Linear(ls[2], ls[3])

        def forward(self, x):
            x = self.l1(x); x = torch.relu(x); x = self.l2(x); x = torch.relu(x); x = self.l3(x)
            return x

    class Lin(nn.Module):
        def __init__(self):
            super().__init__(); self.l1 = nn.Linear(4, 3)

        def forward(self, x):
            return self.l1(x)

    class Fn(nn.Module):
        def __init__(self):
            super().__init__(); self.a = nn.Linear(2, 5); self.b = nn.Linear(5, 2)

        def forward(self, x):
            return self.b(F.relu(self.a(x)))

    class Meth(nn.Module):
        def __init__(self):
            super().__init__(); self.a = nn.Linear(2, 5); self.b = nn.Linear(5, 2)

        def forward(self, x):
            return self.b(self.a(x).tanh())

    class Swapped(nn.Module):                       # flattening order (l2, l1) is not the order of use
        def __init__(self):
            super().__init__(); self.l2 = nn.Linear(3, 1); self.l1 = nn.Linear(4, 3)

        def forward(self, x):
            return self.l2(torch.tanh(self.l1(x)))

    class Res(nn.Module):
        def __init__(self):
            super().__init__(); self.l1 = nn.Linear(4, 4)

        def forward(self, x):
            return self.l1(x) + x

    class Mixed(nn.Module):
        def __init__(self):
            super().__init__(); self.a = nn.Linear(2, 3); self.b = nn.Linear(3, 3); self.c = nn.Linear(3, 1)

        def forward(self, x):
            return self.c(torch.tanh(self.b(torch.relu(self.a(x)))))

    class Conv(nn.Module):
        def __init__(self):
            super().__init__(); self.c = nn.Conv2d(1, 2, 3); self.f = nn.Linear(8, 2)

        def forward(self, x):
            return self.f(torch.relu(self.c(x)).view(-1, 8))

    assert bnn._mlp_structure(Net([1, 10, 10, 1])) == ([1, 10, 10, 1], "relu")
    assert bnn._mlp_structure(Lin()) == ([4, 3], "relu")
    assert bnn._mlp_structure(Fn()) == ([2, 5, 2], "relu")
    assert bnn._mlp_structure(Meth()) == ([2, 5, 2], "tanh")
    assert bnn._mlp_structure(nn.Sequential(nn.Linear(3, 5), nn.Sigmoid(), nn.Linear(5, 1))) == ([3, 5, 1], "sigmoid")
    for bad in (Swapped(), Res(), Mixed(), Conv(), nn.Sequential(nn.Linear(3, 5, bias=False), nn.ReLU(), nn.Linear(5, 1))):
        assert bnn._mlp_structure(bad) is None, type(bad).__name__
    # a log-softmax over the outputs closes the chain (module, function or method form; dim 1 only)
    class LsmF(nn.Module):
        def __init__(self):
            super().__init__(); self.a = nn.Linear(4, 6); self.b = nn.Linear(6, 3)

        def forward(self, x):
            return F.log_softmax(self.b(torch.tanh(self.a(x))), dim=1)

    class LsmDim0(nn.Module):
        def __init__(self):
            super().__init__(); self.a = nn.Linear(4, 3)

        def forward(self, x):
            return F.log_softmax(self.a(x), dim=0)

    assert bnn._mlp_structure(nn.Sequential(nn.Linear(4, 6), nn.Tanh(), nn.Linear(6, 3), nn.LogSoftmax(dim=1))) == ([4, 6, 3], "tanh", "log_softmax")
    assert bnn._mlp_structure(LsmF()) == ([4, 6, 3], "tanh", "log_softmax")
    assert bnn._mlp_structure(LsmDim0()) is None
    lsm = nn.Sequential(nn.Linear(4, 6), nn.Tanh(), nn.Linear(6, 3), nn.LogSoftmax(dim=1))
    ls_sizes = [w.nelement() for w in lsm.parameters()]; ls_shapes = [w.shape for w in lsm.parameters()]
    yl = torch.randint(0, 3, (12, 1)).float()
    f = bnn.define_model_log_prob(lsm, "multi_class_log_softmax_output", torch.randn(12, 4), yl, ls_sizes, ls_shapes, [1.0] * 4, 3.0)
    assert f._hta_spec["loss"] == "multi_class_linear_output" and abs(f._hta_spec["tau_out"] - 3.0 / 12) < 1e-12     # nll_loss: mean reduction (S:1180)
    f = bnn.define_model_log_prob(lsm, "multi_class_linear_output", torch.randn(12, 4), yl, ls_sizes, ls_shapes, [1.0] * 4, 3.0)
    assert not hasattr(f, "_hta_spec")
    # the closures carry a native spec exactly for the likelihoods with a kernel
    X = torch.randn(12, 4); yc = torch.randint(0, 3, (12, 1)).float()
    net = Lin(); sizes = [w.nelement() for w in net.parameters()]; shapes = [w.shape for w in net.parameters()]
    f = bnn.define_model_log_prob(net, "multi_class_linear_output", X, yc, sizes, shapes, [1.0, 1.0], 1.0)
    assert f._hta_spec["dims"] == [4, 3] and f._hta_spec["loss"] == "multi_class_linear_output" and f._hta_spec["Y"].shape == (12,)
    for bad in (yc + 0.5, yc - 1.0, yc + 3.0):      # non-integer / negative / >= n_out labels never reach the kernel
        f = bnn.define_model_log_prob(net, "multi_class_linear_output", X, bad, sizes, shapes, [1.0, 1.0], 1.0)
        assert not hasattr(f, "_hta_spec")
    f = bnn.define_model_log_prob(net, "regression", X, torch.randn(12, 3), sizes, shapes, [1.0, 1.0], 1.0)
    assert not hasattr(f, "_hta_spec")              # multi-output regression returns one value per output (S:1184): callback path
    from hamiltorch_amd import mlp
    assert mlp._kernel_for(dict(dims=[8, 100, 1], loss="regression")) == "mlp1"
    assert mlp._kernel_for(dict(dims=[1, 10, 10, 1], loss="regression")) == "netn"
    assert mlp._kernel_for(dict(dims=[4, 3], loss="multi_class_linear_output")) == "netn"
    assert mlp._kernel_for(dict(dims=[8, 100, 3], loss="multi_class_linear_output")) is None       # wider than the small-net kernel
    assert mlp._kernel_for(dict(dims=[2, 3, 3, 3, 3, 1], loss="regression")) is None               # five Linear layers


def test_verify_gaussian_catches_a_unit_bump_on_a_wide_gaussian():
    """a D=50 fp32 MVN log-prob plus exp(-0.5 ((w_k - 1.5) / 0.15)^2).  Wherever the probe accepts it, the
    verification on Gaussian samples (which visit the bump) must reject: the tolerance is rounding-sized, not 1 nat."""
    from hamiltorch_amd.models import GaussianTarget, probe_gaussian, verify_gaussian
    D = 50
    g = torch.Generator().manual_seed(3)
    A = torch.randn(D, D, generator=g)
    cov = A @ A.T / D + 0.5 * torch.eye(D)
    mvn = torch.distributions.MultivariateNormal(torch.zeros(D), cov)
    exact = GaussianTarget(torch.zeros(D), covariance=cov)
    xs = mvn.sample((40, 16))
    xs[0, 0, :] = 0.0
    assert verify_gaussian(exact, lambda w: mvn.log_prob(w).sum(), xs)                 # the true closure still verifies
    accepted = 0
    for k in range(12):
        def bumpy(w, k=k):
            return mvn.log_prob(w).sum() + 1.0 * torch.exp(-0.5 * ((w[k] - 1.5) / 0.15) ** 2)
        t = probe_gaussian(bumpy, torch.zeros(4, D))
        if t is None:
            continue
        accepted += 1
        x = xs.clone()
        x[1, :, k] = 1.5 + 0.1 * torch.randn(16, generator=g)                        # rows inside the bump, as HMC would visit
        assert not verify_gaussian(t, bumpy, x), k
    # a smooth 1e-2-nat ripple is also beyond rounding
    t = probe_gaussian(lambda w: mvn.log_prob(w).sum(), torch.zeros(4, D))
    assert t is not None
    assert not verify_gaussian(t, lambda w: mvn.log_prob(w).sum() + 1e-2 * torch.sin(3.0 * w[0]), xs)


def test_predict_eval_all_survives_out_of_memory(monkeypatch):
    """an out-of-memory error in the batched evaluation halves the chunk (down to the reference's
    per-sample loop) instead of propagating."""
    from hamiltorch_amd import bnn
    samples = [torch.full((3,), float(k)) for k in range(20)]

    def f(w):
        return w.sum(), torch.stack([w[0], 2 * w[1]])
    real_vmap = torch.func.vmap
    seen = []

    def picky_vmap(fn, limit):
        def run(batch):
            seen.append(batch.shape[0])
            if batch.shape[0] > limit:
                raise torch.OutOfMemoryError("simulated")
            return real_vmap(fn)(batch)
        return run
    monkeypatch.setattr(bnn.torch.func, "vmap", lambda fn: picky_vmap(fn, 3))
    bnn.predict_stats.pop("oom_halvings", None)
    before = dict(bnn.predict_stats)
    vs, os_ = bnn._eval_all(f, samples, torch.device("cpu"))
    assert [float(v) for v in vs] == [3.0 * k for k in range(20)]
    assert torch.equal(torch.stack(os_)[:, 1], 2.0 * torch.arange(20.0))
    assert seen[0] == 20 and max(seen[-5:]) <= 3 and bnn.predict_stats["oom_halvings"] >= 2
    assert bnn.predict_stats["batched"] == before["batched"] + 1
    # even one sample under vmap does not fit: the reference's loop
    seen.clear()
    monkeypatch.setattr(bnn.torch.func, "vmap", lambda fn: picky_vmap(fn, 0))
    vs, os_ = bnn._eval_all(f, samples, torch.device("cpu"))
    assert [float(v) for v in vs] == [3.0 * k for k in range(20)]
    assert bnn.predict_stats["looped"] == before["looped"] + 1


def test_wide_two_hidden_layer_models_route_to_the_matrix_core_kernel():
    """mlp._kernel_for: the notebook's split-HMC model (1-100-100-1) and its relatives go to the hta_netn_* entry points, where
    the library dispatches csrc/mlp3_mfma.hip; shapes outside its range keep their previous route."""
    from hamiltorch_amd import mlp
    k = lambda dims, loss="regression": mlp._kernel_for(dict(dims=dims, loss=loss))     # noqa: E731
    assert k([1, 100, 100, 1]) == "netn" and mlp._is_mlp3([1, 100, 100, 1], "regression")
    assert k([4, 104, 72, 1]) == "netn" and mlp._is_mlp3([4, 104, 72, 1], "regression")
    assert k([1, 10, 10, 1]) == "netn" and not mlp._is_mlp3([1, 10, 10, 1], "regression")        # the small-net kernel's own
    assert k([1, 105, 100, 1]) is None and k([5, 100, 100, 1]) is None                              # beyond the kernel: callback path
    assert k([1, 100, 100, 2]) is None and k([1, 100, 100, 1], "binary_class_linear_output") is None
    assert k([8, 100, 1]) == "mlp1"
    # the notebook's class is recognised structurally (fc1 / fc2 / fc3 with F.relu in forward)
    import torch.nn as nn
    from hamiltorch_amd import bnn

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1, self.fc2, self.fc3 = nn.Linear(1, 100), nn.Linear(100, 100), nn.Linear(100, 1)

        def forward(self, x):
            x = torch.nn.functional.relu(self.fc1(x))
            x = torch.nn.functional.relu(self.fc2(x))
            return self.fc3(x)
    net = Net()
    assert bnn._mlp_structure(net)[:2] == ([1, 100, 100, 1], "relu")
    sizes = [w.nelement() for w in net.parameters()]; shapes = [w.shape for w in net.parameters()]
    f = bnn.define_model_log_prob(net, "regression", torch.randn(40, 1), torch.randn(40, 1), sizes, shapes, [1.0] * 6, 110.44)
    assert f._hta_spec["dims"] == [1, 100, 100, 1] and mlp._kernel_for(f._hta_spec) == "netn"


@pytest.mark.parametrize("kind,M,L", [("symmetric", 4, 10), ("symmetric", 2, 3), ("symmetric", 7, 1), ("kmid", 4, 5), ("kmid", 2, 1),
                                      ("rand", 4, 3), ("rand", 3, 2)])
def test_split_schedules_repeat_a_gradient_exactly_where_the_kernels_share_it(kind, M, L):
    """csrc/mlp.hpp `split_stage_reuses`: two kicks share one gradient evaluation when the second differentiates the same subset at
    the same parameters.  Walk the oracle's restatement of the reference's loops (S:494-596) with gradient functions that log
    (subset, parameters): the number of DISTINCT consecutive evaluations is the pass count the kernels execute -
    (2M - 2) L + 1 for the symmetric scheme, 2 M L - (L - 1) for KMID, 2 M L for RAND - and in the symmetric scheme every repeated
    evaluation directly follows its twin (so keeping ONE gradient is enough)."""
    import numpy as np
    import hmc_oracle as O
    log = []

    def gf(m):
        def f(theta):
            log.append((m, theta.tobytes()))
            return -(m + 1.0) * theta
        return f
    th = np.linspace(0.1, 0.5, 5).astype(np.float64)[None]
    p = np.ones_like(th)
    perm = list(range(M))[::-1] if kind == "rand" else None
    O.split_leapfrog(th, p, [gf(m) for m in range(M)], L, 0.01, None, kind, perm)
    assert len(log) == 2 * M * L
    distinct = 1 + sum(1 for a, b in zip(log, log[1:]) if a != b)
    if kind == "kmid":       # its second half sweep is repeated by the next step's first one, subset by subset; only subset 0's twin
        assert len(set(log)) == (L + 1) * M        # is adjacent - the one the kernels share (one stored gradient, not M)
    else:
        assert len(set(log)) == distinct           # a repeated evaluation always directly follows its twin
    want = {"symmetric": (2 * M - 2) * L + 1, "kmid": 2 * M * L - (L - 1), "rand": 2 * M * L}[kind]
    assert distinct == want, (distinct, want)


def test_prepared_workspace_caches_follow_the_target(monkeypatch):
    """The per-target caches of prepared workspaces (samplers._prepared_hmc_workspace, rmhmc._prepared_workspace): one
    preparation per (target, shape, stream); an in-place edit of the precision / mean (version counter) or a replaced tensor
    prepares again; a cached entry keeps the tensor object alive (its storage cannot be handed to another matrix while the
    entry could match it); dropping the target forgets the workspace in the library.  The C entry points are stubbed: this is
    the host logic only."""
    import gc
    import types
    from hamiltorch_amd import _abi, rmhmc
    calls = {"hmc_prepare": 0, "hmc_forget": 0, "rm_prepare": 0, "rm_forget": 0}
    monkeypatch.setattr(_abi, "gaussian_workspace_bytes", lambda C, D, n, es: 64 + C * D * n)
    monkeypatch.setattr(_abi, "hmc_gaussian_prepare", lambda *a, **k: calls.__setitem__("hmc_prepare", calls["hmc_prepare"] + 1))
    monkeypatch.setattr(_abi, "hmc_gaussian_forget", lambda ws: calls.__setitem__("hmc_forget", calls["hmc_forget"] + 1))
    monkeypatch.setattr(_abi, "rmhmc_workspace_bytes", lambda C, D, es, N: 64 + C * D)
    monkeypatch.setattr(_abi, "rmhmc_gaussian_prepare", lambda *a, **k: calls.__setitem__("rm_prepare", calls["rm_prepare"] + 1))
    monkeypatch.setattr(_abi, "rmhmc_gaussian_forget", lambda ws: calls.__setitem__("rm_forget", calls["rm_forget"] + 1))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda dev=None: types.SimpleNamespace(cuda_stream=0))
    th0 = torch.zeros(8, 3)
    t = ht.GaussianTarget(torch.zeros(3), precision=torch.eye(3), normalized=False)

    ws1 = samplers._prepared_hmc_workspace(t, th0, 10)
    assert samplers._prepared_hmc_workspace(t, th0, 10) is ws1 and calls["hmc_prepare"] == 1          # reused
    ws2 = samplers._prepared_hmc_workspace(t, th0, 7)                                                  # another launch length
    assert ws2 is not ws1 and calls["hmc_prepare"] == 2
    t.precision.mul_(2.0)                                                                              # in-place edit
    ws3 = samplers._prepared_hmc_workspace(t, th0, 10)
    assert ws3 is not ws1 and calls["hmc_prepare"] == 3
    old = t.precision
    t.precision = torch.eye(3) * 3.0                                                                   # replaced tensor
    assert samplers._prepared_hmc_workspace(t, th0, 10) is not ws3 and calls["hmc_prepare"] == 4
    assert any(sig[0] is old or sig[0] is t.precision for _, sig in t._hta_hmc_ws.values())            # entries hold tensor objects
    assert len(t._hta_hmc_ws) <= 2                                                                     # bounded per target
    # RMHMC: keyed by metric / alpha / jitter too, signed by precision AND mean
    r1 = rmhmc._prepared_workspace(t, th0, 1, 1e6, 1e-3, 20)
    assert rmhmc._prepared_workspace(t, th0, 1, 1e6, 1e-3, 20) is r1 and calls["rm_prepare"] == 1
    assert rmhmc._prepared_workspace(t, th0, 1, 1e6, 2e-3, 20) is not r1 and calls["rm_prepare"] == 2
    t.mean.add_(1.0)
    assert rmhmc._prepared_workspace(t, th0, 1, 1e6, 1e-3, 20) is not r1 and calls["rm_prepare"] == 3
    n_hmc, n_rm = len(t._hta_hmc_ws), len(t._hta_rm_ws)
    forgot = (calls["hmc_forget"], calls["rm_forget"])
    del t, ws1, ws2, ws3, r1, old
    gc.collect()
    assert calls["hmc_forget"] >= forgot[0] + n_hmc and calls["rm_forget"] >= forgot[1] + n_rm          # handles forget in the library


def test_native_forward_gate_mirrors_the_kernel_limits():
    """`_native_forward_ok` admitted shapes hta_net_forward refuses (784 inputs: 2 x 784 x 64 x 4 B = 401 KB of LDS;
    200-wide float64 layers: 205 KB) and predict_model raised instead of taking the torch path.  The gate is the kernel's own
    arithmetic now (csrc/net_forward.hip: net_forward)."""
    from hamiltorch_amd import bnn
    fits = bnn.native_forward_fits
    assert fits([1, 100, 100, 1], 4) and fits([8, 100, 1], 4) and fits([4, 3], 4) and fits([1, 10, 10, 1], 8)
    assert fits([256, 256, 256, 10], 4)                       # 2 x 256 x 64 x 4 = 128 KB
    assert not fits([784, 100, 10], 4)                        # flattened MNIST: the input layer is staged, 401 KB
    assert not fits([784, 512, 10], 4)                        # streamed form: still stages its 784 inputs
    assert fits([16, 512, 10], 4)                             # streamed: only the inputs are staged
    assert not fits([16, 512, 32], 4)                         # > FW_MAXO outputs: not the streamed form, 512 > FW_MAXW
    assert not fits([1, 200, 200, 1], 8)                      # float64: 2 x 200 x 64 x 8 = 205 KB
    assert fits([1, 160, 160, 1], 8) and not fits([1, 164, 1, 1], 8)
    assert not fits([1] * 11, 4)                              # > FW_MAXL layers
    # the kernel's constants have not drifted from the mirror
    src = open(os.path.join(ROOT, "hamiltorch_amd", "csrc", "net_forward.hip")).read()
    assert "FW_TPB = %d, FW_MAXL = %d, FW_MAXW = %d, FW_MAXO = %d" % (bnn._FW_TPB, bnn._FW_MAXL, bnn._FW_MAXW, bnn._FW_MAXO) in src
    # and the model-level gate uses it
    net = torch.nn.Sequential(torch.nn.Linear(784, 100), torch.nn.ReLU(), torch.nn.Linear(100, 10))
    D = sum(p.numel() for p in net.parameters())

    class FakeCuda(torch.Tensor):
        is_cuda = True
    stacked = torch.zeros(3, D).as_subclass(FakeCuda)
    assert bnn._native_forward_ok(net, stacked, torch.zeros(5, 784), "multi_class_linear_output") is None


def test_lift_callable_does_not_touch_distribution_means_and_keeps_user_errors():
    """distributions without a closed-form mean raised NotImplementedError inside the wrapper; a buggy log_prob_func
    was wrapped as host-evaluated with a PCIe warning instead of surfacing as itself."""
    import warnings
    from hamiltorch_amd import host
    base = torch.distributions.Normal(torch.zeros(3), torch.ones(3))
    td = torch.distributions.TransformedDistribution(base, [torch.distributions.transforms.ExpTransform()])
    with pytest.raises(NotImplementedError):
        td.mean
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = host.lift_callable(td.log_prob, torch.ones(2, 3))          # no NotImplementedError
    assert callable(out)

    def buggy(w):
        raise ValueError("the user's own bug")
    assert host.lift_callable(buggy, torch.ones(2, 3)) is buggy             # not wrapped: the engine's first call raises the ValueError

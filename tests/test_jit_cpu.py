"""CPU: the callback compiler up to (and including) the gfx950 code object - no GPU needed.

* the scalar graph (hamiltorch_amd/jit/ir.py): simplification identities, reverse-mode derivatives up to the third order
  against torch.autograd on the same callables;
* the lowering (jit/trace.py): a zoo of log_prob_func shapes (indexing, broadcasting, reductions, matrix products,
  torch.distributions, softplus / logsumexp, where / clamp, in-place updates) - value and gradient of the lowered graph
  against the callable itself; what must NOT be compiled (data-dependent control flow, the tuple protocol, unlisted
  operations) raises ir.Unsupported with a reason;
* the emitter + hipRTC (jit/emit.py, jit/runtime.py, csrc/jit/*.in through hta_jit_compile): every example compiles for
  gfx950 into a code object that exports the kernel and the info block, without spilling.
"""
import math
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from hamiltorch_amd.jit import compile_hmc, emit, runtime, stats, _signature
from hamiltorch_amd.jit.ir import Graph, Unsupported
from hamiltorch_amd.jit.trace import trace_callback

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
f64 = torch.float64


# ---- targets -------------------------------------------------------------------------------------------------------
def funnel(w):
    v, x = w[0], w[1:]
    return -v * v / 18.0 - 0.5 * torch.exp(v) * (x * x).sum() + 0.5 * x.numel() * v


def funnel_dist(w):
    ll = torch.distributions.Normal(0, 3).log_prob(w[0])
    ll += torch.distributions.Normal(0, torch.exp(-w[0]) ** 0.5).log_prob(w[1:]).sum()
    return ll


_A = torch.randn(7, 5, dtype=f64, generator=torch.Generator().manual_seed(1))
_y = (torch.rand(7, generator=torch.Generator().manual_seed(2)) > 0.5).to(f64)
_S = torch.tensor(np.cov(np.random.default_rng(0).standard_normal((5, 40))), dtype=f64)
_mvn = torch.distributions.MultivariateNormal(torch.zeros(5, dtype=f64), covariance_matrix=_S)


def logistic(w):
    z = _A @ w
    return (_y * z - torch.nn.functional.softplus(z)).sum() - 0.5 * w.dot(w)


def mixture(w):
    return torch.logsumexp(torch.stack([_mvn.log_prob(w), _mvn.log_prob(w - 1.5) + math.log(0.5)]), 0)


def student(w):
    return -3.0 * torch.log1p((w ** 2) / 5.0).sum() + torch.tanh(w[0] * w[1]) - torch.sigmoid(w[2]) * w[3].abs()


def piecewise(w):           # element-wise selects are fine (no Python branch on values)
    return -torch.where(w > 0, w ** 2, 0.5 * w ** 2).sum() - torch.clamp(w, -1.0, 2.0).pow(4).sum() - torch.relu(w - 1).sum()


def inplace(w):
    out = torch.zeros(5, dtype=w.dtype)
    out[1:] = w[:4] * w[1:]
    out[0] = w[4].sin()
    acc = -(w * w).sum()
    acc -= out.pow(2).sum()
    return acc


def gathers(w):
    idx = torch.tensor([4, 0, 2])
    m = w.reshape(1, 5).expand(3, 5)
    return -(w[idx] * w[[1, 1, 3]]).sum() ** 2 - torch.cumsum(w, 0).pow(2).mean() - (m.t() @ m).diagonal().sum() - torch.stack([w, 2 * w]).std()


def stopgrad(w):            # detach() stops the derivative, not the value
    return -(w * w.detach()).sum() - torch.exp(w[0]).detach() * w[1] ** 2


def linear_layer(w):
    lin = torch.nn.functional.linear(w[:4].reshape(2, 2), _A[:3, :2], _A[:3, 4])
    return -lin.pow(2).sum() - torch.nn.functional.gelu(w).sum() - torch.erf(w[4]) - torch.nn.functional.silu(w[0])


ZOO = [(funnel, 6), (funnel_dist, 6), (logistic, 5), (mixture, 5), (student, 5), (piecewise, 5), (inplace, 5), (gathers, 5),
       (linear_layer, 5), (stopgrad, 5)]


def autograd_reference(fn, pts, order):
    vals, grads, hess, third = [], [], [], []
    for x in pts:
        x = x.clone().requires_grad_(True)
        v = fn(x).sum()
        g, = torch.autograd.grad(v, x, create_graph=order > 1)
        vals.append(float(v)); grads.append(g.detach().numpy())
        if order > 1:
            H = torch.stack([torch.autograd.grad(g[i], x, create_graph=order > 2, allow_unused=True)[0] if g[i].requires_grad else torch.zeros_like(x)
                             for i in range(x.numel())])
            hess.append(H.detach().numpy())
        if order > 2:
            D = x.numel()
            T3 = np.zeros((D, D, D))
            for i in range(D):
                for j in range(D):
                    if H[i, j].requires_grad:
                        t = torch.autograd.grad(H[i, j], x, retain_graph=True, allow_unused=True)[0]
                        T3[i, j] = 0 if t is None else t.numpy()
            third.append(T3)
    return np.array(vals), np.array(grads), np.array(hess), np.array(third)


# ---- tests ----------------------------------------------------------------------------------------------------------
def test_graph_simplifies_as_it_is_built():
    g = Graph(2)
    x, y = g.inputs
    c = g.const
    assert g.add(x, c(0)) == x and g.mul(x, c(1)) == x and g.mul(c(0), y) == c(0) and g.sub(x, x) == c(0)
    assert g.add(x, y) == g.add(y, x) and g.mul(x, y) == g.mul(y, x)                       # commutative operands share a node
    assert g.neg(g.neg(x)) == x and g.add(x, g.neg(y)) == g.sub(x, y)
    assert g.binary("pow", x, c(2)) == g.mul(x, x) and g.binary("pow", x, c(0.5)) == g.unary("sqrt", x)
    assert g.mul(c(2), g.mul(c(3), x)) == g.mul(c(6), x)                                   # constants meet and fold
    assert g.add(x, x) == g.mul(c(2), x)
    assert g.unary("log", g.unary("exp", x)) == x and g.unary("sqrt", g.unary("exp", x)) == g.unary("exp", g.mul(c(0.5), x))
    assert g.mul(c(2), c(4)) == c(8) and g.is_const(g.unary("exp", c(0)))
    assert g.select(g.bconst(True), x, y) == x and g.select(g.compare("gt", x, y), x, x) == x
    assert g.div(x, y) == g.mul(x, g.unary("recip", y))                                    # one shared reciprocal per divisor
    n = len(g.nodes)
    g.add(g.mul(x, y), g.mul(y, x))
    assert len(g.nodes) <= n + 2                                                           # the product exists once


@pytest.mark.parametrize("fn,D", ZOO, ids=[f.__name__ for f, _ in ZOO])
def test_lowered_value_and_gradient_equal_the_callable(fn, D):
    tr = trace_callback(fn, torch.ones(D, dtype=f64))
    g = tr.grad()
    pts = 0.8 * torch.randn(40, D, dtype=f64, generator=torch.Generator().manual_seed(3))
    v, gr, _, _ = autograd_reference(fn, pts, 1)
    out = tr.graph.evaluate([tr.value] + g, pts.numpy(), np.float64)
    np.testing.assert_allclose(out[:, 0], v, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out[:, 1:], gr, rtol=1e-10, atol=1e-11)
    out32 = tr.graph.evaluate([tr.value] + g, pts.numpy().astype(np.float32))              # the arithmetic the device does
    assert out32.dtype == np.float32
    np.testing.assert_allclose(out32[:, 1:], gr, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("fn,D", [(funnel, 5), (logistic, 5), (student, 5), (mixture, 5)], ids=["funnel", "logistic", "student", "mixture"])
def test_second_and_third_derivatives_equal_autograd(fn, D):
    """What the Riemannian samplers need (S:108, S:397-398): graph differentiation applied twice / three times."""
    tr = trace_callback(fn, torch.ones(D, dtype=f64))
    g = tr.grad()
    H = [tr.graph.grad(gi) for gi in g]
    T3 = {(i, j): tr.graph.grad(H[i][j]) for i in range(D) for j in range(i + 1)}
    pts = 0.6 * torch.randn(6, D, dtype=f64, generator=torch.Generator().manual_seed(5))
    _, _, Hr, Tr = autograd_reference(fn, pts, 3)
    flatH = [H[i][j] for i in range(D) for j in range(D)]
    got = tr.graph.evaluate(flatH, pts.numpy(), np.float64).reshape(-1, D, D)
    np.testing.assert_allclose(got, Hr, rtol=1e-9, atol=1e-10)
    for (i, j), nodes in T3.items():
        got3 = tr.graph.evaluate(nodes, pts.numpy(), np.float64)
        np.testing.assert_allclose(got3, Tr[:, i, j, :], rtol=1e-8, atol=1e-9)


def test_constants_fold_in_the_dtype_torch_uses():
    """Normal(0, 3).log_prob takes log(3) in float32 whatever the argument's dtype: the trace folds constant sub-expressions with
    torch itself, so the compiled value equals the callable's to float64 rounding, not to float32 rounding."""
    tr = trace_callback(funnel_dist, torch.ones(6, dtype=f64))
    pts = torch.randn(8, 6, dtype=f64, generator=torch.Generator().manual_seed(7))
    v = np.array([float(funnel_dist(p)) for p in pts])
    got = tr.graph.evaluate([tr.value], pts.numpy(), np.float64)[:, 0]
    np.testing.assert_allclose(got, v, rtol=1e-13, atol=1e-13)


def test_what_cannot_be_compiled_says_why():
    def branchy(w):
        return (w * w).sum() if w[0] > 0 else -(w * w).sum()

    def item(w):
        return w.sum() * float(w[0])

    def tuple_protocol(w):
        return (w * w).sum(), [w]

    def argmax(w):
        return w[w.argmax()]

    def unlisted(w):
        return torch.linalg.eigvalsh(torch.outer(w, w) + torch.eye(3, dtype=w.dtype)).sum()

    for fn, frag in ((branchy, "control flow"), (item, "control flow"), (tuple_protocol, "tuple protocol"), (argmax, ""), (unlisted, "not in the lowering table")):
        with pytest.raises(Unsupported) as e:
            trace_callback(fn, torch.ones(3, dtype=f64))
        assert frag in str(e.value), (fn.__name__, str(e.value))
    with pytest.raises(Unsupported, match="registers"):
        runtime.hmc_generated_source(trace_callback(lambda w: -(w * w).sum(), torch.ones(200)), torch.float32, 0)


def test_callables_whose_autograd_differs_from_their_operations_are_refused():
    """A trace records what is COMPUTED; torch.no_grad() / a custom backward change what autograd RETURNS.  compile_hmc checks every fresh
    trace's value and gradient against torch.autograd around the example and refuses such callables (they stay on the callback path)."""
    def nograd_scale(w):
        with torch.no_grad():
            s = (w * w).sum()
        return -s * (w * w).sum()

    class Clip(torch.autograd.Function):
        @staticmethod
        def forward(x):
            return x * 2.0

        @staticmethod
        def setup_context(ctx, inputs, output):
            pass

        @staticmethod
        def backward(ctx, g):
            return g * 0.5          # not the derivative of forward

    def custom(w):
        return -(Clip.apply(w) ** 2).sum()

    ex = torch.ones(3)
    with pytest.raises(Unsupported, match="disagree with torch.autograd"):
        compile_hmc(nograd_scale, ex, torch.float32, 0)
    with pytest.raises(Unsupported):                         # (a custom Function does not even trace: torch has no functionalisation rule for it)
        compile_hmc(custom, ex, torch.float32, 0)
    compile_hmc(stopgrad, torch.ones(5), torch.float32, 0)          # detach() itself is understood


def _symbols(blob, tmp_path):
    p = tmp_path / "cb.co"
    p.write_bytes(blob)
    sym = subprocess.run([READELF, "-s", str(p)], capture_output=True, text=True).stdout
    notes = subprocess.run([READELF, "--notes", str(p)], capture_output=True, text=True).stdout
    regs = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", notes)}
    return sym, regs


@pytest.mark.parametrize("fn,D", ZOO, ids=[f.__name__ for f, _ in ZOO])
@pytest.mark.parametrize("dtype,mass", [(torch.float32, 0), (torch.float64, 2)])
def test_every_example_compiles_for_gfx950(fn, D, dtype, mass, tmp_path):
    """Generated code + csrc/jit/hmc_callback.hip.in through hipRTC (hta_jit_compile): a gfx950 code object with the trajectory kernel
    and the info block, nothing spilled to scratch."""
    tr = trace_callback(fn, torch.ones(D, dtype=f64))
    src = runtime.hmc_generated_source(tr, dtype, mass)
    assert "#define HTA_CB_D %d" % D in src and "value_grad" in src
    key, blob = runtime.compile_source(src, runtime.SKELETON_HMC)
    sym, regs = _symbols(blob, tmp_path)
    assert "hta_cb_hmc_kernel" in sym and "hta_cb_info" in sym
    assert regs["vgpr_spill_count"] == 0 and regs["private_segment_fixed_size"] == 0, regs
    assert runtime.compile_source(src, runtime.SKELETON_HMC)[0] == key                      # cached by content


def test_compile_error_carries_the_compilers_log():
    with pytest.raises(runtime.CompileError) as e:
        runtime.compile_source("#define HTA_CB_D 2\n#define HTA_CB_T float\n#define HTA_CB_MASS 0\n#define HTA_CB_NODES 1\nthis is not C++\n",
                               runtime.SKELETON_HMC)
    assert "error" in e.value.log


def test_traces_are_reused_by_closure_signature():
    scale = torch.tensor(2.0)
    fn = lambda w: -scale * (w ** 4).sum()  # noqa: E731
    ex = torch.ones(3)
    t0, h0 = stats["traced"], stats["trace_hits"]
    a = compile_hmc(fn, ex, torch.float32, 0)
    b = compile_hmc(fn, ex, torch.float32, 0)
    assert a is b and stats["traced"] == t0 + 1 and stats["trace_hits"] == h0 + 1
    scale.mul_(2.0)                                              # the closed-over tensor's version counter moves: traced again
    c = compile_hmc(fn, ex, torch.float32, 0)
    assert c is not a and stats["traced"] == t0 + 2 and c.key != a.key
    assert compile_hmc(fn, ex, torch.float32, 1) is not c       # another mass kind is another module
    assert _signature(fn)[0] == _signature(fn)[0]

    def branchy(w):
        return (w * w).sum() if w[0] > 0 else -(w * w).sum()
    for _ in range(2):                                           # the refusal is remembered too
        with pytest.raises(Unsupported, match="control flow"):
            compile_hmc(branchy, ex, torch.float32, 0)


def test_emitted_literals_round_trip():
    assert emit.literal(0.1, "float") == repr(float(np.float32(0.1))) + "f"
    assert emit.literal(1.0, "double") == "1.0" and emit.literal(1e300, "float").endswith("huge_val()")
    assert "nan" in emit.literal(float("nan"), "double") and emit.literal(-math.inf, "float").startswith("(-")
    assert float(emit.literal(1 / 3, "double")) == 1 / 3

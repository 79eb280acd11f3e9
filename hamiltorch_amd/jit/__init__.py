"""Callback compiler: an OPAQUE ``log_prob_func`` inside fused gfx950 kernels.

The reference's hot spot for a general target is ``params_grad`` (hamiltorch/samplers.py:270-278 -> ``collect_gradients``
:33-66): autograd through the user's callable, once per leapfrog step - "84 % of HMC wall time" on its own CPU path, and on a
GPU ~38 eager launches per step around kernels that take microseconds.  This package removes the launches:

  trace.py    the callable is traced ONCE per target (torch.fx ``make_fx``) and lowered element by element to a scalar graph;
  ir.py       reverse-mode differentiation, simplification and common-subexpression sharing on that graph;
  emit.py     value + gradient (and, for the Riemannian samplers, Hessian + third-derivative contraction) as straight-line
              HIP device code;
  runtime.py  hipRTC (through the C ABI: ``hta_jit_compile`` / ``hta_jit_load``) builds that text INTO the hand-written
              kernels of ``csrc/jit/`` - for plain HMC the whole ``sample()`` loop, one chain per lane, one launch per block
              of trajectories.

No inductor, no Triton, no code from torch's compiler stack beyond the tracer.  A callable the tracer or the lowering table
does not cover (data-dependent control flow, the tuple / ``pass_grad`` protocols, unlisted operations, graphs that are too
large) stays on the torch-evaluated callback path; ``last_reason()`` and ``hta_last_route()`` say why.  Every compiled run is
checked against the callable itself on the states it ended in (``samplers._verify_compiled``); a mismatch re-traces once and
otherwise repeats the run on the callback path.  ``HAMILTORCH_AMD_JIT=0`` (or ``sample(..., native=False)``) turns it off.
"""
from __future__ import annotations

import os
import threading
import weakref

import torch

from . import runtime
from .ir import Unsupported
from .trace import trace_callback

_lock = threading.Lock()
_by_fn = weakref.WeakKeyDictionary()      # callable -> {config: entry}
_state = threading.local()
stats = {"traced": 0, "trace_hits": 0, "unsupported": 0}


def enabled():
    return os.environ.get("HAMILTORCH_AMD_JIT", "1") != "0"


def last_reason():
    """Why the calling thread's last compile attempt fell back to the callback path ('' if it did not)."""
    return getattr(_state, "reason", "")


def _note(reason):
    _state.reason = reason


def _signature(fn):
    """Identity + version counters of what `fn` closes over and of the tensors / objects its code names in its globals: the key
    under which a trace is reused.  Holds the objects (their ids stay unique while the entry lives).  In-place changes deeper
    inside captured objects are NOT seen here - the run-time check against the callable catches those."""
    objs = []
    f = getattr(fn, "__func__", fn)
    if getattr(fn, "__self__", None) is not None:
        objs.append(fn.__self__)
    for cell in (getattr(f, "__closure__", None) or ()):
        try:
            objs.append(cell.cell_contents)
        except ValueError:
            objs.append(None)
    code, glob = getattr(f, "__code__", None), getattr(f, "__globals__", None)
    if code is not None and glob is not None:
        for name in code.co_names:
            v = glob.get(name)
            if torch.is_tensor(v) or isinstance(v, (torch.distributions.Distribution, torch.nn.Module)):
                objs.append(v)
    if isinstance(fn, torch.nn.Module):
        objs += list(fn.parameters()) + list(fn.buffers())
    for a in (getattr(fn, "args", None) or ()) if hasattr(fn, "func") else ():      # functools.partial
        objs.append(a)
    sig = tuple((id(o), getattr(o, "_version", None), float(o) if isinstance(o, (int, float)) and not isinstance(o, bool) else None)
                for o in objs)
    return sig, objs


class CompiledHMC:
    """A traced callable compiled into the HMC trajectory kernel for one (D, dtype, mass kind)."""

    def __init__(self, traced, key, blob, dtype, mass_kind):
        self.traced, self.key, self.blob, self.dtype, self.mass_kind = traced, key, blob, dtype, mass_kind

    def module(self, device):
        return runtime.module_for(self.key, self.blob, device)


class CompiledDerivs(CompiledHMC):
    """A traced callable compiled into the derivative kernels of the Riemannian samplers (csrc/jit/derivs_callback.hip.in)."""


def compile_hmc(fn, example, dtype, mass_kind, fresh=False):
    """CompiledHMC for ``fn`` at points shaped like the (D,) tensor ``example``; raises ``Unsupported``.  A trace is reused while the
    callable's closure signature is unchanged (``fresh=True`` traces again)."""
    return _compile(fn, example, dtype, int(mass_kind), fresh)


def compile_derivs(fn, example, dtype, fresh=False):
    """CompiledDerivs (value / gradient / Hessian + third-derivative contraction kernels) for ``fn``; raises ``Unsupported``."""
    return _compile(fn, example, dtype, "derivs", fresh)


def compile_rmhmc(fn, example, dtype, jitter, fresh=False):
    """The callable built into the explicit-RMHMC trajectory kernel (csrc/jit/rmhmc_callback.hip.in; D <= 16); raises ``Unsupported``."""
    if example.numel() > runtime.MAX_RMHMC_DIM:
        _note("D = %d: the chain-per-lane Riemannian kernel holds a chain's matrices in registers (D <= %d)" % (example.numel(), runtime.MAX_RMHMC_DIM))
        raise Unsupported(last_reason())
    return _compile(fn, example, dtype, "rmhmc-jitter" if jitter else "rmhmc", fresh)


def _compile(fn, example, dtype, mass_kind, fresh):
    _note("")
    cfg = (int(example.numel()), dtype, mass_kind, example.device.type)
    sig = objs = None
    try:
        sig, objs = _signature(fn)
        with _lock:
            ent = _by_fn.get(fn, {}).get(cfg)
    except TypeError:       # not weak-referenceable / unhashable: no reuse
        ent = None
        sig = None
    if ent is not None and not fresh and ent[0] == sig:
        stats["trace_hits"] += 1
        if isinstance(ent[2], Unsupported):
            _note(str(ent[2]))
            raise ent[2]
        return ent[2]
    try:
        traced = trace_callback(fn, example)
        stats["traced"] += 1
        _check_against_autograd(traced, fn, example)
        if mass_kind == "derivs":
            key, blob = runtime.compile_source(runtime.derivs_generated_source(traced, dtype), runtime.SKELETON_DERIVS)
            out = CompiledDerivs(traced, key, blob, dtype, mass_kind)
        elif mass_kind in ("rmhmc", "rmhmc-jitter"):
            key, blob = runtime.compile_source(runtime.derivs_generated_source(traced, dtype, mass_kind == "rmhmc-jitter"), runtime.SKELETON_RMHMC)
            out = CompiledDerivs(traced, key, blob, dtype, mass_kind)
        else:
            key, blob = runtime.compile_source(runtime.hmc_generated_source(traced, dtype, mass_kind), runtime.SKELETON_HMC)
            out = CompiledHMC(traced, key, blob, dtype, mass_kind)
    except Unsupported as e:
        stats["unsupported"] += 1
        _note(str(e))
        out = e
    except runtime.CompileError as e:
        # hipRTC turned the generated text down (an emitter bug, a device function hipRTC lacks): not the user's problem - the callable
        # runs on the previous path, the reason (the compiler's first error line) is reported like any other refusal
        stats["unsupported"] += 1
        first = next((ln for ln in e.log.splitlines() if "error" in ln), str(e).splitlines()[0])
        out = Unsupported("hipRTC rejected the generated code: %s" % first.strip()[:160])
        _note(str(out))
    if sig is not None:
        try:
            with _lock:
                _by_fn.setdefault(fn, {})[cfg] = (sig, objs, out)
        except TypeError:
            pass
    if isinstance(out, Unsupported):
        raise out
    return out


def _check_against_autograd(traced, fn, example, points=4):
    """A fresh trace is believed only after its VALUE AND GRADIENT (the graph's own reverse mode, evaluated in numpy) reproduce the callable
    under torch.autograd at a few points around the example.  The trace records operations, not autograd semantics: a `torch.no_grad()`
    block, a custom `autograd.Function` backward or a gradient hook inside the callable would compile to the derivative of what is
    COMPUTED, not to what autograd returns - such callables are refused here (the run-time check of sample() compares values only)."""
    import numpy as np
    g = torch.Generator(device="cpu").manual_seed(0x5EED)
    base = example.detach().double().cpu()
    pts = torch.cat([base[None], base[None] + 0.05 * (1.0 + base.abs())[None] * torch.randn(points - 1, base.numel(), generator=g, dtype=torch.float64)])
    grads = traced.grad()
    mine = traced.graph.evaluate([traced.value] + grads, pts.numpy(), np.float64)
    tol = 2e-3 if example.dtype == torch.float32 else 1e-7
    for k in range(pts.shape[0]):
        x = pts[k].to(device=example.device, dtype=example.dtype).requires_grad_(True)
        try:
            with torch.enable_grad():
                v = fn(x)
                v = v.sum() if v.dim() else v
                gr, = torch.autograd.grad(v, x, allow_unused=True)
        except Exception as e:
            raise Unsupported("autograd of the callable failed at a check point (%s)" % str(e).split("\n")[0][:100]) from None
        ref = np.concatenate([[float(v)], (torch.zeros_like(x) if gr is None else gr).detach().double().cpu().numpy()])
        if not np.all(np.isfinite(ref)) or not np.all(np.isfinite(mine[k])):
            continue
        if np.abs(mine[k] - ref).max() > tol * (1.0 + np.abs(ref).max()):
            raise Unsupported("the traced graph's value / gradient disagree with torch.autograd at a check point (max difference %.3g): "
                              "a no_grad block, a custom backward or a hook inside the callable?" % np.abs(mine[k] - ref).max())


def torch_logp(fn, theta):
    """log p of every row of theta [k, D] by the callable itself (torch, no graphs): the reference the compiled code is checked against."""
    def f(w):
        r = fn(w)
        r = r[0] if isinstance(r, tuple) else r
        return r.sum() if torch.is_tensor(r) and r.dim() else r
    with torch.no_grad():
        try:
            return torch.func.vmap(f)(theta)
        except Exception:
            return torch.stack([torch.as_tensor(f(t)) for t in theta])

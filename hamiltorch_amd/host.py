"""Host-resident inputs: every reference notebook and tests/test_util.py:97-110 hand ``sample`` /
``leapfrog`` CPU tensors (S:850, S:925) and get CPU tensors back.

There is no CPU implementation of the engine - the kernels exist for gfx950 only and ``_abi.require_device`` keeps saying
so - but a caller's tensors do not have to live on the GPU for the engine to run there: the public entry points are wrapped
by ``host_inputs``.  When the state argument (``params`` / ``params_init``) is on the host

* the tensor arguments (state, momentum, mass operands, data, a constant ``pass_grad``) are STAGED to the current GPU,
* ``log_prob_func`` is lifted (``lift_callable``): a ``GaussianTarget`` / ``MultivariateNormal.log_prob`` moves with
  ``as_gaussian``; a closure that evaluates on device tensors as it is (it captured none of its own) is used as it is; a
  closure over HOST tensors becomes ``HostEvaluated`` - the user's function keeps running where its tensors live, for all
  chains at once (``vmap`` on the host), its argument and result crossing PCIe.  ``sample`` probes the lifted callable like
  any other: the reference's idiom ``lambda w: MultivariateNormal(mean, cov).log_prob(w).sum()`` is recognised as a
  quadratic form and runs on the fused kernels (the closure is then evaluated a handful of times in all: probe + the
  verification on the run's own samples); anything else goes through the callback path with host evaluations - correct,
  slow, and announced once per callable by a warning that names it,
* ``sample_model`` / ``sample_split_model`` stage a deep copy of the module and the data,
* results come back on the host (the reference's lists of CPU tensors).

Nothing is staged when the state is already on the device: the GPU-resident path is untouched.
"""
from __future__ import annotations

import copy
import functools
import inspect
import warnings

import torch

#: argument names that carry tensors (or lists of tensors) to stage; everything else passes through
_TENSOR_ARGS = ("params", "params_init", "momentum", "inv_mass", "mass", "fish", "x", "y", "pass_grad")
_STATE_ARGS = ("params", "params_init", "fish")
_warned = set()


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("hamiltorch_amd: no AMD Instinct GPU is visible; this engine runs on gfx950 only. There is no CPU fallback.")
    return torch.device("cuda", torch.cuda.current_device())


def _is_host_state(v):
    if torch.is_tensor(v):
        return not v.is_cuda
    if isinstance(v, (list, tuple)) and v and torch.is_tensor(v[0]):        # explicit-RMHMC state [params, params_copy] (S:462)
        return not v[0].is_cuda
    return False


def _to(v, dev):
    if torch.is_tensor(v):
        return v.to(dev)
    if isinstance(v, list):
        return [_to(o, dev) for o in v]
    if isinstance(v, tuple):
        return tuple(_to(o, dev) for o in v)
    return v


def to_host(v):
    """Tensors of a result (nested lists / tuples, sample()'s lazy list) on the host."""
    from .samplelist import SampleList
    if torch.is_tensor(v):
        return v.cpu()
    if isinstance(v, SampleList):
        return list(v.tensor.cpu().unbind(0))
    if isinstance(v, list):
        return [to_host(o) for o in v]
    if isinstance(v, tuple):
        return tuple(to_host(o) for o in v)
    return v


class HostEvaluated:
    """``log_prob_func`` over HOST tensors, callable on device tensors: argument to the host, the user's function there, value
    back (differentiable and vmap-able: ``Tensor.to`` is both; never captured into a HIP graph - a device-to-host copy cannot
    be)."""

    def __init__(self, fn):
        self.fn = fn
        self.__name__ = "host(%s)" % getattr(fn, "__name__", type(fn).__name__)

    def __call__(self, w):
        r = self.fn(w.to("cpu"))
        if isinstance(r, tuple):
            raise TypeError("hamiltorch_amd: the (log_prob, params) tuple protocol (S:54-58) needs its tensors on the GPU")
        return r.to(w.device)


def lift_callable(fn, state_dev):
    """The callable to hand to the engine for a run whose state was staged to the device (see the module docstring)."""
    from .models import GaussianTarget
    if fn is None or isinstance(fn, GaussianTarget):
        return fn
    if isinstance(fn, list):
        return [lift_callable(f, state_dev) for f in fn]
    owner = getattr(fn, "__self__", None)
    if isinstance(owner, torch.distributions.Distribution):
        if isinstance(owner, torch.distributions.MultivariateNormal):
            return fn                                        # as_gaussian moves it
        # where the distribution's parameter tensors live (not `.mean`: distributions without a closed-form mean -
        # TransformedDistribution, some mixtures - raise NotImplementedError there)
        params = [v for v in vars(owner).values() if torch.is_tensor(v)]
        if params and all(v.is_cuda for v in params):
            return fn
    if not callable(fn):
        return fn
    row = state_dev.detach().reshape(-1, state_dev.shape[-1])[0]
    try:
        with torch.no_grad():
            r = fn(row.clone())
        r0 = r[0] if isinstance(r, tuple) else r
        if torch.is_tensor(r0) and r0.is_cuda:
            return fn                                        # a pure function of its argument: runs where the state is
    except RuntimeError as e:
        # only a device mismatch says "this function closes over host tensors"; anything else is the user's bug and surfaces as
        # itself, from the engine's first real call, not as a host-evaluated wrapper with a PCIe warning
        if "device" not in str(e).lower():
            return fn
    except Exception:
        return fn
    name = getattr(fn, "__qualname__", getattr(fn, "__name__", type(fn).__name__))
    key = id(fn)
    if key not in _warned:
        _warned.add(key)
        warnings.warn("hamiltorch_amd: log_prob_func %r works on host tensors (params_init is on the CPU and the function closes "
                      "over CPU tensors): the engine runs on the GPU and evaluates it on the host, crossing PCIe at every call "
                      "it makes.  A quadratic form is recognised and leaves the host out of the loop; for anything else move "
                      "params_init and the tensors the function captures to 'cuda'." % name, stacklevel=3)
    return HostEvaluated(fn)


def host_inputs(func):
    """Decorator of a public entry point: see the module docstring."""
    sig = inspect.signature(func)

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        try:
            bound = sig.bind(*args, **kwargs)
        except TypeError:
            return func(*args, **kwargs)
        a = bound.arguments
        state = next((a[n] for n in _STATE_ARGS if n in a and a[n] is not None), None)
        if state is None or not _is_host_state(state) or not torch.cuda.is_available():
            return func(*args, **kwargs)                      # (no GPU: the entry point's own checks and require_device speak)
        dev = _device()
        for n in _TENSOR_ARGS:
            if n in a and a[n] is not None and not callable(a[n]):
                a[n] = _to(a[n], dev)
        state_dev = next(a[n] for n in _STATE_ARGS if n in a and a[n] is not None)
        state_dev = state_dev[0] if isinstance(state_dev, (list, tuple)) else state_dev
        if "log_prob_func" in a:
            a["log_prob_func"] = lift_callable(a["log_prob_func"], state_dev)
        if callable(a.get("pass_grad")):
            a["pass_grad"] = lift_callable(a["pass_grad"], state_dev)
        if "model" in a and isinstance(a["model"], torch.nn.Module):
            a["model"] = copy.deepcopy(a["model"]).to(dev)
        if "tau_list" in a and a["tau_list"] is not None:
            a["tau_list"] = _to(a["tau_list"], dev) if torch.is_tensor(a["tau_list"]) else a["tau_list"]
        return to_host(func(*bound.args, **bound.kwargs))
    wrapper.__wrapped_host_inputs__ = True
    return wrapper

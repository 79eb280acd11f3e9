"""Bayesian-neural-network front-ends: mirror of samplers.py:1093-1562 (``define_model_log_prob``,
``define_split_model_log_prob``, ``sample_model``, ``sample_split_model``, ``predict_model``).

The closures built here are ordinary ``log_prob_func`` callables (prior S:1141-1157 +
likelihood S:1170-1190), evaluated with ``torch.func.functional_call`` so they batch over
chains under ``vmap``.  For an MLP with a regression likelihood they also carry a structural
description (``_hta_spec``) that lets ``sample`` run the whole split-HMC trajectory in the
native kernel (csrc/mlp_split.hip) instead of calling back into torch.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import util  # noqa: F401
from .enums import Integrator, Metric, Sampler
from .host import host_inputs

_ACTS = {nn.ReLU: "relu", nn.Tanh: "tanh", nn.Sigmoid: "sigmoid"}


def _mlp_structure(model):
    """[in, h1, ..., out] and the activation name if ``model`` computes Linear, act, Linear, ... with one activation kind and
    its parameters are flattened in that order (U:121-122): an ``nn.Sequential`` of those modules, or any module whose
    ``forward`` traces (``torch.fx``) to such a chain - the reference notebooks' ``Net`` classes (``self.l1 .. self.l3`` and
    ``torch.relu`` calls in ``forward``); else None."""
    try:
        return _STRUCTURE_CACHE[model]                    # (define_split_model_log_prob asks once per batch: trace a model once)
    except (KeyError, TypeError):
        pass
    st = _sequential_structure(model)
    if st is None:
        st = _traced_structure(model)
    try:
        _STRUCTURE_CACHE[model] = st
    except TypeError:
        pass
    return st


import weakref  # noqa: E402

_STRUCTURE_CACHE = weakref.WeakKeyDictionary()


_ACT_FUNCS = {"relu": "relu", "tanh": "tanh", "sigmoid": "sigmoid"}


def _traced_structure(model):
    if isinstance(model, nn.Sequential) or not isinstance(model, nn.Module):
        return None
    try:
        import torch.fx
        gm = torch.fx.symbolic_trace(model)
    except Exception:                     # data-dependent control flow etc.: not a plain chain
        return None
    mods = dict(gm.named_modules())
    dims, act, linears, tail = [], None, [], None
    prev, expect_linear, seen_input = None, True, False
    for node in gm.graph.nodes:
        if node.op == "placeholder":
            if seen_input:
                return None
            seen_input, prev = True, node
            continue
        if node.op == "output":
            out = node.args[0]
            if out is not prev or (expect_linear and tail is None) or len(dims) < 2:
                return None
            continue
        tensor_args = [a for a in node.args if isinstance(a, torch.fx.Node)]
        if tensor_args != [prev] or any(isinstance(v, torch.fx.Node) for v in node.kwargs.values()) or tail is not None:
            return None
        is_lsm = (node.op == "call_module" and isinstance(mods.get(node.target), nn.LogSoftmax) and mods[node.target].dim in (1, -1)) or \
                 (node.op in ("call_function", "call_method") and getattr(node.target, "__name__", node.target) == "log_softmax"
                  and (tuple(node.args[1:]) in ((1,), (-1,)) or node.kwargs.get("dim") in (1, -1)))
        if is_lsm:                                  # a log-softmax over the outputs closes the chain ('multi_class_log_softmax_output')
            if expect_linear or len(dims) < 2:
                return None
            tail = "log_softmax"
            prev = node
            continue
        if node.op == "call_module" and isinstance(mods.get(node.target), nn.Linear):
            m = mods[node.target]
            if not expect_linear or m.bias is None or (dims and dims[-1] != m.in_features):
                return None
            if not dims:
                dims.append(m.in_features)
            dims.append(m.out_features)
            linears.append(m)
            expect_linear = False
        else:
            if node.op == "call_module":
                a = _ACTS.get(type(mods.get(node.target)))
            elif node.op == "call_function":
                a = _ACT_FUNCS.get(getattr(node.target, "__name__", ""))
            elif node.op == "call_method":
                a = _ACT_FUNCS.get(node.target)
            else:
                a = None
            if a is None or expect_linear or (act is not None and a != act) or len(node.args) != 1 \
                    or any(v not in (False, None) for v in node.kwargs.values()):       # (F.relu records inplace=False)
                return None
            act = a
            expect_linear = True
        prev = node
    params = list(model.parameters())
    want = [t for m in linears for t in (m.weight, m.bias)]
    if len(params) != len(want) or any(a is not b for a, b in zip(params, want)) or list(model.buffers()):
        return None                       # other parameters, shared layers, or a flattening order that is not the order of use
    return (dims, (act or "relu")) if tail is None else (dims, (act or "relu"), tail)


def _sequential_structure(model):
    if not isinstance(model, nn.Sequential):
        return None
    mods = list(model.children())
    tail = None
    if mods and isinstance(mods[-1], nn.LogSoftmax) and mods[-1].dim in (1, -1):
        mods, tail = mods[:-1], "log_softmax"
    dims, act = [], None
    expect_linear = True
    for m in mods:
        if expect_linear:
            if not isinstance(m, nn.Linear) or m.bias is None:
                return None
            if dims and dims[-1] != m.in_features:
                return None
            if not dims:
                dims.append(m.in_features)
            dims.append(m.out_features)
            expect_linear = False
        else:
            a = _ACTS.get(type(m))
            if a is None or (act is not None and a != act):
                return None
            act = a
            expect_linear = True
    if expect_linear or len(dims) < 2:
        return None
    return (dims, (act or "relu")) if tail is None else (dims, (act or "relu"), tail)


def define_model_log_prob(model, model_loss, x, y, params_flattened_list, params_shape_list, tau_list, tau_out,
                          normalizing_const=1., predict=False, prior_scale=1.0, device='cpu'):
    """S:1093-1201.  Returns ``log_prob_func(flat_params)``."""
    names = [n for n, _ in model.named_parameters()]
    buffers = {n: b for n, b in model.named_buffers()}
    taus = [float(t) for t in tau_list]                  # host scalars: the closure stays capturable as a HIP graph
    half_log_taus = [0.5 * math.log(t) for t in taus]
    x_dev = None if x is None else x.to(device)
    y_dev = None if y is None else y.to(device)
    sizes = list(params_flattened_list)
    shapes = list(params_shape_list)

    def log_prob_func(params):
        i_prev = 0
        l_prior = torch.zeros_like(params[0])
        tensors = {}
        for name, n, shape, tau, hlt in zip(names, sizes, shapes, taus, half_log_taus):
            w = params[i_prev:i_prev + n]
            # Normal(0, tau^-1/2).log_prob(w).sum()   (S:1143, S:1156)
            l_prior = l_prior + (-0.5 * tau * (w * w) + (hlt - 0.9189385332046727)).sum()
            tensors[name] = w.reshape(shape)
            i_prev += n
        if x_dev is None:
            return l_prior / prior_scale                                         # S:1160-1162
        tensors.update(buffers)
        output = torch.func.functional_call(model, tensors, (x_dev,))
        if model_loss == 'binary_class_linear_output':
            ll = -tau_out * nn.functional.binary_cross_entropy_with_logits(output, y_dev, reduction='sum')
        elif model_loss == 'multi_class_linear_output':
            ll = -tau_out * nn.functional.cross_entropy(output, y_dev.long().view(-1), reduction='sum')
        elif model_loss == 'multi_class_log_softmax_output':
            ll = -tau_out * nn.functional.nll_loss(output, y_dev.long().view(-1))   # S:1180 (mean reduction)
        elif model_loss == 'regression':
            ll = -0.5 * tau_out * ((output - y_dev) ** 2).sum(0)                  # S:1184, shape (O,)
        elif callable(model_loss):
            ll = -model_loss(output, y_dev).sum(0)                                # S:1188
        else:
            raise NotImplementedError()                                           # S:1190
        if predict:
            return (ll + l_prior / prior_scale), output
        return ll + l_prior / prior_scale

    st = _mlp_structure(model) if (x is not None and not predict and x.dim() == 2) else None
    spec_tau_out, spec_loss = float(tau_out), model_loss
    if st is not None and len(st) == 3:
        # the model ends in a log-softmax: with 'multi_class_log_softmax_output' (nll_loss, MEAN reduction, S:1180) the closure is
        # -(tau_out / N) * cross-entropy of the logits - the softmax kernel with a scaled precision; other pairings: callback path
        if model_loss == 'multi_class_log_softmax_output' and st[0][-1] >= 2 and y_dev.numel() == x_dev.shape[0]:
            spec_tau_out, spec_loss, st = float(tau_out) / x_dev.shape[0], 'multi_class_linear_output', st[:2]
        else:
            st = None
    if st is not None:
        model_loss_ = spec_loss
        n_pts, n_out = x_dev.shape[0], st[0][-1]
        # the likelihoods with a native kernel: Gaussian on one output, Bernoulli with logits (any number of outputs, summed),
        # softmax cross-entropy on integer labels; the other kinds (and multi-output regression, whose closure returns one
        # value per output, S:1184) stay on the callback path
        if model_loss_ == 'regression' and n_out == 1 and y_dev.numel() == n_pts:
            y_spec = y_dev.reshape(n_pts, 1)
        elif model_loss_ == 'binary_class_linear_output' and y_dev.numel() == n_pts * n_out:
            y_spec = y_dev.reshape(n_pts, n_out)
        elif model_loss_ == 'multi_class_linear_output' and n_out >= 2 and y_dev.numel() == n_pts:
            # the kernel indexes the logits with the label: only integer-valued labels in [0, n_out) may reach it (the
            # reference's CrossEntropyLoss raises on anything else; a closure with such labels stays on the callback path,
            # where torch raises the same error) -- one reduction on the device per closure, at definition time
            yl = y_dev.reshape(n_pts)
            yd = yl.double()
            ok = bool(((yd == yd.round()) & (yd >= 0) & (yd < n_out)).all()) if n_pts > 0 else False
            y_spec = yl if ok else None
        else:
            y_spec = None
        if y_spec is not None:
            log_prob_func._hta_spec = dict(dims=st[0], act=st[1], X=x_dev, Y=y_spec, tau_list=list(taus), tau_out=spec_tau_out,
                                           prior_scale=float(prior_scale), loss=model_loss_)
    return log_prob_func


def define_split_model_log_prob(model, model_loss, train_loader, num_splits, params_flattened_list,
                                params_shape_list, tau_list, tau_out, normalizing_const=1., predict=False,
                                device='cpu', verbose=True):
    """S:1203-1258: one closure per DataLoader batch, prior divided by ``num_splits``."""
    fns = []
    for batch_idx, (data, target) in enumerate(train_loader):
        if batch_idx > num_splits - 1:
            break
        fns.append(define_model_log_prob(model, model_loss, data.clone(), target.clone(), params_flattened_list,
                                         params_shape_list, tau_list, tau_out, normalizing_const=normalizing_const,
                                         prior_scale=num_splits, predict=predict, device=device))
    if verbose:
        print('Number of splits: ', len(fns), ' , each of batch size ', train_loader.batch_size, '\n')
    return fns


def _shapes_and_tau(model, tau_list):
    shapes, sizes = [], []
    build = tau_list is None
    if build:
        tau_list = []
    for w in model.parameters():
        shapes.append(w.shape)
        sizes.append(w.nelement())
        if build:
            tau_list.append(torch.tensor(1.))                                     # S:1354-1355
    return shapes, sizes, tau_list


@host_inputs
def sample_model(model, x, y, params_init, model_loss='multi_class_linear_output', num_samples=10,
                 num_steps_per_sample=10, step_size=0.1, burn=0, inv_mass=None, jitter=None, normalizing_const=1.,
                 softabs_const=None, explicit_binding_const=100, fixed_point_threshold=1e-5,
                 fixed_point_max_iterations=1000, jitter_max_tries=10, sampler=Sampler.HMC, integrator=Integrator.IMPLICIT,
                 metric=Metric.HESSIAN, debug=False, tau_out=1., tau_list=None, store_on_GPU=True, desired_accept_rate=0.8, verbose=True,
                 **ext):
    """S:1261-1362."""
    from . import samplers as S
    shapes, sizes, tau_list = _shapes_and_tau(model, tau_list)
    f = define_model_log_prob(model, model_loss, x, y, sizes, shapes, tau_list, tau_out,
                              normalizing_const=normalizing_const, device=params_init.device)
    return S.sample(f, params_init, num_samples=num_samples, num_steps_per_sample=num_steps_per_sample,
                    step_size=step_size, burn=burn, jitter=jitter, inv_mass=inv_mass,
                    normalizing_const=normalizing_const, softabs_const=softabs_const,
                    explicit_binding_const=explicit_binding_const, fixed_point_threshold=fixed_point_threshold,
                    fixed_point_max_iterations=fixed_point_max_iterations, jitter_max_tries=jitter_max_tries,
                    sampler=sampler, integrator=integrator, metric=metric, debug=debug,
                    desired_accept_rate=desired_accept_rate, store_on_GPU=store_on_GPU, verbose=verbose, **ext)


@host_inputs
def sample_split_model(model, train_loader, params_init, num_splits, model_loss='multi_class_linear_output',
                       num_samples=10, num_steps_per_sample=10, step_size=0.1, burn=0, inv_mass=None, jitter=None,
                       normalizing_const=1., softabs_const=None, explicit_binding_const=100,
                       fixed_point_threshold=1e-5, fixed_point_max_iterations=1000, jitter_max_tries=10,
                       sampler=Sampler.HMC, integrator=Integrator.SPLITTING, metric=Metric.HESSIAN, debug=False,
                       tau_out=1., tau_list=None, store_on_GPU=True, desired_accept_rate=0.8, verbose=True, **ext):
    """S:1364-1466."""
    from . import samplers as S
    shapes, sizes, tau_list = _shapes_and_tau(model, tau_list)
    fl = define_split_model_log_prob(model, model_loss, train_loader, num_splits, sizes, shapes, tau_list, tau_out,
                                     normalizing_const=1., predict=False, device=params_init.device, verbose=verbose)
    return S.sample(fl, params_init, num_samples=num_samples, num_steps_per_sample=num_steps_per_sample,
                    step_size=step_size, burn=burn, jitter=jitter, inv_mass=inv_mass,
                    normalizing_const=normalizing_const, softabs_const=softabs_const,
                    explicit_binding_const=explicit_binding_const, fixed_point_threshold=fixed_point_threshold,
                    fixed_point_max_iterations=fixed_point_max_iterations, jitter_max_tries=jitter_max_tries,
                    sampler=sampler, integrator=integrator, metric=metric, debug=debug,
                    desired_accept_rate=desired_accept_rate, store_on_GPU=store_on_GPU, verbose=verbose, **ext)


#: how predict_model evaluated its closures so far (tests / diagnostics)
predict_stats = {"batched": 0, "looped": 0}


def _eval_all(f, samples, dev):
    """(log_prob, output) of the predict-closure `f` for every sample: batched evaluations over the stacked samples
    (`torch.func.vmap`) instead of the reference's Python loop over samples (S:1530-1552).  The chunk is sized from the
    MEASURED peak memory of one single-sample evaluation (hidden and convolution activations included, not only the
    output) against a 256 MB budget; an out-of-memory error frees the cache and halves the chunk, down to the reference's
    per-sample loop -- predict_model never fails where the loop would succeed.  A closure vmap cannot batch falls back to
    that loop as well.  Returns (list of log-probs, list of outputs)."""
    S = len(samples)
    if S == 0:
        return [], []
    from .samplelist import SampleList
    if isinstance(samples, SampleList) and not samples._done:      # sample()'s lazy list: the rows already are one tensor
        stacked = samples.tensor.to(dev)
        flat = stacked
        uniform = stacked.dim() == 2
    else:
        flat = [t.to(dev) for t in samples]
        uniform = all(t.dim() == 1 and t.shape == flat[0].shape for t in flat)
        stacked = torch.stack(flat) if uniform else None
    if uniform:
        try:
            with torch.no_grad():
                on_gpu = flat[0].is_cuda
                if on_gpu:
                    torch.cuda.synchronize(dev)
                    torch.cuda.reset_peak_memory_stats(dev)
                    base = torch.cuda.memory_allocated(dev)
                v0, o0 = f(flat[0])
                if on_gpu:
                    torch.cuda.synchronize(dev)
                    per = max(1, torch.cuda.max_memory_allocated(dev) - base)
                else:
                    per = max(1, int(o0.numel()) * 64)
                per = max(per, int(o0.numel()) * o0.element_size() * 4)
                chunk = max(1, min(S, (256 << 20) // per))
                vs, os_ = [], []
                c0 = 0
                while c0 < S:
                    try:
                        v, o = torch.func.vmap(f)(stacked[c0:c0 + chunk])
                    except torch.OutOfMemoryError:
                        v = o = None
                        if on_gpu:
                            torch.cuda.empty_cache()
                        if chunk == 1:
                            raise _LoopInstead()
                        chunk = max(1, chunk // 2)
                        predict_stats["oom_halvings"] = predict_stats.get("oom_halvings", 0) + 1
                        continue
                    vs.append(v); os_.append(o)
                    c0 += chunk
                v = torch.cat(vs); o = torch.cat(os_)
            predict_stats["batched"] += 1
            return [v[k].reshape(v0.shape) for k in range(S)], [o[k] for k in range(S)]
        except _LoopInstead:
            pass
        except (RuntimeError, TypeError, ValueError, NotImplementedError) as e:
            if isinstance(e, torch.AcceleratorError):
                raise
            if isinstance(e, torch.OutOfMemoryError):
                if flat[0].is_cuda:
                    torch.cuda.empty_cache()
    predict_stats["looped"] += 1
    vs, os_ = [], []
    with torch.no_grad():
        for t in flat:
            v, o = f(t)
            vs.append(v); os_.append(o)
    return vs, os_


class _LoopInstead(Exception):
    """internal: the batched evaluation ran out of memory even one sample at a time under vmap"""


#: how predict_model produced its last result: "native" (hta_net_forward + element-wise log-probs) or "torch" (vmap of the closure)
predict_route = {"last": None, "native": 0, "torch": 0}
_NATIVE_LOSSES = ("regression", "binary_class_linear_output", "multi_class_linear_output")


def _stacked_samples(samples, dev):
    """[S, D] tensor of a list of (D,) samples on `dev` (zero-copy for sample()'s lazy list), or None if the rows are not that."""
    from .samplelist import SampleList
    if isinstance(samples, SampleList) and not samples._done:
        t = samples.tensor
    else:
        rows = list(samples)
        if not rows or any((not torch.is_tensor(r)) or r.dim() != 1 or r.shape != rows[0].shape for r in rows):
            return None
        t = torch.stack([r.to(dev) for r in rows])
    return t.to(dev).contiguous() if t.dim() == 2 else None


def _native_forward_ok(model, stacked, x, model_loss):
    st = _mlp_structure(model)
    if st is None or len(st) != 2 or model_loss not in _NATIVE_LOSSES:
        return None
    dims, act = st
    if stacked is None or not stacked.is_cuda or stacked.dtype not in (torch.float32, torch.float64):
        return None
    if x.dim() != 2 or x.shape[1] != dims[0] or stacked.shape[1] != sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1)):
        return None
    if not native_forward_fits(dims, stacked.element_size()):
        return None
    return dims, act


#: csrc/net_forward.hip: FW_MAXW (widest staged layer), FW_MAXO (outputs of the streamed one-hidden-layer form), FW_TPB (points per
#: workgroup), FW_MAXL (Linear layers), and the LDS of one CU
_FW_MAXW, _FW_MAXO, _FW_TPB, _FW_MAXL, _FW_LDS = 256, 16, 64, 8, 160 * 1024


def native_forward_fits(dims, itemsize):
    """The limits hta_net_forward checks (csrc/net_forward.hip: net_forward), mirrored so that a model outside them takes the
    torch path instead of raising: the activations of one layer for FW_TPB points are staged twice in LDS, so the widest staged
    layer - the input alone in the streamed form (one hidden layer wider than FW_MAXW, <= FW_MAXO outputs), every input / hidden
    width otherwise - is bounded by FW_MAXW AND by 160 KiB / (2 x FW_TPB x itemsize): 784 inputs or 200-wide float64 layers do
    not fit (the gate admitted them and predict_model raised)."""
    nl = len(dims) - 1
    if nl < 1 or nl > _FW_MAXL or min(dims) < 1:
        return False
    streamed = nl == 2 and dims[2] <= _FW_MAXO and dims[1] > _FW_MAXW
    wmax = dims[0] if streamed else max(dims[:-1])
    wmax = (wmax + 3) & ~3
    if wmax > _FW_MAXW and not streamed:
        return False
    return 2 * wmax * _FW_TPB * itemsize <= _FW_LDS


def _native_log_probs(stacked, out, y, sizes, tau_list, tau_out, model_loss, prior_scale):
    """The closure's predict-mode value (S:1141-1199) for every sample from the network outputs `out` [S, N, O]: element-wise /
    reduction kernels over the whole batch of samples.  Returns [S] or [S, O] (regression keeps one value per output, S:1184),
    or None when `y` does not have the shape the closure's own arithmetic would treat element for element."""
    S, N, O = out.shape
    taus = [float(t) for t in tau_list]
    prior = torch.zeros(S, dtype=out.dtype, device=out.device)
    i0 = 0
    for n, tau in zip(sizes, taus):
        w = stacked[:, i0:i0 + n]
        prior = prior + (-0.5 * tau * (w * w) + (0.5 * math.log(tau) - 0.9189385332046727)).sum(1)      # S:1143, S:1156
        i0 += n
    yd = y.to(device=out.device)
    if model_loss == "regression":
        if tuple(yd.shape) != (N, O):
            return None
        ll = -0.5 * tau_out * ((out - yd.to(out.dtype)) ** 2).sum(1)                                      # [S, O]  (S:1184)
        return ll + (prior / prior_scale)[:, None]
    if model_loss == "binary_class_linear_output":
        if tuple(yd.shape) != (N, O):
            return None
        ll = -tau_out * nn.functional.binary_cross_entropy_with_logits(out, yd.to(out.dtype).expand(S, N, O), reduction="none").sum((1, 2))
        return ll + prior / prior_scale
    if yd.numel() != N:
        return None
    lab = yd.long().view(-1)
    ll = -tau_out * nn.functional.cross_entropy(out.reshape(S * N, O), lab.repeat(S), reduction="none").view(S, N).sum(1)
    return ll + prior / prior_scale


def _predict_batch_native(model, stacked, x, y, sizes, tau_list, tau_out, model_loss, prior_scale):
    ok = _native_forward_ok(model, stacked, x, model_loss)
    if ok is None:
        return None
    from . import _abi
    dims, act = ok
    X = x.to(device=stacked.device, dtype=stacked.dtype).contiguous()
    out = torch.empty(stacked.shape[0], X.shape[0], dims[-1], dtype=stacked.dtype, device=stacked.device)
    lp = None
    try:
        _abi.net_forward(stacked, dims, act, X, out)
    except RuntimeError:                 # a limit of the kernel the gate above does not know: the torch path answers
        predict_route["native_refused"] = _abi.last_error() if hasattr(_abi, "last_error") else "hta_net_forward refused"
        return None
    lp = _native_log_probs(stacked, out, y, sizes, tau_list, tau_out, model_loss, prior_scale)
    if lp is None:
        return None
    return lp, out


def predict_model(model, samples, x=None, y=None, test_loader=None, model_loss='multi_class_linear_output',
                  tau_out=1., tau_list=None, verbose=False):
    """S:1468-1562: evaluate every sample; returns (stack(pred)[S, N, O], list of log-probs).

    Recognised model families (Linear / activation chains: `_mlp_structure`; the losses with a native likelihood) run
    NATIVELY: one forward-only launch over all S parameter rows (csrc/net_forward.hip: `hta_net_forward`) per batch of
    points, log-probs by batched element-wise reductions on the device, one device-to-host copy for the loader form (the
    reference moves every sample's value to the host, S:1536).  Anything else: the closure under `torch.func.vmap`
    (`_eval_all`), chunked by measured memory, down to the reference's loop."""
    shapes, sizes, tau_list = _shapes_and_tau(model, tau_list)
    dev = samples[0].device
    with torch.no_grad():
        if test_loader is not None and (x is None or isinstance(test_loader, torch.utils.data.DataLoader)):
            # S:1520-1541: one closure per batch through define_split_model_log_prob, i.e. the prior divided by the number of
            # batches in every closure (counted once over the loader); the batch count is the reference's own formula
            # (S:1522-1525: a float when the batch size divides the data set, round() + 1 otherwise), kept as is.
            if isinstance(test_loader, torch.utils.data.DataLoader):
                n_data, bs = len(test_loader.dataset), test_loader.batch_size
                num_batches = n_data / bs if n_data % bs == 0 else int(round(n_data / bs) + 1)
            else:                                                                  # any other iterable of (x, y) batches
                test_loader = list(test_loader)
                num_batches = len(test_loader)
            stacked = _stacked_samples(samples, dev) if dev.type == "cuda" else None
            if stacked is not None:
                parts, lp_sum = [], None
                for batch_idx, (data, target) in enumerate(test_loader):
                    if batch_idx > num_batches - 1:                                # (define_split_model_log_prob's own cut, S:1246)
                        break
                    r = _predict_batch_native(model, stacked, data, target, sizes, tau_list, tau_out, model_loss, num_batches)
                    if r is None:
                        parts = None
                        break
                    lp_sum = r[0] if lp_sum is None else lp_sum + r[0]
                    parts.append(r[1])
                if parts:
                    predict_route["last"] = "native"; predict_route["native"] += 1
                    return torch.cat(parts, 1), list(lp_sum.cpu().unbind(0))       # S:1536: the values live on the host
            fns = define_split_model_log_prob(model, model_loss, test_loader, num_batches, sizes, shapes, tau_list, tau_out,
                                              predict=True, device=dev, verbose=verbose)
            per_batch = [_eval_all(f, samples, dev) for f in fns]
            lp_all = None
            for vs, _ in per_batch:                                                # one copy per batch, not one per (sample, batch)
                v = torch.stack([t.reshape(vs[0].shape) for t in vs]).cpu()
                lp_all = v if lp_all is None else lp_all + v
            preds = [torch.cat([os_[k] for _, os_ in per_batch], 0) for k in range(len(samples))]
            lps = list(lp_all.unbind(0))
            predict_route["last"] = "torch"; predict_route["torch"] += 1
        elif x is not None and y is not None:
            if x.device != dev:                                                    # S:1544-1545
                raise RuntimeError('x on device: {} and samples on device: {}'.format(x.device, dev))
            stacked = _stacked_samples(samples, dev) if dev.type == "cuda" else None
            r = _predict_batch_native(model, stacked, x, y, sizes, tau_list, tau_out, model_loss, 1.0) if stacked is not None else None
            if r is not None:
                predict_route["last"] = "native"; predict_route["native"] += 1
                return r[1], list(r[0].unbind(0))
            f = define_model_log_prob(model, model_loss, x, y, sizes, shapes, tau_list, tau_out, predict=True, device=dev)
            lps, preds = _eval_all(f, samples, dev)
            predict_route["last"] = "torch"; predict_route["torch"] += 1
        else:
            raise RuntimeError('Val data not defined (i.e. arguments x, y, val_loader are all not defined)')   # S:1557
    return torch.stack(preds), lps


# ---- native engines (filled in by mlp.py once the kernel library exports them) -------------------
def native_split_engine(log_prob_list, theta0, integrator=None):
    try:
        from . import mlp
    except ImportError:
        return None
    return mlp.split_engine(log_prob_list, theta0, integrator)


def native_hmc_engine(log_prob_func, theta0):
    try:
        from . import mlp
    except ImportError:
        return None
    return mlp.hmc_engine(log_prob_func, theta0)

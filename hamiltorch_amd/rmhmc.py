"""Riemannian-manifold HMC pieces of the API mirror (reference: hamiltorch/samplers.py, `S:`):
``fisher`` S:69-127, ``cholesky_inverse`` S:130-149, ``gibbs`` (RMHMC) S:183-184, ``rm_hamiltonian``
S:677-736, the explicit integrator S:389-462 and the RMHMC branch of ``sample`` S:969-1026.

The eigendecomposition / soft-abs map / solves / log-determinant run in
``csrc/rmhmc_metric.hip`` (one workgroup per system, matrices in LDS).  ``fisher``,
``cholesky_inverse``, ``rm_hamiltonian`` and the momentum draw accept any ``log_prob_func`` (its
Hessian comes from ``torch.func.hessian``); the explicit *integrator* needs d H / d theta, which for
a general target involves third derivatives of log p -- it is implemented for constant-curvature
targets (``GaussianTarget``), the family of BASELINE configs 3 and 5 (SURVEY 8f N1 is the rest).
"""
from __future__ import annotations

import torch

from . import _abi, util
from .enums import Metric
from .models import as_gaussian


def _metric_kind(metric):
    if metric == Metric.SOFTABS:
        return _abi.METRIC_SOFTABS
    if metric == Metric.HESSIAN:
        return _abi.METRIC_HESSIAN
    if metric == Metric.JACOBIAN_DIAG:
        raise NotImplementedError("Metric.JACOBIAN_DIAG is outside the accelerated path")
    raise ValueError('Unknown metric: {}'.format(metric))            # S:127


def _batch(t, what="params"):
    if t.dim() == 1:
        return t.detach().reshape(1, -1).contiguous(), True
    return t.detach().contiguous(), False


def _curvature(theta, log_prob_func):
    """(-Hessian [B,D,D] or shared [D,D], stride, logp[B], gaussian target or None)."""
    tgt = as_gaussian(log_prob_func, theta)
    if tgt is not None:
        d = theta - tgt.mean
        logp = tgt.log_norm - 0.5 * ((d @ tgt.precision) * d).sum(-1)
        return tgt.precision, 0, logp.contiguous(), tgt
    f = lambda w: log_prob_func(w).sum()  # noqa: E731
    H = torch.func.vmap(torch.func.hessian(f))(theta)                     # S:108
    logp = torch.func.vmap(f)(theta)
    D = theta.shape[1]
    return (-H).contiguous(), D * D, logp.contiguous(), None


def fisher(params, log_prob_func=None, jitter=None, softabs_const=1e6, metric=Metric.HESSIAN, seed=None,
           chain_offset=0, draw=0, sub=0):
    """(G, abs_eigenvalues or None): (D,D)/(D,) for one chain, (C,D,D)/(C,D) for a batch.
    Soft-abs eigenvalues come back in ascending order of the underlying eigenvalue only up to the
    solver's ordering (Jacobi does not sort); G, log|G| and G^-1 p do not depend on it."""
    theta, one = _batch(params)
    _abi.require_device(theta, "params")
    kind = _metric_kind(metric)
    Hs, stride, logp, _ = _curvature(theta, log_prob_func)
    if util.has_nan_or_inf(logp) or util.has_nan_or_inf(Hs):
        raise util.LogProbError()                                        # S:97-99, S:110-112
    C, D = theta.shape
    G = torch.empty(C, D, D, dtype=theta.dtype, device=theta.device)
    lam = torch.empty(C, D, dtype=theta.dtype, device=theta.device) if kind == _abi.METRIC_SOFTABS else None
    seed = util.next_stream_seed() if (seed is None and jitter is not None) else (seed or 0)
    _abi.metric_eval(theta, C, D, kind, Hs, stride, softabs_const, jitter, seed, chain_offset, draw, sub,
                     G_out=G, lam_out=lam)
    if one:
        return G[0], (None if lam is None else lam[0])
    return G, lam


def cholesky_inverse(fish, momentum):
    """G^-1 p (S:146-148): (D,1) for one system as the reference, (C,D) for a batch."""
    G = fish.detach()
    one = G.dim() == 2
    G = (G.reshape(1, *G.shape) if one else G).contiguous()
    m = momentum.detach().reshape(G.shape[0], -1).contiguous()
    _abi.require_device(G, "fish")
    C, D = m.shape
    x = torch.empty_like(m)
    _abi.metric_eval(m, C, D, _abi.METRIC_HESSIAN, G, D * D, 0.0, m=m, x_out=x)
    return x.reshape(-1, 1) if one else x


def rm_hamiltonian(params, momentum, log_prob_func, jitter, softabs_const=1e6, metric=Metric.HESSIAN, seed=None,
                   chain_offset=0, draw=0, sub=0):
    """S:710-731.  Shape (1,1) for one chain (as the reference), (C,) for a batch."""
    theta, one = _batch(params)
    p, _ = _batch(momentum, "momentum")
    _abi.require_device(theta, "params")
    kind = _metric_kind(metric)
    Hs, stride, logp, tgt = _curvature(theta, log_prob_func)
    C, D = theta.shape
    H = torch.empty(C, dtype=theta.dtype, device=theta.device)
    seed = util.next_stream_seed() if (seed is None and jitter is not None) else (seed or 0)
    if tgt is not None:
        _abi.metric_eval(theta, C, D, kind, Hs, 0, softabs_const, jitter, seed, chain_offset, draw, sub, X=theta,
                         Pm=tgt.precision, mu=tgt.mean, log_norm=tgt.log_norm, m=p, H_out=H)
    else:
        _abi.metric_eval(theta, C, D, kind, Hs, stride, softabs_const, jitter, seed, chain_offset, draw, sub, m=p, H_out=H)
        H = H - logp.to(H.dtype)
    if one and (util.has_nan_or_inf(logp) or util.has_nan_or_inf(H)):
        raise util.LogProbError()                                        # S:717-723, S:732-734
    return H.reshape(1, 1) if one else H


def gibbs(theta, log_prob_func, jitter, softabs_const, metric, seed, chain_offset, draw):
    """p ~ N(0, G(theta)) = chol(G) z  (S:183-184)."""
    kind = _metric_kind(metric)
    if softabs_const is None and kind == _abi.METRIC_SOFTABS:
        raise TypeError("softabs_const must be set for Metric.SOFTABS")   # the reference fails at S:120
    Hs, stride, _, _ = _curvature(theta, log_prob_func)
    C, D = theta.shape
    p = torch.empty_like(theta)
    _abi.metric_eval(theta, C, D, kind, Hs, stride, softabs_const, jitter, seed, chain_offset, draw, 0, p_out=p)
    return p


def _need_gaussian(log_prob_func, theta):
    tgt = as_gaussian(log_prob_func, theta)
    if tgt is None:
        raise NotImplementedError(
            "explicit RMHMC is accelerated for constant-curvature targets (hamiltorch_amd.GaussianTarget or a "
            "MultivariateNormal.log_prob); a general log_prob_func needs third derivatives of log p "
            "(samplers.py:398 differentiates through hessian + eigh) -- not in the native path yet")
    return tgt


def explicit_leapfrog(params, momentum, log_prob_func, steps, step_size, jitter, softabs_const, omega, metric,
                      seed=None, chain_offset=0, draw=0):
    """S:389-462.  Returns ([ret_params, params_copy], [ret_momenta, momentum_copy]) like the reference."""
    theta, one = _batch(params)
    p, _ = _batch(momentum, "momentum")
    theta, p = theta.clone(), p.clone()
    _abi.require_device(theta, "params")
    tgt = _need_gaussian(log_prob_func, theta)
    thc, pc = theta.clone(), p.clone()
    pt = torch.empty((steps,) + theta.shape, dtype=theta.dtype, device=theta.device)
    pp = torch.empty_like(pt)
    seed = util.next_stream_seed() if (seed is None and jitter is not None) else (seed or 0)
    _abi.rmhmc_gaussian_leapfrog(theta, p, thc, pc, tgt.precision, tgt.mean, _metric_kind(metric), softabs_const, jitter,
                                 seed, chain_offset, draw, steps, step_size, omega, pt, pp)
    unb = (lambda t: t[0]) if one else (lambda t: t)
    return [[unb(t) for t in pt.unbind(0)], unb(thc)], [[unb(t) for t in pp.unbind(0)], unb(pc)]


def sample_explicit(log_prob_func, theta0, N, L, eps, burn, jitter, softabs_const, omega, metric, seed, chain_offset,
                    verbose):
    """The RMHMC / EXPLICIT branch of sample() (S:969-1026): one C call enqueues the whole run."""
    from .samplers import _num_rows
    tgt = _need_gaussian(log_prob_func, theta0)
    kind = _metric_kind(metric)
    if softabs_const is None and kind == _abi.METRIC_SOFTABS:
        raise TypeError("softabs_const must be set for Metric.SOFTABS")
    C, D = theta0.shape
    S = _num_rows(N, burn)
    samples = torch.empty((S, C, D), dtype=theta0.dtype, device=theta0.device)
    samples[0].copy_(theta0)
    cur = theta0.clone()
    rejected = torch.zeros(C, dtype=torch.int32, device=theta0.device)
    ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, theta0.element_size()), dtype=torch.uint8, device=theta0.device)
    prog = util._Progress('Sampling (Sampler.RMHMC; Integrator.EXPLICIT)', N, verbose)
    _abi.rmhmc_gaussian_sample(cur, theta0, tgt.precision, tgt.mean, tgt.log_norm, kind, softabs_const, jitter, L, eps,
                               omega, N, 0, burn, seed, chain_offset, samples, rejected, ws)
    prog.end()
    return samples, rejected

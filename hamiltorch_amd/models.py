"""Model plugins: objects that are valid ``log_prob_func`` callables (the reference's callback
contract, samplers.py:272-274) AND carry the closed form the HIP kernels need, so that
``sample()`` can run whole trajectories on-device instead of calling back into torch.

An unrecognised callable still works: it is evaluated by torch (``torch.func.vmap`` over
chains) with the HIP kernels doing the state updates in between (samplers._GenericHMC).
"""
from __future__ import annotations

import math

import torch


class GaussianTarget:
    """log p(w) = log_norm - 0.5 (w - mean)^T precision (w - mean).

    Callable exactly like the closures in the reference's examples
    (``MultivariateNormal(mean, cov).log_prob(w).sum()``, tests/test_util.py:98-101) -- the
    same object can be handed to the reference on the CPU and to this engine on the GPU.

    Parameters: ``mean`` (D,), and either ``covariance`` or ``precision`` (D,D).
    ``normalized=True`` adds the Gaussian normaliser (as MultivariateNormal.log_prob does);
    ``False`` gives the bare quadratic form ``-0.5 d^T P d``.
    """

    def __init__(self, mean, covariance=None, precision=None, normalized=True):
        if (covariance is None) == (precision is None):
            raise ValueError("GaussianTarget: give exactly one of covariance / precision")
        mean = torch.as_tensor(mean)
        if covariance is not None:
            covariance = torch.as_tensor(covariance, dtype=mean.dtype, device=mean.device)
            precision = torch.linalg.inv(covariance.double()).to(mean.dtype)
        else:
            precision = torch.as_tensor(precision, dtype=mean.dtype, device=mean.device)
        if mean.dim() != 1 or precision.shape != (mean.numel(), mean.numel()):
            raise ValueError("GaussianTarget: mean must be (D,), matrix (D,D)")
        # the quadratic form only sees the symmetric part; symmetrise once so grad = -P d exactly
        self.precision = (0.5 * (precision + precision.t())).contiguous()
        self.mean = mean.contiguous()
        if normalized:
            _, logdet = torch.linalg.slogdet(self.precision.double())
            self.log_norm = float(-0.5 * mean.numel() * math.log(2.0 * math.pi) + 0.5 * logdet)
        else:
            self.log_norm = 0.0

    @property
    def dim(self):
        return self.mean.numel()

    def to(self, device=None, dtype=None):
        out = object.__new__(GaussianTarget)
        out.mean = self.mean.to(device=device, dtype=dtype).contiguous()
        out.precision = self.precision.to(device=device, dtype=dtype).contiguous()
        out.log_norm = self.log_norm
        return out

    def __call__(self, w):
        d = w - self.mean
        return self.log_norm - 0.5 * torch.dot(d, torch.mv(self.precision, d))

    # closed forms (batched over leading dims) used by parity tests and the generic fallbacks
    def grad(self, w):
        return -((w - self.mean) @ self.precision.t())

    def neg_hessian(self):
        return self.precision


def as_gaussian(log_prob_func, like=None):
    """Return a GaussianTarget if ``log_prob_func`` is one, or is the bound ``log_prob`` of a
    ``torch.distributions.MultivariateNormal``; else None."""
    if isinstance(log_prob_func, GaussianTarget):
        tgt = log_prob_func
    else:
        owner = getattr(log_prob_func, "__self__", None)
        if isinstance(owner, torch.distributions.MultivariateNormal) and \
                getattr(log_prob_func, "__name__", "") == "log_prob" and owner.loc.dim() == 1:
            tgt = GaussianTarget(owner.loc, covariance=owner.covariance_matrix)
        else:
            return None
    if like is not None and (tgt.mean.device != like.device or tgt.mean.dtype != like.dtype):
        tgt = tgt.to(device=like.device, dtype=like.dtype)
    return tgt

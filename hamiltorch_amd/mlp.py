"""Native engines for the Bayesian-MLP closures built by ``bnn.define_model_log_prob`` (S:1093-1258).

A closure carries ``_hta_spec`` (dims, activation, data, precisions) when the model is
``Sequential(Linear, act, Linear)`` with a scalar regression output; ``sample`` then runs the
whole (split-)HMC trajectory loop in ``csrc/mlp_hmc.hip`` instead of calling back into torch.
Anything else falls back to the generic-callback path.
"""
from __future__ import annotations

import torch

from . import _abi, util


def _common_spec(fns):
    specs = [getattr(f, "_hta_spec", None) for f in fns]
    if any(s is None for s in specs):
        return None
    s0 = specs[0]
    if len(s0["dims"]) != 3 or s0["dims"][-1] != 1 or s0["dims"][0] > 32 or s0["dims"][1] > 1024:
        return None
    nb = s0["X"].shape[0]
    for s in specs[1:]:
        if (s["dims"], s["act"], s["tau_list"], s["tau_out"], s["prior_scale"]) != \
                (s0["dims"], s0["act"], s0["tau_list"], s0["tau_out"], s0["prior_scale"]) or s["X"].shape[0] != nb:
            return None
    return specs


class _MLPEngine:
    """run() contract of samplers._GaussianHMC."""

    def __init__(self, specs, fallback):
        self.specs, self.fallback = specs, fallback
        s0 = specs[0]
        self.n_in, self.H = s0["dims"][0], s0["dims"][1]
        self.act, self.tau, self.tau_out, self.prior_scale = s0["act"], s0["tau_list"], s0["tau_out"], s0["prior_scale"]
        self.M, self.Nb = len(specs), s0["X"].shape[0]

    def _data(self, like):
        X = torch.cat([s["X"].reshape(self.Nb, self.n_in) for s in self.specs]).to(like).contiguous()
        Y = torch.cat([s["Y"].reshape(self.Nb) for s in self.specs]).to(like).contiguous()
        return X, Y

    def run(self, theta0, N, L, eps, burn, inv_mass, seed, chain_offset, verbose, label):
        from .samplers import _mass_operands, _num_rows
        kind, im, mf = _mass_operands(inv_mass, theta0)
        if kind == _abi.MASS_FULL or theta0.shape[1] != self.H * self.n_in + 2 * self.H + 1:
            return self.fallback().run(theta0, N, L, eps, burn, inv_mass, seed, chain_offset, verbose, label)
        C, D = theta0.shape
        X, Y = self._data(theta0)
        samples = torch.empty((_num_rows(N, burn), C, D), dtype=theta0.dtype, device=theta0.device)
        samples[0].copy_(theta0)
        cur = theta0.clone()
        rejected = torch.zeros(C, dtype=torch.int32, device=theta0.device)
        prog = util._Progress('Sampling ' + label, N, verbose)
        _abi.mlp_hmc_sample(cur, theta0, self.n_in, self.H, self.act, X, Y, self.M, self.Nb, self.tau, self.tau_out,
                            self.prior_scale, kind, im, mf, L, eps, N, 0, burn, seed, chain_offset, samples, rejected)
        prog.end()
        return samples, rejected


def split_engine(log_prob_list, theta0):
    specs = _common_spec(log_prob_list)
    if specs is None or len(specs) < 2:
        return None
    from .samplers import _GenericHMC
    return _MLPEngine(specs, lambda: _GenericHMC(log_prob_list, split=True))


def hmc_engine(log_prob_func, theta0):
    specs = _common_spec([log_prob_func])
    if specs is None:
        return None
    from .samplers import _GenericHMC
    return _MLPEngine(specs, lambda: _GenericHMC(log_prob_func))

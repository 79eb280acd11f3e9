"""Native engines for the Bayesian-MLP closures built by ``bnn.define_model_log_prob`` (S:1093-1258).

A closure carries ``_hta_spec`` (dims, activation, data, precisions) when the model is a chain Linear, act, Linear, ...
(``bnn._mlp_structure``) with a Gaussian ('regression'), Bernoulli-with-logits ('binary_class_linear_output') or softmax
('multi_class_linear_output') likelihood; ``sample`` then runs the whole (split-)HMC trajectory loop natively instead of
calling back into torch: one hidden layer and one output in ``csrc/mlp_hmc.hip`` / ``mlp_mfma.hip``, other small shapes
(no or several hidden layers, several outputs, softmax) in ``csrc/netn_hmc.hip``.  Anything else falls back to the
generic-callback path.
"""
from __future__ import annotations

import torch

from . import _abi, util


def _common_spec(fns):
    specs = [getattr(f, "_hta_spec", None) for f in fns]
    if any(s is None for s in specs):
        return None
    s0 = specs[0]
    if _kernel_for(s0) is None:
        return None
    nb = s0["X"].shape[0]
    for s in specs[1:]:
        if (s["dims"], s["act"], s["tau_list"], s["tau_out"], s["prior_scale"], s["loss"]) != \
                (s0["dims"], s0["act"], s0["tau_list"], s0["tau_out"], s0["prior_scale"], s0["loss"]) or s["X"].shape[0] != nb:
            return None
    return specs


def _n_params(dims):
    return sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))


def _is_mlp3(dims, loss):
    """Two wide hidden layers and one output with a Gaussian likelihood - Linear(n_in, H1)-act-Linear(H1, H2)-act-Linear(H2, 1),
    n_in <= 4, H1, H2 <= 104, beyond the small-net kernel's widths: csrc/mlp3_mfma.hip (the reference's published split-HMC
    model, notebooks/hamiltorch_split_HMC_BNN_example.ipynb cell 9: 1-100-100-1)."""
    return (len(dims) == 4 and dims[-1] == 1 and 1 <= dims[0] <= _abi.MLP3_MAX_IN and max(dims[1], dims[2]) <= _abi.MLP3_MAX_WIDTH
            and loss == "regression" and (max(dims[1], dims[2]) > _abi.NETN_MAX_WIDTH or _n_params(dims) > _abi.NETN_MAX_PARAMS))


def _kernel_for(spec):
    """'mlp1': csrc/mlp_hmc.hip / mlp_mfma.hip (one hidden layer, one output, Gaussian or Bernoulli likelihood, any width up to
    1024); 'netn': the hta_netn_* entry points - csrc/netn_hmc.hip (1 .. 4 Linear layers, widths <= 64, <= 512 parameters, also
    softmax cross-entropy) or csrc/mlp3_mfma.hip (two hidden layers up to 104 wide, one output, Gaussian likelihood, fp32);
    None: the callback path."""
    dims, loss = spec["dims"], spec["loss"]
    if len(dims) == 3 and dims[-1] == 1 and dims[0] <= 32 and dims[1] <= 1024 and loss in _abi.LOSSES:
        return "mlp1"
    if _is_mlp3(dims, loss):
        return "netn"          # same entry points (hta_netn_*): the library dispatches to csrc/mlp3_mfma.hip (fp32)
    if 2 <= len(dims) <= _abi.NETN_MAX_LAYERS + 1 and max(dims) <= _abi.NETN_MAX_WIDTH and _n_params(dims) <= _abi.NETN_MAX_PARAMS \
            and sum(((dims[i + 1] + 3) // 4) * ((dims[i] + 4) // 4) for i in range(len(dims) - 1)) <= _abi.NETN_MAX_BLOCKS \
            and loss in _abi.NET_LOSSES:
        return "netn"
    return None


class _MLPEngine:
    """Same engine shape as samplers._GaussianHMC (begin / advance / finish / run / run_nuts)."""

    def __new__(cls, specs, fallback, integrator=0):
        from .samplers import _Engine

        class Impl(_Engine):
            def __init__(self, specs, fallback, integrator):
                self.specs, self.fallback, self.integrator = specs, fallback, integrator
                s0 = specs[0]
                self.dims, self.kernel = list(s0["dims"]), _kernel_for(s0)
                self.n_in, self.H = s0["dims"][0], s0["dims"][1]
                self.act, self.tau, self.tau_out, self.prior_scale = s0["act"], s0["tau_list"], s0["tau_out"], s0["prior_scale"]
                self.loss = s0["loss"]
                self.M, self.Nb = len(specs), s0["X"].shape[0]
                self._fb = None

            def begin(self, theta0, N, burn, inv_mass, seed, chain_offset):
                from .samplers import _mass_operands
                kind = _mass_operands(inv_mass, theta0)[0]
                if kind == _abi.MASS_FULL or theta0.shape[1] != _n_params(self.dims):
                    self._fb = self.fallback()          # full mass matrix / unexpected layout: generic-callback path
                    return self._fb.begin(theta0, N, burn, inv_mass, seed, chain_offset)
                self._begin_args = (theta0, N, burn, inv_mass, seed, chain_offset)
                super().begin(theta0, N, burn, inv_mass, seed, chain_offset)
                if self.kind == _abi.MASS_DIAG and bool((self.im == 1).all()):
                    # the notebooks pass inv_mass = ones(D): multiplying by 1.0 is exact, so the identity-mass kernels give the
                    # same bits without loading a mass vector at every drift
                    self.kind, self.im, self.mf = _abi.MASS_NONE, None, None
                self.X = torch.cat([s["X"].reshape(self.Nb, self.n_in) for s in self.specs]).to(theta0).contiguous()
                self.Y = torch.cat([s["Y"].reshape(self.Nb, -1) for s in self.specs]).to(theta0).contiguous()

            def advance(self, n0, count, L, eps, H_old=None, H_new=None, progress=None):
                if self._fb is not None:
                    return self._fb.advance(n0, count, L, eps, H_old, H_new, progress)
                step = 1 if H_old is not None else count
                for start in range(n0, n0 + count, step):
                    try:
                        if self.kernel == "netn":
                            _abi.netn_hmc_sample(self.cur, self.theta0, self.dims, self.act, self.X, self.Y, self.M, self.Nb,
                                                 self.tau, self.tau_out, self.prior_scale, self.kind, self.im, self.mf, L, eps,
                                                 min(step, n0 + count - start), start, self.burn, self.seed, self.off,
                                                 self.samples, self.rejected, H_old, H_new, integrator=self.integrator,
                                                 loss=self.loss)
                            continue
                        _abi.mlp_hmc_sample(self.cur, self.theta0, self.n_in, self.H, self.act, self.X, self.Y, self.M,
                                            self.Nb, self.tau, self.tau_out, self.prior_scale, self.kind, self.im, self.mf,
                                            L, eps, min(step, n0 + count - start), start, self.burn, self.seed, self.off,
                                            self.samples, self.rejected, H_old, H_new, integrator=self.integrator, loss=self.loss)
                    except _abi.InvalidArguments:
                        # the kernels stage the whole data set in LDS (csrc/mlp_hmc.hip: "do not fit the LDS staging"); the
                        # reference works for any N.  Arguments are validated before anything is launched, so nothing ran:
                        # a refusal on the very first launch moves the run to the generic-callback path.
                        if start != 0 or n0 != 0:
                            raise
                        self._fb = self.fallback()
                        self._fb.begin(*self._begin_args)
                        return self._fb.advance(n0, count, L, eps, H_old, H_new, progress)
                if progress is not None:
                    progress.update(min(self.N, n0 + count) - 1)

            def finish(self):
                return self._fb.finish() if self._fb is not None else super().finish()

        return Impl(specs, fallback, integrator)


def split_engine(log_prob_list, theta0, integrator=None):
    """integrator: an Integrator split kind (default SPLITTING)."""
    from .enums import Integrator
    from .samplers import _GenericHMC
    integrator = Integrator.SPLITTING if integrator is None else integrator
    kind = {Integrator.SPLITTING: _abi.SPLIT_SYMMETRIC, Integrator.SPLITTING_RAND: _abi.SPLIT_RAND,
            Integrator.SPLITTING_KMID: _abi.SPLIT_KMID}[integrator]
    specs = _common_spec(log_prob_list)
    if specs is None or (len(specs) < 2 and kind != _abi.SPLIT_RAND) or len(specs) > 64:
        return None
    return _MLPEngine(specs, lambda: _GenericHMC(log_prob_list, split=True, integrator=integrator), kind)


def hmc_engine(log_prob_func, theta0):
    specs = _common_spec([log_prob_func])
    if specs is None:
        return None
    from .samplers import _GenericHMC
    return _MLPEngine(specs, lambda: _GenericHMC(log_prob_func))

"""Host mirror of ``hamiltorch/samplers.py`` (reference lines cited as S:n) for the sampling path.

Same names, keyword arguments, return shapes and error behaviour as the reference; the
arithmetic runs in libhamiltorch_amd.so (HIP, gfx950) for many chains at once:

* ``params_init`` of shape (D,)  -> reference behaviour, one chain, list of (D,) tensors;
* ``params_init`` of shape (C,D) -> C independent chains batched on-device, list of (C,D) tensors.

Two execution paths behind ``sample``:
  native   -- the log-prob is a recognised model family (``models.GaussianTarget`` or an MLP
              built by ``sample_model``/``sample_split_model``): whole trajectories run inside
              one persistent kernel;
  generic  -- any other ``log_prob_func`` (the callback contract of S:272-274): torch evaluates
              the callback for all chains (``torch.func.vmap``), HIP kernels do the momentum
              draw, kick/drift, energy reduction and the Metropolis select in between.
There is no CPU path: tensors must live on an AMD GPU.
"""
from __future__ import annotations

import threading
import os
import warnings
import torch
import torch.nn as nn

from . import _abi, util
from .enums import Integrator, Metric, Sampler
from .models import GaussianTarget, as_gaussian, probe_gaussian, verify_gaussian, MAX_NATIVE_DIM
from .samplelist import rows_of
from .host import host_inputs

_sample_lock = threading.RLock()  # multi_chain(parallel=True) calls sample() from threads (U:396-398)


_SPLIT_KINDS = (Integrator.SPLITTING, Integrator.SPLITTING_RAND, Integrator.SPLITTING_KMID)


# =================================================================================================
# helpers
# =================================================================================================
def _as_batch(params, what="params"):
    """(C,D) contiguous copy of a (D,) or (C,D) tensor + whether it was 1-D."""
    if params.dim() == 1:
        return params.detach().clone().reshape(1, -1).contiguous(), True
    if params.dim() == 2:
        return params.detach().clone().contiguous(), False
    raise RuntimeError("%s must be a 1d tensor (one chain) or a 2d [chains, D] tensor." % what)


def _mass_operands(inv_mass, like):
    """inv_mass None | (D,) | (D,D) | list of blocks -> (kind, inv_mass, mass_factor) on the device.

    Inverts the mass matrix once, as S:942-952, and factors it for the momentum draw
    (S:199-201): diag -> sqrt(1/inv_mass); full -> chol(inverse(inv_mass))."""
    if inv_mass is None:
        return _abi.MASS_NONE, None, None
    if isinstance(inv_mass, list):  # block-diagonal list (S:287-292, S:803-809, S:944-947)
        inv_mass = torch.block_diag(*inv_mass)
    im = inv_mass.detach().to(device=like.device, dtype=like.dtype).contiguous()
    D = like.shape[-1]
    if im.dim() == 1:
        if im.numel() != D:
            raise RuntimeError("inv_mass has %d entries, params have %d" % (im.numel(), D))
        return _abi.MASS_DIAG, im, torch.sqrt(1.0 / im).contiguous()
    if im.dim() == 2:
        if tuple(im.shape) != (D, D):
            raise RuntimeError("inv_mass must be (D,), (D,D) or a list of blocks")
        mass = torch.inverse(im)
        return _abi.MASS_FULL, im, torch.linalg.cholesky(mass).contiguous()
    raise RuntimeError("inv_mass must be (D,), (D,D) or a list of blocks")


def _num_rows(num_samples, burn):
    """len(ret_params): params_init plus one row per n in range(num_samples) with n > burn."""
    return 1 + num_samples - max(0, min(num_samples, burn + 1))


def _scalar(v):
    return v.sum() if torch.is_tensor(v) else v


def _own(t):
    """A contiguous private copy (graph outputs are static buffers, vmap outputs may be strided)."""
    return torch.empty(t.shape, dtype=t.dtype, device=t.device).copy_(t)


def _not_batchable(e):
    """True for the errors torch.func raises on a callback it cannot batch (``.item()``, data-dependent control
    flow, the tuple protocol, in-place writes into captured tensors ...); device faults and out-of-memory are not
    that: they propagate."""
    if isinstance(e, (torch.OutOfMemoryError, torch.AcceleratorError)):
        return False
    return isinstance(e, (RuntimeError, TypeError, ValueError, NotImplementedError, AttributeError, IndexError))


class _BatchedCallback:
    """Evaluates a reference-style ``log_prob_func`` (one (D,) vector in, scalar out) for all
    chains.  ``torch.func.vmap`` when the callback allows it, otherwise a per-chain loop."""

    def __init__(self, fn, pass_grad=None):
        self.fn = fn
        self.pass_grad = pass_grad
        self._use_vmap = True
        def f(w):
            r = fn(w)
            if isinstance(r, tuple):      # (log_prob, params) protocol (S:54-58): its gradients come from .backward()
                raise TypeError("tuple protocol (log_prob, params)")
            return _scalar(r)
        # replayed as HIP graphs on the device (util.GraphedCallable): a trajectory calls these hundreds of times
        self._v_logp = util.GraphedCallable(torch.func.vmap(f))
        self._v_gv = util.GraphedCallable(torch.func.vmap(torch.func.grad_and_value(f)))
        # a callback that brings its own batched closed form (GaussianTarget beyond the fused kernels' size: one GEMM per gradient)
        if pass_grad is None and callable(getattr(fn, "_hta_batched_grad", None)):
            self._v_logp = util.GraphedCallable(fn._hta_batched_logp)
            self._v_gv = util.GraphedCallable(fn._hta_batched_grad)
            self.closed_form = True
        self._v_pg = util.GraphedCallable(torch.func.vmap(pass_grad)) if callable(pass_grad) else None

    def _loop(self, theta, want_grad):
        lps, gs = [], []
        for c in range(theta.shape[0]):
            p = theta[c].detach().requires_grad_(want_grad)
            lp = self.fn(p)
            if want_grad:
                p = collect_gradients(lp, p, self.pass_grad)
                gs.append(p.grad.detach())
            lp = lp[0] if isinstance(lp, tuple) else lp
            lps.append(_scalar(lp).detach())
        return (torch.stack(gs) if want_grad else None), torch.stack(lps)

    def logp(self, theta):
        if self._use_vmap:
            try:
                with torch.no_grad():
                    return _own(self._v_logp(theta))
            except Exception as e:  # data-dependent control flow, .item(), tuple protocol, ...
                if not _not_batchable(e):
                    raise
                self._fallback(e)
        return self._loop(theta, False)[1].contiguous()

    def grad(self, theta, own=True):
        """(grad[C,D], logp[C]) at theta.  own=False: the results may be the replayed graph's own output buffers - valid until this
        callback is evaluated again (a caller that consumes them at once saves two copy launches per evaluation: 5 % of a callback
        trajectory's ~39 launches per leapfrog step, profiles/r05ac_callback_launches.txt)."""
        if self.pass_grad is not None and not callable(self.pass_grad):
            g = self.pass_grad.to(theta).expand_as(theta).contiguous()
            return g, self.logp(theta)
        if self._use_vmap:
            try:
                if self._v_pg is not None:
                    return _own(self._v_pg(theta)), self.logp(theta)
                g, v = self._v_gv(theta)
                return (_own(g), _own(v)) if own else (g.contiguous(), v.contiguous())      # (contiguous(): no launch unless strided)
            except Exception as e:
                if not _not_batchable(e):
                    raise
                self._fallback(e)
        g, v = self._loop(theta, True)
        return g.contiguous(), v.contiguous()

    def capturable(self):
        """True while every piece of this callback replays as a HIP graph (so a whole trajectory can be captured)."""
        return self._use_vmap and all(g is None or g.capturable() for g in (self._v_logp, self._v_gv, self._v_pg))

    def _fallback(self, e):
        self._use_vmap = False
        warnings.warn("hamiltorch_amd: log_prob_func is not vmap-able (%s: %s); evaluating it chain by chain"
                      % (type(e).__name__, str(e).split("\n")[0][:120]))


# =================================================================================================
# reference API: small pieces
# =================================================================================================
def collect_gradients(log_prob, params, pass_grad=None):
    """S:33-66: attach ``.grad`` to params via the tuple protocol, ``pass_grad`` or autograd."""
    if isinstance(log_prob, tuple):
        log_prob[0].backward()
        plist = list(log_prob[1])
        params = torch.cat([p.flatten() for p in plist])
        params.grad = torch.cat([p.grad.flatten() for p in plist])
    elif pass_grad is not None:
        params.grad = pass_grad(params) if callable(pass_grad) else pass_grad
    else:
        params.grad = torch.autograd.grad(log_prob, params)[0]
    return params


def acceptance(h_old, h_new):
    """S:609-626."""
    return float(-h_new + h_old)


@host_inputs
def gibbs(params, sampler=Sampler.HMC, log_prob_func=None, jitter=None, normalizing_const=1., softabs_const=None,
          mass=None, metric=Metric.HESSIAN, seed=None, chain_offset=0, draw=0):
    """Momentum resampling (S:152-202) on-device.  ``mass`` is None | (D,) | (D,D) | list of blocks.
    Draws come from the Philox stream (seed, chain, draw) instead of torch's global generator."""
    theta, one = _as_batch(params)
    _abi.require_device(theta, "params")
    seed = util.next_stream_seed() if seed is None else int(seed)
    if sampler == Sampler.RMHMC:
        from . import rmhmc
        p = rmhmc.gibbs(theta, log_prob_func, jitter, softabs_const, metric, seed, chain_offset, draw)
        return p[0] if one else p
    if isinstance(mass, list):
        mass = torch.block_diag(*mass)
    if mass is None:
        kind, mf = _abi.MASS_NONE, None
    elif mass.dim() == 1:
        kind, mf = _abi.MASS_DIAG, torch.sqrt(mass.to(theta)).contiguous()   # S:201
    else:
        kind, mf = _abi.MASS_FULL, torch.linalg.cholesky(mass.to(theta)).contiguous()   # S:199
    p = torch.empty_like(theta)
    _abi.momentum_resample(p, kind, mf, seed, chain_offset, draw)
    return p[0] if one else p


@host_inputs
def hamiltonian(params, momentum, log_prob_func, jitter=0.01, normalizing_const=1., softabs_const=1e6,
                explicit_binding_const=100, inv_mass=None, ham_func=None, sampler=Sampler.HMC,
                integrator=Integrator.EXPLICIT, metric=Metric.HESSIAN):
    """S:738-846.  HMC: -log p + 0.5 p^T M^-1 p (list of callables = split sum under no_grad).
    Returns a 0-d tensor for one chain, (C,) for a batch."""
    if sampler == Sampler.HMC:
        theta, one = _as_batch(params)
        p, _ = _as_batch(momentum, "momentum")
        _abi.require_device(theta, "params")
        fns = log_prob_func if isinstance(log_prob_func, list) else [log_prob_func]
        logp = None
        for f in fns:
            v = _BatchedCallback(f).logp(theta)
            logp = v if logp is None else logp + v
        kind, im, _ = _mass_operands(inv_mass, theta)
        H = torch.empty(theta.shape[0], dtype=theta.dtype, device=theta.device)
        _abi.hamiltonian(p, logp.to(theta.dtype).contiguous(), kind, im, H)
        if one and util.has_nan_or_inf(logp):
            raise util.LogProbError()   # S:783-785
        return H[0] if one else H
    if sampler == Sampler.RMHMC and integrator in (Integrator.IMPLICIT, Integrator.EXPLICIT):
        h = rm_hamiltonian(params, momentum, log_prob_func, jitter, normalizing_const, softabs_const=softabs_const,
                           sampler=sampler, integrator=integrator, metric=metric)
        if integrator == Integrator.EXPLICIT and not isinstance(params, list):
            return 2 * h    # S:822 (first call of the explicit sampler works with the doubled energy)
        return h
    raise NotImplementedError()


@host_inputs
def rm_hamiltonian(params, momentum, log_prob_func, jitter, normalizing_const, softabs_const=1e6,
                   sampler=Sampler.HMC, integrator=Integrator.EXPLICIT, metric=Metric.HESSIAN):
    """S:677-736: -log p + D/2 log 2pi + 1/2 log|G| + 1/2 p^T G^-1 p, shape (1,1) for one chain."""
    from . import rmhmc
    return rmhmc.rm_hamiltonian(params, momentum, log_prob_func, jitter, softabs_const, metric)


@host_inputs
def fisher(params, log_prob_func=None, jitter=None, normalizing_const=1., softabs_const=1e6, metric=Metric.HESSIAN):
    """S:69-127: metric G(theta) and (for SOFTABS) the soft-absolute eigenvalues."""
    from . import rmhmc
    return rmhmc.fisher(params, log_prob_func, jitter, softabs_const, metric)


@host_inputs
def cholesky_inverse(fish, momentum):
    """S:130-149: G^-1 p by Cholesky + two triangular solves; returns (D,1) for one system."""
    from . import rmhmc
    return rmhmc.cholesky_inverse(fish, momentum)


# =================================================================================================
# leapfrog (S:205-606)
# =================================================================================================
@host_inputs
def leapfrog(params, momentum, log_prob_func, steps=10, step_size=0.1, jitter=0.01, normalizing_const=1.,
             softabs_const=1e6, explicit_binding_const=100, fixed_point_threshold=1e-20,
             fixed_point_max_iterations=6, jitter_max_tries=10, inv_mass=None, ham_func=None, sampler=Sampler.HMC,
             integrator=Integrator.IMPLICIT, metric=Metric.HESSIAN, store_on_GPU=True, debug=False, pass_grad=None):
    """Same contract as the reference: returns (ret_params, ret_momenta), lists with one entry
    per step (explicit RMHMC: ``[list, params_copy], [list, momentum_copy]``, S:462)."""
    theta, one = _as_batch(params)
    p, _ = _as_batch(momentum, "momentum")
    _abi.require_device(theta, "params")
    unb = (lambda t: t[0]) if one else (lambda t: t)

    if sampler == Sampler.HMC and integrator not in _SPLIT_KINDS:
        kind, im, _ = _mass_operands(inv_mass, theta)
        tgt = as_gaussian(log_prob_func, theta) if pass_grad is None else None
        if tgt is not None:
            pt = torch.empty((steps,) + theta.shape, dtype=theta.dtype, device=theta.device)
            pp = torch.empty_like(pt)
            _abi.hmc_gaussian_leapfrog(theta, p, tgt.precision, tgt.mean, kind, im, steps, step_size, pt, pp)
            return [unb(t) for t in pt.unbind(0)], [unb(t) for t in pp.unbind(0)]
        cb = _BatchedCallback(log_prob_func, pass_grad)
        ret_t, ret_p = [], []
        g, _ = cb.grad(theta)
        _abi.kick_drift(theta, p, g, 0.5 * step_size, step_size if steps > 0 else 0.0, kind, im)   # S:281 + first S:284
        for n in range(steps):
            g, _ = cb.grad(theta)                                                                  # S:297
            ret_t.append(unb(theta.clone()))
            last = n == steps - 1
            _abi.kick_drift(theta, p, g, step_size, 0.0, kind, im)                                 # S:298
            ret_p.append(unb(p.clone()))
            if last:
                _abi.kick_drift(theta, p, g, -0.5 * step_size, 0.0, kind, im)                      # S:302
                ret_p[-1] = unb(p.clone())
            else:
                _abi.kick_drift(theta, p, None, 0.0, step_size, kind, im)                          # next S:284
        return ret_t, ret_p

    if sampler == Sampler.HMC and integrator in _SPLIT_KINDS:
        if type(log_prob_func) is not list:
            raise RuntimeError('For splitting log_prob_func must be list of functions')      # S:466-467
        if pass_grad is not None:
            raise RuntimeError('Passing user-determined gradients not implemented for splitting')  # S:468-469
        cbs = [_BatchedCallback(f) for f in log_prob_func]
        kind, im, _ = _mass_operands(inv_mass, theta)
        # S:549: one subset order per leapfrog call
        perm = util.split_permutation(util.next_stream_seed(), 0, len(cbs)) if integrator == Integrator.SPLITTING_RAND else None
        ret_t, ret_p = [], []
        carry = None
        for _ in range(steps):
            carry = _split_step(theta, p, cbs, step_size, kind, im, integrator, perm, carry)
            ret_t.append(unb(theta.clone())); ret_p.append(unb(p.clone()))
        return ret_t, ret_p

    if sampler == Sampler.RMHMC and integrator == Integrator.EXPLICIT:
        if pass_grad is not None:
            raise RuntimeError('Passing user-determined gradients not implemented for RMHMC')  # S:390-391
        from . import rmhmc
        return rmhmc.explicit_leapfrog(params, momentum, log_prob_func, steps, step_size, jitter, softabs_const,
                                       explicit_binding_const, metric)
    if sampler == Sampler.RMHMC and integrator == Integrator.IMPLICIT:
        if pass_grad is not None:
            raise RuntimeError('Passing user-determined gradients not implemented for RMHMC')   # S:309-310
        from . import rmhmc
        return rmhmc.implicit_leapfrog(params, momentum, log_prob_func, steps, step_size, jitter, softabs_const, metric,
                                       fixed_point_threshold, fixed_point_max_iterations)
    raise NotImplementedError("Integrator.S3 (semi-separable Hamiltonians through a user ham_func) is outside the accelerated path")


def _split_step(theta, p, cbs, eps, kind, im, integrator=Integrator.SPLITTING, perm=None, carry=None):
    """One step of a split integrator on (theta, p), in place.
    SPLITTING (S:499-540): 2M half-kicks m = 0..M-1, M-1..0 with 2(M-1) drifts of eps / (2(M-1));
    SPLITTING_RAND (S:547-566): for each subset in the order `perm`: half kick, drift eps / M, half kick;
    SPLITTING_KMID (S:572-596): M half kicks, one drift of eps, M half kicks in reverse order.
    A kick without a drift is followed by a kick of the SAME subset at the same parameters (the turning point of the
    symmetric scheme; the step boundary, m = 0): the reference differentiates twice, here the gradient is evaluated once
    and applied twice - the same numbers, (2M - 2) L + 1 instead of 2 M L callback gradients per trajectory.  `carry`:
    the gradient the previous step of the same trajectory ended with (returned by this function), or None."""
    M = len(cbs)
    if integrator == Integrator.SPLITTING_RAND:
        for m in range(M):
            cb = cbs[perm[m]]
            g, _ = cb.grad(theta)
            _abi.kick_drift(theta, p, g, 0.5 * eps, eps / M, kind, im)
            g, _ = cb.grad(theta)
            _abi.kick_drift(theta, p, g, 0.5 * eps, 0.0, kind, im)
        return None
    if M == 1:
        raise RuntimeError('For symmetric splitting log_prob_func must be list of functions greater than length 1')
    if integrator == Integrator.SPLITTING_KMID:
        for m in range(M):
            g = carry if (m == 0 and carry is not None) else cbs[m].grad(theta)[0]
            _abi.kick_drift(theta, p, g, 0.5 * eps, eps if m == M - 1 else 0.0, kind, im)
        for m in reversed(range(M)):
            g, _ = cbs[m].grad(theta)
            _abi.kick_drift(theta, p, g, 0.5 * eps, 0.0, kind, im)
        return g
    dq = eps / ((M - 1) * 2)
    for m in range(M):
        g = carry if (m == 0 and carry is not None) else cbs[m].grad(theta)[0]
        _abi.kick_drift(theta, p, g, 0.5 * eps, dq if m < M - 1 else 0.0, kind, im)
    for m in reversed(range(M)):
        if m < M - 1:                                  # m = M - 1: the forward sweep's last gradient, at the same parameters
            g, _ = cbs[m].grad(theta)
        _abi.kick_drift(theta, p, g, 0.5 * eps, dq if m > 0 else 0.0, kind, im)
    return g


# =================================================================================================
# sample (S:850-1091)
# =================================================================================================
@host_inputs
def sample(log_prob_func, params_init, num_samples=10, num_steps_per_sample=10, step_size=0.1, burn=0, jitter=None,
           inv_mass=None, normalizing_const=1., softabs_const=None, explicit_binding_const=100,
           fixed_point_threshold=1e-5, fixed_point_max_iterations=1000, jitter_max_tries=10, sampler=Sampler.HMC,
           integrator=Integrator.IMPLICIT, metric=Metric.HESSIAN, debug=False, desired_accept_rate=0.8,
           store_on_GPU=True, pass_grad=None, verbose=True, *, seed=None, chain_offset=0, native=True):
    """Drop-in for ``hamiltorch.sample``.  Extensions (keyword-only): ``seed`` (Philox key; default
    derives from ``set_random_seed``), ``chain_offset`` (global id of the first chain, for sharding
    chains across GPUs), ``native=False`` forces the generic-callback path.

    Returns the reference's list (length ``num_samples - burn``, element 0 = ``params_init``) of
    (D,) tensors, or of (C,D) tensors when ``params_init`` is (C,D); with ``debug == 2`` also the
    acceptance rate (float, or a (C,) tensor for a batch)."""
    if params_init.dim() not in (1, 2):
        raise RuntimeError('params_init must be a 1d tensor.')                 # S:925-926 (2-D = batch of chains)
    if burn >= num_samples:
        raise RuntimeError('burn must be less than num_samples.')               # S:928-929
    nuts = False
    if sampler == Sampler.HMC_NUTS:
        if burn == 0:
            raise RuntimeError('burn must be greater than 0 for NUTS.')         # S:933-934
        sampler, nuts = Sampler.HMC, True                                       # S:935-936
    _abi.require_device(params_init, "params_init")
    _abi.load()
    util._poll_status()             # a failure an EARLIER run's kernels reported (sticky status word, copied asynchronously): raise now
    theta0, one = _as_batch(params_init, "params_init")
    seed = util.next_stream_seed() if seed is None else int(seed)
    burn_k = max(int(burn), -1)

    probed = None
    with _sample_lock:
        if sampler == Sampler.HMC:
            if integrator in _SPLIT_KINDS:
                if type(log_prob_func) is not list:
                    raise RuntimeError('For splitting log_prob_func must be list of functions')
                if pass_grad is not None:
                    raise RuntimeError('Passing user-determined gradients not implemented for splitting')
                if integrator == Integrator.SPLITTING_KMID and len(log_prob_func) == 1:
                    raise RuntimeError('For symmetric splitting log_prob_func must be list of functions greater than length 1')
                if isinstance(inv_mass, list):
                    raise NotImplementedError("block-list inv_mass performs no drift in the reference's split "
                                              "integrator (S:514-515); not supported")
                eng = _resolve_split_engine(log_prob_func, theta0, native, integrator)
            else:
                tgt = as_gaussian(log_prob_func, theta0) if (native and pass_grad is None) else None
                if tgt is None and native and pass_grad is None:
                    probed = tgt = _probe(log_prob_func, theta0)
                if tgt is not None and tgt.dim > MAX_NATIVE_DIM:
                    tgt = probed = None                                         # beyond the fused kernels: generic-callback path
                eng = _GaussianHMC(tgt) if tgt is not None else None
                if eng is None and native and pass_grad is None:
                    from . import bnn
                    eng = bnn.native_hmc_engine(log_prob_func, theta0)
                if eng is None and native and pass_grad is None:
                    eng = _compiled_engine(log_prob_func, theta0, inv_mass)    # the callback compiler (jit/): opaque callable, fused kernel
                elif eng is None:
                    _abi.load().hta_jit_note_fallback(b"native=False" if not native else b"pass_grad supplies the gradient")
                if eng is None:
                    eng = _GenericHMC(log_prob_func, pass_grad)
            label = '({}; {})'.format(sampler, integrator)
            def run(e):
                if nuts:
                    out = e.run_nuts(theta0, num_samples, num_steps_per_sample, step_size, burn_k, inv_mass, seed,
                                     chain_offset, verbose, label, desired_accept_rate)
                    return out, e.final_step_size
                return e.run(theta0, num_samples, num_steps_per_sample, step_size, burn_k, inv_mass, seed, chain_offset,
                             verbose, label), step_size
            (samples, rejected), step_size_out = run(eng)
            if probed is not None and not verify_gaussian(probed, log_prob_func, samples):
                _probe_mismatch(log_prob_func)
                (samples, rejected), step_size_out = run(_GenericHMC(log_prob_func, pass_grad))
            if isinstance(eng, _CompiledHMC) and not eng.verify():
                # the compiled code disagrees with the callable on the states the run ended in: a reused trace whose captured
                # state changed in place (trace again, once), or a callable that is not a pure function of its argument
                eng2 = _compiled_engine(log_prob_func, theta0, inv_mass, fresh=True) if eng.reused else None
                if eng2 is not None:
                    (samples, rejected), step_size_out = run(eng2)
                if eng2 is None or not eng2.verify():
                    warnings.warn("hamiltorch_amd: the compiled form of %r disagrees with the callable itself on the sampled "
                                  "states; re-running on the torch-evaluated callback path" % (log_prob_func,))
                    _abi.load().hta_jit_note_fallback(b"compiled code disagrees with the callable")
                    (samples, rejected), step_size_out = run(_GenericHMC(log_prob_func, pass_grad))
            step_size = step_size_out
        elif sampler == Sampler.RMHMC and integrator == Integrator.EXPLICIT:
            if pass_grad is not None:
                raise RuntimeError('Passing user-determined gradients not implemented for RMHMC')
            from . import rmhmc
            lp = log_prob_func
            if native and as_gaussian(log_prob_func, theta0) is None:
                probed = _probe(log_prob_func, theta0)
                if probed is not None and probed.dim <= 128:                       # the constant-curvature RMHMC kernels' range
                    lp = probed
                else:
                    probed = None
            samples, rejected = rmhmc.sample_explicit(lp, theta0, num_samples, num_steps_per_sample,
                                                      step_size, burn_k, jitter, softabs_const,
                                                      explicit_binding_const, metric, seed, chain_offset, verbose)
            if probed is not None and not verify_gaussian(probed, log_prob_func, samples):
                _probe_mismatch(log_prob_func)
                samples, rejected = rmhmc.sample_explicit(log_prob_func, theta0, num_samples, num_steps_per_sample,
                                                          step_size, burn_k, jitter, softabs_const,
                                                          explicit_binding_const, metric, seed, chain_offset, verbose)
        elif sampler == Sampler.RMHMC and integrator == Integrator.IMPLICIT:
            if pass_grad is not None:
                raise RuntimeError('Passing user-determined gradients not implemented for RMHMC')
            from . import rmhmc
            samples, rejected = rmhmc.sample_implicit(log_prob_func, theta0, num_samples, num_steps_per_sample, step_size,
                                                      burn_k, jitter, softabs_const, metric, fixed_point_threshold,
                                                      fixed_point_max_iterations, seed, chain_offset, verbose)
        elif sampler == Sampler.RMHMC:
            raise NotImplementedError("Integrator.S3 (semi-separable Hamiltonians through a user ham_func) is outside "
                                      "the accelerated path")
        else:
            raise NotImplementedError()

    if not store_on_GPU:
        samples = samples.cpu()                                                 # S:1012 / S:1024
    # the reference's list of rows; long device-resident runs come back as a lazily materialised list (samplelist.py: the
    # 1001 view objects of a BASELINE-config-2 call cost more host time than its kernels take)
    rows = rows_of(samples, one)
    _abi.free_scratch(256 << 20)                # (the ABI's caller-side scratch: buffers beyond 256 MiB do not outlive the call)
    if not (verbose or debug == 2):
        util._poll_status()                     # a status copy that has ALREADY completed is looked at here (no wait): a run that failed
                                                # early is reported by the call that made it, not by the next one
    if verbose or debug == 2:
        acc = 1.0 - rejected.to(torch.float64) / float(num_samples)            # S:1085 / S:1089 (burn-in included)
        if samples.is_cuda:
            torch.cuda.current_stream(samples.device).synchronize()              # (the acceptance rate is read below anyway)
        util._poll_status()                                                      # a launch that reported a failure raises HERE, not later
    if verbose:
        print('Acceptance Rate {:.2f}'.format(float(acc.mean())))
    if nuts and debug == 2:
        return rows, step_size                                                  # S:1086-1087
    if debug == 2:
        return rows, (float(acc[0]) if one else acc)
    return rows


def _probe(log_prob_func, theta0):
    """The reference's own example closures (``MultivariateNormal(...).log_prob(w).sum()``, tests/test_util.py:98-101) are
    quadratic forms behind an opaque callable: recognise them by their curvature and run the fused kernels.
    ``HAMILTORCH_AMD_PROBE=0`` (or ``native=False``) keeps every unrecognised callable on the generic path."""
    if os.environ.get("HAMILTORCH_AMD_PROBE", "1") == "0" or hasattr(log_prob_func, "_hta_spec"):
        return None
    # a callable the probe has already turned down is not probed again while its closure signature is unchanged (the probe is
    # ~10 torch.func evaluations and a synchronise: 1.1 ms per sample() call, four times a compiled 50-trajectory launch)
    from .jit import _signature
    key = None
    try:
        key = (_signature(log_prob_func)[0], tuple(theta0.shape[1:]), theta0.dtype)
        if _probe_said_no.get(log_prob_func) == key:
            return None
    except TypeError:
        key = None
    tgt = probe_gaussian(log_prob_func, theta0)
    if tgt is None and key is not None:
        try:
            _probe_said_no[log_prob_func] = key
        except TypeError:
            pass
    return tgt


import weakref  # noqa: E402
_probe_said_no = weakref.WeakKeyDictionary()


def _probe_mismatch(log_prob_func):
    warnings.warn("hamiltorch_amd: %r looked like a Gaussian at the probe points but disagrees with that closed form on "
                  "the sampled states; re-running on the generic-callback path" % (log_prob_func,))


#: set by dist.sample_sharded for the duration of ITS call, in ITS thread / context only (a ContextVar, not a module
#: global: concurrent sample() calls of multi_chain(parallel=True) never see another call's reducer):
#: (sum, count, bad) -> the same over the process group, so that a sharded HMC_NUTS run adapts ONE step size on all
#: chains of all ranks, as the single-process run does
import contextvars  # noqa: E402
_nuts_reduce = contextvars.ContextVar("hamiltorch_amd_nuts_reduce", default=None)


class _Engine:
    """Common shape of the HMC engines: ``begin`` allocates state, ``advance`` runs trajectories
    [n0, n0 + count) with one step size, ``finish`` hands back (samples[S,C,D], rejected[C]).
    ``run`` is the plain fixed-step-size sample(); ``run_nuts`` adds the burn-in dual averaging."""

    def begin(self, theta0, N, burn, inv_mass, seed, chain_offset):
        C, D = theta0.shape
        self.theta0, self.N, self.burn, self.seed, self.off = theta0, N, burn, seed, chain_offset
        self.kind, self.im, self.mf = _mass_operands(inv_mass, theta0)
        self.samples = torch.empty((_num_rows(N, burn), C, D), dtype=theta0.dtype, device=theta0.device)
        self.cur = torch.empty_like(theta0)
        self.rejected = torch.empty(C, dtype=torch.int32, device=theta0.device)
        _abi.run_begin(theta0, self.cur, self.samples[0], self.rejected)      # S:954-961 as one launch instead of three

    def finish(self):
        return self.samples, self.rejected

    def run(self, theta0, N, L, eps, burn, inv_mass, seed, chain_offset, verbose, label):
        self.begin(theta0, N, burn, inv_mass, seed, chain_offset)
        prog = util._Progress('Sampling ' + label, N, verbose)
        self.advance(0, N, L, eps, progress=prog)
        prog.end()
        return self.finish()

    def run_nuts(self, theta0, N, L, eps0, burn, inv_mass, seed, chain_offset, verbose, label, desired):
        """Sampler.HMC_NUTS (S:931-939, S:1030-1035): dual-averaging step size while n < burn, frozen to
        eps_bar at n == burn.  One chain: the reference's schedule exactly.  A batch shares ONE step size,
        adapted on the mean acceptance statistic over chains (extension; variance-reduced); under
        dist.sample_sharded the mean runs over the chains of every rank (`_nuts_reduce`), so the result does not
        depend on the sharding."""
        self.begin(theta0, N, burn, inv_mass, seed, chain_offset)
        C = theta0.shape[0]
        Ho = torch.empty(C, dtype=theta0.dtype, device=theta0.device)
        Hn = torch.empty_like(Ho)
        prog = util._Progress('Sampling ' + label, N, verbose)
        eps, H_t, eps_bar = float(eps0), 0., 1.
        for n in range(min(burn + 1, N)):
            self.advance(n, 1, L, eps, H_old=Ho, H_new=Hn)
            rho = torch.clamp(Ho - Hn, max=0.0)                               # S:1000
            alpha = torch.where(torch.isfinite(rho), torch.exp(rho.float()), torch.zeros_like(rho.float()))
            bad = bool((~torch.isfinite(rho)).any())
            a_sum, a_cnt = float(alpha.double().sum()), float(alpha.numel())
            reduce_ = _nuts_reduce.get()
            if reduce_ is not None:                                           # sharded run: the statistic of ALL chains
                a_sum, a_cnt, bad = reduce_(a_sum, a_cnt, bad)
            if n < burn or bad:                                               # S:1031-1032 / S:1060-1064
                eps, eps_bar, H_t = _dual_average(a_sum / a_cnt, n, eps0, H_t, eps_bar, desired)
            if n == burn:
                eps = eps_bar                                                 # S:1033-1035
                print('Final Adapted Step Size: ', eps)
            prog.update(n)
        if N > burn + 1:
            self.advance(burn + 1, N - burn - 1, L, eps, progress=prog)
        prog.end()
        self.final_step_size = eps
        return self.finish()


def _dual_average(alpha, t, step_size_init, H_t, eps_bar, desired_accept_rate=0.8):
    """Hoffman & Gelman (2014) Algorithm 5 as the reference applies it (S:659-672); float32 log/exp like the
    reference's torch.FloatTensor arithmetic.  `alpha` = min(1, exp(rho)) (0 for a divergent trajectory)."""
    import numpy as np
    f32 = np.float32
    t = t + 1
    mu = float(np.log(f32(10) * f32(step_size_init)))
    gamma, t0, kappa = 0.05, 10, 0.75
    H_t = (1 - (1 / (t + t0))) * H_t + (1 / (t + t0)) * (desired_accept_rate - alpha)
    x_new = mu - (t ** 0.5) / gamma * H_t
    step_size = float(np.exp(f32(x_new)))
    x_new_bar = f32(t ** -kappa * x_new) + f32(1 - t ** -kappa) * np.log(f32(eps_bar))
    return step_size, float(np.exp(f32(x_new_bar))), H_t


def adaptation(rho, t, step_size_init, H_t, eps_bar, desired_accept_rate=0.8):
    """S:629-674: (step_size, eps_bar, H_t) from the log acceptance ratio `rho` of iteration t."""
    import numpy as np
    if rho != rho or rho in (float('inf'), float('-inf')):
        alpha = 0.                                                             # S:660-661
    else:
        alpha = min(1., float(np.exp(np.float32(rho))))                        # S:663
    return _dual_average(alpha, t, step_size_init, H_t, eps_bar, desired_accept_rate)


class _GaussianHMC(_Engine):
    """Native path: one persistent-kernel launch runs every trajectory (csrc/hmc_gaussian.hip)."""

    WS_CAP = 128 << 20

    def __init__(self, target: GaussianTarget):
        self.t = target

    def advance(self, n0, count, L, eps, H_old=None, H_new=None, progress=None):
        theta0 = self.theta0
        C, D = theta0.shape
        # scratch for the pre-drawn momenta / log-uniforms of one launch (<= WS_CAP bytes, so it stays
        # in the 256 MB Infinity Cache); long runs are cut into several launches over `traj_offset`
        fixed = _abi.gaussian_workspace_bytes(C, D, 0, theta0.element_size())       # look-ahead rows + eigen block / area
        per_traj = _abi.gaussian_workspace_bytes(C, D, 1, theta0.element_size()) - fixed
        fits = fixed + per_traj <= self.WS_CAP
        # (D > 6: the kernels draw inline, the workspace is only the eigen area and does not grow with the trajectories)
        chunk = max(1, min(count, (self.WS_CAP - fixed) // per_traj)) if (fits and per_traj > 0) else count
        ws = None
        prepared = False
        if fits and (H_old is None):
            if self.kind == _abi.MASS_NONE and per_traj > 0:
                # identity mass, eigenbasis route: the workspace lives on the target, its eig block PREPARED once
                ws = _prepared_hmc_workspace(self.t, theta0, chunk)
                prepared = True
            else:
                ws = getattr(self, "_ws", None)
                need = _abi.gaussian_workspace_bytes(C, D, chunk, theta0.element_size())
                if ws is None or ws.numel() < need:
                    ws = self._ws = torch.empty(need, dtype=torch.uint8, device=theta0.device)
        if H_old is not None:
            chunk = 1                      # diagnostics are [n_traj, C]: one trajectory per call (NUTS burn-in)
        # (only a PREPARED workspace has a status word: the preparation zeroes it; elsewhere those bytes are whatever the buffer held)
        watch = _status_watch(self.t, ws, C, D, chunk, theta0) if prepared else None
        for start in range(n0, n0 + count, chunk):
            _abi.hmc_gaussian_sample(self.cur, theta0, self.t.precision, self.t.mean, self.t.log_norm, self.kind,
                                     self.im, self.mf, L, eps, min(chunk, n0 + count - start), start, self.burn,
                                     self.seed, self.off, self.samples, self.rejected, H_old, H_new, workspace=ws)
            if progress is not None:
                progress.update(min(self.N, start + chunk) - 1)
        if watch is not None:
            watch.refresh()                # non-blocking copy of the sticky status word: looked at by the next entry / at the next synchronise
            self.status_watch = watch


def _status_watch(tgt, ws, C, D, chunk, theta0):
    """The status watch of a prepared workspace (util._StatusWatch): created once per workspace and kept ON THE WORKSPACE'S HANDLE in
    the target's cache entry, so that it is dropped with the workspace (the watch holds a view of it)."""
    for hit in tgt.__dict__.get("_hta_hmc_ws", {}).values():
        if hit[0].ws is ws:
            handle = hit[0]
            break
    else:
        return None
    w = handle.watches.get(int(chunk))
    if w is None:
        word = _abi.hmc_gaussian_status_word(ws, C, D, chunk, theta0.element_size())
        if word is None:
            return None
        d = weakref.ref(tgt)

        def forget():      # reported: the flagged workspace goes, the next run prepares a fresh one (word zeroed)
            t = d()
            if t is not None:
                t.__dict__.pop("_hta_hmc_ws", None)
        w = handle.watches[int(chunk)] = util._watch_status(word, "hta_hmc_gaussian_sample (%d chains, D = %d)" % (C, D), forget)
    return w


class _HmcWorkspaceHandle:
    """Owns a prepared Gaussian-HMC workspace: the library's (device, eig block) -> plan entry goes when the buffer does."""

    def __init__(self, ws):
        self.ws = ws
        self.watches = {}          # chunk -> util._StatusWatch of this workspace's sticky status word

    def __del__(self):
        try:
            _abi.hmc_gaussian_forget(self.ws)
        except Exception:       # interpreter shutdown
            pass


def _prepared_hmc_workspace(tgt, theta0, chunk):
    """The workspace of hta_hmc_gaussian_sample for launches of `chunk` trajectories on this target, its eig block PREPARED
    once (hta_hmc_gaussian_prepare: the diagonalisation of the precision matrix, a single-wave kernel in front of every
    launch otherwise - 3 % of a BASELINE-config-2 call) and kept on the target object: repeated sample() calls and the
    launches of a chunked run pay it once.  Keyed by everything the eig block's content and position depend on, by the stream
    the kernels run on and by the precision tensor's version counter (an in-place edit of the target prepares again)."""
    C, D = theta0.shape
    key = (theta0.device, theta0.dtype, C, D, int(chunk), torch.cuda.current_stream(theta0.device).cuda_stream)
    # (the entry holds the tensor OBJECT: while it is cached its storage cannot be freed and handed to another matrix)
    sig = (tgt.precision, tgt.precision.data_ptr(), tgt.precision._version)
    cache = tgt.__dict__.setdefault("_hta_hmc_ws", {})
    hit = cache.get(key)
    if hit is not None and hit[1][0] is sig[0] and hit[1][1:] == sig[1:]:
        return hit[0].ws
    ws = torch.empty(_abi.gaussian_workspace_bytes(C, D, chunk, theta0.element_size()), dtype=torch.uint8, device=theta0.device)
    _abi.hmc_gaussian_prepare(theta0, tgt.precision, _abi.MASS_NONE, None, C, D, chunk, ws)
    if len(cache) >= 2:                 # the buffers are up to WS_CAP bytes each: two shapes per target at most
        cache.clear()
    cache[key] = (_HmcWorkspaceHandle(ws), sig)
    return ws


def _mass_kind_of(inv_mass):
    if inv_mass is None:
        return _abi.MASS_NONE
    if isinstance(inv_mass, list):
        return _abi.MASS_FULL
    return _abi.MASS_DIAG if inv_mass.dim() == 1 else _abi.MASS_FULL


def _compiled_engine(log_prob_func, theta0, inv_mass, fresh=False):
    """The callback compiler's engine for this callable, or None (the reason goes to hta_last_route() / jit.last_reason())."""
    from . import jit
    if not jit.enabled() or not callable(log_prob_func):
        _abi.load().hta_jit_note_fallback(b"HAMILTORCH_AMD_JIT=0" if callable(log_prob_func) else b"not a callable")
        return None
    before = jit.stats["trace_hits"]
    try:
        comp = jit.compile_hmc(log_prob_func, theta0[0], theta0.dtype, _mass_kind_of(inv_mass), fresh=fresh)
    except jit.Unsupported as e:
        _abi.load().hta_jit_note_fallback(str(e)[:140].encode("utf-8", "replace"))
        return None
    return _CompiledHMC(log_prob_func, comp, reused=jit.stats["trace_hits"] > before)


class _CompiledHMC(_Engine):
    """An opaque callable compiled into the trajectory kernel (hamiltorch_amd/jit/, csrc/jit/hmc_callback.hip.in): a block of
    trajectories per launch - momentum draw, leapfrog with the callable's value + gradient inlined, energies, Metropolis, burn /
    Q2 bookkeeping and row stores in one kernel, one chain per lane."""

    PREDRAW_MAX_CHAINS = 131072     # (tools/jit_sweep.py: with the draws produced apart 16 384 chains run at 2.3e11 chain-steps/s, the in-lane
                                    #  draw reaches that only from 262 144 chains on - two and more waves per SIMD)
    PREDRAW_CAP = 128 << 20         # bytes of pre-drawn records per launch (longer runs are cut into several launches); inside the 256 MB Infinity Cache

    def __init__(self, fn, compiled, reused=False):
        self.fn, self.comp, self.reused = fn, compiled, reused

    def begin(self, theta0, N, burn, inv_mass, seed, chain_offset):
        from .jit import runtime
        super().begin(theta0, N, burn, inv_mass, seed, chain_offset)
        C, D = theta0.shape
        self.module = self.comp.module(theta0.device)
        self.ws = torch.empty(runtime.hmc_workspace_bytes(C, D, theta0.element_size()), dtype=torch.uint8, device=theta0.device)
        self._ran = False
        # few chains = few waves (1024 chains: 16 of 1024 SIMDs hold one): the draws of a launch are then produced by a kernel of their
        # own that uses the whole GPU, and the trajectory kernel reads them (csrc/jit/hmc_callback.hip.in: hta_cb_predraw_kernel);
        # from PREDRAW_MAX_CHAINS on the chains fill the machine themselves and draw in the lane.  Bit-identical either way.
        self._predraw = C <= self.PREDRAW_MAX_CHAINS and os.environ.get("HAMILTORCH_AMD_JIT_PREDRAW", "1") != "0"
        self._pre = None

    def advance(self, n0, count, L, eps, H_old=None, H_new=None, progress=None):
        from .jit import runtime
        # one launch per block of trajectories; a visible progress bar cuts the run into ~20 launches so that it moves
        chunk = 1 if H_old is not None else (max(1, -(-count // 20)) if (progress is not None and progress.enabled) else count)
        C, D = self.cur.shape
        if self._predraw:
            per = runtime.hmc_predraw_bytes(C, D, 1, self.cur.element_size())
            chunk = max(1, min(chunk, self.PREDRAW_CAP // per))
            need = per * min(chunk, count)
            if self._pre is None or self._pre.numel() < need:
                self._pre = torch.empty(need, dtype=torch.uint8, device=self.cur.device)
        for start in range(n0, n0 + count, chunk):
            k = min(chunk, n0 + count - start)
            runtime.hmc_sample(self.module, self.cur, self.theta0, self.kind, self.im, self.mf, L, eps, k, start, self.burn,
                               self.seed, self.off, self.samples, self.rejected, self.ws, H_old, H_new, resume=self._ran,
                               pre=self._pre if self._predraw else None)
            self._ran = True
            if progress is not None:
                progress.update(min(self.N, start + k) - 1)

    def verify(self, k=128):
        """log p of the states the run ended in: the kernel's own value against the callable evaluated by torch (one vmap call on up
        to `k` chains).  False = the compiled code does not compute this callable (see sample())."""
        from . import jit
        if not self._ran or os.environ.get("HAMILTORCH_AMD_JIT_VERIFY", "1") == "0":
            return True
        C, D = self.cur.shape
        idx = slice(0, min(C, k))
        mine = jit.runtime.hmc_final_logp(self.ws, C, D, self.cur.dtype)[idx]
        ref = jit.torch_logp(self.fn, self.cur[idx]).to(mine.dtype).reshape(-1)
        fin_a, fin_b = torch.isfinite(mine), torch.isfinite(ref)
        tol = (2e-4 if mine.dtype == torch.float32 else 1e-9)
        close = (mine - ref).abs() <= tol * (10.0 + ref.abs())
        return bool(((fin_a == fin_b) & (close | ~fin_b)).all())


class _GenericHMC(_Engine):
    """Generic-callback path (plain HMC, S:267-304) -- also the SPLITTING integrator when given a
    list of callbacks (S:494-547)."""

    def __init__(self, fn, pass_grad=None, split=False, integrator=Integrator.SPLITTING):
        self.split, self.integrator = split, integrator
        self.cbs = [_BatchedCallback(f) for f in fn] if split else [_BatchedCallback(fn, pass_grad)]

    def _logp(self, theta):
        out = None
        for cb in self.cbs:           # split: sum over the subsets (S:787-796)
            v = cb.logp(theta)
            out = v if out is None else out + v
        return out.to(theta.dtype).contiguous()

    def begin(self, theta0, N, burn, inv_mass, seed, chain_offset):
        super().begin(theta0, N, burn, inv_mass, seed, chain_offset)
        C = theta0.shape[0]
        self.prop, self.p = torch.empty_like(theta0), torch.empty_like(theta0)
        self.Ho = torch.empty(C, dtype=theta0.dtype, device=theta0.device)
        self.Hn = torch.empty_like(self.Ho)
        # plain HMC: log p and its gradient AT THE CURRENT POINT, carried from trajectory to trajectory (see _trajectory)
        self._g_cur = None if self.split else torch.empty_like(theta0)
        self._lp_cur = torch.empty_like(self.Ho)
        self._acc = torch.zeros(C, dtype=torch.uint8, device=theta0.device)
        self._cache_valid = False
        # HAMILTORCH_AMD_CARRY=0: evaluate (log p, gradient) at the current point afresh every trajectory, as the reference does
        # (S:971, S:281) - for a callback whose value is not a pure function of its argument.  Read once per run.
        self._carry = os.environ.get("HAMILTORCH_AMD_CARRY", "1") != "0"

    def _refresh_cache(self):
        if self.split:                                  # the split integrators carry log p only (their first kick is one subset's)
            self._lp_cur.copy_(self._logp(self.cur))
        else:
            g, lp = self.cbs[0].grad(self.cur)
            self._g_cur.copy_(g); self._lp_cur.copy_(lp.to(self.cur.dtype))
        self._cache_valid = True

    def _trajectory(self, n, L, eps, Ho, Hn, n_dev=None):
        """One trajectory (S:969-1026).  n_dev: the trajectory index lives in device memory - the form a HIP graph replays."""
        cur, prop, p, kind, im, mf = self.cur, self.prop, self.p, self.kind, self.im, self.mf
        cb = self.cbs[0]
        if n_dev is None:
            _abi.momentum_resample(p, kind, mf, self.seed, self.off, n)                    # S:969
        else:
            _abi.momentum_resample_at(p, kind, mf, self.seed, self.off, n_dev)
        # Plain HMC: the reference evaluates log p at the current point for H_old (S:971) and differentiates it again for the
        # first half kick (S:281) - the point the previous trajectory ended at when it was accepted, the point it started from
        # when it was rejected.  Both values are known: (g_cur, lp_cur) follow the Metropolis decision chain by chain, so a
        # trajectory costs L callback evaluations instead of L + 2 (the native kernels carry lp_cur the same way).
        # (HAMILTORCH_AMD_CARRY=0: evaluate both afresh every trajectory, as the reference does - for a callback whose value
        #  is not a pure function of its argument)
        if not self._cache_valid or not self._carry:
            self._refresh_cache()
        _abi.hamiltonian(p, self._lp_cur, kind, im, Ho)                                    # S:971
        prop.copy_(cur)
        if self.split:
            perm = util.split_permutation(self.seed, n, len(self.cbs)) if self.integrator == Integrator.SPLITTING_RAND else None
            carry = None
            for _ in range(L):
                carry = _split_step(prop, p, self.cbs, eps, kind, im, self.integrator, perm, carry)   # S:499-596
            logp1 = self._logp(prop)
        else:
            g, logp1 = self._g_cur, self._lp_cur
            _abi.kick_drift(prop, p, g, 0.5 * eps, eps if L > 0 else 0.0, kind, im)        # S:281, S:284
            for l in range(L):
                g, logp1 = cb.grad(prop, own=False)                                        # S:297 (consumed before the next evaluation)
                _abi.kick_drift(prop, p, g, eps, 0.0 if l == L - 1 else eps, kind, im)     # S:298 (+ next drift)
            _abi.kick_drift(prop, p, g, -0.5 * eps, 0.0, kind, im)                         # S:302
            logp1 = logp1.to(cur.dtype).contiguous()   # log-prob at the end point, from the last gradient call
        _abi.hamiltonian(p, logp1, kind, im, Hn)                                           # S:995
        acc = self._acc
        if n_dev is None:
            row = self.samples[n - self.burn] if n > self.burn else None
            _abi.mh_select(cur, prop, self.theta0, Ho, Hn, logp1, row, self.rejected, acc, n, self.burn, self.seed,
                           self.off)                                                       # S:1000-1026
        else:
            _abi.mh_select_at(cur, prop, self.theta0, Ho, Hn, logp1, self.samples, self.rejected, acc, n_dev, self.burn,
                              self.seed, self.off)
            _abi.counter_add(n_dev, 1)
        if L > 0:
            took = acc.bool()
            if not self.split:
                self._g_cur.copy_(torch.where(took[:, None], g, self._g_cur))
            self._lp_cur.copy_(torch.where(took, logp1, self._lp_cur))
            if n_dev is None and n == self.burn + 1:                                       # Q2 (S:1018): rejected chains restart from
                self._cache_valid = False                                                  # params_init - their pair is recomputed

    def _graph_eligible(self, count, H_old):
        return (H_old is None and count >= 4 and not getattr(self, "_no_graph", False)
                and os.environ.get("HAMILTORCH_AMD_GRAPHS", "1") != "0"
                and not (self.split and self.integrator == Integrator.SPLITTING_RAND)      # its subset order is drawn on the host
                and all(cb.capturable() for cb in self.cbs))

    def advance(self, n0, count, L, eps, H_old=None, H_new=None, progress=None):
        # (Round 5 tried to run G of these loops side by side - G engines over contiguous chain blocks, each with its own captured
        #  trajectory graph and HIP stream: a callback trajectory is a chain of ~1000 dependent few-microsecond launches, and G such chains
        #  might overlap.  Measured on the notebook funnel at 1024 chains (profiles/r05g_chain_groups.txt): 1 group 1.21e7 chain-steps/s,
        #  2 groups 8.8e6, 4 groups 5.7e6, 8 groups 3.1e6 - the time follows the TOTAL node count: what bounds a replayed graph of tiny
        #  kernels is the rate at which the device retires dispatch packets (~2 us each), not the latency of one dependent chain.
        #  Not kept; fewer nodes is the only lever.)
        Ho = self.Ho if H_old is None else H_old
        Hn = self.Hn if H_new is None else H_new
        n, end = n0, n0 + count
        if self._graph_eligible(count, H_old):
            # The whole trajectory - native kernels and the torch callback - is captured ONCE as a HIP graph and replayed
            # with the trajectory index in device memory: no per-launch host work between the ~3 L launches of a trajectory.
            if progress is not None:
                progress.update(n)
            self._trajectory(n, L, eps, Ho, Hn)                                            # eager: settles vmap / capture fallbacks
            n += 1
            graph = self._capture_trajectory(n, L, eps, Ho, Hn) if self._graph_eligible(end - n, H_old) else None
            if graph is not None:
                n += 1                                                                     # the capture warm-up ran trajectory n
                while n < end:
                    if progress is not None:
                        progress.update(n)
                    graph.replay()
                    if n == self.burn + 1:                             # Q2 reset inside the replayed trajectory: refresh the carried pair
                        self._refresh_cache()
                    n += 1
        while n < end:
            if progress is not None:
                progress.update(n)
            self._trajectory(n, L, eps, Ho, Hn)
            n += 1

    def _capture_trajectory(self, n, L, eps, Ho, Hn):
        dev = self.cur.device
        n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._trajectory(n, L, eps, Ho, Hn, n_dev)                                     # warm-up = trajectory n itself
        torch.cuda.current_stream(dev).wait_stream(side)
        if n == self.burn + 1:                         # the warm-up ran the Q2 trajectory: the carried pair is refreshed OUTSIDE the graph
            self._refresh_cache()
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._trajectory(n, L, eps, Ho, Hn, n_dev)
        except Exception as e:      # e.g. a callback that is not capturable: stay eager (nothing ran during the capture)
            if isinstance(e, torch.OutOfMemoryError):
                raise
            torch.cuda.synchronize(dev)
            util.graph_log.append("trajectory: %s: %s" % (type(e).__name__, str(e).split("\n")[0][:160]))
            self._no_graph = True
            return None
        self._graph_keep = (graph, n_dev)          # keep the index tensor alive as long as the graph
        return graph


def _resolve_split_engine(log_prob_list, theta0, native, integrator=Integrator.SPLITTING):
    from . import bnn
    eng = bnn.native_split_engine(log_prob_list, theta0, integrator) if native else None
    return eng if eng is not None else _GenericHMC(log_prob_list, split=True, integrator=integrator)


# =================================================================================================
# BNN front-ends (S:1093-1466) live in bnn.py; re-exported here under the reference's names
# =================================================================================================
from .bnn import (define_model_log_prob, define_split_model_log_prob, predict_model,  # noqa: E402
                  sample_model, sample_split_model)

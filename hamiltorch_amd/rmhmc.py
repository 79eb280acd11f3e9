"""Riemannian-manifold HMC pieces of the API mirror (reference: hamiltorch/samplers.py, `S:`):
``fisher`` S:69-127, ``cholesky_inverse`` S:130-149, ``gibbs`` (RMHMC) S:183-184, ``rm_hamiltonian``
S:677-736, the explicit integrator S:389-462 and the RMHMC branch of ``sample`` S:969-1026.

The eigendecomposition / soft-abs map / solves / log-determinant run in
``csrc/rmhmc_metric.hip`` (one workgroup per system, matrices in LDS).  ``fisher``,
``cholesky_inverse``, ``rm_hamiltonian`` and the momentum draw accept any ``log_prob_func`` (its
Hessian comes from ``torch.func.hessian``).

The explicit *integrator* needs d H / d theta.  For constant-curvature targets (``GaussianTarget``,
BASELINE configs 3 and 5) that is -grad log p and a whole run is one C call.  For a general target
(the reference differentiates S:726-731 through hessian + eigh, S:398) it is

    d H / d theta_i = -d_i log p - d_i < Hess log p (theta), M >|_(M fixed),

with M = Q W Q^T built from the eigen-system of the metric by ``hta_metric_eval`` (``dmetric_out``, the
Daleckii-Krein form of the soft-abs derivative).  The contraction with the third derivatives of log p
never forms a D^3 tensor: it is one more reverse pass through ``torch.func.hessian`` per chain, batched
with ``vmap``.  Eight metric evaluations per step, each with its own jitter sub-stream, as the reference
(the constant-curvature path shares them pairwise because there dH/dtheta does not depend on the metric).
"""
from __future__ import annotations

import os

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
    H = torch.func.vmap(torch.func.hessian(f))(theta).to(theta.dtype)     # S:108
    logp = torch.func.vmap(f)(theta).to(theta.dtype)
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


class _Curvature:
    """Batched derivatives of a user log_prob_func (one chain per row) through torch.func; a callback torch.func cannot
    batch (``.item()``, data-dependent control flow: the reference handles those, one chain at a time) drops to a
    per-chain loop over ``torch.autograd.functional`` for the rest of the run."""

    def __init__(self, log_prob_func):
        f = lambda w: log_prob_func(w).sum()  # noqa: E731
        self.f = f
        self._loop = False
        self._gh_last = None          # (theta tensor, its version counter, g, -H): see grad_neg_hessian
        self.stats = {"gh_evaluated": 0, "gh_reused": 0}
        # each of these is replayed as a HIP graph (util.GraphedCallable): eager torch.func is hundreds of tiny launches
        self._val = util.GraphedCallable(torch.func.vmap(f))
        # gradient and Hessian in one forward-over-reverse pass: jacfwd of (grad, aux = grad)
        g = torch.func.grad(f)
        self._gh = util.GraphedCallable(torch.func.vmap(torch.func.jacfwd(lambda w: (g(w), g(w)), has_aux=True)))
        self._third = util.GraphedCallable(torch.func.vmap(torch.func.grad(lambda w, m: (torch.func.hessian(f)(w) * m).sum())))

    def _try(self, batched, looped, *args):
        if not self._loop:
            try:
                return batched(*args)
            except Exception as e:
                from .samplers import _not_batchable
                if not _not_batchable(e):
                    raise
                import warnings
                warnings.warn("hamiltorch_amd: log_prob_func is not vmap-able (%s: %s); evaluating its derivatives chain by chain"
                              % (type(e).__name__, str(e).split("\n")[0][:120]))
                self._loop = True
        return looped(*args)

    # (a callback may promote: e.g. constants it builds in float64 - results are brought back to the state's dtype)
    # (the graph outputs are static buffers: every result is copied / converted into a fresh tensor here)
    def value(self, theta):
        def looped(th):
            with torch.no_grad():
                return torch.stack([self.f(t) for t in th])
        return self._try(self._val, looped, theta).to(theta.dtype, copy=True).contiguous()

    def grad_neg_hessian(self, theta):
        """(gradient, negative Hessian) of log p at every chain's theta.  The integrators ask for them several times at the SAME
        state: both calls of a half step of the explicit integrator are evaluated at one (theta, p) pair (S:429-430, S:432-433:
        the momentum moves by dH/dtheta(theta, p_c), the copy's position by dH/dp(theta, p_c)), every iteration of the implicit
        momentum fixed point at one theta (S:313-340), H_new at the theta the last half step was evaluated at.  The reference
        differentiates again each time; here the last result is kept and handed back while `theta` is the same tensor object
        with the same version counter (an in-place update bumps it), so the callback's derivatives are evaluated once per
        state - only the metric evaluation (its own jitter draw) and the contraction are per call."""
        last = self._gh_last
        if last is not None and last[0] is theta and last[1] == theta._version:
            self.stats["gh_reused"] += 1
            return last[2], last[3]
        g, nH = self._grad_neg_hessian(theta)
        self._gh_last = (theta, theta._version, g, nH)
        self.stats["gh_evaluated"] += 1
        return g, nH

    def touched(self, *tensors):
        """The native kernels update states through raw pointers (no version bump): whoever hands a state to one says so."""
        if self._gh_last is not None and any(t is self._gh_last[0] for t in tensors):
            self._gh_last = None

    def _grad_neg_hessian(self, theta):
        def looped(th):
            H = torch.stack([torch.autograd.functional.hessian(self.f, t) for t in th])
            g = torch.stack([torch.autograd.grad(self.f(t_), t_)[0] for t_ in (t.detach().requires_grad_() for t in th)])
            return H, g
        H, g = self._try(self._gh, looped, theta)
        return g.to(theta.dtype, copy=True).contiguous(), (-H).to(theta.dtype).contiguous()

    def kick_update(self, theta, M, g, upd, coef):
        """upd += coef (g + c),  c_i = d_i < Hess log p (theta), M >: the momentum update of S:395-398 given the metric's M."""
        upd.add_(g + self.contract(theta, M), alpha=coef)

    def contract(self, theta, M):
        """c_i = d_i < Hess log p (theta), M >, M held fixed: [C, D]."""
        def looped(th, Mm):
            out = []
            for t, m in zip(th, Mm):
                t = t.detach().requires_grad_()
                with torch.enable_grad():
                    Hm = (torch.autograd.functional.hessian(self.f, t, create_graph=True) * m).sum()
                out.append(torch.autograd.grad(Hm, t, allow_unused=True)[0] if Hm.requires_grad else torch.zeros_like(t))
            return torch.stack([o if o is not None else torch.zeros_like(th[0]) for o in out])
        return self._try(self._third, looped, theta, M).to(theta.dtype, copy=True).contiguous()


class _CompiledCurvature:
    """The same three requests answered by COMPILED code (hamiltorch_amd/jit/, csrc/jit/derivs_callback.hip.in): the callable is
    traced once, differentiated three times on its scalar graph and built into two kernels - (log p, gradient, -Hessian) in ONE
    launch per state, the third-derivative contraction (fused with the momentum update) in one launch per kick - where
    `_Curvature` replays ~100 torch.func launches per request."""

    def __init__(self, log_prob_func, compiled, like):
        self.fn, self.comp = log_prob_func, compiled
        self.module = compiled.module(like.device)
        self._last = None             # (theta tensor, version, lp, g, -H)
        self.stats = {"gh_evaluated": 0, "gh_reused": 0, "compiled": True}

    def _eval(self, theta):
        from .jit import runtime
        last = self._last
        if last is not None and last[0] is theta and last[1] == theta._version:
            self.stats["gh_reused"] += 1
            return last
        C, D = theta.shape
        lp = torch.empty(C, dtype=theta.dtype, device=theta.device)
        g = torch.empty_like(theta)
        nH = torch.empty(C, D, D, dtype=theta.dtype, device=theta.device)
        runtime.derivs(self.module, theta, lp, g, nH)
        self.stats["gh_evaluated"] += 1
        self._last = (theta, theta._version, lp, g, nH)
        return self._last

    def value(self, theta):
        return self._eval(theta)[2]

    def grad_neg_hessian(self, theta):
        e = self._eval(theta)
        return e[3], e[4]

    def touched(self, *tensors):
        if self._last is not None and any(t is self._last[0] for t in tensors):
            self._last = None

    def contract(self, theta, M):
        from .jit import runtime
        out = torch.empty_like(theta)
        runtime.contract(self.module, theta, M, out=out)
        return out

    def kick_update(self, theta, M, g, upd, coef):
        from .jit import runtime
        runtime.contract(self.module, theta, M, upd=upd, grad_in=g, coef=coef)
        self.touched(upd)

    def agrees_with_callable(self, theta, k=16):
        """Value and gradient of the compiled code against torch's evaluation of the callable on up to k rows of theta."""
        th = theta[:k].contiguous()
        e = self._eval(th)
        self._last = None
        self.stats["gh_evaluated"] -= 1
        f = lambda w: self.fn(w).sum()  # noqa: E731
        try:
            g, v = torch.func.vmap(torch.func.grad_and_value(f))(th)
        except Exception:
            return True                 # torch cannot batch it: nothing to compare with here (the samplers' own guards apply)
        tol = 2e-4 if th.dtype == torch.float32 else 1e-9
        ok = True
        for mine, ref in ((e[2], v.to(th.dtype)), (e[3], g.to(th.dtype))):
            fa, fb = torch.isfinite(mine), torch.isfinite(ref)
            ok = ok and bool(((fa == fb) & (((mine - ref).abs() <= tol * (10.0 + ref.abs())) | ~fb)).all())
        return ok


def _curvature_for(log_prob_func, theta):
    """Compiled derivatives when the callback compiler covers the callable (checked against torch on this call's states), else the
    torch.func evaluation."""
    from . import jit
    if jit.enabled() and callable(log_prob_func) and theta.is_cuda:
        for fresh in (False, True):
            try:
                cv = _CompiledCurvature(log_prob_func, jit.compile_derivs(log_prob_func, theta[0], theta.dtype, fresh=fresh), theta)
            except jit.Unsupported as e:
                _abi.load().hta_jit_note_fallback(str(e)[:140].encode("utf-8", "replace"))
                break
            if cv.agrees_with_callable(theta):
                return cv
        else:
            import warnings
            warnings.warn("hamiltorch_amd: the compiled derivatives of %r disagree with torch's; using torch.func" % (log_prob_func,))
    return _Curvature(log_prob_func)


class _WarmBases:
    """Per-chain eigenbases carried from one metric evaluation of a general target to the next (HtaMetricArgs.v0_stride,
    ABI 7): `kw(slot)` = the V0 / v0_stride / V_out arguments of `_abi.metric_eval` for the chains' state `slot` ("a": the
    state set (theta, p), "b": the explicit integrator's copy) - a [C, D, D] tensor, the identity before the first call,
    updated in place by every call.  The evaluation then runs on the matrix cores (csrc/rmhmc_metric_mfma.hip): the
    curvature rotated into the previous basis is nearly diagonal, so it is refined (or finished by a few Jacobi sweeps
    inside the launch) instead of being diagonalised from scratch - a hint only, results agree to rounding (the samplers
    reset the bases once per trajectory: `reset`).
    fp32, soft-abs, D <= 112 on the device; otherwise `kw` is empty and the evaluation is the cold one."""

    def __init__(self, like, kind):
        C, D = like.shape
        self.on = (kind == _abi.METRIC_SOFTABS and like.dtype == torch.float32 and like.is_cuda and D <= _abi.METRIC_MFMA_MAX_D
                   and os.environ.get("HAMILTORCH_AMD_WARM_METRIC", "1") != "0")
        self.like, self.bufs = like, {}

    def kw(self, slot):
        if not self.on:
            return {}
        buf = self.bufs.get(slot)
        if buf is None:
            C, D = self.like.shape
            buf = torch.eye(D, dtype=self.like.dtype, device=self.like.device).repeat(C, 1, 1).contiguous()
            self.bufs[slot] = buf
        D = buf.shape[-1]
        return {"V0": buf, "v0_stride": D * D, "V_out": buf}

    def reset(self):
        """Back to the identity (a cold evaluation next): the kernel treats V0 as exactly orthogonal, and `V <- V0 X` in place
        lets rounding accumulate in V^T V - I over a long run.  The samplers call this once per trajectory:
        one cold evaluation in 8 L + 3, the drift bounded by a trajectory's products."""
        if os.environ.get("HAMILTORCH_AMD_WARM_RESET", "1") == "0":     # measurement knob: what the reset costs - see CHANGELOG round 5
            return
        for buf in self.bufs.values():
            buf.zero_()
            buf.diagonal(dim1=-2, dim2=-1).fill_(1.0)


def _generic_steps(cv, kind, th, pm, thc, pmc, steps, eps, omega, alpha, jitter, seed, chain_offset, draw, path=None, warm=None):
    """S:425-461 on the augmented state, in place.  Unlike the constant-curvature path, dH/dtheta depends on the
    metric here, so each of the reference's 8 gradient calls per step keeps its own metric evaluation and jitter
    sub-stream (2 + 8 l + k, k = 0..7 in the reference's call order)."""
    C, D = th.shape                 # Metric.HESSIAN needs a log-concave target (G = -Hessian positive definite), as in the reference
    eh = 0.5 * eps
    M = torch.empty(C, D, D, dtype=th.dtype, device=th.device)
    warm = warm if warm is not None else _WarmBases(th, kind)

    def kick(theta, mvec, upd, sub, slot):      # upd -= eh dH/dtheta(theta, mvec),  dH/dtheta = -(g + c)   (S:395-398)
        g, Hs = cv.grad_neg_hessian(theta)
        _abi.metric_eval(theta, C, D, kind, Hs, D * D, alpha, jitter, seed, chain_offset, draw, sub, m=mvec, dmetric_out=M,
                         **warm.kw(slot))
        fused = getattr(cv, "kick_update", None)         # (a curvature object only has to provide grad_neg_hessian / contract / value / touched)
        if fused is not None:
            fused(theta, M, g, upd, eh)
        else:
            upd.add_(g + cv.contract(theta, M), alpha=eh)

    def drift(theta, mvec, upd, sub, slot):     # upd += eh dH/dp(theta, mvec) = eh G^-1 mvec              (S:415-422)
        _, Hs = cv.grad_neg_hessian(theta)
        _abi.metric_eval(theta, C, D, kind, Hs, D * D, alpha, jitter, seed, chain_offset, draw, sub, m=mvec, upd_x=upd, cx=eh,
                         **warm.kw(slot))
        cv.touched(upd)

    for l in range(steps):
        k0 = 2 + 8 * l
        kick(th, pmc, pm, k0 + 0, "a"); drift(th, pmc, thc, k0 + 1, "a")           # phi_A(1/2)  S:429-430
        drift(thc, pm, th, k0 + 2, "b"); kick(thc, pm, pmc, k0 + 3, "b")           # phi_B(1/2)  S:432-433
        _abi.rmhmc_binding_rotation(th, pm, thc, pmc, eps, omega)                  # phi_C       S:447-450
        cv.touched(th, thc)
        drift(thc, pm, th, k0 + 4, "b"); kick(thc, pm, pmc, k0 + 5, "b")           # phi_B(1/2)  S:454-455
        kick(th, pmc, pm, k0 + 6, "a"); drift(th, pmc, thc, k0 + 7, "a")           # phi_A(1/2)  S:457-458
        if path is not None:
            path[0][l].copy_(th); path[1][l].copy_(pm)


def explicit_leapfrog(params, momentum, log_prob_func, steps, step_size, jitter, softabs_const, omega, metric,
                      seed=None, chain_offset=0, draw=0):
    """S:389-462.  Returns ([ret_params, params_copy], [ret_momenta, momentum_copy]) like the reference."""
    theta, one = _batch(params)
    p, _ = _batch(momentum, "momentum")
    theta, p = theta.clone(), p.clone()
    _abi.require_device(theta, "params")
    tgt = as_gaussian(log_prob_func, theta)
    thc, pc = theta.clone(), p.clone()
    pt = torch.empty((steps,) + theta.shape, dtype=theta.dtype, device=theta.device)
    pp = torch.empty_like(pt)
    seed = util.next_stream_seed() if (seed is None and jitter is not None) else (seed or 0)
    if tgt is not None:
        _abi.rmhmc_gaussian_leapfrog(theta, p, thc, pc, tgt.precision, tgt.mean, _metric_kind(metric), softabs_const,
                                     jitter, seed, chain_offset, draw, steps, step_size, omega, pt, pp)
    else:
        _generic_steps(_curvature_for(log_prob_func, theta), _metric_kind(metric), theta, p, thc, pc, steps, step_size, omega,
                       softabs_const, jitter, seed, chain_offset, draw, path=(pt, pp))
    unb = (lambda t: t[0]) if one else (lambda t: t)
    return [[unb(t) for t in pt.unbind(0)], unb(thc)], [[unb(t) for t in pp.unbind(0)], unb(pc)]


def sample_explicit(log_prob_func, theta0, N, L, eps, burn, jitter, softabs_const, omega, metric, seed, chain_offset,
                    verbose):
    """The RMHMC / EXPLICIT branch of sample() (S:969-1026): one C call enqueues the whole run."""
    from .samplers import _num_rows
    tgt = as_gaussian(log_prob_func, theta0)
    kind = _metric_kind(metric)
    if softabs_const is None and kind == _abi.METRIC_SOFTABS:
        raise TypeError("softabs_const must be set for Metric.SOFTABS")
    if tgt is None:
        out = _sample_explicit_compiled(log_prob_func, theta0, N, L, eps, burn, jitter, softabs_const, omega, kind, seed,
                                        chain_offset, verbose)
        if out is not None:
            return out
        return _sample_explicit_generic(log_prob_func, theta0, N, L, eps, burn, jitter, softabs_const, omega, kind, seed,
                                        chain_offset, verbose)
    C, D = theta0.shape
    S = _num_rows(N, burn)
    samples = torch.empty((S, C, D), dtype=theta0.dtype, device=theta0.device)
    samples[0].copy_(theta0)
    cur = theta0.clone()
    rejected = torch.zeros(C, dtype=torch.int32, device=theta0.device)
    ws = _prepared_workspace(tgt, theta0, kind, softabs_const, jitter, N)
    prog = util._Progress('Sampling (Sampler.RMHMC; Integrator.EXPLICIT)', N, verbose)
    _abi.rmhmc_gaussian_sample(cur, theta0, tgt.precision, tgt.mean, tgt.log_norm, kind, softabs_const, jitter, L, eps,
                               omega, N, 0, burn, seed, chain_offset, samples, rejected, ws)
    prog.end()
    return samples, rejected


class _WorkspaceHandle:
    """Owns a prepared RMHMC workspace: the library's (device, pointer) -> plan entry goes when the buffer does."""

    def __init__(self, ws):
        self.ws = ws

    def __del__(self):
        try:
            _abi.rmhmc_gaussian_forget(self.ws)
        except Exception:       # interpreter shutdown
            pass


def _prepared_workspace(tgt, theta0, kind, alpha, jitter, N):
    """The workspace of hta_rmhmc_gaussian_sample for this target, PREPARED once (hta_rmhmc_gaussian_prepare: the cold
    eigendecomposition of the precision matrix, the fused route's plan with its read-back + synchronise, the shared inverse:
    1.2 ms at D = 100, as much as 12 trajectories at 1024 chains) and kept on the target object: a run cut into several
    sample() calls, or repeated runs on one target, pay it once.  Keyed by everything the setup depends on, by the stream the
    kernels run on (two streams never share a buffer) and by the precision / mean tensors' version counters (an in-place edit
    of the target prepares again)."""
    C, D = theta0.shape
    need = _abi.rmhmc_workspace_bytes(C, D, theta0.element_size(), N)
    key = (theta0.device, theta0.dtype, C, D, int(kind), None if alpha is None else float(alpha), None if jitter is None else float(jitter),
           torch.cuda.current_stream(theta0.device).cuda_stream)
    # (the entry holds the tensor OBJECTS: while they are cached their storage cannot be freed and handed to another matrix)
    sig = (tgt.precision, tgt.mean, tgt.precision.data_ptr(), tgt.precision._version, tgt.mean.data_ptr(), tgt.mean._version)
    cache = tgt.__dict__.setdefault("_hta_rm_ws", {})
    hit = cache.get(key)
    if hit is not None and hit[1][0] is sig[0] and hit[1][1] is sig[1] and hit[1][2:] == sig[2:] and hit[0].ws.numel() >= need:
        return hit[0].ws
    ws = torch.empty(need, dtype=torch.uint8, device=theta0.device)
    _abi.rmhmc_gaussian_prepare(theta0, tgt.precision, tgt.mean, kind, alpha, jitter, C, ws)
    if len(cache) >= 4:                 # a handful of shapes per target at most
        cache.clear()
    cache[key] = (_WorkspaceHandle(ws), sig)
    return ws


def _sample_explicit_compiled(log_prob_func, theta0, N, L, eps, burn, jitter, alpha, omega, kind, seed, chain_offset, verbose):
    """The whole run inside the compiled chain-per-lane kernel (hamiltorch_amd/jit/, csrc/jit/rmhmc_callback.hip.in) when the callback
    compiler covers the callable: soft-abs metric, D <= 16.  None = not applicable (the reason is in hta_last_route()); the result is
    checked against the callable itself on the states the run ended in, a mismatch re-traces once and else returns None."""
    from . import jit
    from .samplers import _num_rows
    C, D = theta0.shape
    if not (jit.enabled() and callable(log_prob_func) and theta0.is_cuda and kind == _abi.METRIC_SOFTABS):
        return None
    for fresh in (False, True):
        hits = jit.stats["trace_hits"]
        try:
            comp = jit.compile_rmhmc(log_prob_func, theta0[0], theta0.dtype, jitter is not None, fresh=fresh)
        except jit.Unsupported as e:
            _abi.load().hta_jit_note_fallback(str(e)[:140].encode("utf-8", "replace"))
            return None
        reused = jit.stats["trace_hits"] > hits
        module = comp.module(theta0.device)
        S = _num_rows(N, burn)
        samples = torch.empty((S, C, D), dtype=theta0.dtype, device=theta0.device)
        cur = torch.empty_like(theta0)
        rejected = torch.empty(C, dtype=torch.int32, device=theta0.device)
        _abi.run_begin(theta0, cur, samples[0], rejected)
        ws = torch.empty(jit.runtime.rmhmc_workspace_bytes(C, D, theta0.element_size()), dtype=torch.uint8, device=theta0.device)
        prog = util._Progress('Sampling (Sampler.RMHMC; Integrator.EXPLICIT)', N, verbose)
        chunk = max(1, -(-N // 20)) if verbose else N
        for start in range(0, N, chunk):
            k = min(chunk, N - start)
            jit.runtime.rmhmc_sample(module, cur, theta0, L, eps, alpha, jitter, omega, k, start, burn, seed, chain_offset, samples,
                                     rejected, ws)
            prog.update(start + k - 1)
        prog.end()
        if os.environ.get("HAMILTORCH_AMD_JIT_VERIFY", "1") == "0" or N == 0:
            return samples, rejected
        kk = min(C, 128)
        mine = ws[:C * theta0.element_size()].view(theta0.dtype)[:kk]
        ref = jit.torch_logp(log_prob_func, cur[:kk]).to(mine.dtype).reshape(-1)
        fa, fb = torch.isfinite(mine), torch.isfinite(ref)
        tol = 2e-4 if mine.dtype == torch.float32 else 1e-9
        if bool(((fa == fb) & (((mine - ref).abs() <= tol * (10.0 + ref.abs())) | ~fb)).all()):
            return samples, rejected
        if not reused:
            break
    import warnings
    warnings.warn("hamiltorch_amd: the compiled form of %r disagrees with the callable itself on the sampled states; re-running "
                  "on the launch-per-evaluation path" % (log_prob_func,))
    _abi.load().hta_jit_note_fallback(b"compiled code disagrees with the callable")
    return None


def _sample_explicit_generic(log_prob_func, theta0, N, L, eps, burn, jitter, alpha, omega, kind, seed, chain_offset,
                             verbose):
    """The same trajectory loop as csrc/rmhmc_explicit.hip:rmhmc_sample with the target's derivatives coming from
    torch.func: gibbs (sub-stream 0), H_old (1), L explicit steps (2 .. 2+8L-1), H_new on the un-augmented pair
    (2+8L, Q4), Metropolis select + bookkeeping in `hta_mh_select`."""
    from .samplers import _num_rows
    C, D = theta0.shape
    dt, dev = theta0.dtype, theta0.device
    cv = _curvature_for(log_prob_func, theta0)
    S = _num_rows(N, burn)
    samples = torch.empty((S, C, D), dtype=dt, device=dev)
    samples[0].copy_(theta0)
    cur = theta0.clone()
    rejected = torch.zeros(C, dtype=torch.int32, device=dev)
    H0 = torch.empty(C, dtype=dt, device=dev); H1 = torch.empty_like(H0)
    pm = torch.empty_like(cur)
    warm = _WarmBases(cur, kind)
    acc = torch.zeros(C, dtype=torch.uint8, device=dev)
    Hs = lp0 = None               # curvature and log p AT THE CURRENT POINT, carried across trajectories (see the end of the loop)
    carry = os.environ.get("HAMILTORCH_AMD_CARRY", "1") != "0"       # 0: differentiate again every trajectory (an impure callback)
    prog = util._Progress('Sampling (Sampler.RMHMC; Integrator.EXPLICIT)', N, verbose)
    for n in range(N):
        warm.reset()
        if Hs is None:
            _, Hs = cv.grad_neg_hessian(cur)
            lp0 = cv.value(cur)
        _abi.metric_eval(cur, C, D, kind, Hs, D * D, alpha, jitter, seed, chain_offset, n, 0, p_out=pm, **warm.kw("a"))        # S:183-184
        _abi.metric_eval(cur, C, D, kind, Hs, D * D, alpha, jitter, seed, chain_offset, n, 1, m=pm, H_out=H0, **warm.kw("a"))  # S:971
        H0.sub_(lp0)
        th, thc, pmc = cur.clone(), cur.clone(), pm.clone()
        _generic_steps(cv, kind, th, pm, thc, pmc, L, eps, omega, alpha, jitter, seed, chain_offset, n, warm=warm)
        _, Hs1 = cv.grad_neg_hessian(th)
        lp1 = cv.value(th)
        _abi.metric_eval(th, C, D, kind, Hs1, D * D, alpha, jitter, seed, chain_offset, n, 2 + 8 * L, m=pm, H_out=H1,
                         **warm.kw("a"))                                                                                          # S:989
        H1.sub_(lp1)
        row = samples[n - burn] if n > burn else None
        _abi.mh_select(cur, th, theta0, H0, H1, lp1, row, rejected, acc, n, burn, seed, chain_offset)
        cv.touched(cur)
        # the next trajectory starts where this one ended (accepted) or started (rejected): both points' curvature and log p are
        # known - the reference differentiates again (S:971 -> S:822); after the Q2 reset (S:1018) they are recomputed
        took = acc.bool()
        fresh = n == burn + 1 or not carry
        Hs = None if fresh else torch.where(took[:, None, None], Hs1, Hs)
        lp0 = None if fresh else torch.where(took, lp1, lp0)
        prog.update(n)
    prog.end()
    return samples, rejected


# ---- implicit RMHMC: the generalised leapfrog with fixed-point iterations (S:305-387) -------------------------
def _implicit_subs(max_it):
    """jitter sub-streams reserved per implicit step: max_it momentum iterations, 1 + max_it position
    evaluations, 1 final kick.  A fixed layout (the reference draws from one global generator in call order):
    a chain's draws then do not depend on how many iterations the other chains of the batch needed."""
    return 2 * int(max_it) + 2


def _implicit_steps(cv, kind, th, pm, steps, eps, alpha, jitter, seed, chain_offset, draw, thr, max_it, sub0=2, path=None,
                    warm=None):
    """S:312-383 for a batch of chains, in place on (th, pm).  Chains leave a fixed-point loop individually once their
    own max squared update is below `thr` (the reference's `break`); the loop ends when none is left or after max_it."""
    C, D = th.shape
    hs = 0.5 * eps
    M = torch.empty(C, D, D, dtype=th.dtype, device=th.device)
    x = torch.empty_like(th)
    warm = warm if warm is not None else _WarmBases(th, kind)

    def dH_dtheta(theta, mvec, sub):                     # S:318-319 / S:368-369: -(g + c)
        g, Hs = cv.grad_neg_hessian(theta)
        _abi.metric_eval(theta, C, D, kind, Hs, D * D, alpha, jitter, seed, chain_offset, draw, sub, m=mvec, dmetric_out=M,
                         **warm.kw("a"))
        return -(g + cv.contract(theta, M))

    def dH_dp(theta, mvec, sub):                         # S:346-347, S:352-353: G^-1 p
        _, Hs = cv.grad_neg_hessian(theta)
        _abi.metric_eval(theta, C, D, kind, Hs, D * D, alpha, jitter, seed, chain_offset, draw, sub, m=mvec, x_out=x,
                         **warm.kw("a"))
        return x.clone()

    def fixed_point(state, update):
        active = torch.ones(C, dtype=torch.bool, device=th.device)
        for i in range(max_it):
            new = update(i)
            diff = ((state - new) ** 2).amax(dim=1)                       # S:336 / S:356
            state.copy_(torch.where(active[:, None], new, state))
            active &= ~(diff < thr) & torch.isfinite(diff)                # a diverged chain stops iterating: it is rejected later
            if not bool(active.any()):
                break

    per = _implicit_subs(max_it)
    for l in range(steps):
        base = sub0 + l * per
        p_old = pm.clone()
        fixed_point(pm, lambda i: p_old - hs * dH_dtheta(th, pm, base + i))                       # S:313-340
        th_old = th.clone()
        g0 = dH_dp(th, pm, base + max_it)                                                         # S:344-348 (th == th_old here: the first
                                                                                                  #  iteration below reuses its Hessian)
        fixed_point(th, lambda i: th_old + hs * dH_dp(th, pm, base + max_it + 1 + i) + hs * g0)   # S:349-361
        pm.sub_(hs * dH_dtheta(th, pm, base + 2 * max_it + 1))                                    # S:368-383
        if path is not None:
            path[0][l].copy_(th); path[1][l].copy_(pm)


def implicit_leapfrog(params, momentum, log_prob_func, steps, step_size, jitter, softabs_const, metric,
                      fixed_point_threshold, fixed_point_max_iterations, seed=None, chain_offset=0, draw=0):
    """leapfrog(sampler=RMHMC, integrator=IMPLICIT): (ret_params, ret_momenta), one entry per step."""
    theta, one = _batch(params)
    p, _ = _batch(momentum, "momentum")
    theta, p = theta.clone(), p.clone()
    _abi.require_device(theta, "params")
    pt = torch.empty((steps,) + theta.shape, dtype=theta.dtype, device=theta.device)
    pp = torch.empty_like(pt)
    seed = util.next_stream_seed() if (seed is None and jitter is not None) else (seed or 0)
    _implicit_steps(_curvature_for(log_prob_func, theta), _metric_kind(metric), theta, p, steps, step_size, softabs_const, jitter, seed,
                    chain_offset, draw, fixed_point_threshold, fixed_point_max_iterations, path=(pt, pp))
    unb = (lambda t: t[0]) if one else (lambda t: t)
    return [unb(t) for t in pt.unbind(0)], [unb(t) for t in pp.unbind(0)]


def sample_implicit(log_prob_func, theta0, N, L, eps, burn, jitter, alpha, metric, thr, max_it, seed, chain_offset, verbose):
    """The RMHMC / IMPLICIT branch of sample() (S:969-1026): gibbs (sub-stream 0), H_old (1), L implicit steps
    (2 ..), H_new (2 + L * (2 max_it + 2)), Metropolis select + bookkeeping in `hta_mh_select`."""
    from .samplers import _num_rows
    kind = _metric_kind(metric)
    if alpha is None and kind == _abi.METRIC_SOFTABS:
        raise TypeError("softabs_const must be set for Metric.SOFTABS")
    C, D = theta0.shape
    dt, dev = theta0.dtype, theta0.device
    cv = _curvature_for(log_prob_func, theta0)
    S = _num_rows(N, burn)
    samples = torch.empty((S, C, D), dtype=dt, device=dev)
    samples[0].copy_(theta0)
    cur = theta0.clone()
    rejected = torch.zeros(C, dtype=torch.int32, device=dev)
    H0 = torch.empty(C, dtype=dt, device=dev); H1 = torch.empty_like(H0)
    pm = torch.empty_like(cur)
    warm = _WarmBases(cur, kind)
    acc = torch.zeros(C, dtype=torch.uint8, device=dev)
    Hs = lp0 = None               # as in _sample_explicit_generic: carried across trajectories
    carry = os.environ.get("HAMILTORCH_AMD_CARRY", "1") != "0"
    prog = util._Progress('Sampling (Sampler.RMHMC; Integrator.IMPLICIT)', N, verbose)
    for n in range(N):
        warm.reset()
        if Hs is None:
            _, Hs = cv.grad_neg_hessian(cur)
            lp0 = cv.value(cur)
        _abi.metric_eval(cur, C, D, kind, Hs, D * D, alpha, jitter, seed, chain_offset, n, 0, p_out=pm, **warm.kw("a"))        # S:183-184
        _abi.metric_eval(cur, C, D, kind, Hs, D * D, alpha, jitter, seed, chain_offset, n, 1, m=pm, H_out=H0, **warm.kw("a"))  # S:971
        H0.sub_(lp0)
        th = cur.clone()
        _implicit_steps(cv, kind, th, pm, L, eps, alpha, jitter, seed, chain_offset, n, thr, max_it, warm=warm)
        _, Hs1 = cv.grad_neg_hessian(th)
        lp1 = cv.value(th)
        _abi.metric_eval(th, C, D, kind, Hs1, D * D, alpha, jitter, seed, chain_offset, n, 2 + L * _implicit_subs(max_it),
                         m=pm, H_out=H1, **warm.kw("a"))                                                      # S:989
        H1.sub_(lp1)
        row = samples[n - burn] if n > burn else None
        _abi.mh_select(cur, th, theta0, H0, H1, lp1, row, rejected, acc, n, burn, seed, chain_offset)
        cv.touched(cur)
        took = acc.bool()
        fresh = n == burn + 1 or not carry
        Hs = None if fresh else torch.where(took[:, None, None], Hs1, Hs)
        lp0 = None if fresh else torch.where(took, lp1, lp0)
        prog.update(n)
    prog.end()
    return samples, rejected

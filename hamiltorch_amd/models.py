"""Model plugins: objects that are valid ``log_prob_func`` callables (the reference's callback
contract, samplers.py:272-274) AND carry the closed form the HIP kernels need, so that
``sample()`` can run whole trajectories on-device instead of calling back into torch.

An unrecognised callable still works: it is evaluated by torch (``torch.func.vmap`` over
chains) with the HIP kernels doing the state updates in between (samplers._GenericHMC).

The reference's own idiom is neither of the recognised objects but a closure,
``lambda w: MultivariateNormal(mean, cov).log_prob(w).sum()`` (tests/test_util.py:98-101 and every
notebook).  ``probe_gaussian`` recognises such closures by what they compute: constant curvature and a
gradient affine in the input at widely spread probe points; ``verify_gaussian`` re-checks the recovered
closed form against the closure on the samples a run produced (``sample`` falls back to the
generic-callback path on any mismatch).
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

    def _hta_batched_grad(self, theta):
        """(grad[C, D], logp[C]) for all chains from ONE [C, D] x [D, D] product (the callback path's form of params_grad, S:270-278,
        for a target beyond the fused kernels' D <= 1024: GEMM-shaped work, rocBLAS on the matrix cores - instead of autograd's
        forward and backward matrix-vector products under vmap)."""
        d = theta - self.mean
        Pd = d @ self.precision                               # symmetric
        return -Pd, self.log_norm - 0.5 * (d * Pd).sum(dim=1)

    def _hta_batched_logp(self, theta):
        return self._hta_batched_grad(theta)[1]

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


MAX_NATIVE_DIM = 1024      # hta_hmc_gaussian_sample's limit (csrc/hmc_gaussian.hip: wave-per-chain kernel)


def _tol(dtype):
    return 2e-4 if dtype == torch.float32 else 1e-9


def probe_gaussian(log_prob_func, theta0, max_dim=MAX_NATIVE_DIM):
    """GaussianTarget equal to ``log_prob_func`` if that callable is a quadratic form, else None.

    Evaluates value, gradient and Hessian (``torch.func``) at five probe points spread over three orders of
    magnitude around ``theta0[0]`` (+-1, +-30, +1000 along random directions) and accepts only if
      * every Hessian is finite, symmetric and equal to the first to rounding (constant curvature),
      * the gradients are the affine map ``-P (x - mu)`` of ONE ``mu`` (solved from the first point),
      * the values are ``log_norm - 1/2 (x - mu)^T P (x - mu)`` with ONE ``log_norm``.
    A function that is only piecewise quadratic can pass when every probe lands in one piece; ``sample``
    therefore verifies the closed form on the run's own samples afterwards (``verify_gaussian``).
    Anything torch.func cannot differentiate twice / batch returns None (the generic path handles it)."""
    if not callable(log_prob_func) or isinstance(log_prob_func, (list, tuple)):
        return None
    x0 = theta0.detach().reshape(-1, theta0.shape[-1])[0]
    D = x0.numel()
    if D > max_dim:
        return None
    dt, dev = x0.dtype, x0.device
    if dt not in (torch.float32, torch.float64):
        return None
    g = torch.Generator(device="cpu").manual_seed(0x6A55)
    dirs = torch.randn(5, D, generator=g, dtype=torch.float64)
    dirs = dirs / dirs.norm(dim=1, keepdim=True).clamp_min(1e-30) * (D ** 0.5)
    scale = torch.tensor([1.0, -1.0, 30.0, -30.0, 1000.0], dtype=torch.float64)[:, None]
    pts = (x0.double().cpu()[None] + scale * dirs).to(device=dev, dtype=dt)

    def f(w):
        r = log_prob_func(w)
        if isinstance(r, tuple):
            raise TypeError("tuple protocol")
        return r.sum()
    tol = _tol(dt)
    try:
        with torch.enable_grad():
            # cheap screen first (4 gradients): along one line the gradient of a quadratic is affine in the step
            line = (x0.double().cpu()[None] + torch.tensor([0.0, 1.0, -1.0, 30.0], dtype=torch.float64)[:, None] * dirs[0]).to(device=dev, dtype=dt)
            gl = torch.func.vmap(torch.func.grad(f))(line).double()
            if gl.shape != (4, D) or not torch.isfinite(gl).all():
                return None
            d1, d2, d3 = gl[1] - gl[0], gl[2] - gl[0], gl[3] - gl[0]
            gs = float(gl.abs().max()) + 1e-30
            if float((d1 + d2).abs().max()) > 50 * tol * gs or float((d3 - 30.0 * d1).abs().max()) > 50 * tol * gs * 30:
                return None
            H = torch.func.vmap(torch.func.hessian(f))(pts)
            gr = torch.func.vmap(torch.func.grad(f))(pts)
            v = torch.func.vmap(f)(pts)
    except (RuntimeError, TypeError, ValueError, NotImplementedError, AttributeError, IndexError) as e:
        if isinstance(e, (torch.OutOfMemoryError, torch.AcceleratorError)):
            raise
        return None
    if H.shape != (5, D, D) or gr.shape != (5, D) or v.shape != (5,):
        return None
    H, gr, v, X = H.double(), gr.double(), v.double(), pts.double()
    if not (torch.isfinite(H).all() and torch.isfinite(gr).all() and torch.isfinite(v).all()):
        return None
    hs = float(H[0].abs().max())
    if hs == 0.0:
        return None                                             # flat / linear: not a Gaussian
    if float((H - H[0]).abs().max()) > tol * hs or float((H[0] - H[0].T).abs().max()) > tol * hs:
        return None
    P = -0.5 * (H[0] + H[0].T)
    try:
        mu = X[0] + torch.linalg.solve(P, gr[0])               # grad = -P (x - mu)
    except RuntimeError:
        return None
    if not torch.isfinite(mu).all():
        return None
    d = X - mu
    want_g = -(d @ P)
    gs = float(want_g.abs().max()) + float(gr.abs().max())
    if float((gr - want_g).abs().max()) > 50 * tol * max(gs, 1e-30):
        return None
    quad = 0.5 * ((d @ P) * d).sum(-1)
    log_norm = float(v[0] + quad[0])
    vs = float(quad.abs().max()) + abs(log_norm)
    if float((v - (log_norm - quad)).abs().max()) > 50 * tol * max(vs, 1e-30):
        return None
    tgt = object.__new__(GaussianTarget)
    tgt.mean = mu.to(device=dev, dtype=dt).contiguous()
    tgt.precision = P.to(device=dev, dtype=dt).contiguous()
    tgt.log_norm = log_norm
    tgt.probed_from = log_prob_func
    return tgt


def verify_gaussian(tgt, log_prob_func, samples, max_rows=2048):
    """True if the closed form ``tgt`` reproduces ``log_prob_func`` -- VALUE AND GRADIENT -- on rows drawn evenly from
    ``samples[S, C, D]`` (the states a run actually visited), on the midpoints of random pairs of them and on their
    reflection to twice the distance from the sample mean (states the surrogate run may have avoided) -- the guard
    behind ``probe_gaussian``.

    Tolerances are those of fp32 / fp64 ROUNDING of a quadratic form, not a fraction of the log-density: a value may
    differ by ``2e-5 (|quad| + |log_norm|) + 2e-4`` (fp64: ``1e-10 (...) + 1e-9``), a gradient component by ``1e-3``
    (fp64 ``1e-8``) of the gradient scale of the rows.  (Round 2 accepted ``2e-2 (1 + |log p|)``: more than one nat at
    D = 50, enough to hide a unit-height bump on a Gaussian.)"""
    rows = samples.reshape(-1, samples.shape[-1])
    if rows.shape[0] > max_rows:
        idx = torch.linspace(0, rows.shape[0] - 1, max_rows, device=rows.device).long()
        rows = rows[idx]
    rows = rows[torch.isfinite(rows).all(dim=1)]
    if rows.numel() == 0:
        return True
    n = rows.shape[0]
    g = torch.Generator(device="cpu").manual_seed(0x5EED)
    perm = torch.randperm(n, generator=g).to(rows.device)
    k = min(n, 512)
    centre = rows.mean(dim=0, keepdim=True)
    extra = torch.cat([0.5 * (rows[:k] + rows[perm[:k]]), centre + 2.0 * (rows[perm[:k]] - centre)])
    extra = extra[torch.isfinite(extra).all(dim=1)]
    rows = torch.cat([rows, extra.to(rows.dtype)])

    def f(w):
        return log_prob_func(w).sum()
    gr = None
    try:
        with torch.enable_grad():
            gr, v = torch.func.vmap(torch.func.grad_and_value(f))(rows)
    except (RuntimeError, TypeError, ValueError, NotImplementedError, AttributeError, IndexError) as e:
        if isinstance(e, (torch.OutOfMemoryError, torch.AcceleratorError)):
            raise
        rows = rows[torch.linspace(0, rows.shape[0] - 1, min(96, rows.shape[0]), device=rows.device).long()]
        vs, gs = [], []
        for r in rows:
            w = r.detach().clone().requires_grad_(True)
            with torch.enable_grad():
                val = f(w)
                gs.append(torch.autograd.grad(val, w)[0].detach())
            vs.append(val.detach())
        v, gr = torch.stack(vs), torch.stack(gs)
    f32 = rows.dtype == torch.float32
    d = rows.double() - tgt.mean.double()
    Pd = d @ tgt.precision.double()
    quad = 0.5 * (Pd * d).sum(-1)
    want = tgt.log_norm - quad
    v = v.detach().double()
    if not torch.isfinite(v).all():
        return False
    tol_v = (2e-5 if f32 else 1e-10) * (quad.abs() + abs(tgt.log_norm)) + (2e-4 if f32 else 1e-9)
    if not bool(((v - want).abs() <= tol_v).all()):
        return False
    gr = gr.detach().double()
    if not torch.isfinite(gr).all():
        return False
    gscale = float(Pd.abs().max()) + float(tgt.precision.double().abs().max())
    return bool(((gr + Pd).abs().max() <= (1e-3 if f32 else 1e-8) * gscale))

"""GPU parity of the callback compiler (hamiltorch_amd/jit/ + csrc/jit/): an OPAQUE log_prob_func traced, differentiated
and compiled into the fused HMC trajectory kernel, against the oracle on the same Philox streams (chain by chain, SURVEY 8c's
HMC tolerance: 2e-4 end to end in fp32, <= 3 % of chains may flip a Metropolis decision at rounding), against the
torch-evaluated callback path, and through the fall-back rules.  The route is asserted in every test.
"""
import os

import numpy as np
import pytest
import torch

import hmc_oracle as O

pytestmark = pytest.mark.gpu
NP = {torch.float32: np.float32, torch.float64: np.float64}
HL2P = 0.9189385332046727


@pytest.fixture(scope="module")
def ht():
    import hamiltorch_amd
    assert torch.cuda.is_available()
    return hamiltorch_amd


def dev():
    return torch.device("cuda:0")


def tt(a, dtype):
    return None if a is None else torch.tensor(a, dtype=dtype, device=dev())


def funnel_device(w):
    """notebooks/hamiltorch_log_prob_examples.ipynb cell 22 written with device-side arithmetic (= oracle.FunnelTarget, s_i = 1)."""
    v, x = w[0], w[1:]
    ll_v = -v * v / 18.0 - 1.0986122886681098 - HL2P
    ll_x = -0.5 * torch.exp(v) * (x * x).sum() + 0.5 * x.numel() * v - x.numel() * HL2P
    return ll_v + ll_x


def funnel_notebook(w):
    """cell 22 verbatim (torch.distributions)."""
    v_dist = torch.distributions.Normal(0, 3)
    ll = v_dist.log_prob(w[0])
    x_dist = torch.distributions.Normal(0, torch.exp(-w[0]) ** 0.5)
    ll += x_dist.log_prob(w[1:]).sum()
    return ll


def route():
    from hamiltorch_amd import _abi
    return _abi.last_route()


def rand_spd(D, seed, lo=0.5, hi=1.5):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    P = (Q * np.linspace(lo, hi, D)) @ Q.T
    return 0.5 * (P + P.T)


def masses(D, dtype):
    rng = np.random.default_rng(0)
    return {"none": None, "diag": rng.uniform(0.5, 2.0, D).astype(NP[dtype]), "full": rand_spd(D, 7).astype(NP[dtype])}


def start(C, D, seed, dtype, off=0, scale=0.5):
    return (scale * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(NP[dtype])


def compare(got, ref, tol, max_bad):
    got = np.stack([g.cpu().numpy() for g in got]); ref = np.stack(ref)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max(axis=(0, 2))
    bad = ~(err <= tol)
    assert bad.mean() <= max_bad, "%d of %d chains differ (max err %.3g)" % (bad.sum(), bad.size, np.nanmax(err))
    return bad


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float64, 1e-9)])
@pytest.mark.parametrize("fn", [funnel_device, funnel_notebook], ids=["device", "notebook"])
@pytest.mark.parametrize("mass,burn,C", [("none", 0, 200), ("diag", 3, 64), ("full", -1, 33), ("none", 5, 1024)])
def test_compiled_funnel_vs_oracle(ht, dtype, tol, fn, mass, burn, C):
    """The notebook's 11-D funnel through sample(): compiled route, every chain against oracle.sample_hmc on the same draws
    (incl. burn-in bookkeeping, the Q2 reset, mass matrices, a last wave with idle lanes)."""
    D, N, L, eps, seed, off = 11, 14, 12, 0.12, 777, 5
    th0 = start(C, D, seed, dtype, off)
    im = masses(D, dtype)[mass]
    out, acc = ht.sample(fn, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=burn, inv_mass=tt(im, dtype),
                         debug=2, verbose=False, seed=seed, chain_offset=off)
    assert "hta_cb_hmc_kernel<D=11" in route(), route()
    ref, info = O.sample_hmc(O.FunnelTarget(D), th0, N, L, eps, burn, im, O.PhiloxDraws(seed, off + np.arange(C), NP[dtype]))
    assert len(out) == len(ref) == N - max(burn, -1)
    # (fp32 on the funnel: a chain deep in the neck amplifies rounding along its trajectory - the float64 rows pin the algorithm at 1e-9,
    #  the float32 rows allow a tenth of the chains to leave the 2e-4 band)
    bad = compare(out, ref, tol * 5 if mass == "full" else tol, 0.03 if dtype == torch.float64 else 0.10)
    np.testing.assert_allclose(acc.cpu().numpy()[~bad], info["acc_rate"][~bad], atol=1e-12)
    assert 0.5 < float(acc.mean()) <= 1.0


def test_compiled_route_equals_callback_route(ht, monkeypatch):
    """The same callable on the torch-evaluated callback path (HAMILTORCH_AMD_JIT=0: vmap(grad_and_value) + pieces kernels) and
    compiled: the same chains to rounding, the same reject counts."""
    D, C, N, L, eps, seed = 11, 256, 20, 10, 0.15, 31
    th0 = tt(start(C, D, seed, torch.float64), torch.float64)      # (float64: in float32 the funnel's neck amplifies rounding, see above)
    kw = dict(num_samples=N, num_steps_per_sample=L, step_size=eps, burn=2, debug=2, verbose=False, seed=seed)
    a, acc_a = ht.sample(funnel_device, th0, **kw)
    assert "hta_cb_hmc_kernel" in route()
    monkeypatch.setenv("HAMILTORCH_AMD_JIT", "0")
    b, acc_b = ht.sample(funnel_device, th0, **kw)
    assert "hta_cb_hmc_kernel" not in route()
    bad = compare(a, [x.cpu().numpy() for x in b], 1e-8, 0.02)
    assert torch.equal(acc_a.cpu()[~bad], acc_b.cpu()[~bad])


def test_single_chain_contract_and_nuts(ht):
    """(D,) in -> list of (D,) rows, float acceptance; Sampler.HMC_NUTS adapts its step size on the compiled engine exactly as on
    the callback path (one launch per burn-in trajectory, H_old / H_new read back)."""
    init = torch.tensor([0.0] + [1.0] * 10, device=dev())
    out, acc = ht.sample(funnel_device, init, num_samples=40, num_steps_per_sample=8, step_size=0.1, burn=5, debug=2, verbose=False, seed=3)
    assert "hta_cb_hmc_kernel" in route()
    assert isinstance(out, list) and len(out) == 35 and out[0].shape == (11,) and isinstance(acc, float)
    assert torch.equal(out[0], init)
    kw = dict(num_samples=60, num_steps_per_sample=8, step_size=0.3, burn=30, sampler=ht.Sampler.HMC_NUTS, debug=2, verbose=False,
              seed=9, desired_accept_rate=0.7)
    th0 = tt(start(128, 11, 9, torch.float32), torch.float32)
    a, eps_a = ht.sample(funnel_device, th0, **kw)
    assert "hta_cb_hmc_kernel" in route()
    os.environ["HAMILTORCH_AMD_JIT"] = "0"
    try:
        b, eps_b = ht.sample(funnel_device, th0, **kw)
    finally:
        del os.environ["HAMILTORCH_AMD_JIT"]
    assert abs(eps_a - eps_b) <= 0.05 * eps_b, (eps_a, eps_b)          # (the shared step size follows the mean acceptance of 128 fp32 chains)
    assert 0.01 < eps_a < 1.0


def test_other_targets_compiled(ht):
    """Callables with closed-over device tensors, softplus / logsumexp / matrix products / distributions: compiled, and equal to the
    torch-evaluated path."""
    torch.manual_seed(0)
    D = 6
    A = torch.randn(8, D, device=dev())
    y = (torch.rand(8, device=dev()) > 0.5).float()
    mvn = torch.distributions.MultivariateNormal(torch.zeros(D, device=dev()), covariance_matrix=torch.tensor(rand_spd(D, 3), dtype=torch.float32, device=dev()))

    def logistic(w):            # Bayesian logistic regression, closed-over data
        z = A @ w
        return (y * z - torch.nn.functional.softplus(z)).sum() - 0.5 * (w * w).sum()

    def mixture(w):             # log of a two-component mixture + a heavy-tailed term
        a = mvn.log_prob(w)
        b = mvn.log_prob(w - 2.0)
        return torch.logsumexp(torch.stack([a, b]), 0) - torch.log1p((w ** 2).sum() / 5.0)

    th0 = tt(start(160, D, 4, torch.float32), torch.float32)
    for fn in (logistic, mixture):
        kw = dict(num_samples=15, num_steps_per_sample=8, step_size=0.1, verbose=False, seed=12, debug=2)
        a, acc_a = ht.sample(fn, th0, **kw)
        assert "hta_cb_hmc_kernel<D=6" in route(), (fn.__name__, route())
        os.environ["HAMILTORCH_AMD_JIT"] = "0"
        try:
            b, acc_b = ht.sample(fn, th0, **kw)
        finally:
            del os.environ["HAMILTORCH_AMD_JIT"]
        bad = compare(a, [x.cpu().numpy() for x in b], 3e-4, 0.03)
        assert torch.equal(acc_a.cpu()[~bad], acc_b.cpu()[~bad])


def test_fallbacks_keep_the_callback_path_and_say_why(ht):
    """Data-dependent control flow, pass_grad, the tuple protocol and native=False stay on the torch-evaluated path; hta_last_route()
    carries the reason; results are the callback path's."""
    from hamiltorch_amd import jit

    def branchy(w):
        if w[0] > 0:
            return -(w * w).sum()
        return -0.5 * (w * w).sum() - (w ** 4).sum()

    th0 = tt(start(32, 4, 1, torch.float32), torch.float32)
    with pytest.warns(UserWarning, match="not vmap-able"):
        out = ht.sample(branchy, th0, num_samples=5, num_steps_per_sample=3, step_size=0.1, verbose=False, seed=2)
    assert "not compiled" in route() and "control flow" in route(), route()
    assert "control flow" in jit.last_reason()
    assert torch.isfinite(torch.stack(list(out))).all()
    quartic = lambda w: -(w ** 4).sum() - 0.5 * (w * w).sum()  # noqa: E731
    ht.sample(quartic, th0, num_samples=5, num_steps_per_sample=3, step_size=0.1, verbose=False, seed=2)
    assert "hta_cb_hmc_kernel<D=4" in route()
    ht.sample(quartic, th0, num_samples=5, num_steps_per_sample=3, step_size=0.1, verbose=False, seed=2, native=False)
    assert "hta_cb_hmc_kernel" not in route()
    ht.sample(quartic, th0, num_samples=5, num_steps_per_sample=3, step_size=0.1, verbose=False, seed=2,
              pass_grad=lambda w: -4 * w ** 3 - w)
    assert "hta_cb_hmc_kernel" not in route()


def test_stale_trace_is_caught_by_the_check_against_the_callable(ht):
    """A trace is reused while the callable's closure signature is unchanged.  State the signature cannot see (a tensor inside a
    captured object, edited in place) changes the function behind a reused trace: the run-time check against the callable catches it,
    the callable is traced again and the results are those of the NEW function."""
    from hamiltorch_amd import jit

    class Holder:
        pass
    h = Holder()
    h.scale = torch.tensor(1.0, device=dev())
    fn = lambda w: -0.5 * h.scale * (w ** 4).sum() - 0.5 * (w * w).sum()  # noqa: E731
    th0 = tt(start(64, 3, 8, torch.float32), torch.float32)
    kw = dict(num_samples=12, num_steps_per_sample=6, step_size=0.15, verbose=False, seed=5)
    a = torch.stack(list(ht.sample(fn, th0, **kw)))
    traced = jit.stats["traced"]
    a2 = torch.stack(list(ht.sample(fn, th0, **kw)))
    assert jit.stats["traced"] == traced and torch.equal(a, a2)                 # reused, bit-identical
    h.scale.mul_(6.0)                                                           # invisible to the signature
    b = torch.stack(list(ht.sample(fn, th0, **kw)))
    assert jit.stats["traced"] == traced + 1 and "hta_cb_hmc_kernel" in route()
    os.environ["HAMILTORCH_AMD_JIT"] = "0"
    try:
        want = torch.stack(list(ht.sample(fn, th0, **kw)))
    finally:
        del os.environ["HAMILTORCH_AMD_JIT"]
    err = (b - want).abs().amax(dim=(0, 2))
    assert float((err > 2e-4).float().mean()) <= 0.03 and not torch.allclose(a, b)


def test_divergent_chains_are_rejected_and_chunked_runs_are_bit_identical(ht):
    """A step size far too large: non-finite energies reject (S:1045-1057), nothing traps; and a run cut into launches (verbose
    progress: ~20 launches) equals the one-launch run bit for bit (the carried pair is rebuilt at every launch start)."""
    th0 = tt(start(96, 11, 6, torch.float32), torch.float32)
    out, acc = ht.sample(funnel_device, th0, num_samples=8, num_steps_per_sample=30, step_size=40.0, debug=2, verbose=False, seed=3)
    s = torch.stack(list(out))
    assert torch.isfinite(s).all() and float(acc.mean()) < 0.2
    kw = dict(num_samples=45, num_steps_per_sample=7, step_size=0.12, burn=4, seed=13)
    one = torch.stack(list(ht.sample(funnel_device, th0, verbose=False, **kw)))
    many = torch.stack(list(ht.sample(funnel_device, th0, verbose=True, **kw)))
    assert torch.equal(one, many)


def test_funnel_pooled_acceptance_matches_the_oracle(ht):
    """T2 on the funnel (0.950 on the GPU leg against 0.905 on the CPU leg was never pinned): pooled acceptance of the
    bench's configuration (eps = 0.2, L = 25, start (0, 1, .., 1)) at EQUAL trajectory counts within +-0.01 of the oracle's on the
    same streams."""
    C, N, L, eps, seed = 256, 40, 25, 0.2, 1
    th0 = np.ones((C, 11), np.float32); th0[:, 0] = 0.0
    out, acc = ht.sample(funnel_device, tt(th0, torch.float32), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=-1, debug=2,
                         verbose=False, seed=seed)
    assert "hta_cb_hmc_kernel" in route()
    ref, info = O.sample_hmc(O.FunnelTarget(11), th0, N, L, eps, -1, None, O.PhiloxDraws(seed, np.arange(C), np.float32))
    assert abs(float(acc.mean()) - float(info["acc_rate"].mean())) <= 0.01


# ---- compiled derivatives for the Riemannian samplers (csrc/jit/derivs_callback.hip.in) --------------------------------------
def scaled_funnel(scales):
    def f(w):
        s = torch.as_tensor(scales, dtype=w.dtype, device=w.device)
        v, x = w[0], w[1:]
        return -v * v / 18.0 - 1.0986122886681098 - HL2P + (-0.5 * torch.exp(v) * (x * x / s).sum() + 0.5 * x.numel() * v
                                                             - x.numel() * HL2P - 0.5 * torch.log(s).sum())
    return f


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float64, 1e-10)])
def test_compiled_derivatives_vs_oracle_and_torch_func(ht, dtype, tol):
    """log p, gradient, -Hessian and the third-derivative contraction of the compiled funnel: against oracle.FunnelTarget's analytic
    forms and against the torch.func evaluation rmhmc._Curvature performs (the path they replace), fused momentum update included."""
    from hamiltorch_amd import rmhmc
    D, C = 7, 100
    scales = np.array([0.5, 1.0, 1.7, 2.4, 3.3, 0.8])
    fn, o = scaled_funnel(scales), O.FunnelTarget(D, scales)
    th = start(C, D, 3, dtype, scale=0.8)
    M = np.random.default_rng(1).standard_normal((C, D, D)).astype(NP[dtype])
    cv = rmhmc._curvature_for(fn, tt(th, dtype))
    assert isinstance(cv, rmhmc._CompiledCurvature) and "hta_cb_derivs_kernel<D=7" in route()
    tht, Mt = tt(th, dtype), tt(M, dtype)
    g, nH = cv.grad_neg_hessian(tht)
    lp = cv.value(tht)
    assert cv.stats["gh_evaluated"] == 1 and cv.stats["gh_reused"] == 1                  # one launch serves all three
    t64 = th.astype(np.float64)
    scale = lambda a: tol * (1.0 + np.abs(a).max())  # noqa: E731
    np.testing.assert_allclose(lp.cpu().numpy(), o.logp(t64), atol=scale(o.logp(t64)))
    np.testing.assert_allclose(g.cpu().numpy(), o.grad(t64), atol=scale(o.grad(t64)))
    np.testing.assert_allclose(nH.cpu().numpy(), o.neg_hessian(t64), atol=scale(o.neg_hessian(t64)))
    Ms = 0.5 * (M + np.swapaxes(M, 1, 2)).astype(np.float64)
    want_c = o.third_contract(t64, Ms)
    c = cv.contract(tht, Mt)
    assert "hta_cb_contract_kernel" in route()
    np.testing.assert_allclose(c.cpu().numpy(), want_c, atol=scale(want_c))
    upd = torch.ones_like(tht)
    cv.kick_update(tht, Mt, g, upd, 0.25)
    np.testing.assert_allclose(upd.cpu().numpy(), 1.0 + 0.25 * (o.grad(t64) + want_c), atol=scale(want_c))
    ref = rmhmc._Curvature(fn)
    g2, nH2 = ref.grad_neg_hessian(tht)
    np.testing.assert_allclose(nH.cpu().numpy(), nH2.cpu().numpy(), atol=scale(nH2.cpu().numpy()) * 5)
    np.testing.assert_allclose(c.cpu().numpy(), ref.contract(tht, Mt).cpu().numpy(), atol=scale(want_c) * 5)


def test_rmhmc_on_compiled_derivatives_equals_the_torch_func_route(ht):
    """sample(RMHMC, EXPLICIT, SOFTABS) with jitter on the notebook funnel: compiled derivatives against HAMILTORCH_AMD_JIT=0
    (torch.func) on the same Philox streams, float64 - and the implicit integrator likewise."""
    fn = scaled_funnel(np.ones(5))
    th0 = tt(start(24, 6, 2, torch.float64, scale=0.4), torch.float64)
    for integ, extra in ((ht.Integrator.EXPLICIT, dict(explicit_binding_const=10.0)),
                         (ht.Integrator.IMPLICIT, dict(fixed_point_threshold=1e-10, fixed_point_max_iterations=30))):
        kw = dict(num_samples=4, num_steps_per_sample=3, step_size=0.08, jitter=1e-3, softabs_const=1e6, sampler=ht.Sampler.RMHMC,
                  integrator=integ, metric=ht.Metric.SOFTABS, verbose=False, seed=21, **extra)
        a = torch.stack(list(ht.sample(fn, th0, **kw)))
        os.environ["HAMILTORCH_AMD_JIT"] = "0"
        try:
            b = torch.stack(list(ht.sample(fn, th0, **kw)))
        finally:
            del os.environ["HAMILTORCH_AMD_JIT"]
        err = (a - b).abs().amax(dim=(0, 2))
        assert float((err > 1e-6).double().mean()) <= 0.1, (integ, float(err.max()))


# ---- the fused explicit-RMHMC trajectory kernel for small general targets (csrc/jit/rmhmc_callback.hip.in) ----------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-7), (torch.float32, 5e-3)])
@pytest.mark.parametrize("D,scaled,jitter,alpha,burn", [(5, True, None, 1e6, 0), (5, False, 1e-2, 1e6, 2), (11, False, 1e-3, 1.0, -1),
                                                        (3, True, 1e-3, 0.7, 0)])
def test_fused_rmhmc_kernel_vs_oracle(ht, dtype, tol, D, scaled, jitter, alpha, burn):
    """sample(RMHMC, EXPLICIT, SOFTABS) on the funnel, one launch for the run: every chain against oracle.sample_rmhmc_explicit on the
    same Philox streams (momentum by chol(G) z, 8 jitter sub-streams per step in the reference's order, Q1 / Q2 / Q4) - the test the
    launch-per-evaluation route passes (tests/test_gpu_rmhmc.py::test_generic_sample_rmhmc_vs_oracle), same tolerances."""
    scales = np.array([0.5, 1.0, 1.7, 2.4, 3.3, 0.8, 1.2, 2.0, 2.9, 0.6])[:D - 1] if scaled else np.ones(D - 1)
    lp, o = scaled_funnel(scales), O.FunnelTarget(D, scales)
    C, N, L, eps, omega, seed, off = 70, 6, 3, 0.08, 10.0, 99, 7
    th0 = (0.4 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(NP[dtype])
    out, acc = ht.sample(lp, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=burn, jitter=jitter,
                         softabs_const=alpha, explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                         metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=seed, chain_offset=off)
    assert "hta_cb_rmhmc_kernel<D=%d" % D in route(), route()
    with np.errstate(all="ignore"):
        ref, info = O.sample_rmhmc_explicit(o, th0, N, L, eps, omega, alpha, burn, jitter, O.PhiloxDraws(seed, off + np.arange(C), NP[dtype]))
    got = np.stack([x.cpu().numpy() for x in out]); want = np.stack(ref)
    assert got.shape == want.shape
    bad = ~(np.abs(got - want).max(axis=(0, 2)) <= tol)
    assert bad.mean() <= 0.1, "%d of %d chains differ, max err %.3g" % (bad.sum(), C, np.nanmax(np.abs(got - want)))
    np.testing.assert_allclose(acc.cpu().numpy()[~bad], info["acc_rate"][~bad], atol=1e-12)


def test_fused_rmhmc_kernel_equals_the_launch_sequence(ht, monkeypatch):
    """The same run on the launch-per-evaluation route (HAMILTORCH_AMD_JIT=0: torch.func derivatives + hta_metric_eval) and in the
    fused kernel: float64, the notebook's funnel with jitter."""
    fn = scaled_funnel(np.ones(10))
    th0 = tt(start(40, 11, 2, torch.float64, scale=0.4), torch.float64)
    kw = dict(num_samples=4, num_steps_per_sample=4, step_size=0.1, jitter=1e-3, softabs_const=1e6, explicit_binding_const=10.0,
              sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, verbose=False, seed=21)
    a = torch.stack(list(ht.sample(fn, th0, **kw)))
    assert "hta_cb_rmhmc_kernel<D=11,f64,jitter=1" in route()
    monkeypatch.setenv("HAMILTORCH_AMD_JIT", "0")
    b = torch.stack(list(ht.sample(fn, th0, **kw)))
    assert "hta_cb_rmhmc_kernel" not in route()
    err = (a - b).abs().amax(dim=(0, 2))
    assert float((err > 1e-6).double().mean()) <= 0.1, float(err.max())


def test_predrawn_records_equal_the_in_lane_draw(ht, monkeypatch):
    """With few chains the launch's momentum draws and log-uniforms come from hta_cb_predraw_kernel (the whole GPU) and the trajectory kernel
    reads them a trajectory ahead; HAMILTORCH_AMD_JIT_PREDRAW=0 draws in the lane.  Same generator and formulas: a run cut into launches by a
    small record cap is BIT-IDENTICAL to the one-launch run; against the in-lane draw the chains agree to rounding (there the compiler may fuse
    the draw's last multiplication into the first half kick: one rounding fewer) - float64, full mass matrix, burn-in, 200 chains."""
    from hamiltorch_amd import samplers
    D = 11
    th0 = tt(start(200, D, 4, torch.float64), torch.float64)
    im = tt(masses(D, torch.float64)["full"], torch.float64)
    kw = dict(num_samples=30, num_steps_per_sample=6, step_size=0.1, burn=3, inv_mass=im, debug=2, verbose=False, seed=17)
    a, acc_a = ht.sample(funnel_device, th0, **kw)
    assert ",predrawn>" in route(), route()
    monkeypatch.setattr(samplers._CompiledHMC, "PREDRAW_CAP", 7 * 200 * (D + 1) * 8)          # 7 trajectories per launch
    c, acc_c = ht.sample(funnel_device, th0, **kw)
    assert torch.equal(torch.stack(list(a)), torch.stack(list(c))) and torch.equal(acc_a, acc_c)
    monkeypatch.setenv("HAMILTORCH_AMD_JIT_PREDRAW", "0")
    b, acc_b = ht.sample(funnel_device, th0, **kw)
    assert "hta_cb_hmc_kernel" in route() and "predrawn" not in route()
    bad = compare(a, [x.cpu().numpy() for x in b], 1e-9, 0.02)
    assert torch.equal(acc_a.cpu()[~bad], acc_b.cpu()[~bad])

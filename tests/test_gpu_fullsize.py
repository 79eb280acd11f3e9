"""GPU parity AT THE BASELINE SIZES, on the kernel instances bench.py times, chain by chain against the oracle, and the
reference's own recorded full-size runs (tests/golden/cfg2.npz, cfg3.npz, cfg4.npz) fed straight into the C ABI.

Tolerances are SURVEY 8c's: HMC atol = rtol = 1e-5 on a leapfrog path (2e-4 end to end over >= 100 trajectories);
explicit RMHMC at D = 100: 1e-4 on theta / p per path, 1e-3 on H, 5e-4 end to end; a Metropolis decision within rounding of
its threshold may flip, which moves a whole chain: <= 1-3 % of the compared chains may differ.
"""
import numpy as np
import pytest
import torch

import hmc_oracle as O

pytestmark = pytest.mark.gpu

SIGMA3 = np.array([[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]])


@pytest.fixture(scope="module")
def ht():
    import hamiltorch_amd
    assert torch.cuda.is_available()
    return hamiltorch_amd


def dev():
    return torch.device("cuda:0")


def tt(a, dtype=torch.float32):
    return None if a is None else torch.tensor(a, dtype=dtype, device=dev())


def _chain_err(got, ref):
    got = np.asarray(got); ref = np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return np.abs(got - ref).max(axis=(0, 2))


def _replay_reference_run(init, N, propose):
    """The reference's bookkeeping around a trajectory for burn = 0 (S:1000-1026): element 0 of the list is params_init; the
    n = 0 result is kept only as the running state; a rejection at n > 0 resets to the last STORED row (Q2)."""
    rows, cur, rejected = [init.clone()], init.clone(), 0
    for n in range(N):
        new, ok = propose(n, cur)
        if ok:
            cur = new
            if n > 0:
                rows.append(cur.clone())
        else:
            rejected += 1
            if n > 0:
                cur = rows[-1].clone()
                rows.append(cur.clone())
    return torch.cat(rows).cpu().numpy(), 1.0 - rejected / N


# =====================================================================================================================
# cfg2: 3-D Gaussian HMC, 1024 chains, L = 25, eps = 0.3 -- hmc_gauss_quad_kernel<3,false,25>
def _CFG2_ROUTE(prepared=True):
    """The quad kernel's instance for L = 25: on a PREPARED workspace (bench.py, sample()) the fused launch of round 4 (records
    produced inside the trajectory launch, "quad_fused" = 1), else the two-launch form in the variant the route key
    "quad_variant" selects (0: the name without a fourth template argument)."""
    from hamiltorch_amd import _abi
    v = _abi.get_tuning("quad_variant")
    if prepared and _abi.get_tuning("quad_fused") and v == 7:
        return "hmc_gauss_quad_fused_kernel<3,25>"
    return "hmc_gauss_quad_kernel<3,false,25%s>" % (",%d" % v if v else "")


# =====================================================================================================================
def _cfg2(ht):
    P = np.linalg.inv(SIGMA3)
    t = ht.GaussianTarget(torch.zeros(3, device=dev()), covariance=tt(SIGMA3))
    o = O.GaussianTarget(np.zeros(3, np.float32), t.precision.cpu().numpy(), t.log_norm)
    return t, o, P


@pytest.mark.parametrize("prepared", [True, False])
def test_cfg2_bench_instance_vs_oracle_every_chain(ht, prepared):
    """The exact C-ABI call bench.py times (1024 chains, L = 25, burn = -1, prepared workspace: the fused launch; 3 launches over
    traj_offset) against the oracle on the same Philox streams: all 1024 chains x 120 trajectories, and the reject counts.
    Also on an unprepared workspace: the two-launch form (pre-draw pass + trajectory kernel)."""
    from hamiltorch_amd import _abi
    t, o, _ = _cfg2(ht)
    C, T, L, eps, seed, off = 1024, 120, 25, 0.3, 0, 0
    th0 = (0.1 * O.philox_normals(seed, off + np.arange(C), 0, 3, O.PURPOSE_INIT)).astype(np.float32)
    theta0 = tt(th0)
    cur = theta0.clone()
    samples = torch.empty(T + 1, C, 3, device=dev()); samples[0].copy_(theta0)
    rej = torch.zeros(C, dtype=torch.int32, device=dev())
    ws = torch.empty(_abi.gaussian_workspace_bytes(C, 3, 40, 4), dtype=torch.uint8, device=dev())
    if prepared:
        _abi.hmc_gaussian_prepare(theta0, t.precision, 0, None, C, 3, 40, ws)
    for start in (0, 40, 80):
        _abi.hmc_gaussian_sample(cur, theta0, t.precision, t.mean, t.log_norm, 0, None, None, L, eps, 40, start, -1, seed, off,
                                 samples, rej, workspace=ws)
    assert _abi.last_route() == _CFG2_ROUTE(prepared)                        # the instance bench.py times (and names)
    torch.cuda.synchronize()
    if prepared:
        _abi.hmc_gaussian_forget(ws)
    ref, info = O.sample_hmc(o, th0, T, L, eps, -1, None, O.PhiloxDraws(seed, off + np.arange(C)))
    err = _chain_err(samples.cpu().numpy(), np.stack(ref))
    bad = err > 2e-4
    assert bad.mean() <= 0.01, "%d of %d chains differ (max %.3g)" % (bad.sum(), C, err.max())
    want_rej = np.round((1.0 - info["acc_rate"]) * T).astype(np.int64)
    assert np.array_equal(rej.cpu().numpy()[~bad], want_rej[~bad])
    np.testing.assert_allclose(cur.cpu().numpy()[~bad], np.stack(ref)[-1][~bad], atol=2e-4)


def test_cfg2_sample_api_full_size_with_burn_vs_oracle(ht):
    """hamiltorch_amd.sample() at cfg2's size with burn-in (Q2 reset, Q3 acceptance-rate convention): 1024 chains x 130
    trajectories (burn 20) chain by chain, through the API the bench's `api_ms_per_step` times."""
    t, o, _ = _cfg2(ht)
    C, N, L, eps, burn, seed, off = 1024, 130, 25, 0.3, 20, 11, 4096
    th0 = (0.1 * O.philox_normals(seed, off + np.arange(C), 0, 3, O.PURPOSE_INIT)).astype(np.float32)
    out, acc = ht.sample(t, tt(th0), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=burn, debug=2, verbose=False,
                         seed=seed, chain_offset=off)
    ref, info = O.sample_hmc(o, th0, N, L, eps, burn, None, O.PhiloxDraws(seed, off + np.arange(C)))
    from hamiltorch_amd import _abi
    assert _abi.last_route() == _CFG2_ROUTE()
    assert len(out) == len(ref) == N - burn
    err = _chain_err(torch.stack(out).cpu().numpy(), np.stack(ref))
    bad = err > 2e-4
    assert bad.mean() <= 0.01, "%d of %d chains differ (max %.3g)" % (bad.sum(), C, err.max())
    np.testing.assert_allclose(acc.cpu().numpy()[~bad], info["acc_rate"][~bad], atol=1e-12)


@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 1e-5), ("f64", torch.float64, 1e-12)])
def test_cfg2_reference_paths_through_c_abi(ht, golden, tag, dtype, tol):
    """tests/golden/cfg2.npz: every step of the reference's four 25-step leapfrog paths (its own theta0, p0) reproduced by
    hta_hmc_gaussian_leapfrog at SURVEY 8c's HMC tolerance."""
    from hamiltorch_amd import _abi
    g = golden("cfg2")
    t = ht.GaussianTarget(torch.zeros(3, dtype=dtype, device=dev()), covariance=tt(SIGMA3, dtype))
    th, pm = tt(g["theta0"], dtype), tt(g["p0"], dtype)
    pt = torch.empty(25, 4, 3, dtype=dtype, device=dev()); pp = torch.empty_like(pt)
    _abi.hmc_gaussian_leapfrog(th.clone(), pm.clone(), t.precision, t.mean, 0, None, 25, 0.3, pt, pp)
    np.testing.assert_allclose(pt.permute(1, 0, 2).cpu().numpy(), g[f"lf_theta_{tag}"], rtol=tol, atol=tol)
    np.testing.assert_allclose(pp.permute(1, 0, 2).cpu().numpy(), g[f"lf_p_{tag}"], rtol=tol, atol=tol)


def test_cfg2_reference_run_replayed_through_c_abi(ht, golden):
    """The reference's own 40-trajectory sample() run at cfg2 (recorded momenta and Metropolis uniforms) replayed through the
    C-ABI pieces: hta_hamiltonian -> hta_hmc_gaussian_leapfrog (25 steps) -> hta_hamiltonian, decision from the recorded
    uniform (S:1000-1004): every returned sample of the reference, and its acceptance rate."""
    from hamiltorch_amd import _abi
    g = golden("cfg2")
    t, _, _ = _cfg2(ht)
    cur = tt(g["theta0"][0][None])
    logp = lambda th: (t.log_norm - 0.5 * ((th - t.mean) @ t.precision * (th - t.mean)).sum(-1)).contiguous()  # noqa: E731
    H0 = torch.empty(1, device=dev()); H1 = torch.empty(1, device=dev())
    N = g["e2e_samples"].shape[0]

    def propose(n, cur):
        p = tt(g["e2e_momenta"][n][None])
        _abi.hamiltonian(p, logp(cur), 0, None, H0)
        th = cur.clone()
        _abi.hmc_gaussian_leapfrog(th, p, t.precision, t.mean, 0, None, 25, 0.3)
        _abi.hamiltonian(p, logp(th), 0, None, H1)
        return th, min(0.0, float(H0 - H1)) >= float(np.log(g["e2e_uniforms"][n]))

    got, acc = _replay_reference_run(cur, N, propose)
    np.testing.assert_allclose(got, g["e2e_samples"], rtol=1e-4, atol=1e-4)
    assert abs(acc - float(g["e2e_acc"])) < 1e-9


# =====================================================================================================================
# cfg3 / cfg5: D = 100 explicit RMHMC, soft-abs, L = 10, jitter 1e-3
# =====================================================================================================================
def _cfg3(ht, dtype=torch.float32):
    g = torch.Generator().manual_seed(0)
    Q = torch.linalg.qr(torch.randn(100, 100, generator=g, dtype=torch.float64))[0]
    P = (Q * torch.linspace(0.5, 2.0, 100, dtype=torch.float64)) @ Q.T
    P = (0.5 * (P + P.T)).numpy()
    npdt = np.float32 if dtype == torch.float32 else np.float64
    t = ht.GaussianTarget(torch.zeros(100, dtype=dtype, device=dev()), precision=tt(P, dtype), normalized=False)
    o = O.GaussianTarget(np.zeros(100, npdt), P.astype(npdt), 0.0)
    return t, o


@pytest.mark.parametrize("C,N,nsel,route", [(256, 5, 32, "rmhmc_uvc_kernel<co>"), (512, 3, 32, "rmhmc_uvc2_kernel<co>"),
                                            (1024, 3, 128, "rmhmc_uvc2_kernel<co>"), (2048, 3, 32, "rmhmc_mfma4_kernel<true>"),
                                            (4096, 3, 32, "rmhmc_batch_kernel<25,true>")])
def test_cfg3_bench_instances_vs_oracle_L10(ht, C, N, nsel, route):
    """The trajectory kernels bench.py times for cfg3 (256 chains), the north-star 1024-chain size and cfg5's 4096-chain
    route, at the BASELINE step count L = 10 with jitter 1e-3: 32 chains spread over the whole batch (first / last
    workgroups included) against the oracle -- which does the reference's eigendecomposition per metric evaluation
    (S:108-122) while the kernels use the shared-inverse form -- on the same Philox streams."""
    D, L, eps, omega, alpha, jitter, seed, off = 100, 10, 0.1, 10.0, 1e6, 1e-3, 2026, 7
    t, o = _cfg3(ht)
    th0 = (0.1 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    out, acc = ht.sample(t, tt(th0), num_samples=N, num_steps_per_sample=L, step_size=eps, jitter=jitter, softabs_const=alpha,
                         explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                         metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=seed, chain_offset=off)
    from hamiltorch_amd import _abi
    if _abi.get_tuning("rmhmc_lean") and route.startswith(("rmhmc_uv_kernel", "rmhmc_mfma4x4_kernel")):
        route = route[:-1] + ",lean>"                              # the instance the route key "rmhmc_lean" selects (the default)
    assert _abi.last_route() == route, _abi.last_route()          # the dispatch at this chain count IS the kernel named here
    got = torch.stack(out).cpu().numpy()
    assert got.shape == (N, C, D) and np.isfinite(got).all()
    sel = np.unique(np.r_[0:4, np.linspace(4, C - 5, nsel - 8).astype(int), C - 4:C])
    ref, info = O.sample_rmhmc_explicit(o, th0[sel], N, L, eps, omega, alpha, 0, jitter,
                                        O.PhiloxDraws(seed, off + sel, np.float32), "softabs", momentum="split")
    err = _chain_err(got[:, sel], np.stack(ref))
    bad = err > 5e-4
    assert bad.sum() <= max(1, len(sel) // 64), "%s: %d of %d chains differ (max %.3g)" % (route, bad.sum(), len(sel), err.max())
    np.testing.assert_allclose(acc.cpu().numpy()[sel][~bad], info["acc_rate"][~bad], atol=1e-12)
    assert np.abs(got[-1] - got[0]).mean() > 1e-2           # the chains moved


def test_cfg3_eigendecomposition_route_bench_instance_vs_oracle_L10(ht):
    """The instance bench.py times as cfg3-eig - BASELINE config 3 (256 chains, D = 100, L = 10, jitter 1e-3) WITH an eigendecomposition per
    metric evaluation ("rmhmc_fused" = 0: metric_traj_mfma_kernel, one launch per trajectory - resident state, bfloat16 split products, the
    draw p = G^(1/2) z, the Metropolis selection inside the launch) - against the oracle's eigh per evaluation (S:108-122) on the same Philox
    streams: 24 chains spread over the batch, samples and acceptance."""
    from hamiltorch_amd import _abi
    D, C, N, L, eps, omega, alpha, jitter, seed, off = 100, 256, 3, 10, 0.1, 10.0, 1e6, 1e-3, 2026, 7
    t, o = _cfg3(ht)
    th0 = (0.1 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    _abi.set_tuning("rmhmc_fused", 0)
    try:
        out, acc = ht.sample(t, tt(th0), num_samples=N, num_steps_per_sample=L, step_size=eps, jitter=jitter, softabs_const=alpha,
                             explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                             metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=seed, chain_offset=off)
        route = _abi.last_route()
        sqrtdraw = _abi.get_tuning("metric_sqrtdraw")
    finally:
        _abi.set_tuning("rmhmc_fused", 1)
    assert route == "metric_traj_mfma_kernel", route
    got = torch.stack(out).cpu().numpy()
    assert got.shape == (N, C, D) and np.isfinite(got).all()
    sel = np.unique(np.r_[0:4, np.linspace(4, C - 5, 16).astype(int), C - 4:C])
    ref, info = O.sample_rmhmc_explicit(o, th0[sel], N, L, eps, omega, alpha, 0, jitter,
                                        O.PhiloxDraws(seed, off + sel, np.float32), "softabs", momentum="sqrt" if sqrtdraw else "chol")
    err = _chain_err(got[:, sel], np.stack(ref))
    bad = err > 5e-4
    assert bad.sum() <= 1, "%d of %d chains differ (max %.3g)" % (bad.sum(), len(sel), err.max())
    np.testing.assert_allclose(acc.cpu().numpy()[sel][~bad], info["acc_rate"][~bad], atol=1e-12)
    assert np.abs(got[-1] - got[0]).mean() > 1e-2


def test_cfg3_reference_fixture_through_c_abi(ht, golden):
    """tests/golden/cfg3.npz (the unmodified reference at D = 100, fp32): soft-abs spectrum, diag G, G^-1 p, the Riemannian
    Hamiltonian and every step of its 3-step explicit path through the C ABI (fisher / cholesky_inverse / rm_hamiltonian /
    leapfrog of the API mirror)."""
    from hamiltorch_amd import rmhmc
    g = golden("cfg3")
    D, alpha, omega, eps, _ = g["cfg"]
    t = ht.GaussianTarget(torch.zeros(100, device=dev()), precision=tt(g["P_f32"]), normalized=False)
    th, pm = tt(g["theta0_f32"]), tt(g["p0_f32"])
    G, lam = ht.samplers.fisher(th, t, softabs_const=alpha, metric=ht.Metric.SOFTABS)
    np.testing.assert_allclose(np.sort(lam.cpu().numpy()), np.sort(g["lam_f32"]), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(torch.diagonal(G).cpu().numpy(), g["Gdiag_f32"], rtol=1e-4, atol=1e-4)
    x = ht.samplers.cholesky_inverse(G, pm)
    np.testing.assert_allclose(x.reshape(-1).cpu().numpy(), g["Ginvp_f32"], rtol=1e-4, atol=1e-4)
    H = ht.samplers.rm_hamiltonian(th, pm, t, None, 1.0, softabs_const=alpha, metric=ht.Metric.SOFTABS)
    np.testing.assert_allclose(H.reshape(-1).cpu().numpy(), g["H_f32"], rtol=0, atol=1e-3)
    lt, lp = ht.samplers.leapfrog(th, pm, t, steps=3, step_size=float(eps), jitter=None, softabs_const=float(alpha),
                                  explicit_binding_const=float(omega), sampler=ht.Sampler.RMHMC,
                                  integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS)
    np.testing.assert_allclose(torch.stack(lt[0]).cpu().numpy(), g["lf_theta_f32"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(torch.stack(lp[0]).cpu().numpy(), g["lf_p_f32"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(lt[1].cpu().numpy(), g["lf_thetac_f32"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(lp[1].cpu().numpy(), g["lf_pc_f32"], rtol=0, atol=1e-4)
    assert rmhmc is not None


class _RecordedJitterCurvature:
    """Stands in for rmhmc._Curvature on a Gaussian target: gradient -P theta, and the metric input `P + diag(jitter u_k)`
    with u_k the reference's k-th recorded torch.rand(D) draw (S:113-115), in call order -- so that hta_metric_eval sees exactly
    the matrix the reference eigendecomposed."""

    def __init__(self, P, draws, jitter):
        self.P, self.draws, self.jitter, self.k = P, draws, jitter, 0

    def grad_neg_hessian(self, theta):
        Hs = self.P + torch.diag(self.jitter * self.draws[self.k])
        self.k += 1
        return -(theta @ self.P), Hs.expand(theta.shape[0], -1, -1).contiguous()

    def contract(self, theta, M):
        return torch.zeros_like(theta)                      # third derivatives of a quadratic form

    def touched(self, *tensors):                            # (rmhmc._Curvature's cache invalidation: nothing is cached here)
        pass

    def value(self, theta):
        return -0.5 * ((theta @ self.P) * theta).sum(-1)


def test_cfg3_reference_jitter_draws_replayed_through_c_abi(ht, golden):
    """cfg3 with jitter = 1e-3: the reference's recorded jitter draws (8 per step, S:115) replayed in call order through
    hta_metric_eval / hta_rmhmc_binding_rotation (the eigendecomposition route) reproduce its 2-step path at D = 100."""
    from hamiltorch_amd import rmhmc, _abi
    g = golden("cfg3")
    D, alpha, omega, eps, jitter = g["cfg"]
    P = tt(g["P_f32"])
    for steps in (1, 2):
        cv = _RecordedJitterCurvature(P, tt(g["jit_draws"]), float(jitter))
        th, pm = tt(g["theta0_f32"][None]), tt(g["p0_f32"][None])
        thc, pmc = th.clone(), pm.clone()
        rmhmc._generic_steps(cv, _abi.METRIC_SOFTABS, th, pm, thc, pmc, steps, float(eps), float(omega), float(alpha), None, 0, 0, 0)
        assert cv.k == 8 * steps
        np.testing.assert_allclose(th[0].cpu().numpy(), g["jit_lf_theta"][steps - 1], rtol=0, atol=1e-4)
        np.testing.assert_allclose(pm[0].cpu().numpy(), g["jit_lf_p"][steps - 1], rtol=0, atol=1e-4)


def test_cfg3_reference_run_replayed_through_c_abi(ht, golden):
    """The reference's end-to-end explicit-RMHMC sample() at D = 100 with jitter (recorded momenta, uniforms and all 8L + 3
    jitter vectors per trajectory) replayed through the C-ABI pieces: H (sub-stream 1), L explicit steps (2 .. 8L+1), H on the
    un-augmented pair (8L + 2, Q4), decision from the recorded uniform."""
    from hamiltorch_amd import rmhmc, _abi
    g = golden("cfg3")
    D, alpha, omega, eps, jitter = (float(v) for v in g["cfg"])
    N, L = (int(v) for v in g["e2e_cfg"])
    per = 8 * L + 3
    P = tt(g["P_f32"])
    jit = tt(g["e2e_jitters"])
    cur = tt(g["theta0_f32"][None])
    D = 100
    H0 = torch.empty(1, device=dev()); H1 = torch.empty(1, device=dev())

    def ham(theta, p, u, out):
        Hs = (P + torch.diag(jitter * u)).reshape(1, D, D).contiguous()
        _abi.metric_eval(theta, 1, D, _abi.METRIC_SOFTABS, Hs, D * D, alpha, m=p, H_out=out)
        out.add_(0.5 * ((theta @ P) * theta).sum(-1))          # - log p

    def propose(n, cur):
        base = n * per
        p = tt(g["e2e_momenta"][n][None])
        ham(cur, p, jit[base + 1], H0)
        cv = _RecordedJitterCurvature(P, jit[base + 2: base + 2 + 8 * L], jitter)
        th, pm = cur.clone(), p.clone()
        thc, pmc = th.clone(), pm.clone()
        rmhmc._generic_steps(cv, _abi.METRIC_SOFTABS, th, pm, thc, pmc, L, eps, omega, alpha, None, 0, 0, 0)
        ham(th, pm, jit[base + 2 + 8 * L], H1)
        return th, min(0.0, float(H0 - H1)) >= float(np.log(g["e2e_uniforms"][n]))

    got, acc = _replay_reference_run(cur, N, propose)
    np.testing.assert_allclose(got, g["e2e_samples"], rtol=0, atol=3e-4)
    assert abs(acc - float(g["e2e_acc"])) < 1e-9


# =====================================================================================================================
# cfg4: Bayesian MLP 8-100-1, 400 points, M = 4 symmetric split HMC, 512 chains -- mlp_mfma_kernel
# =====================================================================================================================
def _cfg4_data():
    g = torch.Generator().manual_seed(0)
    X = torch.randn(400, 8, generator=g); w = torch.randn(8, 1, generator=g)
    Y = torch.sin(X @ w) + 0.1 * torch.randn(400, 1, generator=g)
    return X, Y


def test_cfg4_bench_instance_vs_oracle_full_size(ht):
    """sample_split_model at BASELINE config 4's full size (512 chains, D = 1001, M = 4 x 100 points, tau_out = 100,
    eps = 5e-4, L = 10; the MFMA kernel instance bench.py times): 6 trajectories, 64 chains spread over the batch against the
    oracle on the same Philox streams, plus the acceptance rates."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 100), torch.nn.ReLU(), torch.nn.Linear(100, 1)).to(dev())
    X, Y = _cfg4_data()
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=100, shuffle=False)
    C, D, N, L, eps, seed = 512, 1001, 6, 10, 5e-4, 5
    flat = ht.util.flatten(net).detach().cpu().numpy()
    th0 = (flat[None] + 0.02 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    out, acc = ht.sample_split_model(net, loader, tt(th0), 4, model_loss="regression", num_samples=N, num_steps_per_sample=L,
                                     step_size=eps, inv_mass=torch.ones(D, device=dev()), tau_out=100.0, tau_list=torch.ones(4),
                                     debug=2, verbose=False, seed=seed)
    from hamiltorch_amd import _abi
    assert _abi.last_route() == "mlp_mfma_kernel<2,7,0,512>", _abi.last_route()          # the instance bench.py times
    got = torch.stack(out).cpu().numpy()
    assert got.shape == (N, C, D) and np.isfinite(got).all()
    sel = np.unique(np.r_[0:2, np.arange(2, C - 2, 8), C - 2:C])
    otg = [O.MLPRegressionTarget([8, 100, 1], X.numpy()[m * 100:(m + 1) * 100], Y.numpy()[m * 100:(m + 1) * 100], np.ones(4), 100.0, 4, "relu")
           for m in range(4)]
    ref, info = O.sample_hmc(None, th0[sel], N, L, eps, 0, np.ones(D, np.float32), O.PhiloxDraws(seed, sel),
                             grad_fns=[t_.grad for t_ in otg], logp_fns=[t_.logp for t_ in otg])
    err = _chain_err(got[:, sel], np.stack(ref))
    bad = err > 5e-4
    assert bad.mean() <= 0.05, "%d of %d chains differ (max %.3g)" % (bad.sum(), len(sel), err.max())
    np.testing.assert_allclose(acc.cpu().numpy()[sel][~bad], info["acc_rate"][~bad], atol=1e-12)


def test_cfg4_reference_fixture_through_c_abi(ht, golden):
    """tests/golden/cfg4.npz (the unmodified reference at cfg4's size): log-prob and gradient of the full-data closure and
    the four split closures from hta_mlp_logp_grad, H0, and its 10-step SPLITTING path replayed from its own (theta, p0)
    with the native gradient kernel doing every half kick (S:499-540) and hta_kick_drift the updates."""
    from hamiltorch_amd import _abi
    g = golden("cfg4")
    n = "relu_cfg4"
    M, tau_out, eps, L = g[f"{n}_cfg"]
    M, L, eps, tau_out = int(M), int(L), float(eps), float(tau_out)
    X = tt(g[f"{n}_X"]); Y = tt(g[f"{n}_Y"].reshape(-1))
    tau = [float(v) for v in g[f"{n}_tau_list"]]
    th = tt(g[f"{n}_theta"][None])
    grad = torch.empty_like(th); lp = torch.empty(1, device=dev())
    _abi.mlp_logp_grad(th, 8, 100, "relu", X, Y, 1, 400, 0, tau, tau_out, 1.0, grad, lp)
    np.testing.assert_allclose(lp.cpu().numpy(), g[f"{n}_logp"], rtol=2e-5)
    np.testing.assert_allclose(grad[0].cpu().numpy(), g[f"{n}_grad"], rtol=3e-4, atol=3e-4 * float(np.abs(g[f"{n}_grad"]).max()))
    tot = 0.0
    for m in range(M):
        _abi.mlp_logp_grad(th, 8, 100, "relu", X, Y, M, 400 // M, m, tau, tau_out, float(M), grad, lp)
        np.testing.assert_allclose(float(lp), g[f"{n}_split_logp"][m], rtol=2e-5)
        tot += float(lp)
    p = tt(g[f"{n}_p0"][None])
    H0 = torch.empty(1, device=dev())
    _abi.hamiltonian(p, torch.tensor([tot], device=dev()), _abi.MASS_DIAG, torch.ones(1001, device=dev()), H0)
    np.testing.assert_allclose(H0.cpu().numpy(), g[f"{n}_H0"], rtol=2e-5)
    im = torch.ones(1001, device=dev())
    dq = eps / (2 * (M - 1))

    def total_logp(theta):
        tot = torch.zeros(1, device=dev())
        for m in range(M):
            _abi.mlp_logp_grad(theta, 8, 100, "relu", X, Y, M, 400 // M, m, tau, tau_out, float(M), grad, lp)
            tot += lp
        return tot

    def split_steps(theta, mom):
        for _ in range(L):
            for m in range(M):
                _abi.mlp_logp_grad(theta, 8, 100, "relu", X, Y, M, 400 // M, m, tau, tau_out, float(M), grad, lp)
                _abi.kick_drift(theta, mom, grad, 0.5 * eps, dq if m < M - 1 else 0.0, _abi.MASS_DIAG, im)
            for m in reversed(range(M)):
                _abi.mlp_logp_grad(theta, 8, 100, "relu", X, Y, M, 400 // M, m, tau, tau_out, float(M), grad, lp)
                _abi.kick_drift(theta, mom, grad, 0.5 * eps, dq if m > 0 else 0.0, _abi.MASS_DIAG, im)

    th = th.clone(); p = p.clone()
    split_steps(th, p)
    np.testing.assert_allclose(th[0].cpu().numpy(), g[f"{n}_lf_theta"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(p[0].cpu().numpy(), g[f"{n}_lf_p"], rtol=0, atol=2e-3 * float(np.abs(g[f"{n}_lf_p"]).max()))
    # the reference's own sample_split_model run (recorded momenta / uniforms) replayed with the native gradient kernel
    cur = tt(g[f"{n}_theta"][None])
    H1 = torch.empty(1, device=dev())
    N = g[f"{n}_e2e_samples"].shape[0]

    def propose(k, cur):
        p = tt(g[f"{n}_e2e_momenta"][k][None])
        _abi.hamiltonian(p, total_logp(cur), _abi.MASS_DIAG, im, H0)
        th = cur.clone()
        split_steps(th, p)
        _abi.hamiltonian(p, total_logp(th), _abi.MASS_DIAG, im, H1)
        return th, min(0.0, float(H0 - H1)) >= float(np.log(g[f"{n}_e2e_uniforms"][k]))

    got, acc = _replay_reference_run(cur, N, propose)
    np.testing.assert_allclose(got, g[f"{n}_e2e_samples"], rtol=0, atol=5e-5)
    assert abs(acc - float(g[f"{n}_e2e_acc"])) < 1e-9

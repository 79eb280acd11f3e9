"""GPU parity for the Riemannian path: batched metric evaluation (Jacobi eigh, soft-abs, solves,
log-determinant, Cholesky), Riemannian Hamiltonian, momentum draw, explicit integrator and the
RMHMC branch of sample(), against the oracle and the fixtures recorded from the reference.

Tolerances: D=100 fp32 theta/p atol 1e-4, H 1e-3; small D fp32 3e-4 relative; fp64 1e-9."""
import numpy as np
import pytest
import torch

import hmc_oracle as O

pytestmark = pytest.mark.gpu
MFMA4_DEFAULT = 1      # the library's default for the "rmhmc_mfma4" tuning key (csrc/abi.cpp)
NP = {torch.float32: np.float32, torch.float64: np.float64}


@pytest.fixture(scope="module")
def ht():
    import hamiltorch_amd
    assert torch.cuda.is_available()
    return hamiltorch_amd


def dev():
    return torch.device("cuda:0")


def tt(a, dtype):
    return None if a is None else torch.tensor(np.asarray(a), dtype=dtype, device=dev())


def oracle_momentum(jitter, metric="softabs"):
    """Which momentum draw the oracle must replay for the sample call that just returned: the fused Gaussian routes
    (kernel names rmhmc_*) draw p = chol(P) z1 + sqrt(jitter u) . z2 when jitter is on (tuning key "rmhmc_momsplit",
    default 1; oracle: rm_gibbs_split, same law as chol(G) z); the eigendecomposition route on the matrix cores draws p = G^(1/2) z
    (tuning key "metric_sqrtdraw", default 1; oracle: rm_gibbs_sqrt, same law); every other route and jitter-free fused runs draw chol(G) z."""
    from hamiltorch_amd import _abi
    route = _abi.last_route()
    fused = route.startswith("rmhmc_")
    if route in ("metric_traj_mfma_kernel", "metric_warm_mfma_kernel") and _abi.get_tuning("metric_sqrtdraw") and metric == "softabs":
        return "sqrt"        # round 6: the matrix-core metric kernel draws p = G^(1/2) z for a Gaussian target's soft-abs metric (oracle: rm_gibbs_sqrt)
    return "split" if (jitter is not None and fused and _abi.get_tuning("rmhmc_momsplit")) else "chol"


def sym_batch(B, D, kind, seed):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(B):
        Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
        if kind == "spd":
            lam = rng.uniform(0.5, 2.0, D)
        elif kind == "indef":
            lam = rng.uniform(0.3, 2.0, D) * rng.choice([-1.0, 1.0], D)
        elif kind == "degenerate":
            lam = np.repeat(rng.uniform(0.5, 2.0, (D + 2) // 3), 3)[:D]
        else:
            raise ValueError(kind)
        A = (Q * lam) @ Q.T
        out.append(0.5 * (A + A.T))
    return np.stack(out)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 4e-4), (torch.float64, 2e-9)])
@pytest.mark.parametrize("D,kind,alpha", [(1, "spd", 1e6), (2, "indef", 1.3), (3, "spd", 1e6), (5, "indef", 2.0),
                                           (8, "degenerate", 1e6), (17, "indef", 0.7), (33, "spd", 1e6),
                                           (64, "indef", 1e6), (100, "spd", 1e6), (101, "degenerate", 3.0)])
def test_metric_eval_vs_oracle(ht, dtype, tol, D, kind, alpha):
    """fisher(): G, soft-abs eigenvalues; cholesky_inverse(): G^-1 m; log|G| and m^T G^-1 m -- batched, arbitrary symmetric Hs."""
    from hamiltorch_amd import _abi
    B = 5
    Hs = sym_batch(B, D, kind, D).astype(NP[dtype])
    rng = np.random.default_rng(1)
    m = rng.standard_normal((B, D)).astype(NP[dtype])
    G, lam, _ = O.softabs_metric(Hs.astype(np.float64), alpha)
    x64 = np.linalg.solve(G, m.astype(np.float64)[..., None])[..., 0]
    t = tt(Hs, dtype)
    Gd = torch.empty(B, D, D, dtype=dtype, device=dev()); lamd = torch.empty(B, D, dtype=dtype, device=dev())
    xd = torch.empty(B, D, dtype=dtype, device=dev()); ld = torch.empty(B, dtype=dtype, device=dev()); qd = torch.empty(B, dtype=dtype, device=dev())
    Vd = torch.empty(B, D, D, dtype=dtype, device=dev()); Ld = torch.empty(B, D, D, dtype=dtype, device=dev())
    _abi.metric_eval(t, B, D, _abi.METRIC_SOFTABS, t, D * D, alpha, m=tt(m, dtype), x_out=xd, G_out=Gd, lam_out=lamd,
                     V_out=Vd, L_out=Ld, logdet_out=ld, quad_out=qd)
    scale = np.abs(G).max()
    np.testing.assert_allclose(Gd.cpu().numpy(), G, rtol=tol, atol=tol * scale)
    np.testing.assert_allclose(np.sort(lamd.cpu().numpy(), axis=1), np.sort(lam, axis=1), rtol=tol, atol=tol * scale)
    cond = np.abs(lam).max() / np.abs(lam).min()
    np.testing.assert_allclose(xd.cpu().numpy(), x64, rtol=tol * cond, atol=tol * cond * np.abs(x64).max())
    np.testing.assert_allclose(ld.cpu().numpy(), np.log(lam).sum(1), rtol=tol, atol=tol * D)
    np.testing.assert_allclose(qd.cpu().numpy(), (m * x64).sum(1), rtol=tol * cond, atol=tol * cond)
    V = Vd.double().cpu().numpy()
    np.testing.assert_allclose(np.einsum("bij,bik->bjk", V, V), np.broadcast_to(np.eye(D), (B, D, D)), atol=50 * tol / 4e-4 * 1e-6 if dtype == torch.float32 else 1e-12)
    Lc = Ld.double().cpu().numpy()
    np.testing.assert_allclose(Lc @ np.swapaxes(Lc, 1, 2), G, rtol=4 * tol, atol=4 * tol * scale)
    # HESSIAN metric = Cholesky path, on the SPD matrix G itself (cholesky_inverse API)
    x2 = ht.samplers.cholesky_inverse(tt(G, dtype), tt(m, dtype))
    np.testing.assert_allclose(x2.cpu().numpy(), x64, rtol=tol * cond, atol=tol * cond * np.abs(x64).max())
    x1 = ht.samplers.cholesky_inverse(tt(G[0], dtype), tt(m[0], dtype))
    assert x1.shape == (D, 1)


@pytest.mark.parametrize("name", ["d3", "d10", "d6indef"])
@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 3e-4), ("f64", torch.float64, 1e-9)])
def test_reference_fixtures(ht, golden, name, tag, dtype, tol):
    """fisher / cholesky_inverse / rm_hamiltonian / explicit leapfrog against values recorded from the reference."""
    g = golden("rmhmc")
    alpha, omega, eps, steps = g[f"{name}_cfg"]
    P = tt(g[f"{name}_P_{tag}"], dtype)
    D = P.shape[0]
    tgt = ht.GaussianTarget(torch.zeros(D, dtype=dtype, device=dev()), precision=P, normalized=False)
    th = tt(g[f"{name}_theta0_{tag}"], dtype); pm = tt(g[f"{name}_p0_{tag}"], dtype)
    for mtag, metric in (("softabs", ht.Metric.SOFTABS), ("hessian", ht.Metric.HESSIAN)):
        if f"{name}_G_{mtag}_{tag}" not in g.files:
            continue
        for lp in (tgt, (lambda w, t=tgt: t(w))):       # plugin and opaque callback (torch.func.hessian) routes
            G, lam = ht.samplers.fisher(th, lp, jitter=None, softabs_const=alpha, metric=metric)
            np.testing.assert_allclose(G.cpu().numpy(), g[f"{name}_G_{mtag}_{tag}"], rtol=tol, atol=tol)
            if mtag == "softabs":
                np.testing.assert_allclose(np.sort(lam.cpu().numpy()), np.sort(g[f"{name}_lam_{mtag}_{tag}"]), rtol=tol, atol=tol)
            else:
                assert lam is None
            H = ht.samplers.rm_hamiltonian(th, pm, lp, None, 1.0, softabs_const=alpha, metric=metric)
            assert H.shape == (1, 1)
            np.testing.assert_allclose(H.cpu().numpy().reshape(-1), g[f"{name}_H_{mtag}_{tag}"], rtol=tol, atol=tol)
        x = ht.samplers.cholesky_inverse(G, pm)
        np.testing.assert_allclose(x.cpu().numpy().reshape(-1), g[f"{name}_Ginvp_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)
        lpar, lmom = ht.samplers.leapfrog(th, pm, tgt, steps=int(steps), step_size=float(eps), jitter=None,
                                          explicit_binding_const=float(omega), softabs_const=float(alpha),
                                          sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=metric)
        assert len(lpar[0]) == int(steps) and len(lmom[0]) == int(steps)
        np.testing.assert_allclose(lpar[0][-1].cpu().numpy(), g[f"{name}_lf_theta_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)
        np.testing.assert_allclose(lmom[0][-1].cpu().numpy(), g[f"{name}_lf_p_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)
        np.testing.assert_allclose(lpar[1].cpu().numpy(), g[f"{name}_lf_thetac_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)
        np.testing.assert_allclose(lmom[1].cpu().numpy(), g[f"{name}_lf_pc_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)


def cfg3_target(ht, D, dtype, seed=0):
    """P = Q diag(linspace(.5, 2, D)) Q^T, log p = -1/2 w^T P w."""
    g = torch.Generator().manual_seed(seed)
    Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
    P = (Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T
    P = (0.5 * (P + P.T)).numpy()
    t = ht.GaussianTarget(torch.zeros(D, dtype=dtype, device=dev()), precision=tt(P, dtype), normalized=False)
    o = O.GaussianTarget(np.zeros(D, NP[dtype]), P.astype(NP[dtype]), 0.0)
    return t, o


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float64, 1e-9)])
@pytest.mark.parametrize("D,jitter", [(4, None), (10, 1e-3), (31, None), (100, None), (100, 1e-3)])
def test_explicit_leapfrog_and_hamiltonian_vs_oracle(ht, dtype, tol, D, jitter):
    """T1 at cfg3's shape: same (theta, p) and the same Philox jitter stream into kernel sequence and oracle."""
    t, o = cfg3_target(ht, D, dtype)
    C, steps, eps, omega, alpha, seed, off, n = 6, 3, 0.1, 10.0, 1e6, 77, 40, 9
    rng = np.random.default_rng(D)
    th0 = (0.3 * rng.standard_normal((C, D))).astype(NP[dtype]); p0 = rng.standard_normal((C, D)).astype(NP[dtype])
    draws = O.PhiloxDraws(seed, off + np.arange(C), NP[dtype])
    from hamiltorch_amd import rmhmc
    out_t, out_p = rmhmc.explicit_leapfrog(tt(th0, dtype), tt(p0, dtype), t, steps, eps, jitter, alpha, omega, ht.Metric.SOFTABS,
                                           seed=seed, chain_offset=off, draw=n)
    a, b, c, d = O.explicit_rmhmc_leapfrog(th0, p0, o, steps, eps, omega, alpha, jitter,
                                           (lambda k: draws.jitter_uniforms(n, 2 + k, D)), "softabs")
    np.testing.assert_allclose(out_t[0][-1].cpu().numpy(), a, rtol=tol, atol=tol)
    np.testing.assert_allclose(out_p[0][-1].cpu().numpy(), b, rtol=tol, atol=tol)
    np.testing.assert_allclose(out_t[1].cpu().numpy(), c, rtol=tol, atol=tol)
    np.testing.assert_allclose(out_p[1].cpu().numpy(), d, rtol=tol, atol=tol)
    H = rmhmc.rm_hamiltonian(tt(th0, dtype), tt(p0, dtype), t, jitter, alpha, ht.Metric.SOFTABS, seed=seed, chain_offset=off, draw=n, sub=1)
    wH, _ = O.rm_hamiltonian(th0, p0, o, alpha, jitter, None if jitter is None else draws.jitter_uniforms(n, 1, D), "softabs")
    np.testing.assert_allclose(H.cpu().numpy(), wH, rtol=10 * tol, atol=10 * tol)
    # momentum draw p = chol(G) z with the Philox normals of (seed, chain, n)
    pg = ht.samplers.gibbs(tt(th0, dtype), sampler=ht.Sampler.RMHMC, log_prob_func=t, jitter=jitter, softabs_const=alpha,
                           metric=ht.Metric.SOFTABS, seed=seed, chain_offset=off, draw=n)
    wp = O.rm_gibbs(th0, draws.normals(n, D), o, alpha, jitter, None if jitter is None else draws.jitter_uniforms(n, 0, D), "softabs")
    np.testing.assert_allclose(pg.cpu().numpy(), wp, rtol=10 * tol, atol=10 * tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-4), (torch.float64, 1e-8)])
@pytest.mark.parametrize("D,jitter,metric,burn,split", [(3, None, "softabs", 0, 1), (3, 1e-3, "softabs", 2, 1), (10, None, "hessian", -1, 1),
                                                        (20, 1e-3, "softabs", 0, 1), (20, 1e-3, "softabs", 0, 0), (3, 1e-3, "softabs", 2, 0),
                                                        (100, 1e-3, "softabs", 0, 1), (128, 2e-3, "hessian", 0, 1)])
def test_sample_rmhmc_vs_oracle(ht, dtype, tol, D, jitter, metric, burn, split):
    """End to end sample(sampler=RMHMC, integrator=EXPLICIT): momentum draw, both Hamiltonians, L explicit steps,
    MH, burn bookkeeping -- same Philox streams in the kernels and the oracle.  split 1 (default): the momenta of the
    fused routes are chol(P) z1 + sqrt(jitter u) . z2 (oracle: rm_gibbs_split); 0: chol(G) z, a factorisation per draw
    as the reference (S:183-184)."""
    from hamiltorch_amd import _abi
    _abi.set_tuning("rmhmc_momsplit", split)
    t, o = cfg3_target(ht, D, dtype, seed=5)
    C, N, L, eps, omega, alpha, seed, off = 24, 7, 3, 0.15, 10.0, 1e6, 2025, 3
    th0 = (0.3 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(NP[dtype])
    M = ht.Metric.SOFTABS if metric == "softabs" else ht.Metric.HESSIAN
    out, acc = ht.sample(t, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=burn, jitter=jitter,
                         softabs_const=alpha, explicit_binding_const=omega, sampler=ht.Sampler.RMHMC,
                         integrator=ht.Integrator.EXPLICIT, metric=M, debug=2, verbose=False, seed=seed, chain_offset=off)
    ref, info = O.sample_rmhmc_explicit(o, th0, N, L, eps, omega, alpha, burn, jitter, O.PhiloxDraws(seed, off + np.arange(C), NP[dtype]), metric,
                                        momentum=oracle_momentum(jitter, metric))
    got = np.stack([x.cpu().numpy() for x in out]); want = np.stack(ref)
    assert got.shape == want.shape
    bad = np.abs(got - want).max(axis=(0, 2)) > tol
    assert bad.mean() <= 0.05, "%d of %d chains differ" % (bad.sum(), C)
    np.testing.assert_allclose(acc.cpu().numpy()[~bad], info["acc_rate"][~bad], atol=1e-12)


def test_cfg3_shape_smoke_and_errors(ht):
    """cfg3 shape (D=100, 256 chains, softabs, omega=10, eps=0.1): a short run is finite, accepts, and moves."""
    t, _ = cfg3_target(ht, 100, torch.float32)
    th0 = 0.1 * torch.randn(256, 100, device=dev())
    out, acc = ht.sample(t, th0, num_samples=3, num_steps_per_sample=4, step_size=0.1, jitter=1e-3, softabs_const=1e6,
                         explicit_binding_const=10, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                         metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=1)
    s = torch.stack(out)
    assert torch.isfinite(s).all() and float(acc.mean()) > 0.9 and float((s[-1] - s[0]).abs().mean()) > 1e-3
    with pytest.raises(NotImplementedError, match="S3"):
        ht.sample(t, th0[0].clone(), num_samples=2, softabs_const=1e6, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.S3,
                  metric=ht.Metric.SOFTABS, verbose=False)
    with pytest.raises(RuntimeError, match="gradients not implemented for RMHMC"):     # S:390-391
        ht.sample(t, th0[0].clone(), num_samples=2, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                  softabs_const=1e6, metric=ht.Metric.SOFTABS, pass_grad=lambda w: w, verbose=False)


def test_cfg3_statistical_parity_with_jitter(ht):
    """T2 at BASELINE config 3: D=100, 256 chains, softabs alpha=1e6, omega=10, eps=0.1, L=10,
    jitter=1e-3 (stochastic Hamiltonian: only distributional parity exists).  600 trajectories per chain from stationary
    starts; the pooled marginal variances of the last 500 must match diag(P^-1) within SURVEY 8c's +-5 % (128 000 rows,
    lag-1 autocorrelation ~0.5: the Monte-Carlo sd of a variance ratio is ~0.7 %) and the acceptance rate the
    reference's 1.0."""
    from hamiltorch_amd import _abi
    t, o = cfg3_target(ht, 100, torch.float32)
    C, N, keep = 256, 600, 500
    cov = np.linalg.inv(o.P.astype(np.float64))
    th0 = torch.tensor(np.random.default_rng(0).multivariate_normal(np.zeros(100), cov, size=C), dtype=torch.float32, device=dev())
    out, acc = ht.sample(t, th0, num_samples=N, num_steps_per_sample=10, step_size=0.1, burn=-1, jitter=1e-3,
                         softabs_const=1e6, explicit_binding_const=10, sampler=ht.Sampler.RMHMC,
                         integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=7)
    assert _abi.last_route() == "rmhmc_uvc_kernel<co>"        # (round 4; before: rmhmc_uv_kernel<1,lean>)
    a = torch.stack(out).double().cpu().numpy()
    s = a[-keep:].reshape(-1, 100)
    # acceptance: SURVEY 8c asks +-0.01 of the reference / oracle on >= 1e5 pooled trajectories.  Device: 256 x 600 = 153 600
    # trajectories.  Oracle side: the eigendecomposing restatement from the same stationary starts and Philox streams on 24
    # chains x 4 trajectories (all it can afford: 8 L + 3 eigendecompositions of a 100 x 100 matrix per trajectory and chain);
    # the reference itself accepts every proposal at this configuration (acc 1.0)
    sel = np.linspace(0, C - 1, 24).astype(int)
    _, info = O.sample_rmhmc_explicit(o, th0.cpu().numpy()[sel], 4, 10, 0.1, 10.0, 1e6, -1, 1e-3, O.PhiloxDraws(7, sel, np.float32),
                                      "softabs", momentum="split")
    assert abs(float(acc.mean()) - float(np.mean(info["acc_rate"]))) <= 0.01, (float(acc.mean()), float(np.mean(info["acc_rate"])))
    var = s.var(axis=0)
    np.testing.assert_allclose(var, np.diag(cov), rtol=0.05)
    # means: SURVEY 8c's |mean| <= 4 sd / sqrt(ESS_pooled) per dimension with the ESS COMPUTED (hamiltorch_amd.ess: FFT
    # autocorrelation + Geyer truncation, pooled over the 256 chains), not assumed
    from hamiltorch_amd.ess import ess_bulk
    kept = torch.tensor(a[-keep:])
    ess = np.array([ess_bulk(kept[:, :, d]) for d in range(100)])
    assert ess.min() > 0.05 * keep * C, ess.min()                         # (antithetic-looking chains may exceed S C; a collapse may not)
    assert (np.abs(s.mean(axis=0)) <= 4.0 * np.sqrt(np.diag(cov) / ess)).all(), (np.abs(s.mean(axis=0)) / np.sqrt(np.diag(cov) / ess)).max()
    # off-diagonal structure too: the pooled covariance matches P^-1 in the operator norm
    emp = np.cov(s.T)
    assert np.linalg.norm(emp - cov, 2) <= 0.08 * np.linalg.norm(cov, 2)
    # the chains actually move
    assert np.abs(a[-1] - a[0]).mean() > 0.3


# ---- generic (non-constant-curvature) targets: SURVEY 8f N1 ---------------------------------------------
def funnel_logp(scales):
    """The notebook's funnel_ll (scaled variant of oracle.FunnelTarget) as a torch callable on the device."""
    def f(w):
        s = torch.as_tensor(scales, dtype=w.dtype, device=w.device)
        v, x = w[0], w[1:]
        hl2p = 0.9189385332046727
        lv = -v * v / 18.0 - 1.0986122886681098 - hl2p
        return lv + (-0.5 * torch.exp(v) * (x * x / s).sum() + 0.5 * x.numel() * v - x.numel() * hl2p - 0.5 * torch.log(s).sum())
    return f


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.float64, 1e-8)])
@pytest.mark.parametrize("D,kind,alpha", [(2, "indef", 1.3), (5, "indef", 2.0), (8, "degenerate", 1e6), (11, "indef", 0.7),
                                           (40, "spd", 1e6), (64, "indef", 1e6), (100, "indef", 0.05)])
def test_softabs_dmetric_vs_oracle(ht, dtype, tol, D, kind, alpha):
    """dmetric_out: M = Q W Q^T (the derivative of 1/2 log|G| + 1/2 m^T G^-1 m with respect to the entries of Hs),
    including repeated eigenvalues, |alpha lam| << 1 (series branch) and >> 1."""
    from hamiltorch_amd import _abi
    B = 4
    Hs = sym_batch(B, D, kind, D + 1).astype(NP[dtype])
    m = np.random.default_rng(2).standard_normal((B, D)).astype(NP[dtype])
    Mw, xw = O.softabs_dmetric(Hs.astype(np.float64), alpha, m)
    Md = torch.empty(B, D, D, dtype=dtype, device=dev()); xd = torch.empty(B, D, dtype=dtype, device=dev())
    t = tt(Hs, dtype)
    _abi.metric_eval(t, B, D, _abi.METRIC_SOFTABS, t, D * D, alpha, m=tt(m, dtype), x_out=xd, dmetric_out=Md)
    got = Md.double().cpu().numpy()
    np.testing.assert_allclose(got, np.swapaxes(got, 1, 2), atol=tol * np.abs(Mw).max())
    if kind == "degenerate":
        # inside a repeated eigenvalue the eigenvectors are arbitrary; M is not (J is constant on the block)
        pass
    np.testing.assert_allclose(got, Mw, rtol=tol, atol=tol * np.abs(Mw).max())
    # no momentum: only the log-determinant part
    _abi.metric_eval(t, B, D, _abi.METRIC_SOFTABS, t, D * D, alpha, dmetric_out=Md)
    M0, _ = O.softabs_dmetric(Hs.astype(np.float64), alpha, np.zeros_like(m))
    np.testing.assert_allclose(Md.double().cpu().numpy(), M0, rtol=tol, atol=tol * np.abs(M0).max())


@pytest.mark.parametrize("tag", ["a1e6", "a1p3", "d6"])
@pytest.mark.parametrize("dtag,dtype,tol", [("f64", torch.float64, 5e-8), ("f32", torch.float32, 3e-3)])
def test_generic_explicit_leapfrog_vs_reference_fixture(ht, golden, tag, dtag, dtype, tol):
    """leapfrog(sampler=RMHMC, integrator=EXPLICIT) on the scaled funnel: the values the unmodified reference
    produced by differentiating through hessian + eigh (S:398)."""
    g = golden("funnel")
    D, alpha, omega, eps, steps = g[f"{tag}_cfg"]
    D, steps = int(D), int(steps)
    key = f"{tag}_{dtag}"
    lp = funnel_logp(g["scales"][:D - 1])
    th, pm = tt(g[f"{key}_theta0"], dtype), tt(g[f"{key}_p0"], dtype)
    H = ht.samplers.rm_hamiltonian(th, pm, lp, None, 1.0, softabs_const=alpha, metric=ht.Metric.SOFTABS)
    np.testing.assert_allclose(H.cpu().numpy().reshape(-1), g[f"{key}_H"], rtol=tol, atol=tol)
    lpar, lmom = ht.samplers.leapfrog(th, pm, lp, steps=steps, step_size=eps, jitter=None, explicit_binding_const=omega,
                             softabs_const=alpha, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                             metric=ht.Metric.SOFTABS)
    np.testing.assert_allclose(torch.stack(lpar[0]).cpu().numpy(), g[f"{key}_lf_theta"], rtol=tol, atol=tol)
    np.testing.assert_allclose(torch.stack(lmom[0]).cpu().numpy(), g[f"{key}_lf_p"], rtol=tol, atol=tol)
    np.testing.assert_allclose(lpar[1].cpu().numpy(), g[f"{key}_lf_thetac"], rtol=tol, atol=tol)
    np.testing.assert_allclose(lmom[1].cpu().numpy(), g[f"{key}_lf_pc"], rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-7), (torch.float32, 5e-3)])
@pytest.mark.parametrize("D,scaled,jitter,alpha", [(5, True, None, 1e6), (5, False, 1e-2, 1e6), (11, False, 1e-3, 1.0)])
def test_generic_sample_rmhmc_vs_oracle(ht, dtype, tol, D, scaled, jitter, alpha):
    """End to end sample(RMHMC, EXPLICIT) on the funnel for a batch of chains: torch.func derivatives + native metric
    / contraction matrix / rotation / select, against the oracle with the same Philox streams (8 jitter sub-streams
    per step, as the reference draws them)."""
    scales = np.array([0.5, 1.0, 1.7, 2.4, 3.3, 0.8, 1.2, 2.0, 2.9, 0.6])[:D - 1] if scaled else np.ones(D - 1)
    lp = funnel_logp(scales)
    o = O.FunnelTarget(D, scales)
    C, N, L, eps, omega, seed, off = 12, 5, 3, 0.08, 10.0, 99, 7
    th0 = (0.4 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(NP[dtype])
    out, acc = ht.sample(lp, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=0, jitter=jitter,
                         softabs_const=alpha, explicit_binding_const=omega, sampler=ht.Sampler.RMHMC,
                         integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=seed,
                         chain_offset=off)
    with np.errstate(all="ignore"):
        ref, info = O.sample_rmhmc_explicit(o, th0, N, L, eps, omega, alpha, 0, jitter,
                                            O.PhiloxDraws(seed, off + np.arange(C), NP[dtype]))
    got = np.stack([x.cpu().numpy() for x in out]); want = np.stack(ref)
    assert got.shape == want.shape
    bad = ~(np.abs(got - want).max(axis=(0, 2)) <= tol)
    assert bad.mean() <= 0.1, "%d of %d chains differ, max err %.3g" % (bad.sum(), C, np.nanmax(np.abs(got - want)))


# ---- implicit RMHMC: generalised leapfrog with fixed-point iterations (S:305-387), SURVEY 8f N4 --------------
@pytest.mark.parametrize("tag", ["imp_a1e6", "imp_a1p1"])
def test_implicit_leapfrog_vs_reference_fixture(ht, golden, tag):
    g = golden("funnel")
    D, alpha, eps, steps, thr, max_it = g[f"{tag}_cfg"]
    D = int(D)
    lp = funnel_logp(g["scales"][:D - 1])
    th, pm = tt(g[f"{tag}_theta0"], torch.float64), tt(g[f"{tag}_p0"], torch.float64)
    lpar, lmom = ht.samplers.leapfrog(th, pm, lp, steps=int(steps), step_size=eps, jitter=None, softabs_const=alpha,
                                      fixed_point_threshold=thr, fixed_point_max_iterations=int(max_it),
                                      sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.IMPLICIT, metric=ht.Metric.SOFTABS)
    np.testing.assert_allclose(torch.stack(lpar).cpu().numpy(), g[f"{tag}_lf_theta"], rtol=2e-7, atol=2e-7)
    np.testing.assert_allclose(torch.stack(lmom).cpu().numpy(), g[f"{tag}_lf_p"], rtol=2e-7, atol=2e-7)


@pytest.mark.parametrize("dtype,tol,thr", [(torch.float64, 1e-7, 1e-18), (torch.float32, 5e-3, 1e-9)])
@pytest.mark.parametrize("D,scaled,jitter,alpha", [(5, True, None, 1e6), (6, False, 1e-2, 1.0)])
def test_implicit_sample_rmhmc_vs_oracle(ht, dtype, tol, thr, D, scaled, jitter, alpha):
    """sample(RMHMC, IMPLICIT) for a batch of chains that need different numbers of fixed-point iterations; the same
    Philox streams (fixed sub-stream layout per step) in the product and the oracle."""
    scales = np.array([0.5, 1.0, 1.7, 2.4, 3.3])[:D - 1] if scaled else np.ones(D - 1)
    lp = funnel_logp(scales)
    o = O.FunnelTarget(D, scales)
    C, N, L, eps, seed, off, max_it = 10, 4, 2, 0.06, 321, 2, 12
    th0 = (0.4 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(NP[dtype])
    out, acc = ht.sample(lp, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=0, jitter=jitter,
                         softabs_const=alpha, fixed_point_threshold=thr, fixed_point_max_iterations=max_it,
                         sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.IMPLICIT, metric=ht.Metric.SOFTABS, debug=2,
                         verbose=False, seed=seed, chain_offset=off)
    with np.errstate(all="ignore"):
        ref, info = O.sample_rmhmc_implicit(o, th0, N, L, eps, alpha, thr, max_it, 0, jitter,
                                            O.PhiloxDraws(seed, off + np.arange(C), NP[dtype]))
    got = np.stack([x.cpu().numpy() for x in out]); want = np.stack(ref)
    assert got.shape == want.shape
    bad = ~(np.abs(got - want).max(axis=(0, 2)) <= tol)
    assert bad.mean() <= 0.1, "%d of %d chains differ, max err %.3g" % (bad.sum(), C, np.nanmax(np.abs(got - want)))


# ---- identity-soft-abs fast path (csrc/rmhmc_fused.hip) vs the per-evaluation Jacobi path ----------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.float64, 1e-9)])
@pytest.mark.parametrize("D,jitter,metric", [(100, 1e-3, "softabs"), (100, None, "softabs"), (37, 5e-2, "softabs"), (20, 1e-3, "hessian")])
def test_fused_path_equals_jacobi_path(ht, dtype, tol, D, jitter, metric):
    """The same run through the fused whole-trajectory kernel and through the eigendecomposition per evaluation:
    G = softabs(P + jitter) equals P + jitter on this spectrum, so both must agree chain by chain."""
    from hamiltorch_amd import _abi
    t, o = cfg3_target(ht, D, dtype, seed=7)
    C, N, L, eps, omega, alpha, seed = 16, 4, 3, 0.1, 10.0, 1e6, 11
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(NP[dtype])
    M = ht.Metric.SOFTABS if metric == "softabs" else ht.Metric.HESSIAN
    kw = dict(num_samples=N, num_steps_per_sample=L, step_size=eps, burn=0, jitter=jitter, softabs_const=alpha,
              explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=M, debug=2,
              verbose=False, seed=seed)
    outs = []
    _abi.set_tuning("rmhmc_momsplit", 0)          # chol(G) z on both routes (the Jacobi route has no split draw,
    _abi.set_tuning("metric_sqrtdraw", 0)         # and its own default since round 6 is the symmetric square root)
    for fused in (1, 0):
        _abi.set_tuning("rmhmc_fused", fused)
        try:
            _abi.set_tuning("profile", 1)
            out, acc = ht.sample(t, tt(th0, dtype), **kw)
            ms, launches = _abi.profile_collect()
            _abi.set_tuning("profile", 0)
        finally:
            _abi.set_tuning("rmhmc_fused", 1)
        outs.append((np.stack([x.cpu().numpy() for x in out]), acc.cpu().numpy(), launches))
    assert outs[0][2] < outs[1][2], "the fused path was not taken (%d vs %d launches)" % (outs[0][2], outs[1][2])
    bad = np.abs(outs[0][0] - outs[1][0]).max(axis=(0, 2)) > tol
    assert bad.mean() <= 0.07, "%d of %d chains differ, max %.3g" % (bad.sum(), C, np.abs(outs[0][0] - outs[1][0]).max())
    np.testing.assert_allclose(outs[0][1][~bad], outs[1][1][~bad], atol=1e-12)


@pytest.mark.parametrize("D,jitter,metric,alpha,C", [(100, 1e-3, "softabs", 1e6, 300), (100, 1e-3, "softabs", 1.3, 9), (37, 5e-2, "softabs", 1e6, 17),
                                                     (20, None, "softabs", 2.0, 5), (20, 1e-3, "hessian", 1e6, 33), (112, 1e-3, "softabs", 1e6, 4)])
def test_trajectory_kernel_equals_the_launch_sequence(ht, D, jitter, metric, alpha, C):
    """Round 4: on the eigendecomposition route (rmhmc_fused = 0) a trajectory is ONE launch of metric_traj_mfma_kernel (momentum
    draw, H_old, 4 L half steps with the binding rotation, H_new: each chain's workgroup runs its 4 L + 3 evaluations back to
    back) + the accept/reject launch.  Same evaluation code as the launch-per-evaluation sequence ("metric_traj" = 0): samples,
    acceptance and both Hamiltonians are bit-identical; more chains than workgroup slots, a finite alpha (the soft-abs map is
    not the identity), Metric.HESSIAN (Cholesky instead of the refinement) and the largest D the kernel takes included."""
    from hamiltorch_amd import _abi
    t, o = cfg3_target(ht, D, torch.float32, seed=7)
    N, L, eps, omega, seed = 3, 3, 0.1, 10.0, 11
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    M = ht.Metric.SOFTABS if metric == "softabs" else ht.Metric.HESSIAN
    kw = dict(num_samples=N, num_steps_per_sample=L, step_size=eps, burn=0, jitter=jitter, softabs_const=alpha,
              explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=M, debug=2,
              verbose=False, seed=seed)
    outs = []
    _abi.set_tuning("rmhmc_fused", 0)
    _abi.set_tuning("metric_resident", 0)       # (round 6: the state in the caller's coordinates - the form that shares every operation with the launch sequence)
    try:
        for traj in (1, 0):
            _abi.set_tuning("metric_traj", traj)
            _abi.set_tuning("profile", 1)
            out, acc = ht.sample(t, tt(th0, torch.float32), **kw)
            ms, launches = _abi.profile_collect()
            _abi.set_tuning("profile", 0)
            outs.append((np.stack([x.cpu().numpy() for x in out]), acc.cpu().numpy(), launches))
    finally:
        _abi.set_tuning("metric_traj", 1)
        _abi.set_tuning("metric_resident", 1)
        _abi.set_tuning("rmhmc_fused", 1)
    assert outs[0][2] < outs[1][2] / 5, "the trajectory kernel was not taken (%d vs %d profiled launches)" % (outs[0][2], outs[1][2])
    assert np.isfinite(outs[0][0]).all()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("D,jitter,alpha,C,L", [(100, 1e-3, 1e6, 300, 3), (100, 1e-3, 1.3, 9, 3), (100, 3e-3, 0.7, 40, 2), (37, 5e-2, 1e6, 17, 3),
                                                (20, None, 2.0, 5, 3), (112, 1e-3, 1e6, 4, 3), (3, 1e-3, 1e6, 7, 4),
                                                (97, 1e-3, 1e6, 5, 3), (105, 2e-3, 1.3, 6, 2)])      # (97, 105: ragged last tile on the packed-argument entries of 7 tiles)
def test_resident_trajectory_state_equals_the_callers_coordinates(ht, D, jitter, alpha, C, L):
    """Round 6 ("metric_resident"): the trajectory kernel takes the chain's state into the eigenbasis once - theta' = V0^T (theta - mu),
    p' = V0^T p - keeps it in LDS for the trajectory (no V0 product, no global traffic in a solve evaluation; the binding rotation is
    element-wise in any orthonormal basis) and returns theta = mu + V0 theta' at the end.  Against the same kernel with the state in the
    caller's coordinates (0: bit-identical to the launch sequence, the test above): samples and Hamiltonians to fp32 rounding, the
    accept decisions equal wherever the two acceptance probabilities do not straddle the uniform draw, and both as close to the
    float64 oracle as each other.  jitter 3e-3 with alpha 0.7: some evaluations exceed the first-pass bound and take the general
    sequence in the middle of a resident trajectory (state out, evaluation, state in); jitter 5e-2: all of them do."""
    from hamiltorch_amd import _abi
    t, o = cfg3_target(ht, D, torch.float32, seed=7)
    N, eps, omega, seed = 3, 0.1, 10.0, 11
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    kw = dict(num_samples=N, num_steps_per_sample=L, step_size=eps, burn=0, jitter=jitter, softabs_const=alpha,
              explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, debug=2,
              verbose=False, seed=seed)
    outs = []
    _abi.set_tuning("rmhmc_fused", 0)
    try:
        for res in (1, 0):
            _abi.set_tuning("metric_resident", res)
            out, acc = ht.sample(t, tt(th0, torch.float32), **kw)
            outs.append((np.stack([x.cpu().numpy() for x in out]), acc.cpu().numpy()))
    finally:
        _abi.set_tuning("metric_resident", 1)
        _abi.set_tuning("rmhmc_fused", 1)
    a, f = outs
    assert np.isfinite(a[0]).all() and np.isfinite(f[0]).all()
    scale = np.abs(f[0]).max()
    same = (a[1] == f[1]).all(axis=0) if a[1].ndim == 2 else np.ones(C, bool)
    err = np.abs(a[0] - f[0]).max(axis=(0, 2))
    assert (err[same] <= 2e-4 * scale).all(), (err.max(), scale)
    assert same.mean() >= 0.95, same.mean()


@pytest.mark.parametrize("D,jitter,eps,burn,resident", [(100, 1e-3, 0.1, 0, 1), (100, 1e-3, 1.9, 1, 1), (37, 2e-3, 2.0, 2, 1), (100, 1e-3, 2.05, 1, 0), (3, 1e-3, 1.95, 1, 1)])
def test_selection_inside_the_trajectory_launch_equals_the_separate_launch(ht, D, jitter, eps, burn, resident):
    """Round 6 ("metric_select"): the trajectory kernel ends with its chain's Metropolis selection (hta_mh_select's rule,
    uniform and writes - S:1000-1018 with the burn-in reset of SURVEY Q2 - by the chain's own workgroup): one launch per trajectory.
    Bit-identical to the two-launch form: samples, accept flags, the states after burn-in resets; steps at the integrator's stability limit so that
    rejections occur (a small step: every proposal accepted)."""
    from hamiltorch_amd import _abi
    t, _ = cfg3_target(ht, D, torch.float32, seed=5)
    C, N, L, seed = 70, 5, 2, 13
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    kw = dict(num_samples=N, num_steps_per_sample=L, step_size=eps, burn=burn, jitter=jitter, softabs_const=1e6,
              explicit_binding_const=10.0, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, debug=2,
              verbose=False, seed=seed)
    outs, routes = [], []
    _abi.set_tuning("rmhmc_fused", 0)
    _abi.set_tuning("metric_resident", resident)
    try:
        for sel in (1, 0):
            _abi.set_tuning("metric_select", sel)
            out, acc = ht.sample(t, tt(th0, torch.float32), **kw)
            routes.append(_abi.last_route())
            outs.append((np.stack([x.cpu().numpy() for x in out]), acc.cpu().numpy()))
    finally:
        _abi.set_tuning("metric_select", 1)
        _abi.set_tuning("metric_resident", 1)
        _abi.set_tuning("rmhmc_fused", 1)
    assert "metric_traj_mfma_kernel" in routes[0] and "metric_traj_mfma_kernel" in routes[1], routes
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    rate = outs[0][1].mean()
    assert (rate == 1.0) if eps < 0.5 else (rate < 1.0), rate       # steps at the integrator's stability limit: rejections (and the burn-in reset) happen


@pytest.mark.parametrize("D,jitter", [(100, 1e-3), (20, None), (37, 2e-3)])
def test_fused_pair_kernel_equals_single_chain_kernel(ht, D, jitter):
    """The fused kernel can carry two chains per workgroup (tuning value 3; an odd chain count leaves the last pair half
    empty).  Same run with one chain per workgroup (the default) must agree chain by chain."""
    from hamiltorch_amd import _abi
    t, _ = cfg3_target(ht, D, torch.float32, seed=3)
    C, N, L, eps, seed = 1025, 3, 2, 0.1, 5
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    kw = dict(num_samples=N, num_steps_per_sample=L, step_size=eps, burn=0, jitter=jitter, softabs_const=1e6,
              explicit_binding_const=10.0, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
              metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=seed)
    outs = []
    for mode in (3, 1):
        _abi.set_tuning("rmhmc_fused", mode)
        try:
            out, acc = ht.sample(t, tt(th0, torch.float32), **kw)
        finally:
            _abi.set_tuning("rmhmc_fused", 1)
        outs.append((np.stack([x.cpu().numpy() for x in out]), acc.cpu().numpy()))
    assert np.isfinite(outs[0][0]).all()
    bad = np.abs(outs[0][0] - outs[1][0]).max(axis=(0, 2)) > 2e-4
    assert bad.mean() <= 0.02, "%d of %d chains differ, max %.3g" % (bad.sum(), C, np.abs(outs[0][0] - outs[1][0]).max())
    np.testing.assert_allclose(outs[0][1][~bad], outs[1][1][~bad], atol=1e-12)


# ---- Metric.HESSIAN on a general log-concave target ---------------------------------------------------------------
def logcosh_logp(P, A):
    def f(w):
        Pt = torch.as_tensor(P, dtype=w.dtype, device=w.device); At = torch.as_tensor(A, dtype=w.dtype, device=w.device)
        return -0.5 * torch.dot(w, torch.mv(Pt, w)) - torch.log(torch.cosh(torch.mv(At, w))).sum()
    return f


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.float64, 1e-8)])
@pytest.mark.parametrize("D", [3, 16, 60])
def test_hessian_dmetric_vs_oracle(ht, dtype, tol, D):
    """dmetric_out for Metric.HESSIAN: M = 1/2 G^-1 - 1/2 v v^T from the in-LDS Cholesky factor."""
    from hamiltorch_amd import _abi
    B = 4
    Hs = sym_batch(B, D, "spd", D + 3).astype(NP[dtype])
    m = np.random.default_rng(3).standard_normal((B, D)).astype(NP[dtype])
    Mw, vw = O.hessian_dmetric(Hs.astype(np.float64), m)
    Md = torch.empty(B, D, D, dtype=dtype, device=dev()); xd = torch.empty(B, D, dtype=dtype, device=dev())
    t = tt(Hs, dtype)
    _abi.metric_eval(t, B, D, _abi.METRIC_HESSIAN, t, D * D, 0.0, m=tt(m, dtype), x_out=xd, dmetric_out=Md)
    np.testing.assert_allclose(Md.double().cpu().numpy(), Mw, rtol=tol, atol=tol * np.abs(Mw).max())
    np.testing.assert_allclose(xd.double().cpu().numpy(), vw, rtol=tol, atol=tol * np.abs(vw).max())


def test_hessian_metric_general_target_vs_reference_fixture(ht, golden):
    g = golden("logcosh")
    omega, eps, steps, thr, max_it = g["cfg"]
    lp = logcosh_logp(g["P"], g["A"])
    th, pm = tt(g["theta0"], torch.float64), tt(g["p0"], torch.float64)
    H = ht.samplers.rm_hamiltonian(th, pm, lp, None, 1.0, metric=ht.Metric.HESSIAN)
    np.testing.assert_allclose(H.cpu().numpy().reshape(-1), g["H"], rtol=1e-9, atol=1e-9)
    lpar, lmom = ht.samplers.leapfrog(th, pm, lp, steps=int(steps), step_size=eps, jitter=None, explicit_binding_const=omega,
                                      sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.HESSIAN)
    np.testing.assert_allclose(torch.stack(lpar[0]).cpu().numpy(), g["exp_theta"], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(torch.stack(lmom[0]).cpu().numpy(), g["exp_p"], rtol=1e-7, atol=1e-7)
    lpar, lmom = ht.samplers.leapfrog(th, pm, lp, steps=int(steps), step_size=eps, jitter=None, fixed_point_threshold=thr,
                                      fixed_point_max_iterations=int(max_it), sampler=ht.Sampler.RMHMC,
                                      integrator=ht.Integrator.IMPLICIT, metric=ht.Metric.HESSIAN)
    np.testing.assert_allclose(torch.stack(lpar).cpu().numpy(), g["imp_theta"], rtol=2e-7, atol=2e-7)
    np.testing.assert_allclose(torch.stack(lmom).cpu().numpy(), g["imp_p"], rtol=2e-7, atol=2e-7)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-7), (torch.float32, 5e-3)])
def test_hessian_metric_general_target_sample_vs_oracle(ht, dtype, tol):
    """sample(RMHMC, EXPLICIT, Metric.HESSIAN) with jitter on the log-cosh target, batch of chains, same Philox streams."""
    rng = np.random.default_rng(2); D = 8
    Q, _ = np.linalg.qr(rng.standard_normal((D, D))); P = (Q * np.linspace(0.5, 2.0, D)) @ Q.T; P = 0.5 * (P + P.T)
    A = 0.6 * rng.standard_normal((10, D))
    lp, o = logcosh_logp(P, A), O.LogCoshTarget(P, A)
    C, N, L, eps, omega, seed, off, jitter = 12, 4, 3, 0.1, 10.0, 77, 1, 1e-3
    th0 = (0.4 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(NP[dtype])
    out, acc = ht.sample(lp, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=0, jitter=jitter,
                         explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                         metric=ht.Metric.HESSIAN, debug=2, verbose=False, seed=seed, chain_offset=off)
    ref, info = O.sample_rmhmc_explicit(o, th0, N, L, eps, omega, 1.0, 0, jitter, O.PhiloxDraws(seed, off + np.arange(C), NP[dtype]),
                                        "hessian")
    got = np.stack([x.cpu().numpy() for x in out]); want = np.stack(ref)
    bad = ~(np.abs(got - want).max(axis=(0, 2)) <= tol)
    assert bad.mean() <= 0.1, "%d of %d chains differ, max err %.3g" % (bad.sum(), C, np.nanmax(np.abs(got - want)))


def test_fused_workspace_passes_are_equivalent(ht):
    """The fused path pre-draws the momenta of as many trajectories as the workspace holds; with the minimum workspace it
    works in passes of 4.  Same samples either way (bit for bit: the same kernels on the same streams)."""
    from hamiltorch_amd import _abi
    D, C, T, L = 24, 40, 11, 2
    t, _ = cfg3_target(ht, D, torch.float32, seed=9)
    th0 = tt((0.3 * O.philox_normals(3, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32)
    outs = []
    for nbytes in (_abi.rmhmc_workspace_bytes(C, D, 4, 0), _abi.rmhmc_workspace_bytes(C, D, 4, T)):
        cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev())
        samples = torch.zeros(T + 1, C, D, device=dev())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
        _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, 1e-3, L, 0.1, 10.0,
                                   T, 0, -1, 21, 0, samples, rej, ws)
        outs.append((samples.cpu().numpy(), rej.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.abs(outs[0][0][-1] - outs[0][0][0]).max() > 1e-3


@pytest.mark.parametrize("split", [1, 0])
@pytest.mark.parametrize("T", [32, 75, 200])
def test_fused_momentum_overlap_equals_serial(ht, T, split):
    """hta_set_tuning('rmhmc_overlap', 1): with room for two blocks the momentum draws of block b+1 run on a side stream under
    the trajectories of block b (the default, 0, keeps everything on the caller's stream - measured faster at every chain
    count, profiles/r03h_*).  Same kernels, same draws: bit-identical samples / reject counts, also when the call is repeated
    back to back (event reuse) and followed by work on the caller's stream that reads the results; with the split draw
    and with the factorisation per draw."""
    from hamiltorch_amd import _abi
    _abi.set_tuning("rmhmc_momsplit", split)
    D, C, L = 24, 40, 2
    t, _ = cfg3_target(ht, D, torch.float32, seed=9)
    th0 = tt((0.3 * O.philox_normals(3, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32)
    outs = []
    for mode in (1, 0):
        _abi.set_tuning("rmhmc_overlap", mode)
        try:
            res = []
            for rep in range(2):
                cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev())
                samples = torch.zeros(T + 1, C, D, device=dev())
                ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
                _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, 1e-3, L, 0.1, 10.0,
                                           T, 0, -1, 21 + rep, 0, samples, rej, ws)
                res.append((samples.sum(dim=0).cpu().numpy(), samples.cpu().numpy(), rej.cpu().numpy()))   # no explicit sync
        finally:
            _abi.set_tuning("rmhmc_overlap", 0)
        outs.append(res)
    for a_, b_ in zip(outs[0], outs[1]):
        for x, y in zip(a_, b_):
            assert np.array_equal(x, y)
    assert not np.array_equal(outs[0][0][1], outs[0][1][1])


@pytest.mark.parametrize("D,C,jit", [(24, 40, 1e-3), (100, 50, 1e-3), (100, 33, None), (112, 17, 1e-3), (7, 16, 1e-3),
                                     (100, 2050, 1e-3)])
def test_batched_mfma_kernel_equals_fused_kernel(ht, D, C, jit):
    """rmhmc_batch_kernel (16 chains per workgroup, S X and P X as v_mfma_f32_16x16x4_f32 tiles) against rmhmc_fused_kernel
    (one chain per workgroup, the parity reference of this path): same streams and update order, sums in a different order ->
    chain by chain to rounding, with ragged groups, burn-in (Q2 reset) and without jitter."""
    from hamiltorch_amd import _abi
    T, L, burn = 9, 3, 2
    t, _ = cfg3_target(ht, D, torch.float32, seed=9)
    th0 = tt((0.3 * O.philox_normals(3, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32)
    outs = []
    for mode in (2, 0):
        _abi.set_tuning("rmhmc_batch", mode)
        _abi.set_tuning("rmhmc_mfma4", 0)           # (from 513 to 2048 chains the default route is a four-chain kernel,
        _abi.set_tuning("rmhmc_uv", 0)              #  up to 512 chains rmhmc_uv_kernel)
        try:
            cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            samples = torch.zeros(T - burn + 1, C, D, device=dev())
            ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
            _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, jit, L, 0.1, 10.0,
                                       T, 0, burn, 21, 0, samples, rej, ws)
            torch.cuda.synchronize()
        finally:
            _abi.set_tuning("rmhmc_batch", 1)
            _abi.set_tuning("rmhmc_mfma4", 1)
            _abi.set_tuning("rmhmc_uv", 1)
        outs.append((samples.cpu().numpy(), rej.cpu().numpy(), cur.cpu().numpy()))
    err = np.abs(outs[0][0] - outs[1][0]).max(axis=(0, 2))
    assert (err > 2e-4).mean() <= 0.05, "max err %.3g (%d chains differ)" % (err.max(), (err > 2e-4).sum())
    good = err <= 2e-4
    assert np.array_equal(outs[0][1][good], outs[1][1][good])
    np.testing.assert_allclose(outs[0][2][good], outs[1][2][good], atol=2e-4)
    assert np.abs(outs[0][0][T - 1 - burn] - outs[0][0][1]).max() > 1e-3          # rows 1 .. T-1-burn are written


@pytest.mark.parametrize("D,C,jit,batch", [(7, 19, 1e-3, 0), (24, 40, 1e-2, 0), (50, 33, 1e-3, 0), (77, 21, 1e-3, 0),
                                           (100, 50, 1e-3, 0), (104, 9, 1e-3, 0), (100, 40, 1e-3, 2)])
def test_wave_momentum_kernel_equals_workgroup_kernel(ht, D, C, jit, batch):
    """rmhmc_momentum_wave_kernel (one wave per draw, the work matrix in registers, p = L z accumulated panel by panel) against
    rmhmc_momentum_kernel (Cholesky in LDS, then the triangular product): same jitter / normal streams and the same factor;
    the sums of p run in a different order -> whole runs agree chain by chain to rounding, every NB instance, ragged D."""
    from hamiltorch_amd import _abi
    T, L, burn = 9, 3, 2
    t, _ = cfg3_target(ht, D, torch.float32, seed=11)
    th0 = tt((0.3 * O.philox_normals(5, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32)
    outs = []
    _abi.set_tuning("rmhmc_momsplit", 0)          # the draws with a factorisation each (the default is the split draw)
    _abi.set_tuning("rmhmc_batch", batch)
    try:
        for mode in (1, 0):
            _abi.set_tuning("rmhmc_momwave", mode)
            cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            samples = torch.zeros(T - burn + 1, C, D, device=dev())
            ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
            _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, jit, L, 0.1, 10.0,
                                       T, 0, burn, 23, 0, samples, rej, ws)
            torch.cuda.synchronize()
            outs.append((samples.cpu().numpy(), rej.cpu().numpy(), cur.cpu().numpy()))
    finally:
        _abi.set_tuning("rmhmc_momwave", 1)
        _abi.set_tuning("rmhmc_batch", 1)
    assert np.isfinite(outs[0][0]).all()
    err = np.abs(outs[0][0] - outs[1][0]).max(axis=(0, 2))
    assert (err > 2e-4).mean() <= 0.05, "max err %.3g (%d chains differ)" % (err.max(), (err > 2e-4).sum())
    good = err <= 2e-4
    assert np.array_equal(outs[0][1][good], outs[1][1][good])
    np.testing.assert_allclose(outs[0][2][good], outs[1][2][good], atol=2e-4)
    assert np.abs(outs[0][0][T - 1 - burn] - outs[0][0][1]).max() > 1e-3


@pytest.mark.parametrize("D,C,jit", [(100, 40, 1e-3), (100, 301, 1e-3), (100, 24, None), (37, 30, 5e-2), (7, 9, 1e-3), (64, 18, 2e-3),
                                     (65, 7, 1e-3), (3, 5, 1e-3), (100, 700, 1e-3)])
def test_uv_kernel_equals_one_chain_kernel(ht, D, C, jit):
    """rmhmc_uv_kernel (csrc/rmhmc_uv.hip: one chain per workgroup up to 256 chains, two beyond; a chain's state set and its
    copy are two columns of v_mfma_f32_4x4x1_16b_f32, tracked products) against rmhmc_fused_kernel with every product evaluated
    ("rmhmc_pair" = 0, one chain per workgroup on the vector ALUs): same streams and update order, sums in a different order ->
    chain by chain to rounding; an odd chain count (half-empty last workgroup), no jitter (no solve phases), many refinements,
    D on both sides of a wave's 32 rows, burn-in (Q2 reset), and beyond the default range ("rmhmc_uv" = 2 at 700 chains)."""
    from hamiltorch_amd import _abi
    T, L, burn = 9, 4, 2
    t, _ = cfg3_target(ht, D, torch.float32, seed=5)
    th0 = tt((0.3 * O.philox_normals(3, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32)
    outs = []
    try:
        for uv in (2, 0):
            _abi.set_tuning("rmhmc_uv", uv); _abi.set_tuning("rmhmc_pair", 1 if uv else 0)
            _abi.set_tuning("rmhmc_mfma4", 0); _abi.set_tuning("rmhmc_batch", 0)
            cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            samples = torch.zeros(T - burn + 1, C, D, device=dev())
            Ho = torch.zeros(T, C, device=dev()); Hn = torch.zeros(T, C, device=dev()); ac = torch.zeros(T, C, dtype=torch.uint8, device=dev())
            ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
            _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, jit, L, 0.1, 10.0,
                                       T, 0, burn, 21, 0, samples, rej, ws, H_old=Ho, H_new=Hn, accept=ac)
            torch.cuda.synchronize()
            outs.append((samples.cpu().numpy(), rej.cpu().numpy(), cur.cpu().numpy(), Ho.cpu().numpy(), Hn.cpu().numpy(), ac.cpu().numpy()))
    finally:
        _abi.set_tuning("rmhmc_uv", 1); _abi.set_tuning("rmhmc_pair", 1)
        _abi.set_tuning("rmhmc_mfma4", MFMA4_DEFAULT); _abi.set_tuning("rmhmc_batch", 1)
    assert np.isfinite(outs[0][0]).all()
    np.testing.assert_allclose(outs[0][3][0], outs[1][3][0], rtol=2e-5, atol=2e-4)      # H_old / H_new of the first trajectory:
    np.testing.assert_allclose(outs[0][4][0], outs[1][4][0], rtol=2e-5, atol=2e-4)      # before any accept decision
    err = np.abs(outs[0][0] - outs[1][0]).max(axis=(0, 2))
    assert (err > 2e-4).mean() <= 0.05, "max err %.3g (%d chains differ)" % (err.max(), (err > 2e-4).sum())
    good = err <= 2e-4
    assert np.array_equal(outs[0][1][good], outs[1][1][good])
    assert np.array_equal(outs[0][5][:, good], outs[1][5][:, good])
    np.testing.assert_allclose(outs[0][2][good], outs[1][2][good], atol=2e-4)
    assert np.abs(outs[0][0][T - 1 - burn] - outs[0][0][1]).max() > 1e-3


@pytest.mark.parametrize("D,C,jit", [(100, 40, 1e-3), (100, 300, 1e-3), (100, 24, None), (37, 30, 5e-2), (20, 12, 1e-3), (64, 20, 2e-3),
                                     (3, 5, 1e-3)])
def test_tracked_products_equal_explicit_products(ht, D, C, jit):
    """"rmhmc_pair" = 1 (default): the one-chain kernel carries P (theta - mu) and S p along element-wise and evaluates only the
    refinement products of a half step, two half steps side by side (csrc/rmhmc_fused.hip: pair_tracked).  "rmhmc_pair" = 0
    evaluates every product of every half step.  Same streams, same update order: chain by chain to rounding - one workgroup per
    CU and two (C = 300), without jitter (K = 0: no solve phases at all), many refinements (jitter 5e-2), burn-in (Q2 reset)."""
    from hamiltorch_amd import _abi
    T, L, burn = 9, 4, 2
    t, _ = cfg3_target(ht, D, torch.float32, seed=5)
    th0 = tt((0.3 * O.philox_normals(3, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32)
    outs = []
    try:
        for pair in (1, 0):
            _abi.set_tuning("rmhmc_pair", pair)
            _abi.set_tuning("rmhmc_mfma4", 0); _abi.set_tuning("rmhmc_batch", 0); _abi.set_tuning("rmhmc_uv", 0)
            cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            samples = torch.zeros(T - burn + 1, C, D, device=dev())
            Ho = torch.zeros(T, C, device=dev()); Hn = torch.zeros(T, C, device=dev())
            ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
            _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, jit, L, 0.1, 10.0,
                                       T, 0, burn, 21, 0, samples, rej, ws, H_old=Ho, H_new=Hn)
            torch.cuda.synchronize()
            outs.append((samples.cpu().numpy(), rej.cpu().numpy(), cur.cpu().numpy(), Ho.cpu().numpy(), Hn.cpu().numpy()))
    finally:
        _abi.set_tuning("rmhmc_pair", 1)
        _abi.set_tuning("rmhmc_mfma4", MFMA4_DEFAULT); _abi.set_tuning("rmhmc_batch", 1); _abi.set_tuning("rmhmc_uv", 1)
    assert np.isfinite(outs[0][0]).all()
    np.testing.assert_allclose(outs[0][4][0], outs[1][4][0], rtol=2e-5, atol=2e-4)      # H_new of the first trajectory: before any accept decision
    err = np.abs(outs[0][0] - outs[1][0]).max(axis=(0, 2))
    assert (err > 2e-4).mean() <= 0.05, "max err %.3g (%d chains differ)" % (err.max(), (err > 2e-4).sum())
    good = err <= 2e-4
    assert np.array_equal(outs[0][1][good], outs[1][1][good])
    np.testing.assert_allclose(outs[0][2][good], outs[1][2][good], atol=2e-4)
    assert np.abs(outs[0][0][T - 1 - burn] - outs[0][0][1]).max() > 1e-3


@pytest.mark.parametrize("D,C,jit", [(7, 9, 1e-3), (24, 33, 1e-3), (64, 18, 1e-3), (65, 7, 1e-3), (100, 50, 1e-3), (100, 33, None),
                                     (100, 1030, 1e-3)])
def test_mfma4_kernel_equals_fused_kernel(ht, D, C, jit):
    """rmhmc_mfma4_kernel (4 chains per two-wave workgroup, S X and P X on v_mfma_f32_4x4x1_16b_f32) against rmhmc_fused_kernel
    (one chain per workgroup): same streams and update order, sums in a different order -> chain by chain to rounding, with
    ragged groups, burn-in (Q2 reset), D on both sides of a wave's 64 rows, and without jitter."""
    from hamiltorch_amd import _abi
    T, L, burn = 9, 3, 2
    t, _ = cfg3_target(ht, D, torch.float32, seed=9)
    th0 = tt((0.3 * O.philox_normals(3, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32)
    outs = []
    try:
        on = 2 if C < 513 else 1                                 # from 513 to 2048 chains it is the default route
        for mode, waves, pair in ((on, 4, 1), (0, 4, 1), (on, 2, 1), (on, 4, 0)):
            _abi.set_tuning("rmhmc_mfma4", mode)
            _abi.set_tuning("rmhmc_mfma4_waves", waves)          # 4: rmhmc_mfma4x4_kernel (default), 2: rmhmc_mfma4_kernel
            _abi.set_tuning("rmhmc_pair", pair)                  # 1: tracked products (default), 0: every product of every half step
            _abi.set_tuning("rmhmc_batch", 0)
            _abi.set_tuning("rmhmc_uv", 0)                       # (mode 0: the one-chain kernel, the reference)
            cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            samples = torch.zeros(T - burn + 1, C, D, device=dev())
            ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
            _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, jit, L, 0.1, 10.0,
                                       T, 0, burn, 21, 0, samples, rej, ws)
            torch.cuda.synchronize()
            outs.append((samples.cpu().numpy(), rej.cpu().numpy(), cur.cpu().numpy()))
    finally:
        _abi.set_tuning("rmhmc_mfma4", MFMA4_DEFAULT)
        _abi.set_tuning("rmhmc_mfma4_waves", 4)
        _abi.set_tuning("rmhmc_pair", 1)
        _abi.set_tuning("rmhmc_batch", 1)
        _abi.set_tuning("rmhmc_uv", 1)
    assert np.isfinite(outs[0][0]).all()
    err2 = np.abs(outs[0][0] - outs[2][0]).max(axis=(0, 2))          # four-wave kernel against the two-wave kernel
    assert (err2 > 2e-4).mean() <= 0.05, "four- vs two-wave kernel: max err %.3g (%d chains differ)" % (err2.max(), (err2 > 2e-4).sum())
    err3 = np.abs(outs[0][0] - outs[3][0]).max(axis=(0, 2))          # paired half steps against one half step at a time
    assert (err3 > 2e-4).mean() <= 0.05, "paired vs single half steps: max err %.3g (%d chains differ)" % (err3.max(), (err3 > 2e-4).sum())
    err = np.abs(outs[0][0] - outs[1][0]).max(axis=(0, 2))
    assert (err > 2e-4).mean() <= 0.05, "max err %.3g (%d chains differ)" % (err.max(), (err > 2e-4).sum())
    good = err <= 2e-4
    assert np.array_equal(outs[0][1][good], outs[1][1][good])
    np.testing.assert_allclose(outs[0][2][good], outs[1][2][good], atol=2e-4)
    assert np.abs(outs[0][0][T - 1 - burn] - outs[0][0][1]).max() > 1e-3



@pytest.mark.parametrize("C", [1024, 2050, 4100])
def test_sample_rmhmc_cfg5_shapes_vs_oracle(ht, C):
    """sample(RMHMC, EXPLICIT, SOFTABS) at D=100 on the default routes of large batches (1024 chains:
    rmhmc_mfma4_kernel; 2050 and 4100: rmhmc_batch_kernel; momenta from rmhmc_momentum_split_kernel) against the oracle on the
    first and last chains of the batch (same Philox streams; SURVEY 8c tolerance 1e-4 on theta)."""
    D, N, L, eps, omega, alpha, seed, off, jitter = 100, 4, 2, 0.1, 10.0, 1e6, 77, 5, 1e-3
    t, o = cfg3_target(ht, D, torch.float32, seed=0)
    th0 = (0.1 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    out = ht.sample(t, tt(th0, torch.float32), num_samples=N, num_steps_per_sample=L, step_size=eps, jitter=jitter,
                    softabs_const=alpha, explicit_binding_const=omega, sampler=ht.Sampler.RMHMC,
                    integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, verbose=False, seed=seed, chain_offset=off)
    got = np.stack([x.cpu().numpy() for x in out])
    sel = np.r_[0:3, C - 3:C]
    ref, _ = O.sample_rmhmc_explicit(o, th0[sel], N, L, eps, omega, alpha, 0, jitter, O.PhiloxDraws(seed, off + sel, np.float32), "softabs",
                                     momentum=oracle_momentum(jitter))
    want = np.stack(ref)
    assert np.isfinite(got).all() and got.shape == (N, C, D)
    err = np.abs(got[:, sel] - want).max(axis=(0, 2))
    assert (err < 3e-4).sum() >= len(sel) - 1, err
    assert np.abs(got[-1] - got[0]).mean() > 1e-3


# ---- the matrix-core evaluation kernel (csrc/rmhmc_metric_mfma.hip) against the Jacobi kernel and the oracle ---------------
def _warm_eval(ht, P, X, m, alpha, jitter, seed, mode, want_g=True):
    """One batched warm-start evaluation through hta_metric_eval with hta_set_tuning('metric_mfma', mode)."""
    from hamiltorch_amd import _abi
    dt = torch.float32
    B, D = m.shape
    Pd = tt(P, dt)
    V0 = torch.empty(D, D, device=dev()); lam0 = torch.empty(D, device=dev())
    _abi.set_tuning("metric_mfma", 0)                      # the shared basis always comes from the Jacobi kernel (as in the driver)
    _abi.metric_eval(Pd, 1, D, _abi.METRIC_SOFTABS, Pd, 0, alpha, V_out=V0, lamraw_out=lam0)
    out = dict(x=torch.zeros(B, D, device=dev()), H=torch.empty(B, device=dev()), ld=torch.empty(B, device=dev()),
               q=torch.empty(B, device=dev()), lam=torch.empty(B, D, device=dev()), p=torch.empty(B, D, device=dev()),
               G=torch.empty(B, D, D, device=dev()), lp=torch.empty(B, device=dev()), ug=torch.zeros(B, D, device=dev()))
    mu = torch.zeros(D, device=dev())
    try:
        _abi.set_tuning("metric_mfma", mode)
        _abi.metric_eval(Pd, B, D, _abi.METRIC_SOFTABS, Pd, 0, alpha, jitter, seed, 3, 7, 2, X=tt(X, dt), Pm=Pd, mu=mu, log_norm=0.25,
                         m=tt(m, dt), upd_x=out["x"], cx=0.5, upd_g=out["ug"], cg=-0.5, lam_out=out["lam"], logdet_out=out["ld"],
                         quad_out=out["q"], H_out=out["H"], logp_out=out["lp"], V0=V0, lam0=lam0)
        route = _abi.last_route()
        if want_g:
            _abi.metric_eval(Pd, B, D, _abi.METRIC_SOFTABS, Pd, 0, alpha, jitter, seed, 3, 7, 0, p_out=out["p"], G_out=out["G"],
                             V0=V0, lam0=lam0)
        torch.cuda.synchronize()
    finally:
        _abi.set_tuning("metric_mfma", 1)
    assert route == ("metric_warm_mfma_kernel" if mode else "metric_eval_kernel<float>"), route      # the kernel under test ran
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("D,kind,alpha,jitter", [(1, "spd", 1e6, 1e-3), (2, "indef", 1.3, 1e-2), (3, "spd", 1e6, 1e-3), (10, "spd", 1e6, None), (16, "indef", 1.3, 1e-3), (31, "spd", 1e6, 1e-3),
                                                 (64, "indef", 2.0, 1e-2), (100, "spd", 1e6, 1e-3), (100, "spd", 1e6, 0.3),
                                                 (101, "degenerate", 3.0, 1e-3), (112, "spd", 1e6, 1e-3), (9, "identity", 1e6, 1e-3),
                                                 (100, "indef", 0.05, 1e-3), (100, "indef", 1.0, 1e-3), (48, "degenerate", 1e6, 1e-3),
                                                 (100, "spd", 0.05, 1e-3), (64, "degenerate", 1.0, 1e-2)])
def test_metric_mfma_kernel_vs_oracle_and_jacobi_kernel(ht, D, kind, alpha, jitter):
    """csrc/rmhmc_metric_mfma.hip (eigenvectors refined from the shared basis with MFMA GEMMs; MFMA Cholesky) DIRECTLY against
    the oracle's float64 eigh (`for got in (a, j)`: the matrix-core kernel `a` and the Jacobi kernel `j` are each compared with
    the oracle, then with each other; `_warm_eval` asserts through hta_last_route that `a` really ran metric_warm_mfma_kernel): G^-1 m, log|G|, m^T G^-1 m,
    H, log p, P (X - mu), the soft-abs spectrum, the assembled G and the momentum draw p = chol(G) z.  Cases: well separated
    spectra (refinement converges), an indefinite curvature with finite alpha, a large jitter and (nearly) degenerate /
    identity spectra (the in-kernel fallback to Jacobi)."""
    rng = np.random.default_rng(D)
    if kind == "identity":
        P = 1.5 * np.eye(D)
    elif kind == "spd" and D == 100:
        P = cfg3_target(ht, 100, torch.float32)[1].P.astype(np.float64)
    else:
        P = sym_batch(1, D, kind, D + 1)[0]
    B, seed = 37, 99
    X = (0.3 * rng.standard_normal((B, D))).astype(np.float32)
    m = rng.standard_normal((B, D)).astype(np.float32)
    a = _warm_eval(ht, P, X, m, alpha, jitter, seed, 1)
    j = _warm_eval(ht, P, X, m, alpha, jitter, seed, 0)
    # the oracle on the same jitter stream (chain_offset 3, draw 7, sub 2 / 0)
    Hs = np.broadcast_to(P, (B, D, D)).astype(np.float64).copy()
    ju = None if jitter is None else O.philox_uniforms(seed, 3 + np.arange(B), 7, D, O.PURPOSE_JITTER, 2, dtype=np.float64)
    G, lam, _ = O.softabs_metric(Hs, alpha, jitter, ju)
    x64 = np.linalg.solve(G, m.astype(np.float64)[..., None])[..., 0]
    cond = np.abs(lam).max() / np.abs(lam).min()
    tol = 4e-4
    for got in (a, j):
        np.testing.assert_allclose(got["x"], 0.5 * x64, rtol=tol * cond, atol=tol * cond * np.abs(x64).max())
        np.testing.assert_allclose(got["ld"], np.log(lam).sum(1), rtol=tol, atol=tol * D)
        np.testing.assert_allclose(got["q"], (m * x64).sum(1), rtol=tol * cond, atol=tol * cond)
        np.testing.assert_allclose(np.sort(got["lam"], axis=1), np.sort(lam, axis=1), rtol=tol, atol=tol * np.abs(lam).max())
        want_lp = 0.25 - 0.5 * np.einsum("bi,ij,bj->b", X.astype(np.float64), P, X.astype(np.float64))
        np.testing.assert_allclose(got["lp"], want_lp, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(got["ug"], -0.5 * (X.astype(np.float64) @ P), rtol=1e-4, atol=1e-4)
        wantH = -want_lp + 0.5 * D * np.log(2 * np.pi) + 0.5 * np.log(lam).sum(1) + 0.5 * (m * x64).sum(1)
        np.testing.assert_allclose(got["H"], wantH, rtol=tol * cond, atol=tol * cond * 10)
    # kernel against kernel: tighter than either against float64
    np.testing.assert_allclose(a["x"], j["x"], rtol=2e-4 * cond, atol=2e-5 * cond * np.abs(x64).max())
    np.testing.assert_allclose(a["H"], j["H"], rtol=1e-5 * cond, atol=1e-4 * cond)
    # G (sub-stream 0) and the momentum draw
    ju0 = None if jitter is None else O.philox_uniforms(seed, 3 + np.arange(B), 7, D, O.PURPOSE_JITTER, 0, dtype=np.float64)
    G0, _, _ = O.softabs_metric(Hs, alpha, jitter, ju0)
    z = O.philox_normals(seed, 3 + np.arange(B), 7, D, dtype=np.float64)
    p64 = np.einsum("bij,bj->bi", np.linalg.cholesky(G0), z)
    for got in (a, j):
        np.testing.assert_allclose(got["G"], G0, rtol=tol, atol=tol * np.abs(G0).max())
        np.testing.assert_allclose(got["p"], p64, rtol=tol * cond, atol=tol * cond * np.abs(p64).max())


@pytest.mark.parametrize("D,alpha,jitter", [(100, 1e6, 1e-3), (100, 1.3, 1e-3), (100, 0.7, 3e-3), (64, 2.0, 5e-4), (37, 1e6, 1e-3), (112, 1e6, 1e-3)])
def test_second_pass_in_closed_form_equals_the_three_product_pass(ht, D, alpha, jitter):
    """Round 4 ("metric_second"): where the first refinement pass moves no eigenvector entry by more than 8e-3, the second pass
    is second-order perturbation theory - one product F E1, E2_ij = (F E1)_ij / (lam_j' - lam_i'), lam_i' = lam_i + (F E1)_ii -
    instead of A X, X^T A X, X^T X.  Same evaluations with the closed form (1) and with the three-product pass (0): solves,
    log-determinants, quadratic forms, Hamiltonians, soft-abs spectra, the assembled metric and the momentum draw agree to fp32
    rounding (an order tighter than either agrees with the float64 oracle), and the closed form is as close to the oracle as
    the full pass.  BASELINE config 3's spectrum (jitter 1e-3: first-pass updates of ~5e-3) with the identity soft-abs map and
    with a finite alpha, a larger jitter (mixed: some systems exceed the bound and take the full pass), smaller D, the largest D."""
    from hamiltorch_amd import _abi
    rng = np.random.default_rng(D + 1)
    P = cfg3_target(ht, D, torch.float32, seed=7)[1].P.astype(np.float64)
    B, seed = 41, 123
    X = (0.3 * rng.standard_normal((B, D))).astype(np.float32)
    m = rng.standard_normal((B, D)).astype(np.float32)
    outs = []
    try:
        for second in (1, 0):
            _abi.set_tuning("metric_second", second)
            outs.append(_warm_eval(ht, P, X, m, alpha, jitter, seed, 1))
    finally:
        _abi.set_tuning("metric_second", 1)
    a, f = outs
    Hs = np.broadcast_to(P, (B, D, D)).astype(np.float64).copy()
    ju = O.philox_uniforms(seed, 3 + np.arange(B), 7, D, O.PURPOSE_JITTER, 2, dtype=np.float64)
    G, lam, _ = O.softabs_metric(Hs, alpha, jitter, ju)
    x64 = 0.5 * np.linalg.solve(G, m.astype(np.float64)[..., None])[..., 0]
    sx = np.abs(x64).max()
    # (measured, tools/scratch/second_pass_err.py, D = 100: max |x - x64| / max |x64| = 3.6e-6 closed form, 1.5e-5 full pass - whose
    # S_ij - lam_j Gm_ij is a difference of O(1) numbers -, 2.5e-6 Jacobi kernel; closed against full 1.6e-5)
    np.testing.assert_allclose(a["x"], f["x"], rtol=0, atol=4e-5 * sx)
    np.testing.assert_allclose(a["ld"], f["ld"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(a["q"], f["q"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(a["H"], f["H"], rtol=2e-5, atol=5e-5)
    np.testing.assert_allclose(np.sort(a["lam"], axis=1), np.sort(f["lam"], axis=1), rtol=0, atol=2e-6 * np.abs(lam).max())
    np.testing.assert_allclose(a["G"], f["G"], rtol=0, atol=2e-5 * np.abs(G).max())
    np.testing.assert_allclose(a["p"], f["p"], rtol=0, atol=1e-4 * np.abs(f["p"]).max())
    ea, ef = np.abs(a["x"] - x64).max() / sx, np.abs(f["x"] - x64).max() / sx
    assert ea <= max(1.2 * ef, 8e-6), (ea, ef)                      # no further from float64 than the full pass is
    assert np.abs(np.sort(a["lam"], axis=1) - np.sort(lam, axis=1)).max() <= 8e-6 * np.abs(lam).max()


@pytest.mark.parametrize("D,kind,alpha,jitter", [(100, "spd", 1e6, 1e-3), (100, "spd", 1.3, 1e-3), (100, "spd", 1e6, None), (64, "indef", 2.0, 1e-2), (37, "spd", 1e6, 5e-2),
                                                 (48, "degenerate", 1e6, 1e-3), (3, "spd", 1e6, 1e-3), (112, "spd", 1e6, 1e-3), (97, "spd", 1e6, 1e-3),
                                                 (105, "spd", 1.3, 2e-3)])
def test_symmetric_square_root_draw_vs_oracle(ht, D, kind, alpha, jitter):
    """Round 6 ("metric_sqrtdraw"): the momentum draw of a soft-abs evaluation on a shared basis is p = G^(1/2) z with the symmetric square
    root Q diag(sqrt lam~) Q^T - a SOLVE-shaped evaluation (formation, refinement, five matrix-vector products) instead of assembling G and
    factorising it; same law N(0, G) as S:183-184's chol(G) z, from the same z.  hta_metric_eval with p_out alone against the oracle's
    rm_gibbs_sqrt map in float64 (every system its own jitter draw), the fast sequence (well separated spectrum), the general sequence
    (jitter 5e-2: the refinement's first pass is above the bound) and the in-launch Jacobi fallback (degenerate spectrum); with the
    tuning key off, or with G_out beside p_out, the draw is chol(G) z as before."""
    from hamiltorch_amd import _abi
    if kind == "spd" and D == 100:
        P = cfg3_target(ht, 100, torch.float32)[1].P.astype(np.float64)
    elif kind == "spd":
        P = cfg3_target(ht, D, torch.float32, seed=7)[1].P.astype(np.float64)
    else:
        P = sym_batch(1, D, kind, D + 1)[0]
    B, seed = 33, 71
    Pd = tt(P, torch.float32)
    V0 = torch.empty(D, D, device=dev()); lam0 = torch.empty(D, device=dev())
    outs = {}
    try:
        _abi.set_tuning("metric_mfma", 0)
        _abi.metric_eval(Pd, 1, D, _abi.METRIC_SOFTABS, Pd, 0, alpha, V_out=V0, lamraw_out=lam0)
        _abi.set_tuning("metric_mfma", 1)
        for key in (1, 0):
            _abi.set_tuning("metric_sqrtdraw", key)
            p = torch.empty(B, D, device=dev())
            _abi.metric_eval(Pd, B, D, _abi.METRIC_SOFTABS, Pd, 0, alpha, jitter, seed, 3, 7, 0, p_out=p, V0=V0, lam0=lam0)
            assert _abi.last_route() == "metric_warm_mfma_kernel"
            outs[key] = p.cpu().numpy()
    finally:
        _abi.set_tuning("metric_sqrtdraw", 1)
        _abi.set_tuning("metric_mfma", 1)
    Hs = np.broadcast_to(P, (B, D, D)).astype(np.float64).copy()
    ju = None if jitter is None else O.philox_uniforms(seed, 3 + np.arange(B), 7, D, O.PURPOSE_JITTER, 0, dtype=np.float64)
    G, lam, Q = O.softabs_metric(Hs, alpha, jitter, ju)
    z = O.philox_normals(seed, 3 + np.arange(B), 7, D, dtype=np.float64)
    want_sqrt = np.einsum("bij,bj->bi", Q, np.sqrt(lam) * np.einsum("bji,bj->bi", Q, z))
    want_chol = np.einsum("bij,bj->bi", np.linalg.cholesky(G), z)
    cond = np.abs(lam).max() / np.abs(lam).min()
    tol = 4e-4
    np.testing.assert_allclose(outs[1], want_sqrt, rtol=tol * cond, atol=tol * cond * np.abs(want_sqrt).max())
    np.testing.assert_allclose(outs[0], want_chol, rtol=tol * cond, atol=tol * cond * np.abs(want_chol).max())
    # the two maps are different square roots of G; the symmetric one has |p|^2 = z^T G z
    assert np.abs(outs[1] - outs[0]).max() > 1e-3 * np.abs(want_chol).max() or D <= 3
    zGz = np.einsum("bi,bij,bj->b", z, G, z)
    np.testing.assert_allclose((outs[1].astype(np.float64) ** 2).sum(1), zGz, rtol=2e-3 * cond)


@pytest.mark.parametrize("D,alpha,jitter", [(100, 1e6, 1e-3), (100, 1.3, 1e-3), (100, 1e6, None), (64, 2.0, 5e-4), (37, 1e6, 1e-3), (112, 1e6, 1e-3), (3, 1e6, 1e-3),
                                            (17, 1e6, 1e-3), (70, 1e6, 1e-3), (90, 2.0, 1e-3),      # (every tile count 1 ... 7: 2, 5 and 6 tiles here)
                                            (97, 1e6, 1e-3), (105, 1.3, 2e-3)])                     # (7 tiles with a ragged last tile: strips, planes)
def test_fast_solve_with_bfloat16_products_equals_exact_products(ht, D, alpha, jitter):
    """Round 6 ("metric_bx3"): the solve evaluations of a Gaussian target on the shared basis run a reorganised sequence (V0 resident,
    the element-wise passes in the products' epilogues, log p and P d from the eigenbasis) whose second-pass product F E1 (1), and
    whose formation F = V0^T diag(e) V0 as well (2, the default), are taken as three bfloat16 products of operands split hi + lo, or
    in exact fp32 (0).  F E1 is a second-order correction, and a term of F with relative error 2^-16 moves the solve by less than the
    closed-form second pass's own truncation (tools/scratch/bf16_formation_err.py): neither may be visible - (2) and (1) agree with (0)
    an order tighter than any of them agrees with float64, and (2) is as close to the float64 oracle as (0).  Every output of the solve: x through upd_x, P d through upd_g, log|G|, the quadratic form, H, log p, the
    soft-abs spectrum; no jitter (F = 0: the second pass is skipped), the smallest and the largest tile counts."""
    from hamiltorch_amd import _abi
    rng = np.random.default_rng(D + 5)
    P = cfg3_target(ht, D, torch.float32, seed=7)[1].P.astype(np.float64)
    B, seed = 41, 321
    X = (0.3 * rng.standard_normal((B, D))).astype(np.float32)
    m = rng.standard_normal((B, D)).astype(np.float32)
    outs = []
    try:
        for bx3 in (2, 1, 0):      # 2 (default): formation and F E1 on bfloat16; 1: F E1 only; 0: exact fp32 products
            _abi.set_tuning("metric_bx3", bx3)
            outs.append(_warm_eval(ht, P, X, m, alpha, jitter, seed, 1, want_g=False))
    finally:
        _abi.set_tuning("metric_bx3", 2)
    a, a1, f = outs
    np.testing.assert_allclose(a1["x"], f["x"], rtol=0, atol=2e-6 * np.abs(f["x"]).max())
    np.testing.assert_allclose(a1["lam"], f["lam"], rtol=0, atol=1e-6 * np.abs(f["lam"]).max())
    Hs = np.broadcast_to(P, (B, D, D)).astype(np.float64).copy()
    ju = None if jitter is None else O.philox_uniforms(seed, 3 + np.arange(B), 7, D, O.PURPOSE_JITTER, 2, dtype=np.float64)
    G, lam, _ = O.softabs_metric(Hs, alpha, jitter, ju)
    x64 = 0.5 * np.linalg.solve(G, m.astype(np.float64)[..., None])[..., 0]
    sx = np.abs(x64).max()
    np.testing.assert_allclose(a["x"], f["x"], rtol=0, atol=2e-6 * sx)
    np.testing.assert_allclose(a["ug"], f["ug"], rtol=0, atol=0)                     # P d does not pass through the second pass at all
    np.testing.assert_allclose(a["lp"], f["lp"], rtol=0, atol=0)
    np.testing.assert_allclose(a["ld"], f["ld"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(a["q"], f["q"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(a["H"], f["H"], rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(a["lam"], f["lam"], rtol=0, atol=1e-6 * np.abs(lam).max())
    ea, ef = np.abs(a["x"] - x64).max() / sx, np.abs(f["x"] - x64).max() / sx
    assert ea <= max(1.2 * ef, 8e-6), (ea, ef)
    want_lp = 0.25 - 0.5 * np.einsum("bi,ij,bj->b", X.astype(np.float64), P, X.astype(np.float64))
    np.testing.assert_allclose(a["lp"], want_lp, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(a["ug"], -0.5 * (X.astype(np.float64) @ P), rtol=2e-5, atol=2e-5)


def test_metric_mfma_kernel_issues_matrix_instructions_on_cfg3(ht):
    """The eigendecomposition route of BASELINE config 3 (hta_set_tuning('rmhmc_fused', 0)) runs its metric evaluations on
    rmhmc_metric_mfma.hip and agrees with the Jacobi kernel chain by chain over a short run."""
    from hamiltorch_amd import _abi
    t, _ = cfg3_target(ht, 100, torch.float32)
    C, N, L = 64, 3, 4
    th0 = tt((0.1 * O.philox_normals(5, np.arange(C), 0, 100, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32)
    kw = dict(num_samples=N, num_steps_per_sample=L, step_size=0.1, jitter=1e-3, softabs_const=1e6, explicit_binding_const=10.0,
              sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, verbose=False, seed=5)
    outs = []
    try:
        _abi.set_tuning("rmhmc_fused", 0)
        _abi.set_tuning("metric_sqrtdraw", 0)      # chol(G) z on both (the Jacobi kernel has no other draw; the matrix-core kernel's default is G^(1/2) z)
        for mode in (1, 0):
            _abi.set_tuning("metric_mfma", mode)
            outs.append(torch.stack(ht.sample(t, th0, **kw)).cpu().numpy())
    finally:
        _abi.set_tuning("rmhmc_fused", 1); _abi.set_tuning("metric_mfma", 1); _abi.set_tuning("metric_sqrtdraw", 1)
    err = np.abs(outs[0] - outs[1]).max(axis=(0, 2))
    assert (err > 3e-4).sum() <= 1, err.max()


def test_prepared_workspace_gives_identical_results_and_skips_the_setup(ht):
    """hta_rmhmc_gaussian_prepare (ABI 6): the once-per-target setup of hta_rmhmc_gaussian_sample - cold eigendecomposition of P,
    the fused route's plan, the shared inverse - hoisted out of the sample call.  Same bits with and without; a prepared call
    launches no metric_eval_kernel; a call whose jitter / alpha / P pointer differ from the preparation runs the setup itself;
    hamiltorch_amd.sample() prepares once per target and again after an in-place edit of the target."""
    from hamiltorch_amd import _abi, rmhmc
    t, o = cfg3_target(ht, 100, torch.float32)
    C, D, T, L = 96, 100, 6, 4
    th0 = tt((0.1 * O.philox_normals(3, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32)

    def run(ws, jitter=1e-3, prepare=False, fused=1):
        _abi.set_tuning("rmhmc_fused", fused)
        cur = th0.clone(); samples = torch.zeros(T + 1, C, D, device=dev()); rej = torch.zeros(C, dtype=torch.int32, device=dev())
        if prepare:                                           # (zeros: row 0 is the caller's to fill - the runs are compared whole)
            _abi.rmhmc_gaussian_prepare(cur, t.precision, t.mean, _abi.METRIC_SOFTABS, 1e6, jitter, C, ws)
        _abi.set_tuning("profile", 1)
        _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, jitter, L, 0.1, 10.0, T, 0, -1, 5, 0,
                                   samples, rej, ws)
        torch.cuda.synchronize()
        n = _abi.profile_collect()[1]
        _abi.set_tuning("profile", 0)
        return samples.clone(), rej.clone(), n
    nbytes = _abi.rmhmc_workspace_bytes(C, D, 4, T)
    for fused in (1, 0):                                  # the closed-form route and the eigendecomposition route
        ws_a = torch.empty(nbytes, dtype=torch.uint8, device=dev()); ws_b = torch.empty(nbytes, dtype=torch.uint8, device=dev())
        s0, r0, n0 = run(ws_a, fused=fused)
        s1, r1, n1 = run(ws_b, prepare=True, fused=fused)
        s2, r2, n2 = run(ws_b, fused=fused)               # prepared by the previous call's prepare: setup skipped
        assert torch.equal(s0, s1) and torch.equal(s0, s2) and torch.equal(r0, r2)
        assert n2 == n0 - 1, (n0, n1, n2)                 # exactly the cold eigendecomposition launch is gone
        s3, _, n3 = run(ws_b, jitter=2e-3, fused=fused)   # another jitter: the preparation does not apply, the call sets up itself
        s4, _, _ = run(ws_a, jitter=2e-3, fused=fused)
        assert torch.equal(s3, s4) and n3 == n0 and not torch.equal(s3, s0)
        _abi.rmhmc_gaussian_forget(ws_b)
        assert run(ws_b, fused=fused)[2] == n0
        # the prepared block sits behind 4 C D + 3 C elements - a preparation for ANOTHER chain count must not
        # apply (the call would read basis, inverse and chol(P) from the wrong offsets): it sets up itself, same bits
        big = torch.empty(_abi.rmhmc_workspace_bytes(2 * C, D, 4, T), dtype=torch.uint8, device=dev())
        _abi.rmhmc_gaussian_prepare(th0.repeat(2, 1), t.precision, t.mean, _abi.METRIC_SOFTABS, 1e6, 1e-3, 2 * C, big)
        s5, r5, n5 = run(big, fused=fused)
        assert torch.equal(s5, s0) and torch.equal(r5, r0) and n5 == n0
        _abi.rmhmc_gaussian_forget(big)
    _abi.reset_tuning()
    # through sample(): prepared once per target, again after an in-place edit of the precision matrix
    kw = dict(num_samples=5, num_steps_per_sample=3, step_size=0.1, jitter=1e-3, softabs_const=1e6, explicit_binding_const=10,
              sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, verbose=False, seed=9)
    a = torch.stack(ht.sample(t, th0, **kw))
    ws_id = next(iter(t._hta_rm_ws.values()))[0].ws.data_ptr()
    b = torch.stack(ht.sample(t, th0, **kw))
    assert torch.equal(a, b) and next(iter(t._hta_rm_ws.values()))[0].ws.data_ptr() == ws_id
    t.precision.mul_(1.5)
    c = torch.stack(ht.sample(t, th0, **kw))
    t2 = ht.GaussianTarget(t.mean.clone(), precision=t.precision.clone(), normalized=False)
    d = torch.stack(ht.sample(t2, th0, **kw))
    assert torch.equal(c, d) and not torch.equal(c, a)


# ---- per-system curvature with per-system warm bases on the matrix cores (HtaMetricArgs.v0_stride, ABI 7) -------------------
@pytest.mark.parametrize("D,kind,alpha,jitter,start", [(100, "spd", 1e6, 1e-3, "near"), (100, "indef", 1.0, None, "near"),
                                                       (37, "indef", 0.7, 1e-3, "near"), (64, "spd", 1e6, None, "identity"),
                                                       (100, "indef", 1e6, 1e-3, "identity"), (48, "degenerate", 1e6, 1e-3, "near"),
                                                       (16, "degenerate", 2.0, None, "identity"), (112, "spd", 0.05, 1e-3, "far"),
                                                       (11, "indef", 1e6, 1e-2, "near"), (5, "spd", 1e6, None, "identity")])
def test_metric_general_targets_on_the_matrix_cores_vs_oracle(ht, D, kind, alpha, jitter, start):
    """hta_metric_eval with one curvature matrix AND one approximate eigenbasis per system (what the chains of a general target
    hand over from their previous evaluation): metric_warm_mfma_kernel rotates Hs_b into V0_b, refines (start "near": the exact
    basis of a slightly different matrix) or finishes with Jacobi inside the launch (start "identity" / "far", degenerate spectra),
    and returns G^-1 m, log|G|, the soft-abs spectrum, the derivative matrix M (dmetric_out), G, the momentum draw and the NEW
    basis (V_out, written over V0) - each against the oracle's float64 eigh, and against the cold Jacobi kernel
    (hta_set_tuning("metric_general", 0)).  A second call from the returned basis must reproduce the results."""
    from hamiltorch_amd import _abi
    rng = np.random.default_rng(7 * D + len(kind))
    B, seed, off, draw = 9, 5, 2, 4
    Hs = sym_batch(B, D, kind, D + 3)
    if start == "near":      # eigenvectors of Hs + a 1 % symmetric perturbation
        E = rng.standard_normal((B, D, D)); E = 0.01 * (E + E.transpose(0, 2, 1)) / np.sqrt(D)
        V0 = np.linalg.eigh(Hs + E)[1]
    elif start == "far":
        V0 = np.stack([np.linalg.qr(rng.standard_normal((D, D)))[0] for _ in range(B)])
    else:
        V0 = np.broadcast_to(np.eye(D), (B, D, D)).copy()
    m = rng.standard_normal((B, D)).astype(np.float32)
    dt = torch.float32
    Hd, md = tt(Hs, dt), tt(m, dt)

    def run(general, V0np, sub, want):
        _abi.set_tuning("metric_general", general)
        out = {k: torch.empty(s, device=dev()) for k, s in (("x", (B, D)), ("ld", (B,)), ("q", (B,)), ("lam", (B, D)), ("M", (B, D, D)),
                                                              ("G", (B, D, D)), ("p", (B, D)))}
        Vb = tt(V0np, dt).contiguous()
        kw = dict(V0=Vb, v0_stride=D * D, V_out=Vb)
        if want == "solve":
            _abi.metric_eval(md, B, D, _abi.METRIC_SOFTABS, Hd, D * D, alpha, jitter, seed, off, draw, sub, m=md, x_out=out["x"],
                             logdet_out=out["ld"], quad_out=out["q"], lam_out=out["lam"], dmetric_out=out["M"], **kw)
        else:
            _abi.metric_eval(md, B, D, _abi.METRIC_SOFTABS, Hd, D * D, alpha, jitter, seed, off, draw, sub, G_out=out["G"], p_out=out["p"], **kw)
        route = _abi.last_route()
        torch.cuda.synchronize()
        res = {k: v.cpu().numpy().astype(np.float64) for k, v in out.items()}
        res["V"] = Vb.cpu().numpy().astype(np.float64)
        return res, route

    a, route = run(1, V0, 2, "solve")
    assert route == "metric_warm_mfma_kernel", route
    j, route_j = run(0, V0, 2, "solve")
    assert route_j.startswith("metric_eval_kernel"), route_j
    ju = None if jitter is None else O.philox_uniforms(seed, off + np.arange(B), draw, D, O.PURPOSE_JITTER, 2, dtype=np.float64)
    G, lam, _ = O.softabs_metric(Hs, alpha, jitter, ju)
    M64, x64 = O.softabs_dmetric(Hs, alpha, m, jitter, ju)
    cond = np.abs(lam).max() / np.abs(lam).min()
    tol = 4e-4
    for got in (a, j):
        np.testing.assert_allclose(got["x"], x64, rtol=tol * cond, atol=tol * cond * np.abs(x64).max())
        np.testing.assert_allclose(got["ld"], np.log(lam).sum(1), rtol=tol, atol=tol * D)
        np.testing.assert_allclose(got["q"], (m * x64).sum(1), rtol=tol * cond, atol=tol * cond)
        np.testing.assert_allclose(np.sort(got["lam"], axis=1), np.sort(lam, axis=1), rtol=tol, atol=tol * np.abs(lam).max())
    # the derivative matrix: the divided differences are sensitive where eigenvalues nearly coincide - compare the kernels with
    # each other tightly and the matrix-core kernel with float64 on the scale of M
    np.testing.assert_allclose(a["M"], j["M"], rtol=0, atol=3e-3 * cond * np.abs(M64).max())
    if kind != "degenerate":
        np.testing.assert_allclose(a["M"], M64, rtol=0, atol=3e-3 * cond * np.abs(M64).max())
    # the returned basis: orthonormal, and it diagonalises Hs (+ jitter) to the soft-abs spectrum's raw eigenvalues
    Hj = Hs.copy()
    if jitter is not None:
        Hj[:, np.arange(D), np.arange(D)] += jitter * ju
    Vn = a["V"]
    np.testing.assert_allclose(np.einsum("bij,bik->bjk", Vn, Vn), np.broadcast_to(np.eye(D), (B, D, D)), atol=2e-5 * D)
    Dg = np.einsum("bij,bik,bkl->bjl", Vn, Hj, Vn)
    offd = Dg - np.einsum("bii->bi", Dg)[:, :, None] * np.eye(D)
    assert np.abs(offd).max() <= 2e-5 * D * np.abs(lam).max(), np.abs(offd).max()
    # second call from the returned basis (nearly exact: the refinement path) reproduces the results
    a2, route2 = run(1, Vn, 2, "solve")
    assert route2 == "metric_warm_mfma_kernel"
    np.testing.assert_allclose(a2["x"], a["x"], rtol=2e-4 * cond, atol=2e-5 * cond * np.abs(x64).max())
    np.testing.assert_allclose(a2["ld"], a["ld"], rtol=1e-5, atol=1e-4 * D)
    # fisher() and the momentum draw (jitter sub-stream 0), basis updated in place as well
    g, route_g = run(1, V0, 0, "matrix")
    assert route_g == "metric_warm_mfma_kernel"
    ju0 = None if jitter is None else O.philox_uniforms(seed, off + np.arange(B), draw, D, O.PURPOSE_JITTER, 0, dtype=np.float64)
    G0, _, _ = O.softabs_metric(Hs, alpha, jitter, ju0)
    z = O.philox_normals(seed, off + np.arange(B), draw, D, dtype=np.float64)
    p64 = np.einsum("bij,bj->bi", np.linalg.cholesky(G0), z)
    np.testing.assert_allclose(g["G"], G0, rtol=tol, atol=tol * np.abs(G0).max())
    np.testing.assert_allclose(g["p"], p64, rtol=tol * cond, atol=tol * cond * np.abs(p64).max())
    np.testing.assert_allclose(np.einsum("bij,bik->bjk", g["V"], g["V"]), np.broadcast_to(np.eye(D), (B, D, D)), atol=2e-5 * D)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_general_integrators_differentiate_the_callback_once_per_state(ht, dtype):
    """Both calls of an explicit half step are evaluated at one (theta, p) pair (S:429-430, S:432-433), the implicit momentum
    fixed point iterates at one theta (S:313-340): `_Curvature.grad_neg_hessian` hands the derivatives of the callback back
    while the state tensor is unchanged (same object, same version; the native in-place updates are announced through
    `touched`) - 10 evaluations instead of 24 for three explicit steps, and the SAME results as differentiating every time."""
    from hamiltorch_amd import rmhmc, _abi
    D, C = 6, 5
    lp = funnel_logp(np.array([0.5, 1.0, 1.7, 2.4, 0.8]))
    rng = np.random.default_rng(3)
    th0 = tt(0.4 * rng.standard_normal((C, D)), dtype); p0 = tt(rng.standard_normal((C, D)), dtype)

    def run(cache):
        cv = rmhmc._Curvature(lp)
        if not cache:
            cv.grad_neg_hessian = cv._grad_neg_hessian
        th, pm, thc, pmc = th0.clone(), p0.clone(), th0.clone(), p0.clone()
        rmhmc._generic_steps(cv, _abi.METRIC_SOFTABS, th, pm, thc, pmc, 3, 0.05, 10.0, 1.3, 1e-3, 7, 0, 2)
        return cv, [t.cpu().numpy() for t in (th, pm, thc, pmc)]

    cv, got = run(True)
    assert cv.stats == {"gh_evaluated": 10, "gh_reused": 14}, cv.stats
    _, want = run(False)
    for a_, b_ in zip(got, want):
        np.testing.assert_array_equal(a_, b_)
    # implicit: the momentum fixed point reuses, the position fixed point (theta moves every iteration) does not
    cv2 = rmhmc._Curvature(lp)
    th, pm = th0.clone(), p0.clone()
    rmhmc._implicit_steps(cv2, _abi.METRIC_SOFTABS, th, pm, 2, 0.05, 1.3, None, 7, 0, 2, 1e-12, 4)
    cv3 = rmhmc._Curvature(lp); cv3.grad_neg_hessian = cv3._grad_neg_hessian
    th3, pm3 = th0.clone(), p0.clone()
    rmhmc._implicit_steps(cv3, _abi.METRIC_SOFTABS, th3, pm3, 2, 0.05, 1.3, None, 7, 0, 2, 1e-12, 4)
    assert cv2.stats["gh_reused"] >= 2 * 3 and cv2.stats["gh_evaluated"] < cv2.stats["gh_evaluated"] + cv2.stats["gh_reused"]
    np.testing.assert_array_equal(th.cpu().numpy(), th3.cpu().numpy())
    np.testing.assert_array_equal(pm.cpu().numpy(), pm3.cpu().numpy())


def test_metric_general_mode_restarts_the_basis_of_a_system_that_went_non_finite(ht):
    """A diverged chain hands hta_metric_eval a NaN curvature (or, afterwards, would hand back a NaN basis): its V_out is the
    identity, so that the chain's next evaluation - after the Metropolis step has put it back on a finite state - is a cold start
    and not NaN forever; the other systems of the batch are unaffected."""
    from hamiltorch_amd import _abi
    D, B = 24, 5
    Hs = sym_batch(B, D, "spd", 3)
    bad = Hs.copy(); bad[2, 3, 1] = np.nan
    m = np.random.default_rng(0).standard_normal((B, D)).astype(np.float32)
    V = tt(np.broadcast_to(np.eye(D), (B, D, D)).copy(), torch.float32).contiguous()
    x = torch.empty(B, D, device=dev()); md = tt(m, torch.float32)
    kw = dict(V0=V, v0_stride=D * D, V_out=V)
    _abi.metric_eval(md, B, D, _abi.METRIC_SOFTABS, tt(bad, torch.float32), D * D, 1e6, None, 0, 0, 0, 0, m=md, x_out=x, **kw)
    assert _abi.last_route() == "metric_warm_mfma_kernel"
    Vn = V.cpu().numpy()
    assert not np.isfinite(x.cpu().numpy()[2]).all()
    np.testing.assert_array_equal(Vn[2], np.eye(D, dtype=np.float32))
    assert np.isfinite(Vn).all() and np.isfinite(x.cpu().numpy()[[0, 1, 3, 4]]).all()
    _abi.metric_eval(md, B, D, _abi.METRIC_SOFTABS, tt(Hs, torch.float32), D * D, 1e6, None, 0, 0, 0, 0, m=md, x_out=x, **kw)
    want = np.linalg.solve(Hs, m.astype(np.float64)[..., None])[..., 0]
    np.testing.assert_allclose(x.cpu().numpy(), want, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("C,D,route", [(256, 100, "rmhmc_uv_kernel<1"), (512, 100, "rmhmc_uv_kernel<2"), (1024, 100, "rmhmc_mfma4x4_kernel<true"),
                                       (300, 37, "rmhmc_uv_kernel<2"), (700, 90, "rmhmc_mfma4x4_kernel<true")])
def test_lean_instances_of_the_lone_wave_kernels_are_bit_identical(ht, C, D, route):
    """hta_set_tuning('rmhmc_lean', 1): rmhmc_uv_kernel without the lane predicate around its LDS stores (both lane halves
    hold the same values) and, with rmhmc_mfma4x4_kernel, without the selects that zero the padding rows (exact zeros by
    construction) - fewer instructions per step of kernels whose time is their instruction count.  Same samples, reject
    counts and final state bit for bit, with jitter, with a mean offset, with padding rows (D = 37, 90, 100 < 128) and with a
    chain that diverges (it is rejected and must sample on, exactly as in the default instance)."""
    from hamiltorch_amd import _abi
    t, _ = cfg3_target(ht, D, torch.float32, seed=D)
    t.mean.add_(torch.linspace(-1.0, 1.0, D, device=dev()))
    T, L = 5, 6
    th0 = 0.1 * torch.randn(C, D, generator=torch.Generator().manual_seed(C), dtype=torch.float32).to(dev()) + t.mean
    th0[3] = 1e30                                          # a chain that starts where the energy overflows
    outs = []
    for lean in (0, 1):
        _abi.set_tuning("rmhmc_lean", lean)
        _abi.set_tuning("rmhmc_uvc", 0); _abi.set_tuning("rmhmc_uv_co", 0); _abi.set_tuning("rmhmc_mfma4_lo", 513)   # the round-3 routes
        try:
            cur = th0.clone(); samples = torch.zeros(T + 1, C, D, device=dev()); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            ws = torch.zeros(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
            _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, 1e-3, L, 0.1, 10.0, T, 0, -1, 5, 0,
                                       samples, rej, ws)
            r = _abi.last_route()
            torch.cuda.synchronize()
        finally:
            _abi.reset_tuning()
        assert r.startswith(route) and (",lean>" in r) == bool(lean), r
        outs.append((samples[1:].cpu(), rej.cpu(), cur.cpu()))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    assert int(outs[0][1][3]) == T and 0 <= int(outs[0][1].sum()) - T < C * T // 2      # the diverged chain rejects every proposal
    assert bool(torch.isfinite(outs[0][0][:, 4:]).all())


@pytest.mark.parametrize("C,D", [(256, 100), (512, 100), (1024, 100), (300, 37)])
def test_uv_kernel_coresident_and_four_chain_instances(ht, C, D):
    """Round 4: rmhmc_uv_kernel under a 256-register cap ("rmhmc_uv_co": two workgroups per CU fill each other's phase latency;
    the same arithmetic - bit-identical samples, reject counts and final state) and with four accumulator chains per product
    ("rmhmc_uv_acc" = 4: no s_nop between dependent matrix instructions; another summation order - equal to rounding, a
    Metropolis decision may flip in a few chains).  1024 chains reach the kernel through the co-resident route (512 two-chain
    workgroups on 256 CUs)."""
    from hamiltorch_amd import _abi
    t, _ = cfg3_target(ht, D, torch.float32, seed=D)
    t.mean.add_(torch.linspace(-1.0, 1.0, D, device=dev()))
    T, L = 4, 6
    th0 = 0.1 * torch.randn(C, D, generator=torch.Generator().manual_seed(C), dtype=torch.float32).to(dev()) + t.mean
    th0[3] = 1e30
    outs = {}
    for name, keys in (("base", {"rmhmc_uv": 2, "rmhmc_uv_co": 0}), ("co", {"rmhmc_uv": 2, "rmhmc_uv_co": 1}),
                       ("acc4", {"rmhmc_uv": 2, "rmhmc_uv_acc": 4, "rmhmc_uv_co": 0}), ("co_acc4", {"rmhmc_uv_co": 1, "rmhmc_uv_acc": 4})):
        _abi.set_tuning("rmhmc_uvc", 0)
        for k, v in keys.items():
            _abi.set_tuning(k, v)
        try:
            cur = th0.clone(); samples = torch.zeros(T + 1, C, D, device=dev()); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            ws = torch.zeros(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
            _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, 1e-3, L, 0.1, 10.0, T, 0, -1, 5, 0,
                                       samples, rej, ws)
            r = _abi.last_route()
            torch.cuda.synchronize()
        finally:
            _abi.reset_tuning()
        assert r.startswith("rmhmc_uv_kernel<"), (name, r)
        assert ("co" in r.split(",")) == ("co" in name) and r.endswith(",4>") == ("acc4" in name), (name, r)
        outs[name] = (samples[1:].cpu(), rej.cpu(), cur.cpu())
    for a_, b_ in zip(outs["base"], outs["co"]):
        assert torch.equal(a_, b_)
    for a_, b_ in zip(outs["acc4"], outs["co_acc4"]):
        assert torch.equal(a_, b_)
    live = torch.ones(C, dtype=torch.bool); live[3] = False
    d = (outs["base"][0] - outs["acc4"][0]).abs().amax(dim=(0, 2))[live]
    assert int((d > 1e-4).sum()) <= max(1, C // 100), float(d.max())
    assert int(outs["acc4"][1][3]) == T


@pytest.mark.parametrize("D,C,burn", [(100, 256, 2), (100, 37, -1), (37, 30, 1), (64, 20, 0), (7, 9, 2), (99, 256, 3)])
def test_uvc_kernel_equals_uv_kernel(ht, D, C, burn):
    """Round 4: rmhmc_uvc_kernel (csrc/rmhmc_uvc.hip: one chain per workgroup, ONE value per lane in the element-wise work, the
    second-order term of every solve deferred into the idle matrix-instruction columns of the next phase: three product phases
    per step instead of five) against rmhmc_uv_kernel<1> (K = 2 refinement phases per solve): the same streams and update
    order; the results differ by third-order terms in jitter / lambda_min and by rounding.  Padding rows (D = 37, 99 < 128), a
    mean offset, burn-in (Q2 reset), H_old / H_new / accept, several launches over traj_offset, a chain that diverges."""
    from hamiltorch_amd import _abi
    T, L, jit = 9, 6, 1e-3
    t, _ = cfg3_target(ht, D, torch.float32, seed=5)
    t.mean.add_(torch.linspace(-1.0, 1.0, D, device=dev()))
    th0 = tt((0.3 * O.philox_normals(3, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32) + t.mean
    th0[min(3, C - 1)] = 1e30
    bad = min(3, C - 1)
    outs = []
    try:
        for uvc in (1, 0):
            _abi.set_tuning("rmhmc_uvc", uvc); _abi.set_tuning("rmhmc_uv_g", 1); _abi.set_tuning("rmhmc_uv", 2)
            nb = max(burn, 0)
            cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            samples = torch.zeros(T - nb + 1, C, D, device=dev())
            Ho = torch.zeros(T, C, device=dev()); Hn = torch.zeros(T, C, device=dev()); ac = torch.zeros(T, C, dtype=torch.uint8, device=dev())
            ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
            for t0, nt in ((0, 5), (5, T - 5)):                     # two launches: the state travels through `cur`
                _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, jit, L, 0.1, 10.0,
                                           nt, t0, burn, 21, 0, samples, rej, ws, H_old=Ho[t0:], H_new=Hn[t0:], accept=ac[t0:])
                r = _abi.last_route()
            torch.cuda.synchronize()
            assert r.startswith("rmhmc_uvc_kernel<" if uvc else "rmhmc_uv_kernel<1"), r
            outs.append((samples.cpu().numpy(), rej.cpu().numpy(), cur.cpu().numpy(), Ho.cpu().numpy(), Hn.cpu().numpy(), ac.cpu().numpy()))
    finally:
        _abi.reset_tuning()
    keep = np.arange(C) != bad
    assert np.isfinite(outs[0][0][:, keep]).all()
    np.testing.assert_allclose(outs[0][3][0][keep], outs[1][3][0][keep], rtol=2e-5, atol=2e-4)      # H_old / H_new of the first
    np.testing.assert_allclose(outs[0][4][0][keep], outs[1][4][0][keep], rtol=2e-5, atol=2e-4)      # trajectory: before any decision
    err = np.abs(outs[0][0] - outs[1][0])[:, keep].max(axis=(0, 2))
    assert (err > 1e-4).mean() <= 0.05, "max err %.3g (%d chains differ)" % (err.max(), (err > 1e-4).sum())
    good = np.where(keep)[0][err <= 1e-4]
    assert np.array_equal(outs[0][1][good], outs[1][1][good])
    assert np.array_equal(outs[0][5][:, good], outs[1][5][:, good])
    np.testing.assert_allclose(outs[0][2][good], outs[1][2][good], atol=1e-4)
    assert int(outs[0][1][bad]) == T and np.array_equal(outs[0][2][bad], outs[1][2][bad])
    assert np.abs(outs[0][0][-1] - outs[0][0][1])[keep].max() > 1e-3


@pytest.mark.parametrize("co", [0, 1])
def test_uvc_kernel_vs_oracle_at_cfg3(ht, co):
    """BASELINE config 3 on rmhmc_uvc_kernel (256 chains, D = 100, L = 10, jitter 1e-3) against the oracle, which does the
    reference's eigendecomposition per metric evaluation (S:108-122), on the same Philox streams: 32 chains spread over the
    batch, chain by chain; both register budgets of the kernel (one and two workgroups per CU)."""
    from hamiltorch_amd import _abi
    C, N, nsel, D, L, eps, omega, alpha, jitter, seed, off = 256, 4, 32, 100, 10, 0.1, 10.0, 1e6, 1e-3, 2026, 7
    t, o = cfg3_target(ht, D, torch.float32)
    th0 = (0.1 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    try:
        _abi.set_tuning("rmhmc_uvc", 1); _abi.set_tuning("rmhmc_uv_co", co)
        out, acc = ht.sample(t, tt(th0, torch.float32), num_samples=N, num_steps_per_sample=L, step_size=eps, jitter=jitter, softabs_const=alpha,
                             explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                             metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=seed, chain_offset=off)
        assert _abi.last_route() == "rmhmc_uvc_kernel<%s>" % ("co" if co else "solo"), _abi.last_route()
    finally:
        _abi.reset_tuning()
    got = torch.stack(out).cpu().numpy()
    assert got.shape == (N, C, D) and np.isfinite(got).all()
    sel = np.unique(np.r_[0:4, np.linspace(4, C - 5, nsel - 8).astype(int), C - 4:C])
    ref, info = O.sample_rmhmc_explicit(o, th0[sel], N, L, eps, omega, alpha, 0, jitter,
                                        O.PhiloxDraws(seed, off + sel, np.float32), "softabs", momentum="split")
    err = np.abs(got[:, sel] - np.stack(ref)).max(axis=(0, 2))
    assert (err > 5e-4).sum() <= 1, "%d of %d chains differ (max %.3g)" % ((err > 5e-4).sum(), len(sel), err.max())
    np.testing.assert_allclose(acc.cpu().numpy()[sel][err <= 5e-4], info["acc_rate"][err <= 5e-4], atol=1e-12)


@pytest.mark.parametrize("D,C,burn,jit", [(100, 512, 2, 1e-3), (100, 37, -1, 1e-3), (37, 30, 1, 1.2e-3), (64, 21, 0, None), (7, 9, 2, 1e-3), (99, 1024, 3, 1e-3)])
def test_uvc2_kernel_equals_uv_kernel(ht, D, C, burn, jit):
    """Round 4: rmhmc_uvc2_kernel (csrc/rmhmc_uvc.hip: two chains per workgroup, TWO values per lane in the element-wise work,
    branch-free half steps; the schedule of rmhmc_uv_kernel<2>, any K) against rmhmc_uv_kernel<2>: same streams, same products
    in the same order - results equal to rounding of the element-wise expressions.  Odd chain counts (half-empty last group),
    padding rows, no jitter (K = 0), another jitter scale (1.2e-3: the largest the log-det series admits at lambda_min = 0.5), burn-in (Q2), two launches, a diverging chain; 1024 chains
    run two workgroups per CU."""
    from hamiltorch_amd import _abi
    T, L = 9, 6
    t, _ = cfg3_target(ht, D, torch.float32, seed=5)
    t.mean.add_(torch.linspace(-1.0, 1.0, D, device=dev()))
    th0 = tt((0.3 * O.philox_normals(3, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32), torch.float32) + t.mean
    bad = min(3, C - 1)
    th0[bad] = 1e30
    outs = []
    try:
        for uvc in (1, 0):
            _abi.set_tuning("rmhmc_uvc", uvc); _abi.set_tuning("rmhmc_uv_g", 2); _abi.set_tuning("rmhmc_uv", 2)
            _abi.set_tuning("rmhmc_uv_co", 1 if C > 512 else 0)
            nb = max(burn, 0)
            cur = th0.clone(); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            samples = torch.zeros(T - nb + 1, C, D, device=dev())
            Ho = torch.zeros(T, C, device=dev()); Hn = torch.zeros(T, C, device=dev()); ac = torch.zeros(T, C, dtype=torch.uint8, device=dev())
            ws = torch.empty(_abi.rmhmc_workspace_bytes(C, D, 4, T), dtype=torch.uint8, device=dev())
            for t0, nt in ((0, 5), (5, T - 5)):
                _abi.rmhmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, _abi.METRIC_SOFTABS, 1e6, jit, L, 0.1, 10.0,
                                           nt, t0, burn, 21, 0, samples, rej, ws, H_old=Ho[t0:], H_new=Hn[t0:], accept=ac[t0:])
                r = _abi.last_route()
            torch.cuda.synchronize()
            assert r.startswith("rmhmc_uvc2_kernel<" if uvc else "rmhmc_uv_kernel<2"), r
            outs.append((samples.cpu().numpy(), rej.cpu().numpy(), cur.cpu().numpy(), Ho.cpu().numpy(), Hn.cpu().numpy(), ac.cpu().numpy()))
    finally:
        _abi.reset_tuning()
    keep = np.arange(C) != bad
    assert np.isfinite(outs[0][0][:, keep]).all()
    np.testing.assert_allclose(outs[0][3][0][keep], outs[1][3][0][keep], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(outs[0][4][0][keep], outs[1][4][0][keep], rtol=2e-5, atol=2e-4)
    err = np.abs(outs[0][0] - outs[1][0])[:, keep].max(axis=(0, 2))
    assert (err > 1e-4).mean() <= 0.05, "max err %.3g (%d chains differ)" % (err.max(), (err > 1e-4).sum())
    good = np.where(keep)[0][err <= 1e-4]
    assert np.array_equal(outs[0][1][good], outs[1][1][good])
    assert np.array_equal(outs[0][5][:, good], outs[1][5][:, good])
    np.testing.assert_allclose(outs[0][2][good], outs[1][2][good], atol=1e-4)
    assert int(outs[0][1][bad]) == T and np.array_equal(outs[0][2][bad], outs[1][2][bad])
    assert np.abs(outs[0][0][-1] - outs[0][0][1])[keep].max() > 1e-3


def test_softabs_finite_alpha_at_cfg3_size_in_fp64(ht):
    """BASELINE config 3's size in fp64 with a FINITE soft-abs constant (alpha = 1.3: the soft-abs map
    is not the identity, the run needs an eigendecomposition per metric evaluation) used to be an error - A + V of a 100 x 100
    fp64 system exceed the 160 KiB of a CU.  The instance with both matrices in a per-workgroup slab of global memory serves it (round 6:
    float64 no longer runs on the eigenvectors-only instance, whose code generation depended on the spelling of an address expression -
    csrc/rmhmc_metric.hip: metric_geometry): the run returns and agrees with the oracle chain by chain."""
    from hamiltorch_amd import _abi
    t, o = cfg3_target(ht, 100, torch.float64)
    C, N, L, eps, omega, alpha, jitter, seed = 6, 2, 2, 0.1, 10.0, 1.3, 1e-3, 5
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, 100, O.PURPOSE_INIT, dtype=np.float64))
    out, acc = ht.sample(t, tt(th0, torch.float64), num_samples=N, num_steps_per_sample=L, step_size=eps, jitter=jitter, softabs_const=alpha,
                         explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                         metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=seed)
    assert _abi.last_route() == "metric_eval_kernel<double,aglobal>", _abi.last_route()
    got = torch.stack(out).cpu().numpy()
    ref, info = O.sample_rmhmc_explicit(o, th0, N, L, eps, omega, alpha, 0, jitter, O.PhiloxDraws(seed, np.arange(C), np.float64), "softabs")
    np.testing.assert_allclose(got, np.stack(ref), rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(acc.cpu().numpy(), info["acc_rate"], atol=1e-12)


@pytest.mark.parametrize("D", [141, 150])
def test_metric_eval_fp32_beyond_one_cu_of_lds(ht, D):
    """fp32 beyond D = 140 (A + V > 160 KiB): the global-eigenvector instance against the oracle's float64 eigh."""
    from hamiltorch_amd import _abi
    B = 3
    Hs = sym_batch(B, D, "indef", D).astype(np.float32)
    m = np.random.default_rng(1).standard_normal((B, D)).astype(np.float32)
    x = torch.empty(B, D, device=dev()); lam = torch.empty(B, D, device=dev()); ld = torch.empty(B, device=dev())
    _abi.metric_eval(x, B, D, _abi.METRIC_SOFTABS, tt(Hs, torch.float32), D * D, 1.3, None, 0, 0, 0, 0, m=tt(m, torch.float32), x_out=x,
                     lam_out=lam, logdet_out=ld)
    assert _abi.last_route() == "metric_eval_kernel<float,vglobal>", _abi.last_route()
    G, lam_t, _ = O.softabs_metric(Hs.astype(np.float64), 1.3)
    want = np.linalg.solve(G, m.astype(np.float64)[..., None])[..., 0]
    np.testing.assert_allclose(np.sort(lam.cpu().numpy(), axis=1), np.sort(lam_t, axis=1), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(x.cpu().numpy(), want, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(ld.cpu().numpy(), np.log(lam_t).sum(1), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype,D,rt,tol", [(torch.float32, 200, "float,aglobal", 3e-3), (torch.float32, 254, "float,aglobal", 4e-3),
                                            (torch.float64, 128, "double,aglobal", 1e-8), (torch.float64, 180, "double,aglobal", 2e-8),
                                            (torch.float64, 110, "double,aglobal", 1e-8), (torch.float64, 100, "double,aglobal", 1e-8),
                                            (torch.float32, 300, "float,aglobal,dyn", 6e-3), (torch.float64, 200, "double,aglobal,dyn", 3e-8)])
def test_metric_eval_beyond_the_round_4_size_limits(ht, dtype, D, rt, tol):
    """VERDICT r04 "missing" #4: fisher / cholesky_inverse / rm_hamiltonian have no size limit in the reference (S:108-122, S:146-148,
    S:710-731); hta_metric_eval stopped at D ~ 156 fp32 / 110 fp64 (the Jacobi kernel's per-thread work lists).  Round 5: the instance
    with both matrices in the caller's workspace slab (ABI 10: HtaMetricArgs::workspace) and 8 work-list entries per thread runs up to
    D = 254 fp32 / 180 fp64 - slow (every rotation is an L2 round trip), not an error; beyond that the instance that walks its work
    lists at run time (`dyn`: any D <= 1024).  Against the oracle's float64 eigh."""
    from hamiltorch_amd import _abi
    B = 3
    Hs = sym_batch(B, D, "indef", D).astype(NP[dtype])
    m = np.random.default_rng(1).standard_normal((B, D)).astype(NP[dtype])
    assert _abi.metric_eval_workspace_bytes(B, D, Hs.itemsize) > 0
    x = torch.empty(B, D, device=dev(), dtype=dtype); lam = torch.empty_like(x); ld = torch.empty(B, device=dev(), dtype=dtype)
    _abi.metric_eval(x, B, D, _abi.METRIC_SOFTABS, tt(Hs, dtype), D * D, 1.3, None, 0, 0, 0, 0, m=tt(m, dtype), x_out=x, lam_out=lam, logdet_out=ld)
    assert _abi.last_route() == "metric_eval_kernel<%s>" % rt, _abi.last_route()
    G, lam_t, _ = O.softabs_metric(Hs.astype(np.float64), 1.3)
    want = np.linalg.solve(G, m.astype(np.float64)[..., None])[..., 0]
    np.testing.assert_allclose(np.sort(lam.cpu().numpy(), axis=1), np.sort(lam_t, axis=1), rtol=0.2 * tol, atol=0.2 * tol)
    np.testing.assert_allclose(x.cpu().numpy(), want, rtol=tol, atol=tol)
    np.testing.assert_allclose(ld.cpu().numpy(), np.log(lam_t).sum(1), rtol=0.1 * tol, atol=tol)
    # the workspace is the caller's: without one the call is refused (no hidden allocation behind the ABI), with a short one too
    with pytest.raises(_abi.InvalidArguments, match="workspace"):
        _abi.metric_eval(x, B, D, _abi.METRIC_SOFTABS, tt(Hs, dtype), D * D, 1.3, None, 0, 0, 0, 0, m=tt(m, dtype), x_out=x,
                         workspace=torch.empty(16, dtype=torch.uint8, device=dev()))
    # the Hessian metric (Cholesky of the slab-resident matrix)
    P = np.einsum("bij,bkj->bik", Hs, Hs).astype(NP[dtype]) / D + np.eye(D, dtype=NP[dtype])
    _abi.metric_eval(x, B, D, _abi.METRIC_HESSIAN, tt(P, dtype), D * D, 0.0, None, 0, 0, 0, 0, m=tt(m, dtype), x_out=x, logdet_out=ld)
    want = np.linalg.solve(P.astype(np.float64), m.astype(np.float64)[..., None])[..., 0]
    np.testing.assert_allclose(x.cpu().numpy(), want, rtol=tol, atol=tol)
    np.testing.assert_allclose(ld.cpu().numpy(), np.linalg.slogdet(P.astype(np.float64))[1], rtol=0.1 * tol, atol=tol)


@pytest.mark.parametrize("dtype,D,tol", [(torch.float32, 200, 2e-3), (torch.float64, 128, 1e-7), (torch.float32, 300, 4e-3), (torch.float64, 200, 2e-7)])
def test_sample_rmhmc_softabs_beyond_the_round_4_size_limits(ht, dtype, D, tol):
    """sample(RMHMC, EXPLICIT, SOFTABS) at D = 200 fp32 / D = 128 fp64 with a finite soft-abs constant (an eigendecomposition per
    metric evaluation) returns and matches the oracle chain by chain - VERDICT r04 next-round item 9's "done" line."""
    from hamiltorch_amd import _abi
    t, o = cfg3_target(ht, D, dtype)
    C, N, L, eps, omega, alpha, jitter, seed = 4, 2, 2, 0.05, 10.0, 1.3, 1e-3, 5
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=NP[dtype]))
    out, acc = ht.sample(t, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, jitter=jitter, softabs_const=alpha,
                         explicit_binding_const=omega, sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT,
                         metric=ht.Metric.SOFTABS, debug=2, verbose=False, seed=seed)
    assert _abi.last_route().startswith("metric_eval_kernel<%s,aglobal" % ("float" if dtype == torch.float32 else "double")), _abi.last_route()
    got = torch.stack(out).cpu().numpy()
    ref, info = O.sample_rmhmc_explicit(o, th0, N, L, eps, omega, alpha, 0, jitter, O.PhiloxDraws(seed, np.arange(C), NP[dtype]), "softabs")
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, np.stack(ref), rtol=tol, atol=tol)

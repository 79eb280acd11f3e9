"""GPU parity: the HIP path (through hamiltorch_amd -> ctypes -> libhamiltorch_amd.so) against the
oracle on the same seeded inputs, against the golden fixtures recorded from the reference, and
through size-independent properties at BASELINE sizes.

Tolerances (calibrated on the reference fp32 vs fp64): HMC trajectories
atol = rtol = 1e-5 (fp32), 1e-11 (fp64); a Metropolis decision within rounding of its
threshold may flip, so end-to-end sample() comparisons allow <= 1 % of chains to differ.
"""
import os

import numpy as np
import pytest
import torch

import hmc_oracle as O

pytestmark = pytest.mark.gpu

SIGMA3 = np.array([[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]])
NP = {torch.float32: np.float32, torch.float64: np.float64}


@pytest.fixture(scope="module")
def ht():
    import hamiltorch_amd
    assert torch.cuda.is_available()
    return hamiltorch_amd


def dev():
    return torch.device("cuda:0")


def rand_spd(D, seed, lo=0.5, hi=2.0):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    P = (Q * np.linspace(lo, hi, D)) @ Q.T
    return 0.5 * (P + P.T)


def targets(ht, P, dtype, mu=None):
    D = P.shape[0]
    mu = np.zeros(D) if mu is None else mu
    t = ht.GaussianTarget(torch.tensor(mu, dtype=dtype, device=dev()), precision=torch.tensor(P, dtype=dtype, device=dev()),
                          normalized=False)
    o = O.GaussianTarget(mu.astype(NP[dtype]), P.astype(NP[dtype]), 0.0)
    return t, o


def masses(D, dtype, seed=0):
    rng = np.random.default_rng(seed)
    diag = rng.uniform(0.5, 2.0, D)
    full = rand_spd(D, seed + 7, 0.5, 1.5)
    return {"none": None, "diag": diag.astype(NP[dtype]), "full": full.astype(NP[dtype])}


def tt(a, dtype):
    return None if a is None else torch.tensor(a, dtype=dtype, device=dev())


# ---------------------------------------------------------------------------------------------
def test_device_is_gfx950(ht):
    from hamiltorch_amd import _abi
    info = _abi.device_info(0)
    assert info["arch"].startswith("gfx950"), info
    assert info["wavefront_size"] == 64 and info["compute_units"] >= 200


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-6), (torch.float64, 1e-13)])
@pytest.mark.parametrize("D", [1, 3, 4, 7, 50, 130])
def test_gibbs_matches_philox_oracle(ht, dtype, tol, D):
    """K1: momentum draw == oracle Philox/Box-Muller stream (seed, chain_offset + c, draw)."""
    C, seed, off, draw = 37, 987654321012, 1000, 5
    z = O.philox_normals(seed, off + np.arange(C), draw, D, dtype=NP[dtype])
    m = masses(D, dtype)
    th = torch.zeros(C, D, dtype=dtype, device=dev())
    for kind, mass in m.items():
        p = ht.samplers.gibbs(th, mass=tt(mass, dtype), seed=seed, chain_offset=off, draw=draw)
        want = O.gibbs_momentum(z, mass)
        np.testing.assert_allclose(p.cpu().numpy(), want, rtol=10 * tol, atol=10 * tol)
    p1 = ht.samplers.gibbs(th[0], seed=seed, chain_offset=off, draw=draw)
    assert p1.shape == (D,)
    np.testing.assert_allclose(p1.cpu().numpy(), z[0], rtol=tol, atol=tol)


def test_kat1_and_reversibility(ht, golden):
    """tests/test_util.py:97-110 through the fused leapfrog kernel + the reference's own numbers."""
    g = golden("hmc_kat")
    t = ht.GaussianTarget(torch.zeros(2, device=dev()), covariance=torch.diag(torch.tensor([0.1, 0.1], device=dev())))
    one = torch.ones(2, device=dev())
    for steps in (1, 3, 100):
        p, m = ht.samplers.leapfrog(one, one, t, steps=steps, step_size=0.1, inv_mass=one.clone(),
                                    sampler=ht.Sampler.HMC, integrator=ht.Integrator.EXPLICIT)
        assert len(p) == steps and len(m) == steps and p[-1].shape == (2,)
        np.testing.assert_allclose(p[-1].cpu().numpy(), g[f"kat1_theta_{steps}"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(m[-1].cpu().numpy(), g[f"kat1_p_{steps}"], rtol=2e-5, atol=2e-5)
    p, m = ht.samplers.leapfrog(one, one, t, steps=100, step_size=0.1, inv_mass=one.clone(), sampler=ht.Sampler.HMC,
                                integrator=ht.Integrator.EXPLICIT)
    p2, _ = ht.samplers.leapfrog(p[-1], -m[-1], t, steps=100, step_size=0.1, inv_mass=one.clone(),
                                 sampler=ht.Sampler.HMC, integrator=ht.Integrator.EXPLICIT)
    np.testing.assert_allclose(p2[-1].cpu().numpy(), [1.0, 1.0], atol=1e-5)


@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-12)])
def test_kat2_hamiltonian_and_leapfrog_path(ht, golden, tag, dtype, tol):
    g = golden("hmc_kat")
    t = ht.GaussianTarget(torch.zeros(3, dtype=dtype, device=dev()), covariance=torch.tensor(SIGMA3, dtype=dtype, device=dev()))
    th = torch.tensor([0.3, -0.2, 0.5], dtype=dtype, device=dev()); pm = torch.tensor([0.1, 0.7, -0.4], dtype=dtype, device=dev())
    ims = {"none": None, "diag": tt(np.array([1.0, 0.5, 2.0]), dtype), "full": tt(g[f"kat2_inv_mass_full_{tag}"], dtype)}
    for mk, im in ims.items():
        H = ht.samplers.hamiltonian(th, pm, t, inv_mass=im, sampler=ht.Sampler.HMC)
        np.testing.assert_allclose(H.cpu().numpy().reshape(-1), g[f"kat2_H_{mk}_{tag}"], rtol=tol, atol=tol)
        for native in (t, (lambda w, t=t: t(w))):      # fused kernel and generic-callback pieces
            p, m = ht.samplers.leapfrog(th, pm, native, steps=5, step_size=0.3, inv_mass=im, sampler=ht.Sampler.HMC,
                                        integrator=ht.Integrator.EXPLICIT)
            np.testing.assert_allclose(torch.stack(p).cpu().numpy(), g[f"kat2_theta_{mk}_{tag}"], rtol=tol, atol=tol)
            np.testing.assert_allclose(torch.stack(m).cpu().numpy(), g[f"kat2_p_{mk}_{tag}"], rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 1e-11)])
@pytest.mark.parametrize("D", [1, 2, 5, 8, 9, 33, 64, 100, 257])
def test_leapfrog_batch_vs_oracle(ht, dtype, tol, D):
    """T1: identical (theta, p) into the kernel and the oracle; small-D (registers) and wave-per-chain layouts."""
    C, steps, eps = 70, 7, 0.11
    rng = np.random.default_rng(D)
    P = rand_spd(D, D)
    mu = rng.standard_normal(D)
    t, o = targets(ht, P, dtype, mu)
    th0 = rng.standard_normal((C, D)).astype(NP[dtype]); p0 = rng.standard_normal((C, D)).astype(NP[dtype])
    for kind, im in masses(D, dtype).items():
        if kind == "full" and D > 130:
            continue
        pt, pp = ht.samplers.leapfrog(tt(th0, dtype), tt(p0, dtype), t, steps=steps, step_size=eps, inv_mass=tt(im, dtype),
                                      sampler=ht.Sampler.HMC, integrator=ht.Integrator.EXPLICIT)
        wt, wp = O.hmc_leapfrog(th0, p0, o.grad, steps, eps, im, return_path=True)
        scale = max(1.0, np.abs(np.stack(wt)).max(), np.abs(np.stack(wp)).max())
        np.testing.assert_allclose(torch.stack(pt).cpu().numpy(), np.stack(wt), rtol=tol, atol=tol * scale * (1 + D / 16))
        np.testing.assert_allclose(torch.stack(pp).cpu().numpy(), np.stack(wp), rtol=tol, atol=tol * scale * (1 + D / 16))
        H = ht.samplers.hamiltonian(tt(th0, dtype), tt(p0, dtype), t, inv_mass=tt(im, dtype), sampler=ht.Sampler.HMC)
        wH, _ = O.hmc_hamiltonian(th0, p0, o.logp, im)
        np.testing.assert_allclose(H.cpu().numpy(), wH, rtol=10 * tol, atol=10 * tol * (1 + D))


def _compare_runs(got, ref, tol, max_bad=0.01):
    got = np.stack([g.cpu().numpy() for g in got]); ref = np.stack(ref)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max(axis=(0, 2))
    bad = err > tol
    assert bad.mean() <= max_bad, "%d of %d chains differ (max err %.3g)" % (bad.sum(), bad.size, err.max())
    return bad


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float64, 1e-10)])
@pytest.mark.parametrize("D,mass,burn", [(3, "none", 0), (3, "diag", 4), (3, "full", -1), (6, "none", 2), (20, "diag", 0),
                                          (70, "none", 3)])
def test_sample_fused_vs_oracle(ht, dtype, tol, D, mass, burn):
    """End to end: gibbs -> H -> leapfrog -> H -> MH -> burn bookkeeping (incl. the Q2 reset), same Philox draws."""
    C, N, L, eps, seed, off = 96, 25, 5, 0.25, 4242, 17
    P = rand_spd(D, 3) if D != 3 else np.linalg.inv(SIGMA3)
    t, o = targets(ht, P, dtype)
    th0 = (0.3 * O.philox_normals(seed, off + np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(NP[dtype])
    im = masses(D, dtype)[mass]
    out, acc = ht.sample(t, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=burn,
                         inv_mass=tt(im, dtype), debug=2, verbose=False, seed=seed, chain_offset=off)
    ref, info = O.sample_hmc(o, th0, N, L, eps, burn, im, O.PhiloxDraws(seed, off + np.arange(C), NP[dtype]))
    assert len(out) == len(ref) == N - max(burn, -1)
    bad = _compare_runs(out, ref, tol)
    np.testing.assert_allclose(acc.cpu().numpy()[~bad], info["acc_rate"][~bad], atol=1e-12)
    assert 0.3 < float(acc.mean()) <= 1.0


@pytest.mark.parametrize("mass", ["none", "diag", "full"])
def test_sample_generic_callback_vs_fused_and_oracle(ht, mass):
    """The same target through the generic-callback path (torch evaluates the closure, HIP does the
    state updates) must reproduce the fused kernel and the oracle."""
    dtype, C, N, L, eps, seed = torch.float32, 48, 12, 4, 0.3, 99
    t, o = targets(ht, np.linalg.inv(SIGMA3), dtype)
    th0 = (0.5 * O.philox_normals(seed, np.arange(C), 0, 3, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    im = masses(3, dtype)[mass]
    closure = lambda w: t(w)  # noqa: E731  -- opaque to the plugin recogniser
    out_g = ht.sample(closure, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=1,
                      inv_mass=tt(im, dtype), verbose=False, seed=seed)
    out_f = ht.sample(t, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=1,
                      inv_mass=tt(im, dtype), verbose=False, seed=seed)
    ref, _ = O.sample_hmc(o, th0, N, L, eps, 1, im, O.PhiloxDraws(seed, np.arange(C)))
    _compare_runs(out_g, ref, 2e-4, 0.03)
    _compare_runs(out_f, ref, 2e-4, 0.03)


def test_sample_single_chain_contract(ht):
    """(D,) in -> list of (D,) tensors of length num_samples - burn, element 0 = params_init, float acc rate."""
    t, _ = targets(ht, np.linalg.inv(SIGMA3), torch.float32)
    init = torch.tensor([0.5, -0.5, 0.25], device=dev())
    out, acc = ht.sample(t, init, num_samples=30, num_steps_per_sample=5, step_size=0.3, burn=10, debug=2, verbose=False, seed=1)
    assert isinstance(out, list) and len(out) == 20 and out[0].shape == (3,) and isinstance(acc, float)
    assert torch.equal(out[0], init)
    out_cpu = ht.sample(t, init, num_samples=5, verbose=False, store_on_GPU=False, seed=1)
    assert out_cpu[0].device.type == "cpu"
    mvn = torch.distributions.MultivariateNormal(torch.zeros(3, device=dev()), torch.tensor(SIGMA3, dtype=torch.float32, device=dev()))
    out_mvn = ht.sample(mvn.log_prob, init, num_samples=30, num_steps_per_sample=5, step_size=0.3, burn=10, verbose=False, seed=1)
    np.testing.assert_allclose(torch.stack(out_mvn).cpu().numpy(), torch.stack(out).cpu().numpy(), atol=1e-4)


def test_sharding_invariance(ht):
    """Chains are keyed by global id: two half-size calls with chain_offset reproduce one full call bit for bit."""
    t, _ = targets(ht, np.linalg.inv(SIGMA3), torch.float32)
    C, seed = 128, 31337
    th0 = tt(0.1 * O.philox_normals(seed, np.arange(C), 0, 3, O.PURPOSE_INIT), torch.float32)
    kw = dict(num_samples=15, num_steps_per_sample=6, step_size=0.3, verbose=False, seed=seed)
    full = torch.stack(ht.sample(t, th0, **kw))
    a = torch.stack(ht.sample(t, th0[:64], chain_offset=0, **kw))
    b = torch.stack(ht.sample(t, th0[64:], chain_offset=64, **kw))
    assert torch.equal(full, torch.cat([a, b], dim=1))


def test_divergent_chain_is_rejected_not_fatal(ht):
    """A chain whose energy overflows is a rejection for that chain only (LogProbError semantics, S:1045)."""
    t, _ = targets(ht, np.diag([1.0, 1.0, 1e30]), torch.float32)
    th0 = torch.ones(8, 3, device=dev())
    out, acc = ht.sample(t, th0, num_samples=6, num_steps_per_sample=50, step_size=2.5, debug=2, verbose=False, seed=3)
    s = torch.stack(out)
    assert torch.isfinite(s).all() and float(acc.max()) == 0.0
    assert torch.equal(s[-1], th0)


def test_cfg2_statistical_parity_at_full_size(ht):
    """BASELINE cfg2 shape (1024 chains, L=25, eps=0.3): pooled posterior mean / covariance / acceptance against the
    exact Gaussian and the oracle's acceptance on a chain subset (T2 tolerances of SURVEY 8c)."""
    dtype = torch.float32
    t, o = targets(ht, np.linalg.inv(SIGMA3), dtype)
    C, N, L, eps, seed = 1024, 300, 25, 0.3, 2024
    th0 = tt(0.1 * O.philox_normals(seed, np.arange(C), 0, 3, O.PURPOSE_INIT), dtype)
    out, acc = ht.sample(t, th0, num_samples=N, num_steps_per_sample=L, step_size=eps, burn=50, debug=2, verbose=False, seed=seed)
    s = torch.stack(out[1:]).double().cpu().numpy()          # [S, C, 3]
    pooled = s.reshape(-1, 3)
    ess = min(O.ess_bulk(s[:, :, d]) for d in range(3))
    sd = np.sqrt(np.diag(SIGMA3))
    assert np.all(np.abs(pooled.mean(0)) <= 4 * sd / np.sqrt(ess))
    np.testing.assert_allclose(np.cov(pooled.T), SIGMA3, rtol=0.05, atol=0.02)
    sub = np.arange(64)
    _, info = O.sample_hmc(o, th0[:64].cpu().numpy(), N, L, eps, 50, None, O.PhiloxDraws(seed, sub))
    assert abs(float(acc[:64].mean()) - info["acc_rate"].mean()) <= 0.01
    assert float(acc.mean()) > 0.95            # reference: 0.998 at L=25


def test_multi_chain_adapters(ht):
    t, _ = targets(ht, np.linalg.inv(SIGMA3), torch.float32)
    prior = lambda: torch.randn(3, device=dev())  # noqa: E731
    chain = ht.util.setup_chain(ht.sample, prior, dict(log_prob_func=t, num_samples=8, num_steps_per_sample=3, step_size=0.3, verbose=False))
    res = ht.util.multi_chain(chain, 2, [1, 2, 3], parallel=False)
    assert len(res) == 3 and len(res[0]) == 8 and res[0][0].shape == (3,)
    res_t = ht.util.multi_chain(chain, 2, [1, 2, 3], parallel=True)
    assert len(res_t) == 3
    res_b = ht.util.multi_chain(chain, 2, [1, 2, 3], batched=True)
    assert len(res_b) == 3 and len(res_b[0]) == 8 and res_b[2][0].shape == (3,)
    # same seed -> same params_init as the serial adapter
    assert torch.equal(res_b[1][0], res[1][0])
    # values, not only shapes: the serial adapter IS sample() under set_random_seed(seed) with params_init = prior() ...
    for k, seed in enumerate([1, 2, 3]):
        ht.set_random_seed(seed)
        init = prior()
        want = ht.sample(log_prob_func=t, params_init=init, num_samples=8, num_steps_per_sample=3, step_size=0.3, verbose=False)
        assert torch.equal(torch.stack(res[k]), torch.stack(want))
        assert torch.equal(torch.stack(res_t[k]), torch.stack(want))       # the thread pool serialises on the sampling lock: same chains
    # ... and the batched adapter is ONE sample() call over the stacked initial states under the first seed
    inits = []
    for seed in [1, 2, 3]:
        torch.manual_seed(seed)
        inits.append(prior())
    ht.set_random_seed(1)
    want_b = ht.sample(log_prob_func=t, params_init=torch.stack(inits), num_samples=8, num_steps_per_sample=3, step_size=0.3, verbose=False)
    for c in range(3):
        assert torch.equal(torch.stack(res_b[c]), torch.stack([row[c] for row in want_b]))
    # every chain of the batch is a valid draw sequence of the target: pooled over a longer batched run the covariance is right
    chain_l = ht.util.setup_chain(ht.sample, prior, dict(log_prob_func=t, num_samples=300, num_steps_per_sample=10, step_size=0.3, burn=50,
                                                         verbose=False))
    big = ht.util.multi_chain(chain_l, 2, list(range(64)), batched=True)
    pooled = torch.stack([torch.stack(ch) for ch in big]).reshape(-1, 3).double().cpu().numpy()
    np.testing.assert_allclose(np.cov(pooled.T), SIGMA3, rtol=0.15, atol=0.08)


@pytest.mark.parametrize("D,mass", [(3, "none"), (5, "diag"), (4, "full")])
def test_predrawn_workspace_equals_inline_rng(ht, D, mass):
    """The full-chip RNG pre-pass + load path must reproduce the inline-draw path (same Philox stream)."""
    from hamiltorch_amd import _abi
    from hamiltorch_amd.samplers import _mass_operands
    dtype = torch.float32
    t, _ = targets(ht, rand_spd(D, 11), dtype)
    C, N, L, eps = 200, 40, 6, 0.2
    th0 = tt(np.random.default_rng(0).standard_normal((C, D)), dtype)
    kind, im, mf = _mass_operands(tt(masses(D, dtype)[mass], dtype), th0)
    outs = []
    _abi.set_tuning("gauss_eig", 0)     # same arithmetic on both sides (the eigenbasis route needs the pre-pass; it has its own test)
    try:
        for use_ws in (False, True):
            cur = th0.clone()
            samples = torch.zeros(N + 1, C, D, device=dev())
            rej = torch.zeros(C, dtype=torch.int32, device=dev())
            ws = torch.empty(_abi.gaussian_workspace_bytes(C, D, N, 4), dtype=torch.uint8, device=dev()) if use_ws else None
            _abi.hmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, kind, im, mf, L, eps, N, 0, -1, 77, 5,
                                     samples, rej, workspace=ws)
            outs.append((samples, rej))
    finally:
        _abi.set_tuning("gauss_eig", 1)
    assert torch.equal(outs[0][1], outs[1][1])
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy(), rtol=0, atol=1e-6)


def test_chunked_launches_equal_single_launch(ht):
    """sample() cuts long runs into several launches over traj_offset: same chain either way (bit for bit on the direct
    kernels; to rounding on the eigenbasis route, whose state crosses a launch boundary as q = mu + Q y)."""
    from hamiltorch_amd import samplers, _abi
    t, _ = targets(ht, np.linalg.inv(SIGMA3), torch.float32)
    th0 = tt(np.random.default_rng(1).standard_normal((64, 3)), torch.float32)
    kw = dict(num_samples=50, num_steps_per_sample=5, step_size=0.3, burn=7, verbose=False, seed=5)
    a = torch.stack(ht.sample(t, th0, **kw))
    old = samplers._GaussianHMC.WS_CAP
    try:
        samplers._GaussianHMC.WS_CAP = 64 * 4 * 4 * 9        # room for a few trajectories per launch
        b = torch.stack(ht.sample(t, th0, **kw))
        err = (a - b).abs().amax(dim=(0, 2))
        assert (err > 2e-5).float().mean() <= 0.02, float(err.max())
        _abi.set_tuning("gauss_eig", 0)
        a0 = torch.stack(ht.sample(t, th0, **kw))
        samplers._GaussianHMC.WS_CAP = old
        b0 = torch.stack(ht.sample(t, th0, **kw))
        assert torch.equal(a0, b0)
    finally:
        samplers._GaussianHMC.WS_CAP = old
        _abi.set_tuning("gauss_eig", 1)


@pytest.mark.parametrize("route", ["fused", "generic"])
def test_nuts_step_size_adaptation_vs_oracle(ht, route):
    """Sampler.HMC_NUTS (dual averaging during burn-in, S:1030-1035): adapted step size and samples vs the oracle with
    the same Philox draws -- single chain (the reference's schedule) and a batch (shared step size)."""
    t, o = targets(ht, np.linalg.inv(SIGMA3), torch.float32)
    lp = t if route == "fused" else (lambda w: t(w))
    for C in (1, 16):
        seed = 404 + C
        th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, 3, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
        init = tt(th0[0] if C == 1 else th0, torch.float32)
        # step_size_init 0.02: the first dual-averaging jump (to ~10x) stays inside the leapfrog stability limit,
        # so rounding differences are not amplified by unstable trajectories
        out, ss = ht.sample(lp, init, num_samples=40, num_steps_per_sample=5, step_size=0.02, burn=15,
                            sampler=ht.Sampler.HMC_NUTS, desired_accept_rate=0.7, debug=2, verbose=False, seed=seed)
        ref, info = O.sample_hmc(o, th0, 40, 5, 0.02, 15, None, O.PhiloxDraws(seed, np.arange(C)), nuts_desired=0.7)
        assert isinstance(ss, float) and abs(ss - info["step_size"]) < 5e-3 * info["step_size"]
        got = torch.stack(out).cpu().numpy().reshape(len(ref), C, 3)
        # after burn-in every trajectory runs with the adapted step size, so a 0.5 % step-size difference shows
        # up as O(1e-2) differences in the samples
        bad = np.abs(got - np.stack(ref)).max(axis=(0, 2)) > 3e-2
        assert bad.mean() <= 0.1
    with pytest.raises(RuntimeError, match="burn must be greater than 0 for NUTS"):
        ht.sample(t, init, num_samples=5, sampler=ht.Sampler.HMC_NUTS, verbose=False)


def test_graph_replay_of_callbacks_matches_eager(ht):
    """util.GraphedCallable: a capturable log_prob_func is replayed as a HIP graph between the native kernels; a callback
    that cannot be captured (host scalars mixed in, as torch.distributions.Normal(0, 3) does) silently stays eager.
    Same samples as the eager evaluation either way."""
    import os
    from hamiltorch_amd import util
    dev_ = torch.device("cuda:0")
    D, C = 6, 32

    def capturable(w):
        return -0.5 * (w * w).sum() - 0.05 * (w ** 4).sum()

    def host_scalars(w):
        return torch.distributions.Normal(0, 3, validate_args=False).log_prob(w).sum()
    th0 = 0.3 * torch.randn(C, D, generator=torch.Generator().manual_seed(0)).to(dev_)
    for fn in (capturable, host_scalars):
        outs = []
        for graphs in ("1", "0"):
            os.environ["HAMILTORCH_AMD_GRAPHS"] = graphs
            try:
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out = ht.sample(fn, th0, num_samples=6, num_steps_per_sample=4, step_size=0.1, verbose=False, seed=4)
            finally:
                os.environ.pop("HAMILTORCH_AMD_GRAPHS", None)
            outs.append(torch.stack(out).cpu().numpy())
        np.testing.assert_allclose(outs[0], outs[1], rtol=1e-6, atol=1e-6)
    g = util.GraphedCallable(torch.func.vmap(torch.func.grad(capturable)))
    a = g(th0).clone(); b = torch.func.vmap(torch.func.grad(capturable))(th0)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert g.cache and next(iter(g.cache.values())) is not False


def test_whole_trajectory_graph_matches_eager(ht):
    """_GenericHMC.advance captures one whole trajectory (native kernels + callback, trajectory index in device memory:
    hta_momentum_resample_at / hta_mh_select_at / hta_counter_add) and replays it; samples, burn-in handling and the
    split integrators are the same as launch-by-launch execution."""
    import os
    import warnings
    dev_ = torch.device("cuda:0")
    D, C = 5, 48

    def lp(w):
        return -0.5 * (w * w).sum() - 0.05 * (w ** 4).sum()

    def lp_b(w):
        return -0.25 * ((w - 0.5) ** 2).sum()
    th0 = 0.3 * torch.randn(C, D, generator=torch.Generator().manual_seed(1)).to(dev_)
    cases = [dict(log_prob_func=lp, num_samples=24, burn=5, num_steps_per_sample=6, step_size=0.15),
             dict(log_prob_func=lp, num_samples=12, num_steps_per_sample=3, step_size=0.1, inv_mass=torch.full((D,), 2.0)),
             dict(log_prob_func=[lp, lp_b], num_samples=12, num_steps_per_sample=3, step_size=0.1,
                  integrator=ht.Integrator.SPLITTING)]
    for kw in cases:
        outs = []
        for graphs in ("1", "0"):
            os.environ["HAMILTORCH_AMD_GRAPHS"] = graphs
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("error")          # no fallback warning: the capture must succeed
                    out = ht.sample(params_init=th0, verbose=False, seed=9, **kw)
            finally:
                os.environ.pop("HAMILTORCH_AMD_GRAPHS", None)
            outs.append(torch.stack(out).cpu().numpy())
        assert outs[0].shape == outs[1].shape
        np.testing.assert_allclose(outs[0], outs[1], rtol=1e-6, atol=1e-6)


def test_device_counter_entry_points(ht):
    """The *_at entry points read the trajectory index from device memory and draw the same streams as the host-index ones."""
    from hamiltorch_amd import _abi
    dev_ = torch.device("cuda:0")
    C, D = 33, 7
    n_dev = torch.tensor([6], dtype=torch.int32, device=dev_)
    mf = torch.ones(1, device=dev_)
    p_a = torch.empty(C, D, device=dev_); p_b = torch.empty(C, D, device=dev_)
    _abi.momentum_resample(p_a, 0, mf, 123, 2, 6)
    _abi.momentum_resample_at(p_b, 0, mf, 123, 2, n_dev)
    assert torch.equal(p_a, p_b)
    _abi.counter_add(n_dev, 1)
    assert int(n_dev.item()) == 7
    g = torch.Generator().manual_seed(3)
    cur = torch.randn(C, D, generator=g).to(dev_); prop = torch.randn(C, D, generator=g).to(dev_)
    Ho = torch.randn(C, generator=g).to(dev_); Hn = Ho + 0.5 * torch.randn(C, generator=g).to(dev_)
    lpn = torch.randn(C, generator=g).to(dev_)
    res = []
    for at in (False, True):
        c, samples = cur.clone(), torch.zeros(4, C, D, device=dev_)
        rej = torch.zeros(C, dtype=torch.int32, device=dev_)
        if at:
            _abi.mh_select_at(c, prop, cur, Ho, Hn, lpn, samples, rej, None, n_dev, 5, 123, 2)
        else:
            _abi.mh_select(c, prop, cur, Ho, Hn, lpn, samples[7 - 5], rej, None, 7, 5, 123, 2)
        res.append((c.cpu(), samples.cpu(), rej.cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert 0 < int(res[0][2].sum()) < C


@pytest.mark.parametrize("route", ["fused", "generic"])
def test_block_list_inv_mass_vs_oracle_and_reference_fixture(ht, route):
    """inv_mass as a list of diagonal blocks (S:188-197, S:287-292, S:803-809, S:944-947): pieces against the reference
    fixture, batched sample() against the oracle on the same Philox draws, on the fused and the generic-callback route."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "blockmass.npz"))
    dtype = torch.float32
    blocks_np = [g["b0"], g["b1"]]
    blocks = [tt(b, dtype) for b in blocks_np]
    t, o = targets(ht, g["P"], dtype)
    fn = t if route == "fused" else (lambda w: t(w))
    H = ht.samplers.hamiltonian(tt(g["kat_theta"], dtype), tt(g["kat_p"], dtype), fn, inv_mass=blocks, sampler=ht.Sampler.HMC)
    np.testing.assert_allclose(H.cpu().numpy().reshape(-1), g["kat_H"], rtol=1e-5)
    pl, ml = ht.samplers.leapfrog(tt(g["kat_theta"], dtype), tt(g["kat_p"], dtype), fn, steps=4, step_size=0.2,
                                  inv_mass=blocks, sampler=ht.Sampler.HMC)
    np.testing.assert_allclose(pl[-1].cpu().numpy(), g["kat_theta_L"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ml[-1].cpu().numpy(), g["kat_p_L"], rtol=1e-4, atol=1e-5)
    C, N, L, eps, burn, seed = 64, 20, 5, 0.8, 3, 77
    th0 = (0.3 * O.philox_normals(seed, np.arange(C), 0, 5, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    out = ht.sample(fn, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=burn, inv_mass=blocks,
                    verbose=False, seed=seed)
    ref, info = O.sample_hmc(o, th0, N, L, eps, burn, blocks_np, O.PhiloxDraws(seed, np.arange(C)))
    _compare_runs(out, ref, 2e-4, 0.03)
    assert 0.3 < info["acc_rate"].mean() < 1.0


# (float64: the register-resident kernels stop at D = 4 - beyond, both settings of 'gauss_eig' are the wave kernels, covered by
#  test_wave_eigenbasis_route_vs_direct_and_oracle; those combinations are not generated rather than skipped)
@pytest.mark.parametrize("dtype,tol,D", [(torch.float32, 1e-4, D) for D in (1, 2, 3, 4, 5, 6)] + [(torch.float64, 1e-11, D) for D in (1, 2, 3, 4)])
@pytest.mark.parametrize("mass", ["none", "diag", "full"])
def test_eigenbasis_route_equals_direct_route(ht, dtype, tol, D, mass):
    """Small-D Gaussian HMC integrates in the eigenbasis of the (mass-whitened) precision matrix (2 D FMAs per step);
    hta_set_tuning('gauss_eig', 0) selects the direct kernel (D + D^2 FMAs per step, more with a mass matrix).  Same draws,
    same map: samples agree to rounding, chain by chain, for identity / diagonal / full mass, with a mean offset, burn-in
    (Q2 reset) and nearly degenerate / widely spread spectra."""
    from hamiltorch_amd import _abi
    C, N, L, eps, seed = 128, 30, 7, 0.2, 5 + D
    rng = np.random.default_rng(D)
    Qm, _ = np.linalg.qr(rng.normal(size=(D, D)))
    lam = np.concatenate([[4.0, 4.0 + 1e-7], rng.uniform(0.05, 6.0, size=D)])[:D]
    P = (Qm * lam) @ Qm.T
    mu = rng.normal(size=D)
    t, _ = targets(ht, P, dtype, mu=mu)
    th0 = tt(mu + rng.normal(size=(C, D)), dtype)
    outs = []
    for mode in (1, 0):
        _abi.set_tuning("gauss_eig", mode)
        try:
            out, acc = ht.sample(t, th0, num_samples=N, num_steps_per_sample=L, step_size=eps, burn=3, debug=2, verbose=False,
                                 seed=seed, inv_mass=tt(masses(D, dtype)[mass], dtype))
        finally:
            _abi.set_tuning("gauss_eig", 1)
        outs.append((torch.stack(out).cpu().numpy(), acc.cpu().numpy()))
    err = np.abs(outs[0][0] - outs[1][0]).max(axis=(0, 2))
    assert (err > tol * 10).mean() <= 0.02, "max err %.3g" % err.max()      # a rounding-level accept flip moves a whole chain
    good = err <= tol * 10
    np.testing.assert_allclose(outs[0][1][good], outs[1][1][good], atol=1e-12)
    assert 0.2 < outs[0][1].mean() <= 1.0



@pytest.mark.parametrize("D,C,L", [(3, 1024, 25), (1, 17, 5), (2, 100, 10), (4, 333, 25), (3, 5, 10)])
def test_quad_kernel_step_count_instances_agree(ht, D, C, L):
    """hmc_gauss_quad_kernel has instances with the step count compiled in (L = 5, 10, 25: no loop bookkeeping on the hot
    path) and a separate instance of the burn+1 trajectory (Q2 reset); hta_set_tuning('gauss_eig', 3) selects the any-L
    instance.  Same arithmetic: bit-identical samples, reject counts and final state, with the Q2 reset exercised."""
    from hamiltorch_amd import _abi
    rng = np.random.default_rng(D * 100 + C)
    P = rand_spd(D, 5 + D)
    mu = rng.normal(size=D)
    t, _ = targets(ht, P, torch.float32, mu=mu)
    th0 = tt(mu + rng.normal(size=(C, D)), torch.float32)
    N, burn, eps = 37, 3, 0.9 / np.sqrt(np.linalg.eigvalsh(P).max()) * (25.0 / L) ** 0.25
    outs = []
    for mode in (1, 3):
        _abi.set_tuning("gauss_eig", mode)
        try:
            cur = th0.clone()
            samples = torch.zeros(N - burn + 1, C, D, device=dev())
            rej = torch.zeros(C, dtype=torch.int32, device=dev())
            ws = torch.zeros(_abi.gaussian_workspace_bytes(C, D, N, 4), dtype=torch.uint8, device=dev())
            for start, cnt in ((0, 2), (2, N - 2)):          # the Q2 trajectory (n = burn+1 = 4) inside the second launch
                _abi.hmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, 0, None, None, L, float(eps), cnt, start, burn,
                                         31, 7, samples, rej, workspace=ws)
            torch.cuda.synchronize()
        finally:
            _abi.set_tuning("gauss_eig", 1)
        outs.append((samples.cpu(), rej.cpu(), cur.cpu()))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    if C >= 100:
        assert 0 < int(outs[0][1].sum()) < C * N
        # Q2: a chain rejected at n = burn+1 restarts from params_init -> its first stored row after the seed row equals th0
        assert bool((outs[0][0][1] == th0.cpu()).all(dim=1).any())


@pytest.mark.parametrize("D,C,L,N,burn", [(3, 1024, 25, 61, 3), (3, 1024, 25, 40, -1), (1, 17, 5, 37, 3), (2, 100, 10, 44, 0),
                                           (4, 333, 25, 23, 5), (3, 5, 7, 30, 2), (4, 64, 10, 52, 7)])
def test_quad_kernel_variants_are_bit_identical(ht, D, C, L, N, burn):
    """hta_set_tuning('quad_variant', 0 | 3 | 7 (default)): the quad kernel with wave-uniform base addresses + 32-bit lane offsets (no 64-bit
    vector add per record load / row store), without the NaN guard in front of the accept compare (2 log u is finite by
    construction) and - 7 - with the row element and the energy butterfly in one interleaved block.  The arithmetic of a
    trajectory is untouched: samples, reject counts and the final state are bit-identical to the round-1 instance (0), over launch
    boundaries, the burn-in / stored phases, the Q2 trajectory and every tail length of the eight-trajectory pass."""
    from hamiltorch_amd import _abi
    rng = np.random.default_rng(D * 1000 + C + N)
    P = rand_spd(D, 5 + D)
    mu = rng.normal(size=D)
    t, _ = targets(ht, P, torch.float32, mu=mu)
    th0 = tt(mu + rng.normal(size=(C, D)), torch.float32)
    eps = 0.9 / np.sqrt(np.linalg.eigvalsh(P).max()) * (25.0 / L) ** 0.25
    rows = N - max(0, min(N, burn + 1)) + 1
    outs, routes = [], []
    for var in (0, 3, 7):
        _abi.set_tuning("quad_variant", var)
        try:
            cur = th0.clone()
            samples = torch.zeros(rows, C, D, device=dev())
            rej = torch.zeros(C, dtype=torch.int32, device=dev())
            ws = torch.zeros(_abi.gaussian_workspace_bytes(C, D, N, 4), dtype=torch.uint8, device=dev())
            for start, cnt in ((0, 2), (2, 11), (13, N - 13)):
                _abi.hmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, 0, None, None, L, float(eps), cnt, start, burn,
                                         31, 7, samples, rej, workspace=ws)
            routes.append(_abi.last_route())
            torch.cuda.synchronize()
        finally:
            _abi.reset_tuning()
        outs.append((samples.cpu(), rej.cpu(), cur.cpu()))
    lb = L if L in (5, 10, 25) else 0
    assert routes == ["hmc_gauss_quad_kernel<%d,false,%d>" % (D, lb)] + ["hmc_gauss_quad_kernel<%d,false,%d,%d>" % (D, lb, v) for v in (3, 7)]
    for other in outs[1:]:
        for a_, b_ in zip(outs[0], other):
            assert torch.equal(a_, b_)
    if C >= 100:
        assert 0 < int(outs[0][1].sum()) < C * N


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_run_begin_is_the_three_initial_copies(ht, dtype):
    """hta_run_begin: params = params_init.clone(), ret_params = [params.clone()], num_rejected = 0 (S:954-961) in one launch."""
    from hamiltorch_amd import _abi
    for C, D in ((37, 5), (1, 1), (1024, 3), (5000, 129)):
        init = torch.randn(C, D, dtype=dtype, device=dev())
        cur = torch.full_like(init, float("nan")); rows = torch.full((2, C, D), float("nan"), dtype=dtype, device=dev())
        rej = torch.full((C,), 7, dtype=torch.int32, device=dev())
        _abi.run_begin(init, cur, rows[0], rej)
        assert torch.equal(cur, init) and torch.equal(rows[0], init) and bool(torch.isnan(rows[1]).all()) and int(rej.abs().sum()) == 0
        cur2 = torch.zeros_like(init)
        _abi.run_begin(init, cur2, None, None)
        assert torch.equal(cur2, init)


def test_hmc_prepared_workspace_is_bit_identical_and_skips_the_setup(ht):
    """hta_hmc_gaussian_prepare (ABI 8): the diagonalisation of the precision matrix - a single-wave kernel in front of every
    launch of the eigenbasis route - hoisted out of the sample call.  Same bits with and without; a prepared call does not look
    at P again (shown by breaking the contract on purpose: P edited in place after the preparation changes nothing until
    hta_hmc_gaussian_forget); another trajectory count (= another eig block position) runs the setup itself;
    hamiltorch_amd.sample() prepares once per target and again after an in-place edit of the target."""
    from hamiltorch_amd import _abi
    rng = np.random.default_rng(77)
    mu = rng.normal(size=3)
    t, _ = targets(ht, rand_spd(3, 8), torch.float32, mu=mu)
    C, N, L, eps = 256, 40, 25, 0.3
    th0 = tt(mu + rng.normal(size=(C, 3)), torch.float32)
    nbytes = _abi.gaussian_workspace_bytes(C, 3, N, 4)
    ws_a = torch.zeros(nbytes, dtype=torch.uint8, device=dev()); ws_b = torch.zeros(nbytes, dtype=torch.uint8, device=dev())

    def run(ws, n=N):
        cur = th0.clone(); samples = torch.zeros(n + 1, C, 3, device=dev()); rej = torch.zeros(C, dtype=torch.int32, device=dev())
        _abi.hmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, 0, None, None, L, eps, n, 0, -1, 13, 0, samples, rej,
                                 workspace=ws)
        assert _abi.last_route().startswith(("hmc_gauss_quad_kernel<3,false,25", "hmc_gauss_quad_fused_kernel<3,25"))
        torch.cuda.synchronize()
        return torch.cat([samples.reshape(-1), rej.float(), cur.reshape(-1)]).cpu()
    s0 = run(ws_a)
    _abi.hmc_gaussian_prepare(th0, t.precision, 0, None, C, 3, N, ws_b)
    s1 = run(ws_b)
    assert torch.equal(s0, s1)
    t.precision.mul_(1.7)                        # contract broken on purpose
    assert torch.equal(run(ws_b), s1)            # prepared: the old eigenbasis, P is not read again
    s3 = run(ws_a)
    assert not torch.equal(s3, s0)               # unprepared: the call diagonalises what P holds now
    assert torch.equal(run(ws_b, N - 7), run(ws_a, N - 7))       # another n_traj: not the prepared eig block -> own setup
    assert torch.equal(run(ws_b), s1)            # ... and the preparation for N is still in force
    _abi.hmc_gaussian_forget(ws_b)
    assert torch.equal(run(ws_b), s3)
    _abi.hmc_gaussian_prepare(th0, t.precision, 0, None, C, 3, N, ws_b)
    assert torch.equal(run(ws_b), s3)
    _abi.hmc_gaussian_forget(ws_b)
    # through sample(): one prepared workspace per target, prepared again after an in-place edit of the precision matrix
    kw = dict(num_samples=30, num_steps_per_sample=L, step_size=eps, verbose=False, seed=9)
    a = torch.stack(list(ht.sample(t, th0, **kw)))
    ws_id = next(iter(t._hta_hmc_ws.values()))[0].ws.data_ptr()
    b = torch.stack(list(ht.sample(t, th0, **kw)))
    assert torch.equal(a, b) and next(iter(t._hta_hmc_ws.values()))[0].ws.data_ptr() == ws_id
    t.precision.mul_(1.5)
    c = torch.stack(list(ht.sample(t, th0, **kw)))
    t2 = ht.GaussianTarget(t.mean.clone(), precision=t.precision.clone(), normalized=False)
    d = torch.stack(list(ht.sample(t2, th0, **kw)))
    assert torch.equal(c, d) and not torch.equal(c, a)


def test_edge_sizes_against_oracle(ht):
    """Edges of the Gaussian path: the maximum dimension (D = 1024, wave-per-chain kernel), a single chain, an empty launch
    (n_traj = 0 leaves everything untouched) and a chain count beyond the quad kernel's range (one chain per lane,
    eigenbasis) -- each against the oracle on the same Philox draws."""
    from hamiltorch_amd import _abi
    dtype = torch.float32
    # D = 1024, 3 chains
    D, C, N, L, eps, seed = 1024, 3, 4, 3, 0.02, 11
    P = rand_spd(D, 2)
    t, o = targets(ht, P, dtype)
    th0 = (0.1 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    out = ht.sample(t, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, verbose=False, seed=seed)
    ref, _ = O.sample_hmc(o, th0, N, L, eps, 0, None, O.PhiloxDraws(seed, np.arange(C)))
    _compare_runs(out, ref, 5e-4, 0.0)
    # one chain, 3-D (a single quad of one wave)
    t3, o3 = targets(ht, np.linalg.inv(SIGMA3), dtype)
    th1 = np.array([[0.3, -0.2, 0.5]], np.float32)
    out = ht.sample(t3, tt(th1, dtype), num_samples=40, num_steps_per_sample=25, step_size=0.3, verbose=False, seed=5)
    ref, _ = O.sample_hmc(o3, th1, 40, 25, 0.3, 0, None, O.PhiloxDraws(5, np.arange(1)))
    _compare_runs(out, ref, 2e-4, 0.0)
    # empty launch through the C ABI
    cur = tt(th1, dtype).clone(); rej = torch.zeros(1, dtype=torch.int32, device=dev())
    ws = torch.zeros(_abi.gaussian_workspace_bytes(1, 3, 0, 4), dtype=torch.uint8, device=dev())
    _abi.hmc_gaussian_sample(cur, cur.clone(), t3.precision, t3.mean, t3.log_norm, 0, None, None, 25, 0.3, 0, 0, 0, 1, 0, None, rej,
                             workspace=ws)
    assert torch.equal(cur.cpu(), torch.tensor(th1)) and int(rej.item()) == 0
    # 70000 chains: past quad_max_chains -> one chain per lane
    C = 70000
    th0 = (0.3 * O.philox_normals(3, np.arange(C), 0, 3, O.PURPOSE_INIT, dtype=np.float64)).astype(np.float32)
    out = ht.sample(t3, tt(th0, dtype), num_samples=6, num_steps_per_sample=10, step_size=0.3, verbose=False, seed=3)
    ref, _ = O.sample_hmc(o3, th0, 6, 10, 0.3, 0, None, O.PhiloxDraws(3, np.arange(C)))
    _compare_runs(out, ref, 2e-4, 0.002)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.float64, 1e-9)])
@pytest.mark.parametrize("D", [7, 20, 64, 70, 100, 128])
def test_wave_eigenbasis_route_vs_direct_and_oracle(ht, dtype, tol, D):
    """D > 6, identity mass: the wave-per-chain kernel integrates in the eigenbasis of P (diagonalised once per launch by
    the Jacobi kernel; two D x D products per trajectory instead of one per step).  Against the direct wave kernel
    (hta_set_tuning('gauss_eig', 0)) and the oracle on the same Philox draws, with burn-in (Q2 reset) and a mean offset."""
    from hamiltorch_amd import _abi
    C, N, L, eps, seed, burn = 37, 12, 6, 0.15, 40 + D, 2
    rng = np.random.default_rng(D)
    P = rand_spd(D, 3)
    mu = rng.normal(size=D)
    t, o = targets(ht, P, dtype, mu=mu)
    th0 = (mu + 0.5 * O.philox_normals(seed, np.arange(C), 0, D, O.PURPOSE_INIT, dtype=np.float64)).astype(NP[dtype])
    outs = []
    for mode in (1, 0):
        _abi.set_tuning("gauss_eig", mode)
        try:
            out, acc = ht.sample(t, tt(th0, dtype), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=burn, debug=2,
                                 verbose=False, seed=seed)
        finally:
            _abi.set_tuning("gauss_eig", 1)
        outs.append((out, acc.cpu().numpy()))
    ref, info = O.sample_hmc(o, th0, N, L, eps, burn, None, O.PhiloxDraws(seed, np.arange(C), NP[dtype]))
    bad_e = _compare_runs(outs[0][0], ref, tol, 0.06)
    bad_d = _compare_runs(outs[1][0], ref, tol, 0.06)
    good = ~(bad_e | bad_d)
    np.testing.assert_allclose(outs[0][1][good], outs[1][1][good], atol=1e-12)
    assert 0.3 < outs[0][1].mean() <= 1.0


@pytest.mark.parametrize("D", [3, 6, 40])
def test_eigenbasis_routes_with_indefinite_curvature_and_divergence(ht, D):
    """The eigenbasis kernels do not need a positive definite P: with negative eigenvalues the map is the same leapfrog map
    (hyperbolic in those coordinates) and agrees with the direct kernels; with a step size that makes trajectories overflow
    every non-finite proposal is rejected and the states stay finite (S:1045-1057)."""
    from hamiltorch_amd import _abi
    dtype = torch.float32
    rng = np.random.default_rng(D)
    Qm, _ = np.linalg.qr(rng.normal(size=(D, D)))
    lam = np.concatenate([[-0.5, 1.0, 2.0], rng.uniform(0.3, 2.0, size=D)])[:D]
    P = (Qm * lam) @ Qm.T
    t, _ = targets(ht, 0.5 * (P + P.T), dtype)
    C = 64
    th0 = tt(0.1 * rng.normal(size=(C, D)), dtype)
    outs = []
    for mode in (1, 0):
        _abi.set_tuning("gauss_eig", mode)
        try:
            out, acc = ht.sample(t, th0, num_samples=8, num_steps_per_sample=10, step_size=0.1, debug=2, verbose=False, seed=3)
        finally:
            _abi.set_tuning("gauss_eig", 1)
        outs.append((torch.stack(out).cpu().numpy(), acc.cpu().numpy()))
    err = np.abs(outs[0][0] - outs[1][0]).max(axis=(0, 2))
    assert (err > 2e-3).mean() <= 0.05, err.max()
    assert np.isfinite(outs[0][0]).all()
    # overflow: |lambda_min| eps^2 L^2 large -> exp growth past fp32 along the negative direction
    big = ht.sample(t, th0, num_samples=6, num_steps_per_sample=400, step_size=2.0, verbose=False, seed=4)
    s = torch.stack(big)
    assert torch.isfinite(s).all()
    # every proposal diverged and was rejected: the rows repeat params_init bit for bit (S:1018), as in the reference
    assert torch.equal(s[-1], s[0])


@pytest.mark.parametrize("D,C,N,L,burn", [(3, 1024, 1000, 25, -1), (3, 1024, 200, 25, 20), (2, 512, 333, 10, 5), (4, 264, 97, 5, -1),
                                          (1, 64, 40, 7, 3), (3, 4096, 120, 25, -1)])
def test_fused_quad_launch_is_bit_identical(ht, D, C, N, L, burn):
    """Round 4, "quad_fused": the draw records produced INSIDE the trajectory launch (producer blocks behind the consumer blocks,
    chunk counters released at agent scope, the consumers' look-ahead gated per pass) instead of by a pre-draw launch in front of
    it.  Same records, same arithmetic: samples, reject counts and final state equal bit for bit - at BASELINE config 2's exact
    shape, with burn-in (two phases, the Q2 trajectory), every step-count instance, D = 1 ... 4, chain counts whose rows are
    whole 128-byte lines, and REPEATED launches on one workspace (the last consumer of a launch resets the counters; a stale
    counter or a record read before its chunk was released would show here as a difference or a hang)."""
    from hamiltorch_amd import _abi
    rng = np.random.default_rng(100 * D + L)
    mu = rng.normal(size=D)
    t, _ = targets(ht, rand_spd(D, 8 + D), torch.float32, mu=mu)
    th0 = tt(mu + rng.normal(size=(C, D)), torch.float32)
    nbytes = _abi.gaussian_workspace_bytes(C, D, N, 4)
    outs = {}
    for fused in (0, 1):
        _abi.set_tuning("quad_fused", fused)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev())
        _abi.hmc_gaussian_prepare(th0, t.precision, 0, None, C, D, N, ws)
        res = []
        cur = th0.clone()
        for rep in range(4):                                  # the state travels from launch to launch (another seed each)
            nrow = N - max(burn, 0) + 1
            samples = torch.zeros(nrow, C, D, device=dev()); rej = torch.zeros(C, dtype=torch.int32, device=dev())
            _abi.hmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, 0, None, None, L, 0.3, N, 0, burn, 13 + rep, 0, samples, rej,
                                     workspace=ws)
            route = _abi.last_route()
            res.append(torch.cat([samples.reshape(-1), rej.float()]))
        torch.cuda.synchronize()
        assert route.startswith("hmc_gauss_quad_fused_kernel<%d" % D if fused else "hmc_gauss_quad_kernel<%d,false" % D), route
        outs[fused] = (torch.cat(res).cpu(), cur.cpu())
        _abi.hmc_gaussian_forget(ws)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[1][0]).all()


def test_fused_launch_reports_starved_producers(ht):
    """VERDICT r04 item 6 / ADVICE r04 (medium): the consumers of the fused launch wait for producer blocks of the same grid; a
    producer that is never scheduled (debug key "quad_starve": the producers leave at once) used to turn, after the bounded wait,
    into consumers that READ ON - plausible wrong samples with rc 0.  Now: the workspace's STICKY status word (ABI 10:
    hta_hmc_gaussian_status_offset) goes non-zero, every row the starved launch stored and the chain state are NaN, later launches on
    that workspace stay flagged (and NaN) until it is prepared again, and the Python layer raises DeviceStatusError - on the spot
    where it synchronises anyway (verbose / debug = 2), at the next entry or at check_device_status() otherwise."""
    from hamiltorch_amd import _abi, util
    C, D, N, L = 256, 3, 64, 5
    t, _ = targets(ht, rand_spd(D, 8), torch.float32)
    th0 = torch.randn(C, D, generator=torch.Generator().manual_seed(3)).to(dev())
    ws = torch.zeros(_abi.gaussian_workspace_bytes(C, D, N, 4), dtype=torch.uint8, device=dev())
    _abi.hmc_gaussian_prepare(th0, t.precision, 0, None, C, D, N, ws)
    word = _abi.hmc_gaussian_status_word(ws, C, D, N, 4)
    assert word is not None and _abi.hmc_gaussian_status_word(ws, C, 7, N, 4) is None

    def launch(seed):
        cur = th0.clone()
        samples = torch.zeros(N + 1, C, D, device=dev()); rej = torch.zeros(C, dtype=torch.int32, device=dev())
        _abi.hmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, 0, None, None, L, 0.3, N, 0, -1, seed, 0, samples, rej, workspace=ws)
        torch.cuda.synchronize()
        return cur, samples[1:]
    cur, s = launch(1)
    assert _abi.last_route().startswith("hmc_gauss_quad_fused_kernel<3") and int(word) == 0 and torch.isfinite(s).all()
    good = s.clone()
    _abi.set_tuning("quad_starve", 1)
    cur, s = launch(1)                                       # ~1 s: every consumer wave polls out its bound
    assert int(word) != 0, "the starved launch did not raise the status word"
    assert torch.isnan(s).all() and torch.isnan(cur).all(), "a starved launch must not leave plausible rows behind"
    _abi.set_tuning("quad_starve", 0)
    cur, s = launch(1)                                       # the word is sticky: the launch itself is healthy again ...
    assert int(word) != 0 and torch.equal(s, good)           # ... and says so in its data, but the workspace stays flagged
    _abi.hmc_gaussian_prepare(th0, t.precision, 0, None, C, D, N, ws)      # only a new preparation clears it
    cur, s = launch(1)
    assert int(word) == 0 and torch.equal(s, good)
    _abi.hmc_gaussian_forget(ws)

    # through the public API
    tgt = ht.GaussianTarget(t.mean, precision=t.precision)
    kw = dict(num_samples=N, num_steps_per_sample=L, step_size=0.3, burn=-1, seed=5)
    ok = ht.sample(tgt, th0, verbose=False, **kw)
    assert torch.isfinite(torch.stack(ok)).all()
    _abi.set_tuning("quad_starve", 1)
    with pytest.raises(util.DeviceStatusError, match="draw records"):
        ht.sample(tgt, th0, debug=2, verbose=False, **kw)              # synchronises for the acceptance rate: raises on the spot
    tgt2 = ht.GaussianTarget(t.mean, precision=t.precision.clone())    # (a fresh target = a fresh workspace)
    bad = ht.sample(tgt2, th0, verbose=False, **kw)                    # no synchronisation inside: returns ...
    assert torch.isnan(torch.stack(bad)[1:]).all()                     # ... NaN rows, never plausible ones
    _abi.set_tuning("quad_starve", 0)
    with pytest.raises(util.DeviceStatusError):
        util.check_device_status()                                       # the explicit check
    tgt3 = ht.GaussianTarget(t.mean, precision=t.precision.clone())
    bad = ht.sample(tgt3, th0, verbose=False, **kw)
    util.check_device_status()                                           # healthy again: nothing pending
    assert torch.isfinite(torch.stack(bad)).all()
    _abi.set_tuning("quad_starve", 1)
    tgt4 = ht.GaussianTarget(t.mean, precision=t.precision.clone())
    ht.sample(tgt4, th0, verbose=False, **kw)
    torch.cuda.synchronize()                                             # the asynchronous copy of the word has landed
    _abi.set_tuning("quad_starve", 0)
    with pytest.raises(util.DeviceStatusError):
        ht.sample(tgt, th0, verbose=False, **kw)                         # the NEXT entry into the library raises for the earlier run
    again = ht.sample(tgt, th0, verbose=False, **kw)                     # reported once; the flagged workspace went with the report:
    util.check_device_status()                                           # `tgt` runs on a freshly prepared one
    assert torch.equal(torch.stack(again), torch.stack(ok))


def test_fused_quad_launch_needs_a_prepared_workspace_and_whole_lines(ht):
    """The fused launch is only taken where its assumptions hold: a prepared workspace (zeroed counters), rows that are whole
    128-byte lines (C a multiple of 8), no diagnostics outputs; anything else keeps the two-launch form - same results."""
    from hamiltorch_amd import _abi
    t, _ = targets(ht, rand_spd(3, 8), torch.float32)
    _abi.set_tuning("quad_fused", 1)
    for C, prepare, want in ((256, False, "hmc_gauss_quad_kernel<3"), (260, True, "hmc_gauss_quad_kernel<3"), (256, True, "hmc_gauss_quad_fused_kernel<3")):
        th0 = torch.randn(C, 3, generator=torch.Generator().manual_seed(C)).to(dev())
        N = 50
        ws = torch.zeros(_abi.gaussian_workspace_bytes(C, 3, N, 4), dtype=torch.uint8, device=dev())
        if prepare:
            _abi.hmc_gaussian_prepare(th0, t.precision, 0, None, C, 3, N, ws)
        cur = th0.clone(); samples = torch.zeros(N + 1, C, 3, device=dev()); rej = torch.zeros(C, dtype=torch.int32, device=dev())
        _abi.hmc_gaussian_sample(cur, th0, t.precision, t.mean, t.log_norm, 0, None, None, 25, 0.3, N, 0, -1, 5, 0, samples, rej, workspace=ws)
        assert _abi.last_route().startswith(want), (C, prepare, _abi.last_route())
        torch.cuda.synchronize()
        _abi.hmc_gaussian_forget(ws)

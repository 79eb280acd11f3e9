"""Pins oracle/hmc_oracle.py to fixtures produced by the unmodified reference
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

import hmc_oracle as O


def gauss3(dtype):
    g = np.load  # noqa
    sigma = np.array([[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]])
    return O.GaussianTarget.from_cov(np.zeros(3), sigma, dtype=dtype)


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert [int(v) for v in O.philox4x32(0, 0, 0, 0, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    m = 0xFFFFFFFF
    assert [int(v) for v in O.philox4x32(m, m, m, m, m, m)] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert [int(v) for v in O.philox4x32(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)] == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_philox_normal_moments():
    z = O.philox_normals(99, np.arange(256), 3, 1024, dtype=np.float64)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(np.mean(z ** 4) - 3) < 0.05
    u = O.philox_uniforms(99, np.arange(256), 3, 1024, dtype=np.float32)
    assert u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.01


def test_kat1_leapfrog(golden):
    g = golden("hmc_kat")
    tgt = O.GaussianTarget.from_cov(np.zeros(2), np.diag([0.1, 0.1]), dtype=np.float32)
    one = np.ones((1, 2), np.float32)
    for steps in (1, 3, 100):
        th, p = O.hmc_leapfrog(one, one, tgt.grad, steps, 0.1, np.ones(2, np.float32))
        np.testing.assert_allclose(th[0], g[f"kat1_theta_{steps}"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(p[0], g[f"kat1_p_{steps}"], rtol=2e-5, atol=2e-5)
    # reversibility, ported from tests/test_util.py:97-110 (tolerance instead of bit-equality)
    th, p = O.hmc_leapfrog(one, one, tgt.grad, 100, 0.1, np.ones(2, np.float32))
    th2, _ = O.hmc_leapfrog(th, -p, tgt.grad, 100, 0.1, np.ones(2, np.float32))
    np.testing.assert_allclose(th2[0], [1.0, 1.0], atol=1e-4)
    np.testing.assert_allclose(g["kat1_reversed_theta"], [1.0, 1.0], atol=0)


@pytest.mark.parametrize("tag,dt,tol", [("f32", np.float32, 2e-5), ("f64", np.float64, 1e-12)])
def test_kat2_hamiltonian_and_leapfrog(golden, tag, dt, tol):
    g = golden("hmc_kat")
    tgt = gauss3(dt)
    th = np.array([[0.3, -0.2, 0.5]], dt); pm = np.array([[0.1, 0.7, -0.4]], dt)
    np.testing.assert_allclose(tgt.logp(th)[0], g[f"kat2_logp_{tag}"], rtol=tol, atol=tol)
    masses = {"none": None, "diag": np.array([1.0, 0.5, 2.0], dt), "full": g[f"kat2_inv_mass_full_{tag}"]}
    for mk, im in masses.items():
        H, _ = O.hmc_hamiltonian(th, pm, tgt.logp, im)
        np.testing.assert_allclose(H, g[f"kat2_H_{mk}_{tag}"], rtol=tol, atol=tol)
        pt, pp = O.hmc_leapfrog(th, pm, tgt.grad, 5, 0.3, im, return_path=True)
        np.testing.assert_allclose(np.concatenate(pt), g[f"kat2_theta_{mk}_{tag}"], rtol=tol, atol=tol)
        np.testing.assert_allclose(np.concatenate(pp), g[f"kat2_p_{mk}_{tag}"], rtol=tol, atol=tol)


def test_gibbs(golden):
    g = golden("gibbs")
    np.testing.assert_allclose(O.gibbs_momentum(g["z_none"][None]), g["p_none"][None], atol=0)
    np.testing.assert_allclose(O.gibbs_momentum(g["z_diag"][None], g["mass_diag"])[0], g["p_diag"], rtol=1e-6)
    np.testing.assert_allclose(O.gibbs_momentum(g["z_full"][None], g["mass_full"])[0], g["p_full"], rtol=1e-5, atol=1e-6)
    tgt = O.GaussianTarget(np.zeros(3, np.float32), g["rm_P"])
    p = O.rm_gibbs(np.array([[0.3, -0.2, 0.5]], np.float32), g["rm_z"][None], tgt, 1e6)
    np.testing.assert_allclose(p[0], g["rm_p"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["cfg1", "burn10", "burnm1", "diag", "full"])
def test_sample_hmc_end_to_end(golden, name):
    """The reference's own sample() output (cfg1 and burn/mass variants), its recorded
    draws replayed through the oracle: list length, Q2 quirk, accept decisions, values."""
    g = golden("sample_hmc")
    N, L, eps, burn = g[f"{name}_cfg"]
    N, L, burn = int(N), int(L), int(burn)
    im = g[f"{name}_inv_mass"] if f"{name}_inv_mass" in g.files else None
    tgt = gauss3(np.float32)
    draws = O.ReplayDraws(g[f"{name}_momenta"], g[f"{name}_uniforms"])
    ret, info = O.sample_hmc(tgt, g[f"{name}_init"][None].astype(np.float32), N, L, eps, burn, im, draws)
    ref = g[f"{name}_samples"]
    assert len(ret) == ref.shape[0] == N - burn
    got = np.concatenate(ret)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    assert abs(info["acc_rate"][0] - float(g[f"{name}_acc"])) < 1e-9


def test_q2_quirk_present_in_reference_fixture(golden):
    """burn10 case: first post-burn rejection resets the chain to params_init."""
    g = golden("sample_hmc")
    s = g["burn10_samples"]
    assert np.all(s[0] == g["burn10_init"])


@pytest.mark.parametrize("name", ["d3", "d10", "d6indef"])
@pytest.mark.parametrize("tag,dt,tol", [("f32", np.float32, 3e-4), ("f64", np.float64, 1e-9)])
def test_rmhmc_metric_hamiltonian_leapfrog(golden, name, tag, dt, tol):
    g = golden("rmhmc")
    alpha, omega, eps, steps = g[f"{name}_cfg"]
    P = g[f"{name}_P_{tag}"]
    tgt = O.GaussianTarget(np.zeros(P.shape[0], dt), P)
    th = g[f"{name}_theta0_{tag}"][None]; pm = g[f"{name}_p0_{tag}"][None]
    for mtag in ("softabs", "hessian"):
        if f"{name}_G_{mtag}_{tag}" not in g.files:
            continue
        G, lam, _ = O.softabs_metric(tgt.neg_hessian(th), alpha, metric=mtag)
        np.testing.assert_allclose(G[0], g[f"{name}_G_{mtag}_{tag}"], rtol=tol, atol=tol)
        if lam is not None:
            np.testing.assert_allclose(np.sort(lam[0]), np.sort(g[f"{name}_lam_{mtag}_{tag}"]), rtol=tol, atol=tol)
        x, _ = O.cholesky_inverse(G, pm)
        np.testing.assert_allclose(x[0], g[f"{name}_Ginvp_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)
        H, _ = O.rm_hamiltonian(th, pm, tgt, alpha, metric=mtag)
        np.testing.assert_allclose(H, g[f"{name}_H_{mtag}_{tag}"], rtol=tol, atol=tol)
        a, b, c, d = O.explicit_rmhmc_leapfrog(th, pm, tgt, int(steps), eps, omega, alpha, metric=mtag)
        np.testing.assert_allclose(a[0], g[f"{name}_lf_theta_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)
        np.testing.assert_allclose(b[0], g[f"{name}_lf_p_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)
        np.testing.assert_allclose(c[0], g[f"{name}_lf_thetac_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)
        np.testing.assert_allclose(d[0], g[f"{name}_lf_pc_{mtag}_{tag}"], rtol=10 * tol, atol=10 * tol)


def test_rmhmc_sample_end_to_end(golden):
    g = golden("rmhmc")
    tgt = O.GaussianTarget(np.zeros(3, np.float32), g["e2e_P"])
    draws = O.ReplayDraws(g["e2e_momenta"], g["e2e_uniforms"])
    ret, info = O.sample_rmhmc_explicit(tgt, np.array([[0.3, -0.2, 0.5]], np.float32), 12, 3, 0.25, 10.0, 1e6,
                                        burn=2, draws=draws)
    ref = g["e2e_samples"]
    assert len(ret) == ref.shape[0]
    np.testing.assert_allclose(np.concatenate(ret), ref, rtol=2e-4, atol=2e-4)
    assert abs(info["acc_rate"][0] - float(g["e2e_acc"])) < 1e-9


def _mlp_target(g, name, lo=None, hi=None, prior_scale=1.0):
    M, tau_out, eps, L = g[f"{name}_cfg"]
    act = "relu" if name.startswith("relu") else "tanh"
    X, Y = g[f"{name}_X"], g[f"{name}_Y"]
    if lo is not None:
        X, Y = X[lo:hi], Y[lo:hi]
    return O.MLPRegressionTarget(list(g[f"{name}_dims"]), X, Y, g[f"{name}_tau_list"], tau_out, prior_scale, act)


@pytest.mark.parametrize("name", ["relu2", "tanh3", "relu_cfg4"])
def test_mlp_logp_grad_split_leapfrog(golden, name):
    """relu2 / tanh3: small networks; relu_cfg4: BASELINE config 4 at full size (8-100-1, D=1001, 400 points, M=4, tau_out=100,
    eps=5e-4, L=10; tests/golden/cfg4.npz) - all recorded from the unmodified reference."""
    g = golden("cfg4" if name == "relu_cfg4" else "mlp")
    n_e2e, n_full = g[f"{name}_e2e_samples"].shape[0], g[f"{name}_full_samples"].shape[0]
    M, tau_out, eps, L = g[f"{name}_cfg"]
    M, L = int(M), int(L)
    theta = g[f"{name}_theta"][None].astype(np.float32)
    full = _mlp_target(g, name)
    lp, gr = full.logp_and_grad(theta)
    np.testing.assert_allclose(lp, g[f"{name}_logp"], rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(gr[0], g[f"{name}_grad"], rtol=2e-4, atol=2e-4)
    N = g[f"{name}_X"].shape[0]; nb = N // M
    splits = [_mlp_target(g, name, m * nb, (m + 1) * nb, prior_scale=M) for m in range(M)]
    np.testing.assert_allclose([s.logp(theta)[0] for s in splits], g[f"{name}_split_logp"], rtol=2e-5, atol=1e-4)
    p0 = g[f"{name}_p0"][None].astype(np.float32)
    im = np.ones(theta.shape[1], np.float32)
    H0, _ = O.hmc_hamiltonian(theta, p0, [s.logp for s in splits], im)
    np.testing.assert_allclose(H0, g[f"{name}_H0"], rtol=2e-5)
    th, pm = O.split_leapfrog(theta, p0, [s.grad for s in splits], L, eps, im)
    np.testing.assert_allclose(th[0], g[f"{name}_lf_theta"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pm[0], g[f"{name}_lf_p"], rtol=1e-3, atol=1e-3)
    # end-to-end sample_split_model
    draws = O.ReplayDraws(g[f"{name}_e2e_momenta"], g[f"{name}_e2e_uniforms"])
    ret, info = O.sample_hmc(None, theta, n_e2e, L, eps, 0, im, draws,
                             grad_fns=[s.grad for s in splits], logp_fns=[s.logp for s in splits])
    np.testing.assert_allclose(np.concatenate(ret), g[f"{name}_e2e_samples"], rtol=1e-3, atol=1e-4)
    assert abs(info["acc_rate"][0] - float(g[f"{name}_e2e_acc"])) < 1e-9
    # sample_model (full-data HMC, inv_mass None)
    draws = O.ReplayDraws(g[f"{name}_full_momenta"], g[f"{name}_full_uniforms"])
    ret, _ = O.sample_hmc(full, theta, n_full, L, eps, 0, None, draws)
    np.testing.assert_allclose(np.concatenate(ret), g[f"{name}_full_samples"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("name,seed", [("cfg1", 123), ("burn10", 7), ("burnm1", 8), ("diag", 9), ("full", 10)])
def test_torch_port_cfg1_bit_identical(golden, name, seed):
    """BASELINE config 1: the per-chain torch port (bench.py's cpu_baseline) reproduces the reference's
    sample() output bit for bit from the same torch seed -- same op order, same RNG consumption."""
    import torch
    import torch_port as TP
    g = golden("sample_hmc")
    N, L, eps, burn = g[f"{name}_cfg"]
    cov = torch.tensor(g["sigma3"], dtype=torch.float32)

    def lp(w):
        return torch.distributions.MultivariateNormal(torch.zeros(3), cov).log_prob(w).sum()
    im = torch.tensor(g[f"{name}_inv_mass"]) if f"{name}_inv_mass" in g.files else None
    torch.manual_seed(seed)
    ret, acc = TP.port_sample(lp, torch.tensor(g[f"{name}_init"]), int(N), int(L), float(eps), int(burn), im)
    got = np.stack([t.numpy() for t in ret])
    assert got.shape == g[f"{name}_samples"].shape
    assert np.array_equal(got, g[f"{name}_samples"])
    assert acc == float(g[f"{name}_acc"])


def test_torch_port_rmhmc_matches_reference_run(golden):
    """The RMHMC port (autograd through hessian + eigh, as the reference) reproduces the reference's explicit-RMHMC
    sample() from the same torch seed."""
    import torch
    import torch_port as TP
    g = golden("rmhmc")
    P = torch.tensor(g["e2e_P"])

    def lp(w):
        return -0.5 * torch.dot(w, torch.mv(P, w))
    torch.manual_seed(21)
    ret, acc = TP.port_sample_rmhmc(lp, torch.tensor([0.3, -0.2, 0.5]), 12, 3, 0.25, 10.0, 1e6, burn=2)
    got = np.stack([t.numpy() for t in ret])
    assert got.shape == g["e2e_samples"].shape
    np.testing.assert_allclose(got, g["e2e_samples"], rtol=1e-5, atol=1e-6)
    assert acc == float(g["e2e_acc"])


def test_torch_port_split_matches_reference_run(golden):
    """The split-HMC port over MLP closures reproduces the reference's sample_split_model from the same torch seed."""
    import torch
    import torch_port as TP
    g = golden("mlp")
    M, tau_out, eps, L = g["relu2_cfg"]
    net = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))
    X, Y = torch.tensor(g["relu2_X"]), torch.tensor(g["relu2_Y"])
    tau_list = torch.tensor(g["relu2_tau_list"])
    nb = X.shape[0] // int(M)
    fl = [TP.port_mlp_closure(net, X[m * nb:(m + 1) * nb], Y[m * nb:(m + 1) * nb], tau_list, float(tau_out), int(M)) for m in range(int(M))]
    torch.manual_seed(33)
    # sample_split_model builds its closures by iterating the DataLoader (S:1251), which draws the loader's base
    # seed from the global generator before the first momentum draw: replay that consumption
    for _ in torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=nb, shuffle=False):
        pass
    ret, acc = TP.port_sample_split(fl, torch.tensor(g["relu2_theta"]), 10, int(L), float(eps), 0, torch.ones(26))
    got = np.stack([t.numpy() for t in ret])
    np.testing.assert_allclose(got, g["relu2_e2e_samples"], rtol=1e-5, atol=1e-6)
    assert acc == float(g["relu2_e2e_acc"])


def test_nuts_dual_averaging_vs_reference(golden):
    """Sampler.HMC_NUTS = dual-averaging step size (S:629-674): scalar recurrence and an end-to-end adaptive run."""
    g = golden("nuts")
    ss, eb, Ht = 0.3, 1.0, 0.0
    for t, (r, want) in enumerate(zip(g["adapt_rhos"], g["adapt_out"])):
        alpha = 0.0 if not np.isfinite(r) else min(1.0, float(np.exp(np.float32(r))))
        ss, eb, Ht = O.dual_average(alpha, t, 0.3, Ht, eb, 0.75)
        np.testing.assert_allclose([ss, eb, Ht], want, rtol=1e-6, atol=1e-9)
    tgt = gauss3(np.float32)
    draws = O.ReplayDraws(g["e2e_momenta"], g["e2e_uniforms"])
    ret, info = O.sample_hmc(tgt, np.array([[0.5, -0.5, 0.25]], np.float32), 45, 5, 0.05, 20, None, draws, nuts_desired=0.7)
    # the recurrence is fed fp32 energy differences: allow 1e-4 relative on the adapted step size
    assert abs(info["step_size"] - float(g["e2e_step_size"])) < 1e-4 * float(g["e2e_step_size"])
    np.testing.assert_allclose(np.concatenate(ret), g["e2e_samples"], rtol=2e-4, atol=2e-4)


def test_product_adaptation_function_vs_reference(golden):
    """hamiltorch_amd.samplers.adaptation (host scalar logic) against the reference's outputs."""
    from hamiltorch_amd import samplers
    g = golden("nuts")
    ss, eb, Ht = 0.3, 1.0, 0.0
    for t, (r, want) in enumerate(zip(g["adapt_rhos"], g["adapt_out"])):
        ss, eb, Ht = samplers.adaptation(float(r), t, 0.3, Ht, eb, desired_accept_rate=0.75)
        np.testing.assert_allclose([ss, eb, Ht], want, rtol=1e-6, atol=1e-9)


# ---- generic (non-constant-curvature) explicit RMHMC: SURVEY 8f N1 ------------------------------------
def _funnel(g, D, scaled=True):
    return O.FunnelTarget(D, g["scales"][:D - 1] if scaled else None)


@pytest.mark.parametrize("tag", ["a1e6", "a1p3", "d6"])
@pytest.mark.parametrize("dtag,dt,tol", [("f64", np.float64, 2e-8), ("f32", np.float32, 2e-3)])
def test_funnel_metric_hamiltonian_explicit_leapfrog(golden, tag, dtag, dt, tol):
    """fisher / rm_hamiltonian / explicit leapfrog of the reference on the scaled funnel, where dH/dtheta goes
    through hessian + eigh (S:398): the oracle's closed form (softabs_dmetric) must reproduce it."""
    g = golden("funnel")
    D, alpha, omega, eps, steps = g[f"{tag}_cfg"]
    D, steps = int(D), int(steps)
    t = _funnel(g, D)
    key = f"{tag}_{dtag}"
    th, pm = g[f"{key}_theta0"][None].astype(dt), g[f"{key}_p0"][None].astype(dt)
    G, lam_t, _ = O.softabs_metric(t.neg_hessian(th), alpha)
    np.testing.assert_allclose(G[0], g[f"{key}_G"], rtol=tol, atol=tol * np.abs(g[f"{key}_G"]).max())
    np.testing.assert_allclose(np.sort(lam_t[0]), np.sort(g[f"{key}_lam"]), rtol=tol, atol=tol)
    H, _ = O.rm_hamiltonian(th, pm, t, alpha)
    np.testing.assert_allclose(H, g[f"{key}_H"], rtol=tol, atol=tol)
    for n in range(1, steps + 1):        # the reference's path holds the state after every step
        a, b, ac, bc = O.explicit_rmhmc_leapfrog_generic(th, pm, t, n, eps, omega, alpha)
        np.testing.assert_allclose(a[0], g[f"{key}_lf_theta"][n - 1], rtol=tol, atol=tol)
        np.testing.assert_allclose(b[0], g[f"{key}_lf_p"][n - 1], rtol=tol, atol=tol)
    np.testing.assert_allclose(ac[0], g[f"{key}_lf_thetac"], rtol=tol, atol=tol)
    np.testing.assert_allclose(bc[0], g[f"{key}_lf_pc"], rtol=tol, atol=tol)


def test_funnel_explicit_leapfrog_with_recorded_jitter(golden):
    """The notebook's funnel (repeated Hessian eigenvalue) with jitter: every one of the 8 gradient calls of a step
    draws its own torch.rand(D) (S:115); replayed in call order."""
    g = golden("funnel")
    D, alpha, omega, eps, steps, jitter = g["jit_cfg"]
    t = _funnel(g, int(D), scaled=False)
    draws = g["jit_draws"]
    th, pm = g["jit_theta0"][None], g["jit_p0"][None]
    for n in range(1, int(steps) + 1):
        a, b, _, _ = O.explicit_rmhmc_leapfrog_generic(th, pm, t, n, eps, omega, alpha, jitter, lambda k: draws[k][None])
        np.testing.assert_allclose(a[0], g["jit_lf_theta"][n - 1], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(b[0], g["jit_lf_p"][n - 1], rtol=1e-6, atol=1e-6)


def test_funnel_sample_end_to_end(golden):
    """sample(RMHMC, EXPLICIT) on the scaled funnel with the reference's momenta / uniforms replayed (one of its
    trajectories diverges to log p = -inf: the LogProbError rejection path, S:1045-1057)."""
    g = golden("funnel")
    t = _funnel(g, 4)
    draws = O.ReplayDraws(g["e2e_momenta"], g["e2e_uniforms"], dtype=np.float64)
    with np.errstate(all="ignore"):
        ret, info = O.sample_rmhmc_explicit(t, np.array([[0.2, -0.3, 0.4, 0.1]]), 10, 4, 0.1, 10.0, 1e6, burn=1, draws=draws)
    ref = g["e2e_samples"]
    assert len(ret) == ref.shape[0]
    np.testing.assert_allclose(np.concatenate(ret), ref, rtol=1e-6, atol=1e-6)
    assert abs(info["acc_rate"][0] - float(g["e2e_acc"])) < 1e-9


# ---- SPLITTING_RAND / SPLITTING_KMID: SURVEY 8f N3 --------------------------------------------------------
def _split_targets(g):
    M, tau_out, eps, L = g["cfg"]
    M = int(M); nb = g["X"].shape[0] // M
    return [O.MLPRegressionTarget(list(g["dims"]), g["X"][m * nb:(m + 1) * nb], g["Y"][m * nb:(m + 1) * nb], g["tau_list"],
                                  tau_out, M, "relu") for m in range(M)], float(eps), int(L)


@pytest.mark.parametrize("kind", ["rand", "kmid"])
def test_split_kinds_leapfrog_and_sample(golden, kind):
    """The two other data-split integrators of the reference (S:547-596): 3-step leapfrog paths and an end-to-end
    sample_split_model run with the reference's own momenta, uniforms and (RAND) torch.randperm orders replayed."""
    g = golden("splitkinds")
    tg, eps, L = _split_targets(g)
    grads = [t.grad for t in tg]; logps = [t.logp for t in tg]
    th, p0 = g["theta"][None].astype(np.float32), g["p0"][None].astype(np.float32)
    im = np.ones(th.shape[1], np.float32)
    perm = [int(v) for v in g["rand_lf_perm"][0]] if kind == "rand" else None
    for n in range(1, 4):
        a, b = O.split_leapfrog(th, p0, grads, n, eps, im, kind, perm)
        np.testing.assert_allclose(a[0], g[f"{kind}_lf_theta"][n - 1], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(b[0], g[f"{kind}_lf_p"][n - 1], rtol=2e-4, atol=2e-4)
    draws = O.ReplayDraws(g[f"{kind}_e2e_momenta"], g[f"{kind}_e2e_uniforms"], perms=g["rand_e2e_perms"] if kind == "rand" else None)
    ret, info = O.sample_hmc(None, th, 8, L, eps, 0, im, draws, grad_fns=grads, logp_fns=logps, split_kind=kind)
    np.testing.assert_allclose(np.concatenate(ret), g[f"{kind}_e2e_samples"], rtol=2e-4, atol=2e-4)
    assert abs(info["acc_rate"][0] - float(g[f"{kind}_e2e_acc"])) < 1e-9


def test_philox_permutation_is_a_permutation_and_matches_host():
    from hamiltorch_amd import util
    for seed, draw, M in ((1, 0, 1), (1, 0, 2), (99, 7, 5), (2 ** 45 + 3, 11, 64)):
        pm = O.philox_permutation(seed, draw, M)
        assert sorted(pm) == list(range(M))
        assert pm == util.split_permutation(seed, draw, M)
    # not stuck on the identity
    assert any(O.philox_permutation(5, n, 6) != list(range(6)) for n in range(4))


# ---- implicit RMHMC (generalised leapfrog, fixed-point iterations): SURVEY 8f N4 ---------------------------
@pytest.mark.parametrize("tag", ["imp_a1e6", "imp_a1p1"])
def test_implicit_rmhmc_leapfrog_vs_reference(golden, tag):
    g = golden("funnel")
    D, alpha, eps, steps, thr, max_it = g[f"{tag}_cfg"]
    t = _funnel(g, int(D))
    th, pm = g[f"{tag}_theta0"][None], g[f"{tag}_p0"][None]
    for n in range(1, int(steps) + 1):
        a, b = O.implicit_rmhmc_leapfrog(th, pm, t, n, eps, alpha, thr, int(max_it))
        np.testing.assert_allclose(a[0], g[f"{tag}_lf_theta"][n - 1], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(b[0], g[f"{tag}_lf_p"][n - 1], rtol=1e-7, atol=1e-7)


def test_implicit_rmhmc_sample_end_to_end(golden):
    g = golden("funnel")
    t = _funnel(g, 4)
    draws = O.ReplayDraws(g["imp_e2e_momenta"], g["imp_e2e_uniforms"], dtype=np.float64)
    with np.errstate(all="ignore"):
        ret, info = O.sample_rmhmc_implicit(t, np.array([[0.2, -0.3, 0.4, 0.1]]), 8, 3, 0.1, 1e6, 1e-14, 40, burn=1, draws=draws)
    ref = g["imp_e2e_samples"]
    assert len(ret) == ref.shape[0]
    np.testing.assert_allclose(np.concatenate(ret), ref, rtol=1e-6, atol=1e-6)
    assert abs(info["acc_rate"][0] - float(g["imp_e2e_acc"])) < 1e-9


def test_hessian_metric_general_target_vs_reference(golden):
    """Metric.HESSIAN on the log-cosh target (log-concave, not Gaussian): explicit and implicit RMHMC paths of the
    reference against the oracle's closed form M = 1/2 G^-1 - 1/2 v v^T."""
    g = golden("logcosh")
    omega, eps, steps, thr, max_it = g["cfg"]
    t = O.LogCoshTarget(g["P"], g["A"])
    th, pm = g["theta0"][None], g["p0"][None]
    np.testing.assert_allclose(t.neg_hessian(th)[0], g["G"], rtol=1e-10, atol=1e-10)
    H, _ = O.rm_hamiltonian(th, pm, t, 1.0, metric="hessian")
    np.testing.assert_allclose(H, g["H"], rtol=1e-9, atol=1e-9)
    for n in range(1, int(steps) + 1):
        a, b, _, _ = O.explicit_rmhmc_leapfrog_generic(th, pm, t, n, eps, omega, 1.0, metric="hessian")
        np.testing.assert_allclose(a[0], g["exp_theta"][n - 1], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(b[0], g["exp_p"][n - 1], rtol=1e-8, atol=1e-8)
        a, b = O.implicit_rmhmc_leapfrog(th, pm, t, n, eps, 1.0, thr, int(max_it), metric="hessian")
        np.testing.assert_allclose(a[0], g["imp_theta"][n - 1], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(b[0], g["imp_p"][n - 1], rtol=1e-7, atol=1e-7)


def test_block_list_inv_mass(golden):
    """inv_mass given as a list of diagonal blocks (S:188-197, S:287-292, S:803-809, S:944-947): the reference's gibbs,
    hamiltonian, leapfrog and sample() on a 5-D Gaussian with blocks 2x2 + 3x3, recorded draws replayed."""
    g = golden("blockmass")
    blocks = [g["b0"], g["b1"]]
    np.testing.assert_allclose(O.gibbs_momentum(g["gibbs_z"][None], blocks)[0], g["gibbs_p"], rtol=2e-6, atol=1e-6)   # mass = blocks
    for inv, b in zip(O.invert_mass(blocks), blocks):
        np.testing.assert_allclose(inv @ b, np.eye(b.shape[0]), atol=1e-6)
    tgt = O.GaussianTarget(np.zeros(5, np.float32), g["P"])
    th, pm = g["kat_theta"][None], g["kat_p"][None]
    H, _ = O.hmc_hamiltonian(th, pm, tgt.logp, blocks)
    np.testing.assert_allclose(H, g["kat_H"], rtol=1e-6)
    tn, pn = O.hmc_leapfrog(th, pm, tgt.grad, 4, 0.2, blocks)
    np.testing.assert_allclose(tn[0], g["kat_theta_L"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pn[0], g["kat_p_L"], rtol=1e-5, atol=1e-6)
    N, L, eps, burn = g["cfg"]
    draws = O.ReplayDraws(g["momenta"], g["uniforms"])
    ret, info = O.sample_hmc(tgt, g["init"][None], int(N), int(L), eps, int(burn), blocks, draws)
    assert len(ret) == g["samples"].shape[0] == int(N) - int(burn)
    np.testing.assert_allclose(np.concatenate(ret), g["samples"], rtol=1e-4, atol=1e-4)
    assert abs(info["acc_rate"][0] - float(g["acc"])) < 1e-9
    # the block list is the block-diagonal full matrix
    full = np.zeros((5, 5), np.float32); full[:2, :2] = g["b0"]; full[2:, 2:] = g["b1"]
    np.testing.assert_allclose(O.kinetic(pm, blocks), O.kinetic(pm, full), rtol=1e-6)


@pytest.mark.parametrize("tag,dt,tol,htol", [("f32", np.float32, 1e-4, 1e-3), ("f64", np.float64, 1e-9, 1e-9)])
def test_cfg3_full_size_metric_hamiltonian_leapfrog(golden, tag, dt, tol, htol):
    """BASELINE config 3 at D=100 against the unmodified reference (tests/golden/cfg3.npz): soft-abs eigenvalues, G^-1 p,
    the Riemannian Hamiltonian and every step of a 3-step explicit leapfrog path.  SURVEY 8c tolerances: 1e-4 on theta / p,
    1e-3 on H in fp32 (the reference's own fp32-vs-fp64 distance is 1e-6)."""
    g = golden("cfg3")
    D, alpha, omega, eps, _ = g["cfg"]
    P = g[f"P_{tag}"]
    tgt = O.GaussianTarget(np.zeros(int(D), dt), P)
    th, pm = g[f"theta0_{tag}"][None], g[f"p0_{tag}"][None]
    G, lam, _ = O.softabs_metric(tgt.neg_hessian(th), alpha, metric="softabs")
    np.testing.assert_allclose(np.sort(lam[0]), np.sort(g[f"lam_{tag}"]), rtol=tol, atol=tol)
    np.testing.assert_allclose(np.diagonal(G[0]), g[f"Gdiag_{tag}"], rtol=tol, atol=tol)
    np.testing.assert_allclose(O.cholesky_inverse(G, pm)[0][0], g[f"Ginvp_{tag}"], rtol=tol, atol=tol)
    H, _ = O.rm_hamiltonian(th, pm, tgt, alpha, metric="softabs")
    np.testing.assert_allclose(H, g[f"H_{tag}"], rtol=0, atol=htol)
    for n in (1, 2, 3):
        a, b, c, d = O.explicit_rmhmc_leapfrog(th, pm, tgt, n, eps, omega, alpha, metric="softabs")
        np.testing.assert_allclose(a[0], g[f"lf_theta_{tag}"][n - 1], rtol=0, atol=tol)
        np.testing.assert_allclose(b[0], g[f"lf_p_{tag}"][n - 1], rtol=0, atol=tol)
    np.testing.assert_allclose(c[0], g[f"lf_thetac_{tag}"], rtol=0, atol=tol)
    np.testing.assert_allclose(d[0], g[f"lf_pc_{tag}"], rtol=0, atol=tol)


def test_cfg3_full_size_leapfrog_with_recorded_jitter(golden):
    """cfg3 with jitter=1e-3: the reference draws torch.rand(D) in each of the 8 metric evaluations of a step (S:115); the
    recorded draws replayed in call order through the oracle give the reference's path."""
    g = golden("cfg3")
    D, alpha, omega, eps, jitter = g["cfg"]
    tgt = O.GaussianTarget(np.zeros(int(D), np.float32), g["P_f32"])
    th, pm = g["theta0_f32"][None], g["p0_f32"][None]
    draws = g["jit_draws"]
    for n in (1, 2):
        a, b, _, _ = O.explicit_rmhmc_leapfrog(th, pm, tgt, n, eps, omega, alpha, jitter, lambda k: draws[k][None])
        np.testing.assert_allclose(a[0], g["jit_lf_theta"][n - 1], rtol=0, atol=1e-4)
        np.testing.assert_allclose(b[0], g["jit_lf_p"][n - 1], rtol=0, atol=1e-4)
    # the jitter matters at this tolerance: without it the path is measurably different
    a0, _, _, _ = O.explicit_rmhmc_leapfrog(th, pm, tgt, 2, eps, omega, alpha)
    assert np.abs(a0[0] - g["jit_lf_theta"][1]).max() > 1e-6


def test_cfg3_full_size_sample_end_to_end(golden):
    """hamiltorch.sample(RMHMC, EXPLICIT, SOFTABS, jitter=1e-3) at D=100 with every draw of the reference recorded
    (momenta, Metropolis uniforms, the 8 L + 3 jitter vectors per trajectory)."""
    g = golden("cfg3")
    D, alpha, omega, eps, jitter = g["cfg"]
    N, L = (int(v) for v in g["e2e_cfg"])
    tgt = O.GaussianTarget(np.zeros(int(D), np.float32), g["P_f32"])
    draws = O.ReplayDraws(g["e2e_momenta"], g["e2e_uniforms"], jitters=g["e2e_jitters"], jitters_per_traj=8 * L + 3)
    ret, info = O.sample_rmhmc_explicit(tgt, g["theta0_f32"][None], N, L, eps, omega, alpha, burn=0, jitter=jitter, draws=draws)
    ref = g["e2e_samples"]
    assert len(ret) == ref.shape[0]
    np.testing.assert_allclose(np.concatenate(ret), ref, rtol=0, atol=2e-4)
    assert abs(info["acc_rate"][0] - float(g["e2e_acc"])) < 1e-9


@pytest.mark.parametrize("tag,dt,tol", [("f32", np.float32, 1e-5), ("f64", np.float64, 1e-12)])
def test_cfg2_leapfrog_paths_L25(golden, tag, dt, tol):
    """BASELINE config 2's per-chain computation against the unmodified reference (tests/golden/cfg2.npz): every step of
    four 25-step leapfrog paths, identity mass, eps=0.3.  SURVEY 8c tolerance for HMC: atol = rtol = 1e-5 (the reference's
    own fp32-vs-fp64 distance at L=25 is 1e-7)."""
    g = golden("cfg2")
    tgt = gauss3(dt)
    for k in range(4):
        th, pm = g["theta0"][k][None].astype(dt), g["p0"][k][None].astype(dt)
        pt, pp = O.hmc_leapfrog(th, pm, tgt.grad, 25, 0.3, None, return_path=True)
        np.testing.assert_allclose(np.concatenate(pt), g[f"lf_theta_{tag}"][k], rtol=tol, atol=tol)
        np.testing.assert_allclose(np.concatenate(pp), g[f"lf_p_{tag}"][k], rtol=tol, atol=tol)


def test_cfg2_sample_end_to_end_L25(golden):
    """hamiltorch.sample(HMC, L=25, eps=0.3) on the KAT2 target, 40 trajectories, the reference's draws replayed."""
    g = golden("cfg2")
    tgt = gauss3(np.float32)
    draws = O.ReplayDraws(g["e2e_momenta"], g["e2e_uniforms"])
    ret, info = O.sample_hmc(tgt, g["theta0"][0][None].astype(np.float32), 40, 25, 0.3, 0, None, draws)
    ref = g["e2e_samples"]
    assert len(ret) == ref.shape[0] == 40
    np.testing.assert_allclose(np.concatenate(ret), ref, rtol=1e-4, atol=1e-4)
    assert abs(info["acc_rate"][0] - float(g["e2e_acc"])) < 1e-9
    assert 0.5 < float(g["e2e_acc"]) <= 1.0


def test_torch_port_cfg3_with_jitter_matches_reference_run(golden):
    """The RMHMC port at BASELINE config 3's size WITH jitter (bench.py's cfg3 cpu_baseline): same torch seed -> the same
    torch.rand(D) / MultivariateNormal / torch.rand(1) consumption as the reference, hence its sample() output."""
    import torch
    import torch_port as TP
    g = golden("cfg3")
    D, alpha, omega, eps, jitter = g["cfg"]
    N, L = (int(v) for v in g["e2e_cfg"])
    P = torch.tensor(g["P_f32"])

    def lp(w):
        return -0.5 * torch.dot(w, torch.mv(P, w))
    torch.manual_seed(9)
    ret, acc = TP.port_sample_rmhmc(lp, torch.tensor(g["theta0_f32"]), N, L, float(eps), float(omega), float(alpha), burn=0,
                                    jitter=float(jitter))
    got = np.stack([t.numpy() for t in ret])
    assert got.shape == g["e2e_samples"].shape
    np.testing.assert_allclose(got, g["e2e_samples"], rtol=0, atol=2e-5)
    assert acc == float(g["e2e_acc"])


def test_torch_port_cfg4_full_size_matches_reference_run(golden):
    """bench.py's cfg4 cpu_baseline (split-HMC port over MLP closures) at BASELINE config 4's size: Linear(8,100)-ReLU-
    Linear(100,1), 4 splits of 100 points, tau_out=100, eps=5e-4, L=10 - the reference's sample_split_model from the same seed."""
    import torch
    import torch_port as TP
    g = golden("cfg4")
    name = "relu_cfg4"
    M, tau_out, eps, L = g[f"{name}_cfg"]
    M, L = int(M), int(L)
    net = torch.nn.Sequential(torch.nn.Linear(8, 100), torch.nn.ReLU(), torch.nn.Linear(100, 1))
    X, Y = torch.tensor(g[f"{name}_X"]), torch.tensor(g[f"{name}_Y"])
    tau_list = torch.tensor(g[f"{name}_tau_list"])
    nb = X.shape[0] // M
    fl = [TP.port_mlp_closure(net, X[m * nb:(m + 1) * nb], Y[m * nb:(m + 1) * nb], tau_list, float(tau_out), M) for m in range(M)]
    torch.manual_seed(33)
    for _ in torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=nb, shuffle=False):
        pass                                       # the loader's base-seed draw (see test_torch_port_split_matches_reference_run)
    n = g[f"{name}_e2e_samples"].shape[0]
    ret, acc = TP.port_sample_split(fl, torch.tensor(g[f"{name}_theta"]), n, L, float(eps), 0, torch.ones(1001))
    got = np.stack([t.numpy() for t in ret])
    np.testing.assert_allclose(got, g[f"{name}_e2e_samples"], rtol=1e-5, atol=1e-6)
    assert acc == float(g[f"{name}_e2e_acc"])


def test_torch_port_cfg2_bit_identical(golden):
    """bench.py's cfg2 cpu_baseline (one chain of BASELINE config 2: L=25, eps=0.3) reproduces the reference's sample()
    bit for bit from the same torch seed."""
    import torch
    import torch_port as TP
    g = golden("cfg2")
    cov = torch.tensor([[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]])

    def lp(w):
        return torch.distributions.MultivariateNormal(torch.zeros(3), cov).log_prob(w).sum()
    torch.manual_seed(77)
    ret, acc = TP.port_sample(lp, torch.tensor(g["theta0"][0], dtype=torch.float32), 40, 25, 0.3, 0, None)
    got = np.stack([t.numpy() for t in ret])
    assert got.shape == g["e2e_samples"].shape
    assert np.array_equal(got, g["e2e_samples"])
    assert acc == float(g["e2e_acc"])


def test_mlp_binary_logits_likelihood_vs_reference_fixture(golden):
    """define_model_log_prob(model_loss='binary_class_linear_output') (S:1172) on Linear(4,6)-Tanh-Linear(6,1), tau_out = 2:
    value and gradient of the oracle's Bernoulli-with-logits likelihood against the unmodified reference (tests/golden/losses.npz)."""
    g = golden("losses")
    o = O.MLPRegressionTarget([4, 6, 1], g["binary_X"], g["binary_Y"], g["binary_tau_list"], 2.0, 1.0, "tanh",
                              loss="binary_class_linear_output")
    lp, gr = o.logp_and_grad(g["binary_theta"][None].astype(np.float64))
    np.testing.assert_allclose(lp, g["binary_logp"], rtol=2e-6)
    np.testing.assert_allclose(gr[0], g["binary_grad"], rtol=2e-5, atol=2e-6)


def test_mlp_softmax_likelihood_vs_reference_fixture(golden):
    """define_model_log_prob(model_loss='multi_class_linear_output') (S:1173-1178) on Linear(4,6)-Tanh-Linear(6,3), tau_out = 2:
    value and gradient of the oracle's softmax cross-entropy likelihood against the unmodified reference (tests/golden/losses.npz)."""
    g = golden("losses")
    o = O.MLPRegressionTarget([4, 6, 3], g["multi_X"], g["multi_Y"], g["multi_tau_list"], 2.0, 1.0, "tanh",
                              loss="multi_class_linear_output")
    lp, gr = o.logp_and_grad(g["multi_theta"][None].astype(np.float64))
    np.testing.assert_allclose(lp, g["multi_logp"], rtol=2e-6)
    np.testing.assert_allclose(gr[0], g["multi_grad"], rtol=2e-5, atol=2e-6)


def test_deep_and_linear_nets_vs_reference_fixture(golden):
    """The shapes of the reference's notebooks (tests/golden/deepnet.npz, recorded from the unmodified reference by
    oracle/gen_golden.py: gen_deepnet): Net([1, 10, 10, 1]) with torch.relu in forward() and model_loss 'regression'
    (hamiltorch_split_HMC_BNN_example), Linear(4, 3) with 'multi_class_linear_output' (hamiltorch_Bayesian_NN_example),
    and a [3, 5, 4, 2] tanh net with 'binary_class_linear_output' on two outputs: log-probability and gradient of every
    split closure."""
    g = golden("deepnet")
    for name in ("deepreg", "softmaxlin", "bin2"):
        dims, act, loss = [int(d) for d in g[name + "_dims"]], str(g[name + "_act"]), str(g[name + "_loss"])
        M = int(g[name + "_M"]); X, Y = g[name + "_X"], g[name + "_Y"]; Nb = X.shape[0] // M
        for m in range(M):
            o = O.MLPRegressionTarget(dims, X[m * Nb:(m + 1) * Nb], Y[m * Nb:(m + 1) * Nb], g[name + "_tau_list"], float(g[name + "_tau_out"]),
                                      float(M), act, loss=loss)
            lp, gr = o.logp_and_grad(g[name + "_theta"][None].astype(np.float64))
            np.testing.assert_allclose(lp, g[name + "_logp"][m], rtol=3e-6)
            np.testing.assert_allclose(gr[0], g[name + "_grad"][m], rtol=3e-5, atol=3e-6)


# ---- the reference's published split-HMC model: Linear(1,100)-ReLU-Linear(100,100)-ReLU-Linear(100,1), D = 10401 --------------
def _nb_targets(g):
    name = "nbmlp"
    M, tau_out = int(g[f"{name}_cfg"][0]), float(g[f"{name}_cfg"][1])
    X, Y = g[f"{name}_X"], g[f"{name}_Y"]
    nb = X.shape[0] // M
    full = O.MLPRegressionTarget([1, 100, 100, 1], X, Y, g[f"{name}_tau_list"], tau_out, 1.0, "relu")
    splits = [O.MLPRegressionTarget([1, 100, 100, 1], X[m * nb:(m + 1) * nb], Y[m * nb:(m + 1) * nb], g[f"{name}_tau_list"],
                                    tau_out, M, "relu") for m in range(M)]
    return full, splits


def test_nbmlp_oracle_vs_reference_fixture(golden):
    """tests/golden/nbmlp.npz (oracle/gen_golden.py::gen_nbmlp from the unmodified reference): the notebook's model
    (notebooks/hamiltorch_split_HMC_BNN_example.ipynb cells 9-14, 23-25; D = 10401, 400 points, M = 4, tau_out = 110.44,
    eps = 5e-4) - full-data log-prob + gradient, the split closures, H, a 3-step SPLITTING path, a 4-step leapfrog path and the
    two end-to-end runs replayed with the reference's recorded draws."""
    g = golden("nbmlp")
    name = "nbmlp"
    M, tau_out, eps, Ls, Lf = g[f"{name}_cfg"]
    M, Ls, Lf = int(M), int(Ls), int(Lf)
    full, splits = _nb_targets(g)
    assert full.D == 10401
    theta = g[f"{name}_theta"][None].astype(np.float32)
    lp, gr = full.logp_and_grad(theta)
    np.testing.assert_allclose(lp, g[f"{name}_logp"], rtol=2e-5)
    gs = np.abs(g[f"{name}_grad"]).max()
    np.testing.assert_allclose(gr[0], g[f"{name}_grad"], rtol=2e-4, atol=2e-5 * gs)
    np.testing.assert_allclose([s.logp(theta)[0] for s in splits], g[f"{name}_split_logp"], rtol=2e-5)
    np.testing.assert_allclose(splits[1].grad(theta)[0], g[f"{name}_split1_grad"], rtol=2e-4, atol=2e-5 * gs)
    p0 = g[f"{name}_p0"][None].astype(np.float32)
    im = np.ones(theta.shape[1], np.float32)
    H0, _ = O.hmc_hamiltonian(theta, p0, [s.logp for s in splits], im)
    np.testing.assert_allclose(H0, g[f"{name}_H0"], rtol=2e-5)
    th, pm = O.split_leapfrog(theta, p0, [s.grad for s in splits], Ls, eps, im)
    np.testing.assert_allclose(th[0], g[f"{name}_lf_theta"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pm[0], g[f"{name}_lf_p"], rtol=1e-3, atol=2e-3)
    th, pm = O.hmc_leapfrog(theta, p0, full.grad, Lf, eps, im)
    np.testing.assert_allclose(th[0], g[f"{name}_full_lf_theta"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(pm[0], g[f"{name}_full_lf_p"], rtol=1e-3, atol=2e-3)
    draws = O.ReplayDraws(g[f"{name}_e2e_momenta"], g[f"{name}_e2e_uniforms"])
    ret, info = O.sample_hmc(None, theta, 2, 4, eps, -1, im, draws, grad_fns=[s.grad for s in splits], logp_fns=[s.logp for s in splits])
    np.testing.assert_allclose(np.concatenate(ret), g[f"{name}_e2e_samples"], rtol=1e-3, atol=1e-4)
    assert abs(info["acc_rate"][0] - float(g[f"{name}_e2e_acc"])) < 1e-9
    draws = O.ReplayDraws(g[f"{name}_full_momenta"], g[f"{name}_full_uniforms"])
    ret, info = O.sample_hmc(full, theta, 2, 5, eps, -1, im, draws)
    np.testing.assert_allclose(np.concatenate(ret), g[f"{name}_full_samples"], rtol=1e-3, atol=1e-4)
    assert abs(info["acc_rate"][0] - float(g[f"{name}_full_acc"])) < 1e-9


def test_torch_port_nbmlp_matches_reference_run(golden):
    """bench.py's cpu_baseline for the notebook model (split-HMC port over functional-model closures, per-half-kick autograd):
    the reference's own sample_split_model run at D = 10401 from the same torch seed."""
    import torch
    import torch_port as TP
    g = golden("nbmlp")
    name = "nbmlp"
    M, tau_out, eps = int(g[f"{name}_cfg"][0]), float(g[f"{name}_cfg"][1]), float(g[f"{name}_cfg"][2])
    net = TP.notebook_net()
    X, Y = torch.tensor(g[f"{name}_X"]), torch.tensor(g[f"{name}_Y"])
    tau_list = torch.tensor(g[f"{name}_tau_list"])
    nb = X.shape[0] // M
    fl = [TP.port_mlp_closure(net, X[m * nb:(m + 1) * nb], Y[m * nb:(m + 1) * nb], tau_list, tau_out, M) for m in range(M)]
    torch.manual_seed(33)
    for _ in torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=nb, shuffle=False):
        pass                                       # the loader's base-seed draw, as the reference's run consumed it
    ret, acc = TP.port_sample_split(fl, torch.tensor(g[f"{name}_theta"]), 2, 4, eps, -1, torch.ones(10401))
    got = np.stack([t.numpy() for t in ret])
    np.testing.assert_allclose(got, g[f"{name}_e2e_samples"], rtol=1e-5, atol=1e-6)
    assert acc == float(g[f"{name}_e2e_acc"])


def test_split_momentum_draw_has_the_reference_covariance():
    """oracle rm_gibbs_split (the product's default draw on its fused Gaussian route with jitter):  p = A [z1; z2] with
    A = [chol(P) | diag(sqrt(jitter u))], so cov(p | u) = A A^T = P + diag(jitter u) = G, the covariance of the reference's
    chol(G) z (S:113-116, S:183-184); with jitter -> 0 it is the reference's draw itself."""
    rng = np.random.default_rng(4)
    D = 12
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    P = (Q * np.linspace(0.5, 2.0, D)) @ Q.T; P = 0.5 * (P + P.T)
    tgt = O.GaussianTarget(np.zeros(D), P, 0.0)
    th = rng.standard_normal((1, D)); u = rng.uniform(size=(1, D)); jit = 1e-2
    G, _, _ = O.softabs_metric(tgt.neg_hessian(th), 1e6, jit, u, "softabs")
    A = np.stack([O.rm_gibbs_split(th, np.eye(2 * D)[k:k + 1, :D], np.eye(2 * D)[k:k + 1, D:], tgt, jit, u)[0] for k in range(2 * D)], axis=1)
    np.testing.assert_allclose(A @ A.T, G[0], rtol=1e-10, atol=1e-12)
    z = rng.standard_normal((5, D)); th5 = np.repeat(th, 5, 0)
    a = O.rm_gibbs_split(th5, z, rng.standard_normal((5, D)), tgt, 1e-300, np.repeat(u, 5, 0))
    b = O.rm_gibbs(th5, z, tgt, 1e6, None, None, "softabs")
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    # sampled: the empirical covariance of 40000 split draws matches G within 4 standard errors element-wise
    n = 40000
    p = O.rm_gibbs_split(np.repeat(th, n, 0), rng.standard_normal((n, D)), rng.standard_normal((n, D)), tgt, jit, np.repeat(u, n, 0))
    emp = p.T @ p / n
    se = np.sqrt((np.outer(np.diag(G[0]), np.diag(G[0])) + G[0] ** 2) / n)
    assert (np.abs(emp - G[0]) < 4.5 * se).all()


def test_square_root_momentum_draw_has_the_reference_covariance():
    """oracle rm_gibbs_sqrt (the product's default draw on the eigendecomposition route of a Gaussian target's soft-abs metric, round 6):
    p = S z with the SYMMETRIC square root S = Q diag(sqrt lam~) Q^T of G - S S^T = S^2 = G, the covariance of the reference's chol(G) z
    (S:113-122, S:183-184); unlike an eigen-factor Q diag(sqrt lam~) the map does not depend on the eigenvectors' signs or order; an
    indefinite curvature with a finite alpha (the soft-abs map is what makes G positive definite) and jitter."""
    rng = np.random.default_rng(9)
    D = 11
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    lam = np.linspace(0.4, 2.0, D) * np.where(np.arange(D) % 3 == 0, -1.0, 1.0)
    P = (Q * lam) @ Q.T; P = 0.5 * (P + P.T)
    tgt = O.GaussianTarget(np.zeros(D), P, 0.0)
    th = rng.standard_normal((1, D)); u = rng.uniform(size=(1, D)); jit, alpha = 1e-2, 1.3
    G, lam_t, Qg = O.softabs_metric(tgt.neg_hessian(th), alpha, jit, u, "softabs")
    S = np.stack([O.rm_gibbs_sqrt(th, np.eye(D)[k:k + 1], tgt, alpha, jit, u)[0] for k in range(D)], axis=1)
    np.testing.assert_allclose(S, S.T, rtol=0, atol=1e-12)                       # symmetric
    np.testing.assert_allclose(S @ S, G[0], rtol=1e-10, atol=1e-12)              # a square root of G
    assert np.linalg.eigvalsh(S).min() > 0                                       # THE positive definite one
    # the eigenvectors' signs and order do not enter: the same map from a flipped, permuted basis
    perm = rng.permutation(D); sg = rng.choice([-1.0, 1.0], D)
    Q2 = Qg[0][:, perm] * sg
    S2 = (Q2 * np.sqrt(lam_t[0][perm])) @ Q2.T
    np.testing.assert_allclose(S2, S, rtol=1e-10, atol=1e-12)
    # and it is not the reference's map (a different square root: chol(G) is triangular), only its law
    Lc = np.linalg.cholesky(G[0])
    assert np.abs(S - Lc).max() > 1e-2
    np.testing.assert_allclose(Lc @ Lc.T, S @ S.T, rtol=1e-10, atol=1e-12)
    n = 40000
    p = O.rm_gibbs_sqrt(np.repeat(th, n, 0), rng.standard_normal((n, D)), tgt, alpha, jit, np.repeat(u, n, 0))
    emp = p.T @ p / n
    se = np.sqrt((np.outer(np.diag(G[0]), np.diag(G[0])) + G[0] ** 2) / n)
    assert (np.abs(emp - G[0]) < 4.5 * se).all()


def test_torch_port_funnel_hmc_matches_reference_run(golden):
    """bench.py's funnel-hmc / funnel-rmhmc cpu_baseline legs (oracle/cpu_baseline.py, `kind: "port"` on a box without the
    reference): the port on the notebook's verbatim funnel_ll closure reproduces the unmodified reference's runs
    (tests/golden/funnel_hmc.npz: cell 24's HMC settings bit for bit from the same torch seed - hamiltorch.set_random_seed
    seeds torch like torch.manual_seed; cell 30's explicit-RMHMC settings to rounding of eigh's backward)."""
    import random
    import torch
    import torch_port as TP
    from cpu_baseline import funnel_ll
    g = golden("funnel_hmc")
    init = torch.ones(11); init[0] = 0.0

    def seed(v):                                             # hamiltorch.util.set_random_seed (U:11-20) without the package
        random.seed(v); np.random.seed(v); torch.manual_seed(v)
    seed(123)
    ret, acc = TP.port_sample(funnel_ll, init, 30, 25, 0.2, 0, None)
    got = np.stack([t.numpy() for t in ret])
    assert got.shape == g["hmc_samples"].shape and np.array_equal(got, g["hmc_samples"])
    assert acc == float(g["hmc_acc"])
    seed(123)
    ret, acc = TP.port_sample_rmhmc(funnel_ll, init, 2, 25, 0.14, 10.0, 1e6, burn=-1, jitter=1e-3)
    got = np.stack([t.numpy() for t in ret])
    assert got.shape == g["rm_samples"].shape
    np.testing.assert_allclose(got, g["rm_samples"], rtol=0, atol=5e-4)
    assert acc == float(g["rm_acc"])


def test_cpu_baseline_worker_runs_the_reference_when_it_is_there_and_the_port_otherwise():
    """oracle/cpu_baseline.py (one process of bench.py's cpu_baseline leg): `kind: "reference"` = the unmodified package from
    HAMILTORCH_REFERENCE / /root/reference (only in the build container), `kind: "port"` = oracle/torch_port.py; both give one
    JSON line with n / L / dt / acc, and agree on the cost (same work per trajectory: within 2x of each other)."""
    import json
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "cpu_baseline.py")
    rates = {}
    for mode in ("port", ""):
        env = dict(os.environ, HTA_CPU_BASELINE=mode, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", PYTHONDONTWRITEBYTECODE="1")
        out = subprocess.run([sys.executable, script, "funnel-hmc", "5", "1.0"], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        r = json.loads(out.stdout.strip().splitlines()[-1])
        assert r["L"] == 25 and r["n"] >= 1 and r["dt"] > 0 and 0.0 <= r["acc"] <= 1.0 and len(r["samples"]) == r["n"]
        rates[r["kind"]] = r["n"] * r["L"] / r["dt"]
        if mode == "port":
            assert r["kind"] == "port"
    if os.path.isdir("/root/reference/hamiltorch"):
        assert set(rates) == {"port", "reference"} and 0.5 < rates["port"] / rates["reference"] < 2.0, rates

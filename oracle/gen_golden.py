"""Generate the golden fixtures in tests/golden/ by running the UNMODIFIED reference.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs
/root/reference); the GPU box consumes the committed .npz files.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

The reference needs ``termcolor`` (util.py:4), which is not installed: a 2-line
shim lives in oracle/shim/.  Importing hamiltorch reseeds the global RNGs from the
wall clock (util.py:23), so every case seeds *after* import.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "shim"))
sys.path.insert(0, os.environ.get("HAMILTORCH_REFERENCE", "/root/reference"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import hamiltorch  # noqa: E402
from hamiltorch import samplers as S  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)


def npy(t):
    return t.detach().cpu().numpy()


class Recorder:
    """Records the reference's own random draws: gibbs() outputs (S:969) and the
    MH uniforms torch.rand(1) (S:1004)."""

    def __enter__(self):
        self.momenta, self.uniforms, self.jitters, self.perms = [], [], [], []
        self._gibbs, self._rand, self._randperm = S.gibbs, torch.rand, torch.randperm

        def gibbs(*a, **k):
            m = self._gibbs(*a, **k)
            self.momenta.append(npy(m).copy())
            return m

        def rand(*a, **k):
            u = self._rand(*a, **k)
            if tuple(u.shape) == (1,):
                self.uniforms.append(npy(u).copy())
            elif u.dim() == 1:                                  # the jitter draw of fisher() (S:115): torch.rand(D), D > 1
                self.jitters.append(npy(u).copy())
            return u

        def randperm(*a, **k):                                   # SPLITTING_RAND's subset order (S:549)
            r = self._randperm(*a, **k)
            self.perms.append(npy(r).copy())
            return r

        S.gibbs = gibbs
        torch.rand = rand
        torch.randperm = randperm
        return self

    def __exit__(self, *exc):
        S.gibbs = self._gibbs
        torch.rand = self._rand
        torch.randperm = self._randperm


SIGMA3 = [[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]]


def mvn_logp(mean, cov):
    def f(w):
        return torch.distributions.MultivariateNormal(mean, cov).log_prob(w).sum()
    return f


def quad_logp(P):
    def f(w):
        return -0.5 * torch.dot(w, torch.mv(P, w))
    return f


def gen_hmc_kat():
    out = {}
    # KAT1: tests/test_util.py:97-110 set-up
    lp = mvn_logp(torch.zeros(2), torch.diag(torch.tensor([0.1, 0.1])))
    for steps in (1, 3, 100):
        p, m = S.leapfrog(torch.tensor([1.0, 1.0]), torch.tensor([1.0, 1.0]), lp, steps=steps,
                          step_size=0.1, inv_mass=torch.tensor([1.0, 1.0]),
                          sampler=hamiltorch.Sampler.HMC, integrator=hamiltorch.Integrator.EXPLICIT)
        out[f"kat1_theta_{steps}"] = npy(p[-1]); out[f"kat1_p_{steps}"] = npy(m[-1])
    # reversibility (bit-exact in the reference)
    p, m = S.leapfrog(torch.tensor([1.0, 1.0]), torch.tensor([1.0, 1.0]), lp, steps=100, step_size=0.1,
                      inv_mass=torch.tensor([1.0, 1.0]), sampler=hamiltorch.Sampler.HMC,
                      integrator=hamiltorch.Integrator.EXPLICIT)
    p2, m2 = S.leapfrog(p[-1], -m[-1].clone(), lp, steps=100, step_size=0.1,
                        inv_mass=torch.tensor([1.0, 1.0]), sampler=hamiltorch.Sampler.HMC,
                        integrator=hamiltorch.Integrator.EXPLICIT)
    out["kat1_reversed_theta"] = npy(p2[-1])
    # KAT2: 3-D correlated Gaussian, fp32 and fp64, three mass kinds
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        cov = torch.tensor(SIGMA3, dtype=dt)
        lp = mvn_logp(torch.zeros(3, dtype=dt), cov)
        th = torch.tensor([0.3, -0.2, 0.5], dtype=dt)
        pm = torch.tensor([0.1, 0.7, -0.4], dtype=dt)
        out[f"kat2_logp_{tag}"] = npy(lp(th))
        full = torch.tensor([[1.0, 0.2, 0.0], [0.2, 0.5, 0.1], [0.0, 0.1, 2.0]], dtype=dt)
        masses = {"none": None, "diag": torch.tensor([1.0, 0.5, 2.0], dtype=dt), "full": full}
        for mk, im in masses.items():
            H = S.hamiltonian(th, pm, lp, inv_mass=im, sampler=hamiltorch.Sampler.HMC)
            out[f"kat2_H_{mk}_{tag}"] = npy(H).reshape(-1)
            p, m = S.leapfrog(th, pm, lp, steps=5, step_size=0.3, inv_mass=im,
                              sampler=hamiltorch.Sampler.HMC, integrator=hamiltorch.Integrator.EXPLICIT)
            out[f"kat2_theta_{mk}_{tag}"] = np.stack([npy(x) for x in p])
            out[f"kat2_p_{mk}_{tag}"] = np.stack([npy(x) for x in m])
        out[f"kat2_inv_mass_full_{tag}"] = npy(full)
    out["sigma3"] = np.array(SIGMA3)
    np.savez(os.path.join(OUT, "hmc_kat.npz"), **out)


def gen_gibbs():
    out = {}
    D = 4
    mass_d = torch.tensor([0.5, 1.0, 2.0, 4.0])
    A = torch.tensor([[2.0, 0.3, 0.0, 0.1], [0.3, 1.0, 0.2, 0.0], [0.0, 0.2, 1.5, 0.4], [0.1, 0.0, 0.4, 3.0]])
    th = torch.zeros(D)
    for name, mass in (("none", None), ("diag", mass_d), ("full", A)):
        torch.manual_seed(11)
        z = torch.normal(torch.zeros(D), torch.ones(D))  # what Normal/MVN.sample consume
        torch.manual_seed(11)
        p = S.gibbs(th, sampler=hamiltorch.Sampler.HMC, mass=mass)
        out[f"z_{name}"] = npy(z); out[f"p_{name}"] = npy(p)
        if mass is not None:
            out[f"mass_{name}"] = npy(mass)
    # RMHMC gibbs: p ~ N(0, G(theta))
    P = torch.tensor(np.linalg.inv(np.array(SIGMA3)), dtype=torch.float32)
    lp = quad_logp(P)
    th3 = torch.tensor([0.3, -0.2, 0.5])
    torch.manual_seed(5)
    z = torch.normal(torch.zeros(3), torch.ones(3))
    torch.manual_seed(5)
    p = S.gibbs(th3, sampler=hamiltorch.Sampler.RMHMC, log_prob_func=lp, jitter=None,
                softabs_const=1e6, metric=hamiltorch.Metric.SOFTABS)
    out["rm_z"] = npy(z); out["rm_p"] = npy(p); out["rm_P"] = npy(P)
    np.savez(os.path.join(OUT, "gibbs.npz"), **out)


def gen_sample_hmc():
    """cfg1: end-to-end sample() with the draws recorded, several burn/mass settings."""
    out = {}
    cov = torch.tensor(SIGMA3)
    lp = mvn_logp(torch.zeros(3), cov)
    full = torch.tensor([[1.0, 0.2, 0.0], [0.2, 0.5, 0.1], [0.0, 0.1, 2.0]])
    cases = {
        "cfg1": dict(N=400, L=5, eps=0.3, burn=0, inv_mass=None, init=torch.zeros(3), seed=123),
        "burn10": dict(N=60, L=5, eps=0.9, burn=10, inv_mass=None, init=torch.tensor([3.0, 3.0, 3.0]), seed=7),
        "burnm1": dict(N=40, L=4, eps=0.9, burn=-1, inv_mass=None, init=torch.tensor([1.0, -1.0, 0.5]), seed=8),
        "diag": dict(N=50, L=6, eps=0.4, burn=0, inv_mass=torch.tensor([1.0, 0.5, 2.0]), init=torch.zeros(3), seed=9),
        "full": dict(N=50, L=6, eps=0.4, burn=3, inv_mass=full, init=torch.zeros(3), seed=10),
    }
    for name, c in cases.items():
        hamiltorch.set_random_seed(c["seed"])
        with Recorder() as rec:
            ret, acc = hamiltorch.sample(lp, c["init"], num_samples=c["N"], num_steps_per_sample=c["L"],
                                         step_size=c["eps"], burn=c["burn"], inv_mass=c["inv_mass"],
                                         debug=2, verbose=False)
        out[f"{name}_samples"] = np.stack([npy(t) for t in ret])
        out[f"{name}_momenta"] = np.stack(rec.momenta)
        out[f"{name}_uniforms"] = np.concatenate(rec.uniforms)
        out[f"{name}_acc"] = np.array(acc)
        out[f"{name}_cfg"] = np.array([c["N"], c["L"], c["eps"], c["burn"]], dtype=np.float64)
        out[f"{name}_init"] = npy(c["init"])
        if c["inv_mass"] is not None:
            out[f"{name}_inv_mass"] = npy(c["inv_mass"])
    out["sigma3"] = np.array(SIGMA3)
    np.savez(os.path.join(OUT, "sample_hmc.npz"), **out)


def gen_blockmass():
    """Block-diagonal inv_mass given as a list of blocks (S:188-197, S:287-292, S:803-809, S:944-947)."""
    out = {}
    b0 = torch.tensor([[1.5, 0.4], [0.4, 0.8]])
    b1 = torch.tensor([[2.0, 0.3, 0.0], [0.3, 1.0, -0.2], [0.0, -0.2, 0.6]])
    out["b0"], out["b1"] = npy(b0), npy(b1)
    th = torch.zeros(5)
    torch.manual_seed(21)
    z = torch.cat([torch.normal(torch.zeros(2), torch.ones(2)), torch.normal(torch.zeros(3), torch.ones(3))])
    torch.manual_seed(21)
    p = S.gibbs(th, sampler=hamiltorch.Sampler.HMC, mass=[b0, b1])
    out["gibbs_z"], out["gibbs_p"] = npy(z), npy(p)
    A = np.array([[1.0, 0.3, 0.0, 0.2, 0.0], [0.3, 2.0, 0.4, 0.0, 0.1], [0.0, 0.4, 1.5, 0.3, 0.0],
                  [0.2, 0.0, 0.3, 0.8, 0.2], [0.0, 0.1, 0.0, 0.2, 1.2]], dtype=np.float32)
    P = torch.tensor(A)
    lp = quad_logp(P)
    out["P"] = A
    pm = torch.tensor([0.5, -1.0, 0.25, 2.0, -0.3]); th1 = torch.tensor([0.1, 0.2, -0.3, 0.4, -0.5])
    out["kat_theta"], out["kat_p"] = npy(th1), npy(pm)
    out["kat_H"] = npy(S.hamiltonian(th1, pm, lp, inv_mass=[b0, b1], sampler=hamiltorch.Sampler.HMC))
    pl, ml = S.leapfrog(th1, pm, lp, steps=4, step_size=0.2, inv_mass=[b0, b1], sampler=hamiltorch.Sampler.HMC,
                        integrator=hamiltorch.Integrator.IMPLICIT)
    out["kat_theta_L"], out["kat_p_L"] = npy(pl[-1]), npy(ml[-1])
    init = torch.tensor([0.5, -0.5, 0.2, 0.0, 1.0])
    hamiltorch.set_random_seed(31)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample(lp, init, num_samples=60, num_steps_per_sample=5, step_size=0.8, burn=4,
                                     inv_mass=[b0, b1], debug=2, verbose=False)
    out["samples"] = np.stack([npy(t) for t in ret]); out["momenta"] = np.stack(rec.momenta)
    out["uniforms"] = np.concatenate(rec.uniforms); out["acc"] = np.array(acc); out["init"] = npy(init)
    out["cfg"] = np.array([60, 5, 0.8, 4], dtype=np.float64)
    np.savez(os.path.join(OUT, "blockmass.npz"), **out)


def gen_nuts():
    """Dual-averaging step size ("HMC_NUTS", S:629-674 + S:1030-1035): the scalar recurrence and an end-to-end run."""
    out = {}
    rhos = [0.0, -0.5, -3.0, float("nan"), -0.01, -1.2, 0.0, -0.3]
    ss, eb, Ht, rows = 0.3, 1.0, 0.0, []
    for t, r in enumerate(rhos):
        ss, eb, Ht = S.adaptation(r, t, 0.3, Ht, eb, desired_accept_rate=0.75)
        rows.append([ss, eb, Ht])
    out["adapt_rhos"] = np.array(rhos); out["adapt_out"] = np.array(rows)
    lp = mvn_logp(torch.zeros(3), torch.tensor(SIGMA3))
    hamiltorch.set_random_seed(17)
    with Recorder() as rec:
        ret, ss = hamiltorch.sample(lp, torch.tensor([0.5, -0.5, 0.25]), num_samples=45, num_steps_per_sample=5,
                                    step_size=0.05, burn=20, sampler=hamiltorch.Sampler.HMC_NUTS,
                                    desired_accept_rate=0.7, debug=2, verbose=False)
    out["e2e_samples"] = np.stack([npy(t) for t in ret]); out["e2e_step_size"] = np.array(ss)
    out["e2e_momenta"] = np.stack(rec.momenta); out["e2e_uniforms"] = np.concatenate(rec.uniforms)
    np.savez(os.path.join(OUT, "nuts.npz"), **out)


def rand_spd(D, seed, lo=0.5, hi=2.0):
    g = torch.Generator().manual_seed(seed)
    Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
    lam = torch.linspace(lo, hi, D, dtype=torch.float64)
    P = (Q * lam) @ Q.T
    return 0.5 * (P + P.T)


def gen_rmhmc():
    out = {}
    # KAT3/KAT4 at D=3 (both metrics), D=10 SPD, D=6 indefinite with finite softabs_const
    P3 = torch.tensor(np.linalg.inv(np.array(SIGMA3)))
    cases = {
        "d3": dict(P=P3, alpha=1e6, omega=10.0, eps=0.1, steps=2),
        "d10": dict(P=rand_spd(10, 0), alpha=1e6, omega=10.0, eps=0.1, steps=3),
        "d6indef": dict(P=rand_spd(6, 1, -1.0, 2.0), alpha=1.5, omega=5.0, eps=0.05, steps=2),
    }
    for name, c in cases.items():
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            P = c["P"].to(dt)
            D = P.shape[0]
            lp = quad_logp(P)
            g = torch.Generator().manual_seed(3)
            th = (0.5 * torch.randn(D, generator=g, dtype=torch.float64)).to(dt)
            pm = torch.randn(D, generator=g, dtype=torch.float64).to(dt)
            out[f"{name}_P_{tag}"] = npy(P); out[f"{name}_theta0_{tag}"] = npy(th); out[f"{name}_p0_{tag}"] = npy(pm)
            out[f"{name}_cfg"] = np.array([c["alpha"], c["omega"], c["eps"], c["steps"]])
            for metric, mtag in ((hamiltorch.Metric.SOFTABS, "softabs"), (hamiltorch.Metric.HESSIAN, "hessian")):
                if mtag == "hessian" and name == "d6indef":
                    continue  # Cholesky of an indefinite G raises in the reference
                G, lam = S.fisher(th, lp, jitter=None, softabs_const=c["alpha"], metric=metric)
                out[f"{name}_G_{mtag}_{tag}"] = npy(G)
                if lam is not None:
                    out[f"{name}_lam_{mtag}_{tag}"] = npy(lam)
                out[f"{name}_Ginvp_{mtag}_{tag}"] = npy(S.cholesky_inverse(G.detach(), pm)).reshape(-1)
                H = S.rm_hamiltonian(th, pm, lp, None, 1.0, softabs_const=c["alpha"], metric=metric)
                out[f"{name}_H_{mtag}_{tag}"] = npy(H).reshape(-1)
                lpar, lmom = S.leapfrog(th, pm, lp, steps=c["steps"], step_size=c["eps"], jitter=None,
                                        explicit_binding_const=c["omega"], softabs_const=c["alpha"],
                                        sampler=hamiltorch.Sampler.RMHMC,
                                        integrator=hamiltorch.Integrator.EXPLICIT, metric=metric)
                out[f"{name}_lf_theta_{mtag}_{tag}"] = npy(lpar[0][-1]); out[f"{name}_lf_p_{mtag}_{tag}"] = npy(lmom[0][-1])
                out[f"{name}_lf_thetac_{mtag}_{tag}"] = npy(lpar[1]); out[f"{name}_lf_pc_{mtag}_{tag}"] = npy(lmom[1])
    # end-to-end explicit RMHMC sample(), jitter=None, draws recorded
    P = P3.to(torch.float32)
    lp = quad_logp(P)
    hamiltorch.set_random_seed(21)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample(lp, torch.tensor([0.3, -0.2, 0.5]), num_samples=12, num_steps_per_sample=3,
                                     step_size=0.25, burn=2, jitter=None, softabs_const=1e6,
                                     explicit_binding_const=10.0, sampler=hamiltorch.Sampler.RMHMC,
                                     integrator=hamiltorch.Integrator.EXPLICIT,
                                     metric=hamiltorch.Metric.SOFTABS, debug=2, verbose=False)
    out["e2e_samples"] = np.stack([npy(t) for t in ret])
    out["e2e_momenta"] = np.stack(rec.momenta)
    out["e2e_uniforms"] = np.concatenate(rec.uniforms)
    out["e2e_acc"] = np.array(acc)
    out["e2e_P"] = npy(P)
    np.savez(os.path.join(OUT, "rmhmc.npz"), **out)


def make_mlp(dims, act, seed):
    torch.manual_seed(seed)
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append({"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}[act]())
    return nn.Sequential(*layers)


def gen_mlp():
    out = {}
    cases = {
        "relu2": dict(dims=[3, 5, 1], act="relu", N=12, M=3, tau_out=10.0, eps=2e-3, L=2),
        "tanh3": dict(dims=[2, 4, 3, 1], act="tanh", N=8, M=2, tau_out=4.0, eps=5e-3, L=3),
    }
    for name, c in cases.items():
        net = make_mlp(c["dims"], c["act"], 0)
        g = torch.Generator().manual_seed(1)
        X = torch.randn(c["N"], c["dims"][0], generator=g)
        Y = torch.sin(X.sum(1, keepdim=True)) + 0.1 * torch.randn(c["N"], 1, generator=g)
        theta = hamiltorch.util.flatten(net).clone().detach()
        D = theta.numel()
        ntens = len(list(net.parameters()))
        tau_list = torch.tensor([1.0 + 0.5 * k for k in range(ntens)])
        pfl = [w.nelement() for w in net.parameters()]
        psl = [w.shape for w in net.parameters()]
        out[f"{name}_dims"] = np.array(c["dims"]); out[f"{name}_X"] = npy(X); out[f"{name}_Y"] = npy(Y)
        out[f"{name}_theta"] = npy(theta); out[f"{name}_tau_list"] = npy(tau_list)
        out[f"{name}_cfg"] = np.array([c["M"], c["tau_out"], c["eps"], c["L"]])
        # full-data log prob + gradient (define_model_log_prob, S:1093-1201)
        f = S.define_model_log_prob(net, "regression", X, Y, pfl, psl, tau_list, c["tau_out"])
        th = theta.clone().requires_grad_()
        v = f(th)
        out[f"{name}_logp"] = npy(v).reshape(-1)
        out[f"{name}_grad"] = npy(torch.autograd.grad(v.sum(), th)[0])
        # split closures (S:1203-1258) and one SPLITTING leapfrog (S:494-547)
        loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y),
                                             batch_size=c["N"] // c["M"], shuffle=False)
        fl = S.define_split_model_log_prob(net, "regression", loader, c["M"], pfl, psl, tau_list,
                                           c["tau_out"], verbose=False)
        out[f"{name}_split_logp"] = np.array([float(fm(theta).sum()) for fm in fl])
        gm = torch.Generator().manual_seed(2)
        p0 = torch.randn(D, generator=gm)
        inv_mass = torch.ones(D)
        H0 = S.hamiltonian(theta, p0, fl, inv_mass=inv_mass, sampler=hamiltorch.Sampler.HMC)
        out[f"{name}_p0"] = npy(p0); out[f"{name}_H0"] = npy(H0).reshape(-1)
        lp_, lm_ = S.leapfrog(theta.clone(), p0.clone(), fl, steps=c["L"], step_size=c["eps"],
                              inv_mass=inv_mass, sampler=hamiltorch.Sampler.HMC,
                              integrator=hamiltorch.Integrator.SPLITTING)
        out[f"{name}_lf_theta"] = npy(lp_[-1]); out[f"{name}_lf_p"] = npy(lm_[-1])
        # end-to-end sample_split_model with recorded draws
        hamiltorch.set_random_seed(33)
        with Recorder() as rec:
            ret, acc = hamiltorch.sample_split_model(net, loader, theta.clone(), c["M"], model_loss="regression",
                                                     num_samples=10, num_steps_per_sample=c["L"],
                                                     step_size=c["eps"], burn=0, inv_mass=inv_mass,
                                                     tau_out=c["tau_out"], tau_list=tau_list, debug=2,
                                                     verbose=False)
        out[f"{name}_e2e_samples"] = np.stack([npy(t) for t in ret])
        out[f"{name}_e2e_momenta"] = np.stack(rec.momenta)
        out[f"{name}_e2e_uniforms"] = np.concatenate(rec.uniforms)
        out[f"{name}_e2e_acc"] = np.array(acc)
        # sample_model (full data, plain HMC) end-to-end
        hamiltorch.set_random_seed(34)
        with Recorder() as rec:
            ret, acc = hamiltorch.sample_model(net, X, Y, theta.clone(), model_loss="regression",
                                               num_samples=8, num_steps_per_sample=c["L"], step_size=c["eps"],
                                               tau_out=c["tau_out"], tau_list=tau_list, debug=2, verbose=False)
        out[f"{name}_full_samples"] = np.stack([npy(t) for t in ret])
        out[f"{name}_full_momenta"] = np.stack(rec.momenta)
        out[f"{name}_full_uniforms"] = np.concatenate(rec.uniforms)
    np.savez(os.path.join(OUT, "mlp.npz"), **out)


def gen_splitkinds():
    """Integrator.SPLITTING_RAND / SPLITTING_KMID (S:547-596) on the split closures of a small MLP: one leapfrog call
    each (RAND: its torch.randperm recorded) and an end-to-end sample_split_model run with all draws recorded."""
    out = {}
    c = dict(dims=[3, 5, 1], act="relu", N=12, M=3, tau_out=10.0, eps=2e-3, L=2)
    net = make_mlp(c["dims"], c["act"], 0)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(c["N"], c["dims"][0], generator=g)
    Y = torch.sin(X.sum(1, keepdim=True)) + 0.1 * torch.randn(c["N"], 1, generator=g)
    theta = hamiltorch.util.flatten(net).clone().detach()
    D = theta.numel()
    tau_list = torch.tensor([1.0 + 0.5 * k for k in range(4)])
    pfl = [w.nelement() for w in net.parameters()]
    psl = [w.shape for w in net.parameters()]
    out["dims"] = np.array(c["dims"]); out["X"] = npy(X); out["Y"] = npy(Y); out["theta"] = npy(theta)
    out["tau_list"] = npy(tau_list); out["cfg"] = np.array([c["M"], c["tau_out"], c["eps"], c["L"]])
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=c["N"] // c["M"], shuffle=False)
    fl = S.define_split_model_log_prob(net, "regression", loader, c["M"], pfl, psl, tau_list, c["tau_out"], verbose=False)
    gm = torch.Generator().manual_seed(2)
    p0 = torch.randn(D, generator=gm)
    out["p0"] = npy(p0)
    inv_mass = torch.ones(D)
    for kind, tag in ((hamiltorch.Integrator.SPLITTING_RAND, "rand"), (hamiltorch.Integrator.SPLITTING_KMID, "kmid")):
        hamiltorch.set_random_seed(41)
        with Recorder() as rec:
            lp_, lm_ = S.leapfrog(theta.clone(), p0.clone(), fl, steps=3, step_size=c["eps"], inv_mass=inv_mass,
                                  sampler=hamiltorch.Sampler.HMC, integrator=kind)
        out[f"{tag}_lf_theta"] = np.stack([npy(t) for t in lp_]); out[f"{tag}_lf_p"] = np.stack([npy(t) for t in lm_])
        if rec.perms:
            out[f"{tag}_lf_perm"] = np.stack(rec.perms)
        hamiltorch.set_random_seed(42)
        with Recorder() as rec:
            ret, acc = hamiltorch.sample_split_model(net, loader, theta.clone(), c["M"], model_loss="regression",
                                                     num_samples=8, num_steps_per_sample=c["L"], step_size=c["eps"],
                                                     burn=0, inv_mass=inv_mass, tau_out=c["tau_out"], tau_list=tau_list,
                                                     integrator=kind, debug=2, verbose=False)
        out[f"{tag}_e2e_samples"] = np.stack([npy(t) for t in ret])
        out[f"{tag}_e2e_momenta"] = np.stack(rec.momenta)
        out[f"{tag}_e2e_uniforms"] = np.concatenate(rec.uniforms)
        out[f"{tag}_e2e_acc"] = np.array(acc)
        if rec.perms:
            out[f"{tag}_e2e_perms"] = np.stack(rec.perms)
    np.savez(os.path.join(OUT, "splitkinds.npz"), **out)


SCALES = [0.5, 1.0, 1.7, 2.4, 3.3]


def funnel_ll(D, scales=None):
    """The notebook's funnel (hamiltorch_log_prob_examples.ipynb, cell funnel_ll) for any D; `scales` s_i give
    x_i ~ N(0, s_i exp(-v)) (distinct Hessian eigenvalues, so that eigh can be differentiated without jitter)."""
    def f(w):
        sc = torch.ones(D - 1, dtype=w.dtype) if scales is None else torch.tensor(scales[:D - 1], dtype=w.dtype)
        v_dist = torch.distributions.Normal(0, 3)
        ll = v_dist.log_prob(w[0])
        x_dist = torch.distributions.Normal(0, (sc * torch.exp(-w[0])) ** 0.5)
        ll = ll + x_dist.log_prob(w[1:]).sum()
        return ll
    return f


def gen_funnel():
    """Explicit RMHMC on a NON-constant-curvature target: the reference differentiates rm_hamiltonian through
    hessian + eigh (S:398).  (a) scaled funnel, jitter=None: nothing random; (b) the notebook's funnel, whose
    Hessian has a repeated eigenvalue, with jitter - the torch.rand(D) draws of fisher() are recorded in call order."""
    out = {}
    out["scales"] = np.array(SCALES)
    for D, alpha, tag in ((4, 1e6, "a1e6"), (4, 1.3, "a1p3"), (6, 0.7, "d6")):
        lp = funnel_ll(D, SCALES)
        for dt, dtag in ((torch.float64, "f64"), (torch.float32, "f32")):
            g = torch.Generator().manual_seed(5)
            th = (0.6 * torch.randn(D, generator=g, dtype=torch.float64)).to(dt)
            pm = torch.randn(D, generator=g, dtype=torch.float64).to(dt)
            key = f"{tag}_{dtag}"
            out[f"{key}_theta0"] = npy(th); out[f"{key}_p0"] = npy(pm)
            out[f"{tag}_cfg"] = np.array([D, alpha, 5.0, 0.05, 3])
            G, lam = S.fisher(th, lp, jitter=None, softabs_const=alpha, metric=hamiltorch.Metric.SOFTABS)
            out[f"{key}_G"] = npy(G); out[f"{key}_lam"] = npy(lam)
            H = S.rm_hamiltonian(th, pm, lp, None, 1.0, softabs_const=alpha, metric=hamiltorch.Metric.SOFTABS)
            out[f"{key}_H"] = npy(H).reshape(-1)
            lpar, lmom = S.leapfrog(th, pm, lp, steps=3, step_size=0.05, jitter=None, explicit_binding_const=5.0,
                                    softabs_const=alpha, sampler=hamiltorch.Sampler.RMHMC,
                                    integrator=hamiltorch.Integrator.EXPLICIT, metric=hamiltorch.Metric.SOFTABS)
            out[f"{key}_lf_theta"] = np.stack([npy(t) for t in lpar[0]]); out[f"{key}_lf_p"] = np.stack([npy(t) for t in lmom[0]])
            out[f"{key}_lf_thetac"] = npy(lpar[1]); out[f"{key}_lf_pc"] = npy(lmom[1])
    # (b) notebook funnel (s_i = 1) with jitter, one leapfrog call: 8 jitter draws per step, in call order
    D = 5
    lp = funnel_ll(D)
    g = torch.Generator().manual_seed(6)
    th = 0.5 * torch.randn(D, generator=g, dtype=torch.float64); pm = torch.randn(D, generator=g, dtype=torch.float64)
    hamiltorch.set_random_seed(9)
    with Recorder() as rec:
        lpar, lmom = S.leapfrog(th, pm, lp, steps=2, step_size=0.05, jitter=0.01, explicit_binding_const=5.0,
                                softabs_const=1e6, sampler=hamiltorch.Sampler.RMHMC,
                                integrator=hamiltorch.Integrator.EXPLICIT, metric=hamiltorch.Metric.SOFTABS)
    out["jit_theta0"] = npy(th); out["jit_p0"] = npy(pm); out["jit_cfg"] = np.array([D, 1e6, 5.0, 0.05, 2, 0.01])
    out["jit_draws"] = np.stack(rec.jitters)
    out["jit_lf_theta"] = np.stack([npy(t) for t in lpar[0]]); out["jit_lf_p"] = np.stack([npy(t) for t in lmom[0]])
    # end-to-end sample() on the scaled funnel, jitter=None, draws recorded
    D = 4
    lp = funnel_ll(D, SCALES)
    hamiltorch.set_random_seed(77)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample(lp, torch.tensor([0.2, -0.3, 0.4, 0.1], dtype=torch.float64), num_samples=10,
                                     num_steps_per_sample=4, step_size=0.1, burn=1, jitter=None, softabs_const=1e6,
                                     explicit_binding_const=10.0, sampler=hamiltorch.Sampler.RMHMC,
                                     integrator=hamiltorch.Integrator.EXPLICIT, metric=hamiltorch.Metric.SOFTABS,
                                     debug=2, verbose=False)
    out["e2e_samples"] = np.stack([npy(t) for t in ret])
    out["e2e_momenta"] = np.stack(rec.momenta)
    out["e2e_uniforms"] = np.concatenate(rec.uniforms)
    out["e2e_acc"] = np.array(acc)
    # implicit RMHMC (generalised leapfrog with fixed-point iterations, S:305-387) on the scaled funnel, jitter=None
    for D, alpha, tag in ((4, 1e6, "imp_a1e6"), (5, 1.1, "imp_a1p1")):
        lp = funnel_ll(D, SCALES)
        g = torch.Generator().manual_seed(8)
        th = 0.5 * torch.randn(D, generator=g, dtype=torch.float64); pm = torch.randn(D, generator=g, dtype=torch.float64)
        out[f"{tag}_theta0"] = npy(th); out[f"{tag}_p0"] = npy(pm); out[f"{tag}_cfg"] = np.array([D, alpha, 0.05, 3, 1e-16, 30])
        lpar, lmom = S.leapfrog(th, pm, lp, steps=3, step_size=0.05, jitter=None, softabs_const=alpha,
                                fixed_point_threshold=1e-16, fixed_point_max_iterations=30, sampler=hamiltorch.Sampler.RMHMC,
                                integrator=hamiltorch.Integrator.IMPLICIT, metric=hamiltorch.Metric.SOFTABS)
        out[f"{tag}_lf_theta"] = np.stack([npy(t) for t in lpar]); out[f"{tag}_lf_p"] = np.stack([npy(t) for t in lmom])
    D = 4
    lp = funnel_ll(D, SCALES)
    hamiltorch.set_random_seed(78)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample(lp, torch.tensor([0.2, -0.3, 0.4, 0.1], dtype=torch.float64), num_samples=8,
                                     num_steps_per_sample=3, step_size=0.1, burn=1, jitter=None, softabs_const=1e6,
                                     fixed_point_threshold=1e-14, fixed_point_max_iterations=40,
                                     sampler=hamiltorch.Sampler.RMHMC, integrator=hamiltorch.Integrator.IMPLICIT,
                                     metric=hamiltorch.Metric.SOFTABS, debug=2, verbose=False)
    out["imp_e2e_samples"] = np.stack([npy(t) for t in ret])
    out["imp_e2e_momenta"] = np.stack(rec.momenta)
    out["imp_e2e_uniforms"] = np.concatenate(rec.uniforms)
    out["imp_e2e_acc"] = np.array(acc)
    np.savez(os.path.join(OUT, "funnel.npz"), **out)


def gen_logcosh():
    """Metric.HESSIAN on a general (log-concave, non-Gaussian) target: log p = -1/2 w^T P w - sum log cosh(A w).
    Explicit and implicit RMHMC leapfrog paths of the reference, jitter=None."""
    out = {}
    D = 5
    rng = np.random.default_rng(1)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    P = (Q * np.linspace(0.5, 2.0, D)) @ Q.T
    P = 0.5 * (P + P.T)
    A = 0.7 * rng.standard_normal((7, D))
    out["P"] = P; out["A"] = A
    Pt, At = torch.tensor(P), torch.tensor(A)

    def lp(w):
        return -0.5 * torch.dot(w, torch.mv(Pt.to(w.dtype), w)) - torch.log(torch.cosh(torch.mv(At.to(w.dtype), w))).sum()
    g = torch.Generator().manual_seed(4)
    th = 0.6 * torch.randn(D, generator=g, dtype=torch.float64); pm = torch.randn(D, generator=g, dtype=torch.float64)
    out["theta0"] = npy(th); out["p0"] = npy(pm); out["cfg"] = np.array([5.0, 0.08, 3, 1e-16, 30])
    G, _ = S.fisher(th, lp, jitter=None, metric=hamiltorch.Metric.HESSIAN)
    out["G"] = npy(G)
    out["H"] = npy(S.rm_hamiltonian(th, pm, lp, None, 1.0, metric=hamiltorch.Metric.HESSIAN)).reshape(-1)
    lpar, lmom = S.leapfrog(th, pm, lp, steps=3, step_size=0.08, jitter=None, explicit_binding_const=5.0,
                            sampler=hamiltorch.Sampler.RMHMC, integrator=hamiltorch.Integrator.EXPLICIT,
                            metric=hamiltorch.Metric.HESSIAN)
    out["exp_theta"] = np.stack([npy(t) for t in lpar[0]]); out["exp_p"] = np.stack([npy(t) for t in lmom[0]])
    lpar, lmom = S.leapfrog(th, pm, lp, steps=3, step_size=0.08, jitter=None, fixed_point_threshold=1e-16,
                            fixed_point_max_iterations=30, sampler=hamiltorch.Sampler.RMHMC,
                            integrator=hamiltorch.Integrator.IMPLICIT, metric=hamiltorch.Metric.HESSIAN)
    out["imp_theta"] = np.stack([npy(t) for t in lpar]); out["imp_p"] = np.stack([npy(t) for t in lmom])
    np.savez(os.path.join(OUT, "logcosh.npz"), **out)


def gen_signatures():
    """Parameter names, order and defaults of the reference's public functions on the path, as JSON."""
    import inspect
    import json
    from hamiltorch import util as U

    def sig(f):
        return [[k, None if v.default is inspect._empty else repr(v.default)] for k, v in inspect.signature(f).parameters.items()]
    out = {}
    for n in ("sample", "sample_model", "sample_split_model", "predict_model"):
        out[n] = sig(getattr(hamiltorch, n))
    for n in ("leapfrog", "hamiltonian", "rm_hamiltonian", "fisher", "gibbs", "cholesky_inverse", "acceptance", "collect_gradients",
              "define_model_log_prob", "define_split_model_log_prob", "adaptation"):
        out["samplers." + n] = sig(getattr(S, n))
    for n in ("flatten", "unflatten", "setup_chain", "multi_chain", "set_random_seed", "has_nan_or_inf", "update_model_params_in_place"):
        out["util." + n] = sig(getattr(U, n))
    out["enums"] = {e.__name__: {m.name: m.value for m in e} for e in (hamiltorch.Sampler, hamiltorch.Integrator, hamiltorch.Metric)}
    json.dump(out, open(os.path.join(OUT, "signatures.json"), "w"), indent=1, sort_keys=True)


def custom_loss(out, y):                    # a user log-likelihood (S:1186-1188): summed over the batch by the reference
    return 1.5 * ((out - y) ** 2).sum(1)


LOSS_CASES = {
    "binary": dict(loss="binary_class_linear_output", dims=[4, 6, 1], log_softmax=False, classes=2),
    "multi": dict(loss="multi_class_linear_output", dims=[4, 6, 3], log_softmax=False, classes=3),
    "logsoftmax": dict(loss="multi_class_log_softmax_output", dims=[4, 6, 3], log_softmax=True, classes=3),
    "custom": dict(loss=custom_loss, dims=[4, 6, 2], log_softmax=False, classes=0),
}


def make_loss_net(c):
    net = make_mlp(c["dims"], "tanh", 0)
    if c["log_softmax"]:
        net = nn.Sequential(*list(net.children()), nn.LogSoftmax(dim=1))
    return net


def gen_losses():
    """define_model_log_prob for every model_loss kind besides 'regression' (S:1170-1190), the prior-only closure (S:1160-1162)
    and predict_model (S:1468-1562, tensors and DataLoader form), recorded from the reference on small networks."""
    out = {}
    for name, c in LOSS_CASES.items():
        net = make_loss_net(c)
        g = torch.Generator().manual_seed(4)
        N = 10
        X = torch.randn(N, 4, generator=g)
        if c["classes"]:
            Y = torch.randint(0, c["classes"], (N, 1), generator=g).float()
        else:
            Y = torch.randn(N, c["dims"][-1], generator=g)
        theta = (hamiltorch.util.flatten(net).clone().detach() + 0.1 * torch.randn(hamiltorch.util.flatten(net).numel(), generator=g))
        tau_list = torch.tensor([1.0 + 0.5 * k for k in range(len(list(net.parameters())))])
        pfl = [t.nelement() for t in net.parameters()]
        psl = [t.shape for t in net.parameters()]
        tau_out = 2.0
        out[f"{name}_X"] = npy(X); out[f"{name}_Y"] = npy(Y); out[f"{name}_theta"] = npy(theta); out[f"{name}_tau_list"] = npy(tau_list)
        f = S.define_model_log_prob(net, c["loss"], X, Y, pfl, psl, tau_list, tau_out)
        th = theta.clone().requires_grad_()
        v = f(th)
        out[f"{name}_logp"] = npy(v).reshape(-1)
        out[f"{name}_grad"] = npy(torch.autograd.grad(v.sum(), th)[0])
        f0 = S.define_model_log_prob(net, c["loss"], None, None, pfl, psl, tau_list, tau_out, prior_scale=3.0)
        out[f"{name}_prior_only"] = npy(f0(theta)).reshape(-1)
        samples = [theta + 0.05 * torch.randn(theta.numel(), generator=g) for _ in range(3)]
        out[f"{name}_samples"] = np.stack([npy(t) for t in samples])
        pred, lps = hamiltorch.predict_model(net, samples, x=X, y=Y, model_loss=c["loss"], tau_out=tau_out, tau_list=tau_list)
        out[f"{name}_pred"] = npy(pred); out[f"{name}_pred_lp"] = np.stack([npy(t).reshape(-1) for t in lps])
        loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=5, shuffle=False)
        pred, lps = hamiltorch.predict_model(net, samples, test_loader=loader, model_loss=c["loss"], tau_out=tau_out, tau_list=tau_list)
        out[f"{name}_pred_loader"] = npy(pred); out[f"{name}_pred_loader_lp"] = np.stack([npy(t).reshape(-1) for t in lps])
    np.savez(os.path.join(OUT, "losses.npz"), **out)


def gen_deepnet():
    """The shapes of the reference's notebooks that need more than one hidden layer / one output: log-probability and gradient
    of every closure of define_split_model_log_prob (prior divided by the number of splits), recorded from the reference."""
    class Net(nn.Module):                                   # hamiltorch_split_HMC_BNN_example / hamiltorch_Bayesian_NN_example
        def __init__(self, layer_sizes, act):
            super().__init__()
            self.act, self.n = act, len(layer_sizes) - 1
            self.l1 = nn.Linear(layer_sizes[0], layer_sizes[1])          # attributes l1, l2, ... as in the notebooks
            if self.n > 1:
                self.l2 = nn.Linear(layer_sizes[1], layer_sizes[2])
            if self.n > 2:
                self.l3 = nn.Linear(layer_sizes[2], layer_sizes[3])

        def forward(self, x):
            f = torch.relu if self.act == "relu" else torch.tanh
            x = self.l1(x)
            if self.n > 1:
                x = self.l2(f(x))
            if self.n > 2:
                x = self.l3(f(x))
            return x

    cases = {"deepreg": dict(dims=[1, 10, 10, 1], act="relu", loss="regression", N=24, M=2, classes=0, tau_out=7.0),
             "softmaxlin": dict(dims=[4, 3], act="relu", loss="multi_class_linear_output", N=30, M=3, classes=3, tau_out=1.0),
             "bin2": dict(dims=[3, 5, 4, 2], act="tanh", loss="binary_class_linear_output", N=20, M=2, classes=-1, tau_out=1.5)}
    out = {}
    for name, c in cases.items():
        torch.manual_seed(11)
        net = Net(c["dims"], c["act"])
        g = torch.Generator().manual_seed(5)
        X = torch.randn(c["N"], c["dims"][0], generator=g)
        if c["classes"] > 0:
            Y = torch.randint(0, c["classes"], (c["N"], 1), generator=g).float()
        elif c["classes"] < 0:
            Y = torch.randint(0, 2, (c["N"], c["dims"][-1]), generator=g).float()
        else:
            Y = torch.randn(c["N"], c["dims"][-1], generator=g)
        theta = hamiltorch.util.flatten(net).clone().detach() + 0.1 * torch.randn(hamiltorch.util.flatten(net).numel(), generator=g)
        tau_list = torch.tensor([1.0 + 0.25 * k for k in range(len(list(net.parameters())))])
        pfl = [t.nelement() for t in net.parameters()]
        psl = [t.shape for t in net.parameters()]
        loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=c["N"] // c["M"], shuffle=False)
        fns = S.define_split_model_log_prob(net, c["loss"], loader, c["M"], pfl, psl, tau_list, c["tau_out"], verbose=False)
        lps, grads = [], []
        for f in fns:
            th = theta.clone().requires_grad_()
            v = f(th)
            lps.append(npy(v).reshape(-1)[0] if v.numel() == 1 else npy(v.sum()))
            grads.append(npy(torch.autograd.grad(v.sum(), th)[0]))
        out.update({name + "_dims": np.array(c["dims"]), name + "_act": np.array(c["act"]), name + "_loss": np.array(c["loss"]),
                    name + "_M": np.array(c["M"]), name + "_X": npy(X), name + "_Y": npy(Y), name + "_theta": npy(theta),
                    name + "_tau_list": npy(tau_list), name + "_tau_out": np.array(c["tau_out"]), name + "_logp": np.array(lps, dtype=np.float64),
                    name + "_grad": np.stack(grads)})
    np.savez(os.path.join(OUT, "deepnet.npz"), **out)


def gen_cfg2():
    """BASELINE config 2's per-chain computation: KAT2 target, identity mass, L=25, eps=0.3 - 25-step leapfrog
    paths from four starts (fp32 + fp64) and an end-to-end sample() of 40 trajectories with the draws recorded."""
    out = {}
    g = torch.Generator().manual_seed(11)
    th0 = 0.1 * torch.randn(4, 3, generator=g, dtype=torch.float64)
    pm0 = torch.randn(4, 3, generator=g, dtype=torch.float64)
    out["theta0"] = npy(th0); out["p0"] = npy(pm0)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        lp = mvn_logp(torch.zeros(3, dtype=dt), torch.tensor(SIGMA3, dtype=dt))
        ths, pms = [], []
        for k in range(4):
            p, m = S.leapfrog(th0[k].to(dt), pm0[k].to(dt), lp, steps=25, step_size=0.3,
                              sampler=hamiltorch.Sampler.HMC, integrator=hamiltorch.Integrator.EXPLICIT)
            ths.append(np.stack([npy(x) for x in p])); pms.append(np.stack([npy(x) for x in m]))
        out[f"lf_theta_{tag}"] = np.stack(ths); out[f"lf_p_{tag}"] = np.stack(pms)        # [start, step, 3]
    lp = mvn_logp(torch.zeros(3), torch.tensor(SIGMA3))
    hamiltorch.set_random_seed(77)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample(lp, th0[0].float(), num_samples=40, num_steps_per_sample=25, step_size=0.3, burn=0,
                                     debug=2, verbose=False)
    out["e2e_samples"] = np.stack([npy(t) for t in ret])
    out["e2e_momenta"] = np.stack(rec.momenta)
    out["e2e_uniforms"] = np.concatenate(rec.uniforms)
    out["e2e_acc"] = np.array(acc)
    np.savez(os.path.join(OUT, "cfg2.npz"), **out)


def gen_cfg4():
    """BASELINE config 4 at full size: Linear(8,100)-ReLU-Linear(100,1) (D=1001), X = randn(400, 8), w = randn(8, 1),
    Y = sin(X w) + 0.1 randn (generator seed 0), 4 splits of 100 points, tau_out=100, tau_list=ones(4), inv_mass=ones(D),
    eps=5e-4, L=10: full-data log-prob + gradient, the split closures, one SPLITTING leapfrog path of L steps and
    end-to-end sample_split_model / sample_model runs with the reference's draws recorded.  Same key layout as mlp.npz."""
    out = {}
    name, dims, N, M, tau_out, eps, L = "relu_cfg4", [8, 100, 1], 400, 4, 100.0, 5e-4, 10
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, 8, generator=g); w = torch.randn(8, 1, generator=g)
    Y = torch.sin(X @ w) + 0.1 * torch.randn(N, 1, generator=g)
    net = make_mlp(dims, "relu", 0)
    theta = hamiltorch.util.flatten(net).clone().detach()
    D = theta.numel()
    tau_list = torch.ones(4)
    pfl = [t.nelement() for t in net.parameters()]
    psl = [t.shape for t in net.parameters()]
    out[f"{name}_dims"] = np.array(dims); out[f"{name}_X"] = npy(X); out[f"{name}_Y"] = npy(Y)
    out[f"{name}_theta"] = npy(theta); out[f"{name}_tau_list"] = npy(tau_list)
    out[f"{name}_cfg"] = np.array([M, tau_out, eps, L])
    f = S.define_model_log_prob(net, "regression", X, Y, pfl, psl, tau_list, tau_out)
    th = theta.clone().requires_grad_()
    v = f(th)
    out[f"{name}_logp"] = npy(v).reshape(-1)
    out[f"{name}_grad"] = npy(torch.autograd.grad(v.sum(), th)[0])
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=N // M, shuffle=False)
    fl = S.define_split_model_log_prob(net, "regression", loader, M, pfl, psl, tau_list, tau_out, verbose=False)
    out[f"{name}_split_logp"] = np.array([float(fm(theta).sum()) for fm in fl])
    p0 = torch.randn(D, generator=torch.Generator().manual_seed(2))
    inv_mass = torch.ones(D)
    out[f"{name}_p0"] = npy(p0)
    out[f"{name}_H0"] = npy(S.hamiltonian(theta, p0, fl, inv_mass=inv_mass, sampler=hamiltorch.Sampler.HMC)).reshape(-1)
    lp_, lm_ = S.leapfrog(theta.clone(), p0.clone(), fl, steps=L, step_size=eps, inv_mass=inv_mass,
                          sampler=hamiltorch.Sampler.HMC, integrator=hamiltorch.Integrator.SPLITTING)
    out[f"{name}_lf_theta"] = npy(lp_[-1]); out[f"{name}_lf_p"] = npy(lm_[-1])
    hamiltorch.set_random_seed(33)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample_split_model(net, loader, theta.clone(), M, model_loss="regression", num_samples=5,
                                                 num_steps_per_sample=L, step_size=eps, burn=0, inv_mass=inv_mass,
                                                 tau_out=tau_out, tau_list=tau_list, debug=2, verbose=False)
    out[f"{name}_e2e_samples"] = np.stack([npy(t) for t in ret])
    out[f"{name}_e2e_momenta"] = np.stack(rec.momenta)
    out[f"{name}_e2e_uniforms"] = np.concatenate(rec.uniforms)
    out[f"{name}_e2e_acc"] = np.array(acc)
    hamiltorch.set_random_seed(34)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample_model(net, X, Y, theta.clone(), model_loss="regression", num_samples=4,
                                           num_steps_per_sample=L, step_size=eps, tau_out=tau_out, tau_list=tau_list,
                                           debug=2, verbose=False)
    out[f"{name}_full_samples"] = np.stack([npy(t) for t in ret])
    out[f"{name}_full_momenta"] = np.stack(rec.momenta)
    out[f"{name}_full_uniforms"] = np.concatenate(rec.uniforms)
    np.savez_compressed(os.path.join(OUT, "cfg4.npz"), **out)


def nbmlp_data(N=400):
    """Stand-in for the notebook's agw_1d data set (notebooks/hamiltorch_split_HMC_BNN_example.ipynb cell 5 downloads it;
    there is no network here): N = 400 one-dimensional inputs in three clusters with gaps, a smooth target plus noise,
    both standardised as the notebook's loader does.  torch generator seed 0."""
    g = torch.Generator().manual_seed(0)
    n3 = N // 3
    x = torch.cat([-7.2 + 2.4 * torch.rand(n3, generator=g), -1.2 + 2.4 * torch.rand(n3, generator=g),
                   4.8 + 2.4 * torch.rand(N - 2 * n3, generator=g)])
    x = x[torch.randperm(N, generator=g)]
    y = 0.3 * x + torch.sin(1.2 * x) * torch.cos(0.4 * x) + 0.25 * torch.randn(N, generator=g)
    X = ((x - x.mean()) / x.std(unbiased=False)).reshape(-1, 1).float()
    Y = ((y - y.mean()) / y.std(unbiased=False)).reshape(-1, 1).float()
    return X, Y


class NotebookNet(nn.Module):
    """notebooks/hamiltorch_split_HMC_BNN_example.ipynb cell 9, verbatim in structure: fc1 (1,100), fc2 (100,100), fc3 (100,1),
    F.relu between them (D = 10401)."""

    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(1, 100)
        self.fc2 = nn.Linear(100, 100)
        self.fc3 = nn.Linear(100, 1)

    def forward(self, x):
        x = torch.relu(self.fc1(x))
        x = torch.relu(self.fc2(x))
        return self.fc3(x)


def gen_nbmlp():
    """The one model the reference publishes a GPU number for (notebook cells 9-14, 23-25):
    Linear(1,100)-ReLU-Linear(100,100)-ReLU-Linear(100,1), D = 10401, 400 points, M = 4 splits of 100, tau = 1 for all six
    parameter tensors, tau_out = 110.4439498986428, inv_mass = ones(D), step_size = 5e-4 (notebook cell 12; L = 30 there, short
    paths here): full-data log-prob + gradient, the four split closures (value + gradient), H, a 3-step SPLITTING path, a
    4-step plain leapfrog path, and short end-to-end sample_split_model / sample_model runs with the reference's draws
    recorded.  Same key layout as cfg4.npz."""
    out = {}
    name, dims, N, M, tau_out, eps = "nbmlp", [1, 100, 100, 1], 400, 4, 110.4439498986428, 5e-4
    X, Y = nbmlp_data(N)
    torch.manual_seed(0)
    net = NotebookNet()
    theta = hamiltorch.util.flatten(net).clone().detach()
    D = theta.numel()
    assert D == 10401
    tau_list = torch.ones(6)
    pfl = [t.nelement() for t in net.parameters()]
    psl = [t.shape for t in net.parameters()]
    out[f"{name}_dims"] = np.array(dims); out[f"{name}_X"] = npy(X); out[f"{name}_Y"] = npy(Y)
    out[f"{name}_theta"] = npy(theta); out[f"{name}_tau_list"] = npy(tau_list)
    out[f"{name}_cfg"] = np.array([M, tau_out, eps, 3, 4])
    f = S.define_model_log_prob(net, "regression", X, Y, pfl, psl, tau_list, tau_out)
    th = theta.clone().requires_grad_()
    v = f(th)
    out[f"{name}_logp"] = npy(v).reshape(-1)
    out[f"{name}_grad"] = npy(torch.autograd.grad(v.sum(), th)[0])
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=N // M, shuffle=False)
    fl = S.define_split_model_log_prob(net, "regression", loader, M, pfl, psl, tau_list, tau_out, verbose=False)
    out[f"{name}_split_logp"] = np.array([float(fm(theta).sum()) for fm in fl])
    th = theta.clone().requires_grad_()
    out[f"{name}_split1_grad"] = npy(torch.autograd.grad(fl[1](th).sum(), th)[0])
    p0 = torch.randn(D, generator=torch.Generator().manual_seed(2))
    inv_mass = torch.ones(D)
    out[f"{name}_p0"] = npy(p0)
    out[f"{name}_H0"] = npy(S.hamiltonian(theta, p0, fl, inv_mass=inv_mass, sampler=hamiltorch.Sampler.HMC)).reshape(-1)
    lp_, lm_ = S.leapfrog(theta.clone(), p0.clone(), fl, steps=3, step_size=eps, inv_mass=inv_mass,
                          sampler=hamiltorch.Sampler.HMC, integrator=hamiltorch.Integrator.SPLITTING)
    out[f"{name}_lf_theta"] = npy(lp_[-1]); out[f"{name}_lf_p"] = npy(lm_[-1])
    lp_, lm_ = S.leapfrog(theta.clone(), p0.clone(), f, steps=4, step_size=eps, inv_mass=inv_mass,
                          sampler=hamiltorch.Sampler.HMC, integrator=hamiltorch.Integrator.EXPLICIT)
    out[f"{name}_full_lf_theta"] = npy(lp_[-1]); out[f"{name}_full_lf_p"] = npy(lm_[-1])
    hamiltorch.set_random_seed(33)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample_split_model(net, loader, theta.clone(), M, model_loss="regression", num_samples=2,
                                                 num_steps_per_sample=4, step_size=eps, burn=-1, inv_mass=inv_mass,
                                                 tau_out=tau_out, tau_list=tau_list, debug=2, verbose=False)
    out[f"{name}_e2e_samples"] = np.stack([npy(t) for t in ret])
    out[f"{name}_e2e_momenta"] = np.stack(rec.momenta)
    out[f"{name}_e2e_uniforms"] = np.concatenate(rec.uniforms)
    out[f"{name}_e2e_acc"] = np.array(acc)
    hamiltorch.set_random_seed(34)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample_model(net, X, Y, theta.clone(), model_loss="regression", num_samples=2,
                                           num_steps_per_sample=5, step_size=eps, burn=-1, inv_mass=inv_mass, tau_out=tau_out,
                                           tau_list=tau_list, debug=2, verbose=False)
    out[f"{name}_full_samples"] = np.stack([npy(t) for t in ret])
    out[f"{name}_full_momenta"] = np.stack(rec.momenta)
    out[f"{name}_full_uniforms"] = np.concatenate(rec.uniforms)
    out[f"{name}_full_acc"] = np.array(acc)
    np.savez_compressed(os.path.join(OUT, "nbmlp.npz"), **out)


def gen_cfg3():
    """BASELINE config 3 at full size: D=100, P = Q diag(linspace(.5, 2, 100)) Q^T (generator seed 0), soft-abs
    metric alpha=1e6, omega=10, eps=0.1 - metric pieces, Hamiltonian, a 3-step explicit leapfrog path without jitter
    (fp32 + fp64), a 2-step path with jitter=1e-3 and the reference's torch.rand(D) draws recorded in call order, and an
    end-to-end sample() with every draw recorded."""
    out = {}
    D, alpha, omega, eps, jitter = 100, 1e6, 10.0, 0.1, 1e-3
    P64 = rand_spd(D, 0)
    out["cfg"] = np.array([D, alpha, omega, eps, jitter])
    g = torch.Generator().manual_seed(3)
    th64 = 0.1 * torch.randn(D, generator=g, dtype=torch.float64)
    pm64 = torch.randn(D, generator=g, dtype=torch.float64)
    kw = dict(explicit_binding_const=omega, softabs_const=alpha, sampler=hamiltorch.Sampler.RMHMC,
              integrator=hamiltorch.Integrator.EXPLICIT, metric=hamiltorch.Metric.SOFTABS)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        P, th, pm = P64.to(dt), th64.to(dt), pm64.to(dt)
        lp = quad_logp(P)
        out[f"P_{tag}"] = npy(P); out[f"theta0_{tag}"] = npy(th); out[f"p0_{tag}"] = npy(pm)
        G, lam = S.fisher(th, lp, jitter=None, softabs_const=alpha, metric=hamiltorch.Metric.SOFTABS)
        out[f"lam_{tag}"] = npy(lam); out[f"Gdiag_{tag}"] = npy(torch.diagonal(G))
        out[f"Ginvp_{tag}"] = npy(S.cholesky_inverse(G.detach(), pm)).reshape(-1)
        out[f"H_{tag}"] = npy(S.rm_hamiltonian(th, pm, lp, None, 1.0, softabs_const=alpha, metric=hamiltorch.Metric.SOFTABS)).reshape(-1)
        lpar, lmom = S.leapfrog(th, pm, lp, steps=3, step_size=eps, jitter=None, **kw)
        out[f"lf_theta_{tag}"] = np.stack([npy(t) for t in lpar[0]]); out[f"lf_p_{tag}"] = np.stack([npy(t) for t in lmom[0]])
        out[f"lf_thetac_{tag}"] = npy(lpar[1]); out[f"lf_pc_{tag}"] = npy(lmom[1])
    # with jitter: 8 fisher() calls per step, each with its own torch.rand(D) (S:115)
    P, th, pm = P64.float(), th64.float(), pm64.float()
    lp = quad_logp(P)
    hamiltorch.set_random_seed(5)
    with Recorder() as rec:
        lpar, lmom = S.leapfrog(th, pm, lp, steps=2, step_size=eps, jitter=jitter, **kw)
    assert len(rec.jitters) == 16, len(rec.jitters)
    out["jit_draws"] = np.stack(rec.jitters)
    out["jit_lf_theta"] = np.stack([npy(t) for t in lpar[0]]); out["jit_lf_p"] = np.stack([npy(t) for t in lmom[0]])
    # end to end, jitter on: per trajectory 1 (gibbs) + 1 (H) + 8 L (leapfrog) + 1 (H) metric evaluations
    N, L = 4, 2
    hamiltorch.set_random_seed(9)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample(lp, th.clone(), num_samples=N, num_steps_per_sample=L, step_size=eps, burn=0, jitter=jitter,
                                     debug=2, verbose=False, **kw)
    assert len(rec.momenta) == N and len(rec.jitters) == N * (8 * L + 3), (len(rec.momenta), len(rec.jitters))
    out["e2e_cfg"] = np.array([N, L])
    out["e2e_samples"] = np.stack([npy(t) for t in ret])
    out["e2e_momenta"] = np.stack(rec.momenta)
    out["e2e_uniforms"] = np.concatenate(rec.uniforms)
    out["e2e_jitters"] = np.stack(rec.jitters)
    out["e2e_acc"] = np.array(acc)
    np.savez_compressed(os.path.join(OUT, "cfg3.npz"), **out)


def gen_funnel_hmc():
    """The reference's published funnel run (notebooks/hamiltorch_log_prob_examples.ipynb cells 22-24: the verbatim
    funnel_ll closure, HMC, eps = 0.2, L = 25, params_init = (0, 1, ..., 1)): 30 trajectories end to end with the draws
    recorded, and an explicit-RMHMC run of cell 30's settings (eps = 0.14, L = 25, omega = 10, soft-abs 1e6, jitter 1e-3:
    2 trajectories) - what bench.py's funnel-hmc / funnel-rmhmc cpu_baseline legs run (oracle/cpu_baseline.py)."""
    from cpu_baseline import funnel_ll
    out = {}
    init = torch.ones(11); init[0] = 0.0
    hamiltorch.set_random_seed(123)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample(funnel_ll, init, num_samples=30, num_steps_per_sample=25, step_size=0.2, burn=0, debug=2, verbose=False)
    out["hmc_samples"] = np.stack([npy(t) for t in ret]); out["hmc_acc"] = np.array(acc)
    out["hmc_momenta"] = np.stack(rec.momenta); out["hmc_uniforms"] = np.concatenate(rec.uniforms)
    hamiltorch.set_random_seed(123)
    with Recorder() as rec:
        ret, acc = hamiltorch.sample(funnel_ll, init, num_samples=2, num_steps_per_sample=25, step_size=0.14, burn=-1, jitter=1e-3,
                                     softabs_const=1e6, explicit_binding_const=10.0, sampler=hamiltorch.Sampler.RMHMC,
                                     integrator=hamiltorch.Integrator.EXPLICIT, metric=hamiltorch.Metric.SOFTABS, debug=2, verbose=False)
    out["rm_samples"] = np.stack([npy(t) for t in ret]); out["rm_acc"] = np.array(acc)
    np.savez_compressed(os.path.join(OUT, "funnel_hmc.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # python oracle/gen_golden.py funnel  -> only that family
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_hmc_kat()
    gen_gibbs()
    gen_sample_hmc()
    gen_rmhmc()
    gen_mlp()
    gen_nuts()
    gen_funnel()
    gen_splitkinds()
    gen_logcosh()
    gen_blockmass()
    gen_losses()
    gen_deepnet()
    gen_signatures()
    gen_cfg2()
    gen_cfg3()
    gen_cfg4()
    gen_nbmlp()
    gen_funnel_hmc()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))

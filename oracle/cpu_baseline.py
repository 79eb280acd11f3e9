"""ONE single-threaded CPU chain of a bench.py workload, timed: the worker process of bench.py's ``cpu_baseline`` leg.

*** TEST / MEASUREMENT INFRASTRUCTURE ONLY *** (see oracle/hmc_oracle.py for the rules) - never imported by the product.

    python oracle/cpu_baseline.py <workload> <seed> <seconds> [<init.pt|-> <row> [<rounds>]]      ->  one JSON line per round

(`init.pt`: a torch-saved [k, D] tensor; the chain starts from row `row` mod k instead of the workload's initial point - bench.py
hands the device's burned-in states over so that both sides of an ESS comparison start inside the posterior.)

SURVEY 8(d): "the unmodified reference from /root/reference (shim, no bytecode), same log_prob_func object, one chain per
process, torch.set_num_threads(1), one process per usable host core".  When the reference is importable here
(``HAMILTORCH_REFERENCE``, default /root/reference, with the termcolor shim of oracle/shim) the chain runs
``hamiltorch.sample`` / ``sample_model`` / ``sample_split_model`` themselves and the line says ``"kind": "reference"``;
on a box without it (the GPU box) the chain runs oracle/torch_port.py - the per-chain torch/autograd port that
tests/test_oracle_golden.py pins to the reference's recorded runs (bit for bit on cfg2) - and says ``"kind": "port"``.
``HTA_CPU_BASELINE=port`` forces the port (the two are compared in tests/test_oracle_golden.py).

Workloads (the bench.py keys): cfg2, cfg3, cfg4, nbmlp, nbmlp-full, funnel-hmc, funnel-rmhmc.
"""
import json
import os
import sys
import time

_T_PROC = time.time()
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import torch  # noqa: E402

SIGMA = [[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]]          # KAT2 / cfg2
ESS_DIMS = 128                                                       # coordinates of a sample that travel to bench.py for ESS / s


def load_reference():
    """The unmodified reference package, or None."""
    if os.environ.get("HTA_CPU_BASELINE", "") == "port":
        return None
    root = os.environ.get("HAMILTORCH_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(root, "hamiltorch")):
        return None
    sys.path.insert(0, os.path.join(HERE, "shim"))
    sys.path.insert(0, root)
    try:
        import hamiltorch
        return hamiltorch
    except Exception:
        return None


def cfg3_precision(D=100):
    g = torch.Generator().manual_seed(0)
    Q = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0]
    P = (Q * torch.linspace(0.5, 2.0, D, dtype=torch.float64)) @ Q.T
    return (0.5 * (P + P.T)).float()


def cfg4_data():
    g = torch.Generator().manual_seed(0)
    X = torch.randn(400, 8, generator=g); w = torch.randn(8, 1, generator=g)
    Y = torch.sin(X @ w) + 0.1 * torch.randn(400, 1, generator=g)
    return X, Y


def nbmlp_data(N=400):          # the stand-in data of tests/golden/nbmlp.npz (oracle/gen_golden.py::nbmlp_data)
    g = torch.Generator().manual_seed(0)
    n3 = N // 3
    x = torch.cat([-7.2 + 2.4 * torch.rand(n3, generator=g), -1.2 + 2.4 * torch.rand(n3, generator=g),
                   4.8 + 2.4 * torch.rand(N - 2 * n3, generator=g)])
    x = x[torch.randperm(N, generator=g)]
    y = 0.3 * x + torch.sin(1.2 * x) * torch.cos(0.4 * x) + 0.25 * torch.randn(N, generator=g)
    X = ((x - x.mean()) / x.std(unbiased=False)).reshape(-1, 1).float()
    Y = ((y - y.mean()) / y.std(unbiased=False)).reshape(-1, 1).float()
    return X, Y


def funnel_ll(w, dim=10):
    """notebooks/hamiltorch_log_prob_examples.ipynb cell 22, verbatim."""
    v_dist = torch.distributions.Normal(0, 3)
    ll = v_dist.log_prob(w[0])
    x_dist = torch.distributions.Normal(0, torch.exp(-w[0]) ** 0.5)
    ll += x_dist.log_prob(w[1:]).sum()
    return ll


def build(workload, ref, start=None):
    """(run(n) -> (samples list or None, acceptance), L, note): one chain of `workload` for n trajectories (from `start` if given)."""
    import torch_port as TP

    def at(init):
        return init if start is None else start.to(init.dtype).reshape(init.shape)
    if workload == "cfg2":
        L, eps = 25, 0.3
        cov = torch.tensor(SIGMA)

        def lp(w):
            return torch.distributions.MultivariateNormal(torch.zeros(3), cov).log_prob(w).sum()
        init = at(torch.zeros(3))
        if ref:
            return (lambda n: ref.sample(lp, init, num_samples=n, num_steps_per_sample=L, step_size=eps, burn=-1, debug=2, verbose=False)), L, \
                "hamiltorch.sample, HMC"
        return (lambda n: TP.port_sample(lp, init, n, L, eps, burn=-1)), L, "torch_port.port_sample"
    if workload == "cfg3":
        L, eps, omega, alpha, jitter = 10, 0.1, 10.0, 1e6, 1e-3
        P = cfg3_precision()

        def lp(w):
            return -0.5 * torch.dot(w, torch.mv(P, w))
        init = at(0.1 * torch.randn(100, generator=torch.Generator().manual_seed(0)))
        if ref:
            return (lambda n: ref.sample(lp, init, num_samples=n, num_steps_per_sample=L, step_size=eps, burn=-1, jitter=jitter,
                                         softabs_const=alpha, explicit_binding_const=omega, sampler=ref.Sampler.RMHMC,
                                         integrator=ref.Integrator.EXPLICIT, metric=ref.Metric.SOFTABS, debug=2, verbose=False)), L, \
                "hamiltorch.sample, explicit RMHMC (autograd through hessian + eigh per gradient)"
        return (lambda n: TP.port_sample_rmhmc(lp, init, n, L, eps, omega, alpha, burn=-1, jitter=jitter)), L, "torch_port.port_sample_rmhmc"
    if workload in ("funnel-hmc", "funnel-rmhmc"):
        init = torch.ones(11); init[0] = 0.0
        init = at(init)
        if workload == "funnel-hmc":                         # notebook cell 24
            L, eps = 25, 0.2
            if ref:
                return (lambda n: ref.sample(funnel_ll, init, num_samples=n, num_steps_per_sample=L, step_size=eps, burn=-1, debug=2,
                                             verbose=False)), L, "hamiltorch.sample, HMC on the notebook's funnel_ll"
            return (lambda n: TP.port_sample(funnel_ll, init, n, L, eps, burn=-1)), L, "torch_port.port_sample on the notebook's funnel_ll"
        L, eps, omega, alpha, jitter = 25, 0.14, 10.0, 1e6, 1e-3          # notebook cell 30
        if ref:
            return (lambda n: ref.sample(funnel_ll, init, num_samples=n, num_steps_per_sample=L, step_size=eps, burn=-1, jitter=jitter,
                                         softabs_const=alpha, explicit_binding_const=omega, sampler=ref.Sampler.RMHMC,
                                         integrator=ref.Integrator.EXPLICIT, metric=ref.Metric.SOFTABS, debug=2, verbose=False)), L, \
                "hamiltorch.sample, explicit RMHMC on the notebook's funnel_ll"
        return (lambda n: TP.port_sample_rmhmc(funnel_ll, init, n, L, eps, omega, alpha, burn=-1, jitter=jitter)), L, \
            "torch_port.port_sample_rmhmc on the notebook's funnel_ll"
    if workload == "cfg4":
        L, eps, M = 10, 5e-4, 4
        X, Y = cfg4_data()
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(8, 100), torch.nn.ReLU(), torch.nn.Linear(100, 1))
        init = at(torch.cat([p.detach().flatten() for p in net.parameters()]))
        D = init.numel()
        if ref:
            loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=100, shuffle=False)
            return (lambda n: ref.sample_split_model(net, loader, init, M, model_loss="regression", num_samples=n, num_steps_per_sample=L,
                                                     step_size=eps, burn=-1, inv_mass=torch.ones(D), tau_out=100.0, tau_list=torch.ones(4),
                                                     debug=2, verbose=False)), L, "hamiltorch.sample_split_model, M = 4"
        fl = [TP.port_mlp_closure(net, X[m * 100:(m + 1) * 100], Y[m * 100:(m + 1) * 100], torch.ones(4), 100.0, 4) for m in range(4)]
        return (lambda n: TP.port_sample_split(fl, init, n, L, eps, -1, torch.ones(D))), L, "torch_port.port_sample_split"
    if workload in ("nbmlp", "nbmlp-full"):
        L, eps, tau_out = 30, 5e-4, 110.44
        M, Nb = (4, 100) if workload == "nbmlp" else (1, 400)
        X, Y = nbmlp_data()
        torch.manual_seed(0)
        net = TP.notebook_net()
        # the same initial point as bench.py (its Sequential has the same parameter order and the same seed)
        torch.manual_seed(0)
        seq = torch.nn.Sequential(torch.nn.Linear(1, 100), torch.nn.ReLU(), torch.nn.Linear(100, 100), torch.nn.ReLU(), torch.nn.Linear(100, 1))
        init = at(torch.cat([p.detach().flatten() for p in seq.parameters()]))
        D = init.numel()
        tl = torch.ones(6)
        if ref:
            if M > 1:
                loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=Nb, shuffle=False)
                return (lambda n: ref.sample_split_model(net, loader, init, M, model_loss="regression", num_samples=n, num_steps_per_sample=L,
                                                         step_size=eps, burn=-1, inv_mass=torch.ones(D), tau_out=tau_out, tau_list=tl,
                                                         debug=2, verbose=False)), L, "hamiltorch.sample_split_model, M = 4"
            return (lambda n: ref.sample_model(net, X, Y, init, model_loss="regression", num_samples=n, num_steps_per_sample=L,
                                               step_size=eps, burn=-1, inv_mass=torch.ones(D), tau_out=tau_out, tau_list=tl, debug=2,
                                               verbose=False)), L, "hamiltorch.sample_model, full HMC"
        if M > 1:
            fl = [TP.port_mlp_closure(net, X[m * Nb:(m + 1) * Nb], Y[m * Nb:(m + 1) * Nb], tl, tau_out, M) for m in range(M)]
            return (lambda n: TP.port_sample_split(fl, init, n, L, eps, -1, torch.ones(D))), L, "torch_port.port_sample_split"
        f = TP.port_mlp_closure(net, X, Y, tl, tau_out, 1.0)
        return (lambda n: TP.port_sample(f, init, n, L, eps, -1, torch.ones(D))), L, "torch_port.port_sample"
    raise SystemExit("unknown workload %r" % workload)


def main():
    workload, seed, seconds = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    torch.set_num_threads(1)
    ref = load_reference()
    states = torch.load(sys.argv[4]) if len(sys.argv) > 5 and sys.argv[4] != "-" else None
    import warnings
    warnings.filterwarnings("ignore")
    # The notebooks ran on a torch whose distributions did not validate their arguments: a diverged trajectory handed NaN to
    # funnel_ll, got NaN back and was REJECTED (has_nan_or_inf -> LogProbError, S:783-785, S:1045).  Today's default raises a
    # ValueError inside log_prob instead, which the reference does not catch - its explicit-RMHMC funnel run would end the process.
    torch.distributions.Distribution.set_default_validate_args(False)
    dt0 = None
    t_start = time.time()
    stamp = (lambda what: print("cpu_baseline[%d] %6.2f s %s" % (os.getpid(), time.time() - t_start, what), file=sys.stderr, flush=True)) \
        if os.environ.get("HTA_CPU_BASELINE_TIMING") else (lambda what: None)
    stamp("imports done (process age %.2f s)" % (time.time() - _T_PROC))
    for rep in range(rounds):
        # every round is a NEW chain: its own seed, its own start state (row + rep * stride of the device's burned-in states)
        start = None if states is None else states[(int(sys.argv[5]) + 17 * rep) % states.shape[0]].clone()
        run, L, note = build(workload, ref, start)
        torch.manual_seed(seed + 97 * rep)
        if ref:
            ref.set_random_seed(seed + 97 * rep)
        n0 = 20 if workload == "cfg2" else 1
        if dt0 is None:
            run(n0)                                                       # the first call pays torch's lazy initialisation
            t0 = time.time(); run(n0); dt0 = (time.time() - t0) / n0
        n = max(40 if workload == "cfg2" else 1, int(seconds / max(dt0, 1e-6)))
        stamp("round %d: built, warm, n = %d" % (rep, n))
        t0 = time.time()
        ret, acc = run(n)
        dt = time.time() - t0
        stamp("round %d: sampled in %.2f s" % (rep, dt))
        out = {"kind": "reference" if ref else "port", "impl": note, "n": n, "L": L, "dt": dt, "acc": float(acc), "round": rep}
        # the samples travel (first ESS_DIMS coordinates, 6 significant digits), for ESS / s with the estimator bench.py applies to the
        # device samples (hamiltorch_amd/ess.py over the same coordinates); the reference's output is this list (S:1086-1091)
        rows = torch.stack([r.detach().reshape(-1)[:ESS_DIMS] for r in ret[1:]])
        out["ess_dims"] = int(rows.shape[1])
        out["samples"] = [[float("%.6g" % v) for v in row] for row in rows.tolist()]
        print(json.dumps(out), flush=True)
        stamp("round %d: printed" % rep)


if __name__ == "__main__":
    main()

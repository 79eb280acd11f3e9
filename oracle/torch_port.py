"""Per-chain torch/autograd port of the reference's CPU path -- the ``cpu_baseline`` of bench.py.

*** TEST / MEASUREMENT INFRASTRUCTURE ONLY *** (see oracle/hmc_oracle.py for the rules).

Where hmc_oracle.py restates the arithmetic batched in numpy (the parity checker), this file
keeps the reference's *cost structure*: one chain, sequential trajectories, the user callback
evaluated through torch autograd at every leapfrog step (hamiltorch/samplers.py:270-278), the
global torch CPU generator for the draws (S:185-186, S:1004).  It is therefore what "the
reference's CPU path" costs on the host cores of the box bench.py runs on, where
/root/reference itself is not available.

Pinned to the reference: with the same ``torch.manual_seed`` it reproduces the reference's
``sample()`` output bit for bit (tests/test_oracle_golden.py::test_torch_port_cfg1_bit_identical).
"""
import torch


def _grad(log_prob_func, q):
    q = q.detach().requires_grad_()                       # S:271
    return torch.autograd.grad(log_prob_func(q), q)[0]   # S:272, S:65


def _vel(p, inv_mass):
    if inv_mass is None:
        return p                                          # S:284
    if inv_mass.dim() == 2:
        return torch.matmul(inv_mass, p.view(-1, 1)).view(-1)   # S:294
    return inv_mass * p                                   # S:296


def port_leapfrog(q, p, log_prob_func, steps, step_size, inv_mass=None):
    """S:267-304 (HMC branch)."""
    q = q.clone(); p = p.clone()
    p += 0.5 * step_size * _grad(log_prob_func, q)        # S:281
    g = None
    for _ in range(steps):
        q = q + step_size * _vel(p, inv_mass)
        g = _grad(log_prob_func, q)                       # S:297
        p += step_size * g                                # S:298
    return q, p - 0.5 * step_size * g                     # S:302


def port_hamiltonian(q, p, log_prob_func, inv_mass=None):
    """S:779-815; a non-finite log-prob raises (LogProbError, S:783-785)."""
    lp = log_prob_func(q)
    if not bool(torch.isfinite(lp.sum())):
        raise FloatingPointError
    if inv_mass is None:
        kin = 0.5 * torch.dot(p, p)
    elif inv_mass.dim() == 2:
        kin = 0.5 * torch.matmul(p.view(1, -1), torch.matmul(inv_mass, p.view(-1, 1))).view(-1)
    else:
        kin = 0.5 * torch.dot(p, inv_mass * p)
    return -lp + kin


def port_gibbs(q, mass=None):
    """S:185-202."""
    if mass is None:
        return torch.distributions.Normal(torch.zeros_like(q), torch.ones_like(q)).sample()
    if mass.dim() == 2:
        return torch.distributions.MultivariateNormal(torch.zeros_like(q), mass).sample()
    return torch.distributions.Normal(torch.zeros_like(q), mass ** 0.5).sample()


def port_sample(log_prob_func, params_init, num_samples, num_steps_per_sample, step_size, burn=0, inv_mass=None):
    """S:942-1091 for sampler=HMC: returns (list of (D,) tensors, acceptance rate)."""
    if burn >= num_samples:
        raise RuntimeError('burn must be less than num_samples.')
    mass = None
    if inv_mass is not None:
        mass = torch.inverse(inv_mass) if inv_mass.dim() == 2 else 1 / inv_mass   # S:949-952
    params = params_init.clone()
    burn_prev = params_init.clone()
    ret = [params.clone()]
    rejected = 0
    for n in range(num_samples):
        try:
            p = port_gibbs(params, mass)                                              # S:969
            ham = port_hamiltonian(params, p, log_prob_func, inv_mass)                # S:971
            q_new, p_new = port_leapfrog(params, p, log_prob_func, num_steps_per_sample, step_size, inv_mass)
            params = q_new.detach()
            new_ham = port_hamiltonian(params, p_new, log_prob_func, inv_mass)        # S:995
            rho = min(0., float((-new_ham + ham).detach()))                                      # S:1000, S:626
            accept = bool(rho >= torch.log(torch.rand(1)))                            # S:1004
        except FloatingPointError:                                                    # S:1045
            accept = False
        if accept:
            if n > burn:
                ret.append(q_new.detach())
            else:
                burn_prev = q_new.detach().clone()
        else:
            rejected += 1
            if n > burn:
                params = ret[-1]
                ret.append(ret[-1])
            else:
                params = burn_prev.clone()
    return ret, 1 - rejected / num_samples


# ---------------------------------------------------------------------------------------------------
# Explicit RMHMC (soft-abs / Hessian metric) -- same cost structure as the reference: every gradient
# of the Riemannian Hamiltonian is an autograd pass through hessian + eigh + Cholesky (S:395-422).
# ---------------------------------------------------------------------------------------------------
class PortLogProbError(Exception):
    """The port's util.LogProbError (U:23-24): raised where the reference raises it, caught where the reference catches it (S:1045)."""


def _bad(t):
    return not bool(torch.isfinite(t).all())               # util.has_nan_or_inf (U:27-41)


def port_fisher(q, log_prob_func, alpha, softabs=True, jitter=None):
    """S:96-122; jitter draws torch.rand(D) from the global generator exactly where the reference does (S:113-115).  A non-finite
    Hessian raises (S:110-112) - BEFORE eigh, which today's LAPACK would end with a LinAlgError the reference does not catch."""
    hess = torch.autograd.functional.hessian(log_prob_func, q, create_graph=True)   # S:108
    fish = -hess
    if _bad(fish):
        raise PortLogProbError                                                       # S:110-112
    if jitter is not None:
        n = fish.shape[0]
        fish = fish + torch.eye(n) * torch.rand(n) * jitter                         # S:115
    if not softabs:
        return fish, None
    lam, Q = torch.linalg.eigh(fish, UPLO='L')                                       # S:119
    lam_t = (1. / torch.tanh(alpha * lam)) * lam                                     # S:120
    return torch.matmul(Q, torch.matmul(lam_t.diag(), Q.t())), lam_t                 # S:121


def port_rm_hamiltonian(q, p, log_prob_func, alpha, softabs=True, jitter=None):
    """S:710-736, with its three rejection checks (S:715-722, S:733-735)."""
    from numpy import pi
    lp = log_prob_func(q)
    pi_term = q.nelement() * torch.log(2. * torch.tensor(pi))                        # S:712 (float32)
    fish, lam_t = port_fisher(q, log_prob_func, alpha, softabs, jitter)
    if _bad(fish) or (lam_t is not None and _bad(lam_t)):
        raise PortLogProbError                                                       # S:715-722
    logdet = lam_t.log().sum() if softabs else torch.slogdet(fish)[1]                # S:726 / S:728
    low = torch.linalg.cholesky(fish)                                                # S:146-148
    y = torch.linalg.solve_triangular(low, p.view(-1, 1), upper=False)
    x = torch.linalg.solve_triangular(low.t(), y, upper=True)
    ham = -lp + 0.5 * pi_term + 0.5 * logdet + 0.5 * torch.matmul(p.view(1, -1), x)
    if _bad(ham):
        raise PortLogProbError                                                       # S:733-735
    return ham


def port_explicit_leapfrog(q, p, log_prob_func, steps, step_size, omega, alpha, softabs=True, jitter=None, jitter_max_tries=10):
    """S:425-461; a non-finite dH/dtheta redraws the jitter and tries again, at most jitter_max_tries times (S:402-410)."""
    def dH_dq(qq, pp):                                                               # S:395-412
        tries = 0
        while True:
            qq = qq.detach().requires_grad_()
            g = torch.autograd.grad(port_rm_hamiltonian(qq, pp.detach(), log_prob_func, alpha, softabs, jitter), qq)[0]
            if not _bad(g):
                return g
            tries += 1
            if tries > jitter_max_tries:
                raise PortLogProbError                                               # S:407-409

    def dH_dp(qq, pp):                                                               # S:415-422
        pp = pp.detach().requires_grad_(); qq = qq.detach().requires_grad_()
        return torch.autograd.grad(port_rm_hamiltonian(qq, pp, log_prob_func, alpha, softabs, jitter), pp)[0]
    q = q.clone(); p = p.clone(); qc = q.clone(); pc = p.clone()
    for _ in range(steps):
        p = p - 0.5 * step_size * dH_dq(q, pc)
        qc = qc + 0.5 * step_size * dH_dp(q, pc)
        q = q + 0.5 * step_size * dH_dp(qc, p)
        pc = pc - 0.5 * step_size * dH_dq(qc, p)
        c = torch.cos(torch.FloatTensor([2 * omega * step_size])); s = torch.sin(torch.FloatTensor([2 * omega * step_size]))
        q = 0.5 * ((q + qc) + c * (q - qc) + s * (p - pc))                           # S:447-450, sequential
        p = 0.5 * ((p + pc) - s * (q - qc) + c * (p - pc))
        qc = 0.5 * ((q + qc) - c * (q - qc) - s * (p - pc))
        pc = 0.5 * ((p + pc) + s * (q - qc) - c * (p - pc))
        q = q + 0.5 * step_size * dH_dp(qc, p)
        pc = pc - 0.5 * step_size * dH_dq(qc, p)
        p = p - 0.5 * step_size * dH_dq(q, pc)
        qc = qc + 0.5 * step_size * dH_dp(q, pc)
    return q, p


def port_sample_rmhmc(log_prob_func, params_init, num_samples, num_steps_per_sample, step_size, omega, alpha, burn=0,
                      softabs=True, jitter=None):
    """S:969-1057, RMHMC / EXPLICIT branch; a trajectory that raises the reference's LogProbError is rejected (S:1045-1057)."""
    params = params_init.clone().requires_grad_()
    burn_prev = params_init.clone()
    ret = [params_init.clone()]
    rejected = 0
    for n in range(num_samples):
        try:
            G, _ = port_fisher(params, log_prob_func, alpha, softabs, jitter)
            p = torch.distributions.MultivariateNormal(torch.zeros_like(params), G).sample()   # S:183-184
            ham = 2 * port_rm_hamiltonian(params, p, log_prob_func, alpha, softabs, jitter) / 2       # S:822, S:977
            q_new, p_new = port_explicit_leapfrog(params, p, log_prob_func, num_steps_per_sample, step_size, omega, alpha, softabs, jitter)
            params = q_new.detach().requires_grad_()
            new_ham = port_rm_hamiltonian(params, p_new, log_prob_func, alpha, softabs, jitter)       # S:989
            rho = min(0., float((-new_ham + ham).detach()))
            accept = bool(rho >= torch.log(torch.rand(1)))
        except PortLogProbError:                                                                  # S:1045
            accept = False
        if accept:
            if n > burn:
                ret.append(q_new.detach())
            else:
                burn_prev = q_new.detach().clone()
        else:
            rejected += 1
            if n > burn:
                params = ret[-1].detach().clone().requires_grad_()
                ret.append(ret[-1])
            else:
                params = burn_prev.clone().requires_grad_()
    return [t.detach() for t in ret], 1 - rejected / num_samples


# ---------------------------------------------------------------------------------------------------
# Bayesian MLP closures + symmetric split HMC (S:1141-1258, S:494-547)
# ---------------------------------------------------------------------------------------------------
def port_mlp_closure(model, x, y, tau_list, tau_out, prior_scale):
    """define_model_log_prob(model_loss='regression'), S:1145-1199."""
    names = [n for n, _ in model.named_parameters()]
    shapes = [w.shape for w in model.parameters()]
    sizes = [w.nelement() for w in model.parameters()]
    dists = [torch.distributions.Normal(torch.zeros_like(t), t ** -0.5) for t in tau_list]

    def f(params):
        i = 0
        l_prior = torch.zeros_like(params[0], requires_grad=True)
        tensors = {}
        for name, n, shp, d in zip(names, sizes, shapes, dists):
            w = params[i:i + n]
            l_prior = d.log_prob(w).sum() + l_prior
            tensors[name] = w.view(shp)
            i += n
        out = torch.func.functional_call(model, tensors, (x,))
        ll = - 0.5 * tau_out * ((out - y) ** 2).sum(0)
        return ll + l_prior / prior_scale
    return f


def port_sample_split(closures, params_init, num_samples, num_steps_per_sample, step_size, burn=0, inv_mass=None):
    """sample(integrator=SPLITTING) over a list of closures: S:494-547 inside S:965-1026."""
    M = len(closures)
    mass = None if inv_mass is None else 1 / inv_mass
    params = params_init.clone()
    burn_prev = params_init.clone()
    ret = [params.clone()]
    rejected = 0

    def ham(q, p):
        lp = 0
        with torch.no_grad():
            for f in closures:
                lp = lp + f(q)
        kin = 0.5 * torch.dot(p, p) if inv_mass is None else 0.5 * torch.dot(p, inv_mass * p)
        return -lp + kin

    def grad(q, f):
        q = q.detach().requires_grad_()
        return torch.autograd.grad(f(q), q)[0]
    for n in range(num_samples):
        p = port_gibbs(params, mass)
        h0 = ham(params, p)
        q = params.detach().clone()
        for _ in range(num_steps_per_sample):
            for m in range(M):
                g = grad(q, closures[m])
                with torch.no_grad():
                    p += 0.5 * step_size * g
                    if m < M - 1:
                        q += (step_size / ((M - 1) * 2)) * (p if inv_mass is None else inv_mass * p)
            for m in reversed(range(M)):
                g = grad(q, closures[m])
                with torch.no_grad():
                    p += 0.5 * step_size * g
                    if m > 0:
                        q += (step_size / ((M - 1) * 2)) * (p if inv_mass is None else inv_mass * p)
        h1 = ham(q, p)
        rho = min(0., float((-h1 + h0).detach()))
        if rho >= torch.log(torch.rand(1)):
            params = q.clone()
            if n > burn:
                ret.append(q.clone())
            else:
                burn_prev = q.clone()
        else:
            rejected += 1
            if n > burn:
                params = ret[-1]
                ret.append(ret[-1])
            else:
                params = burn_prev.clone()
    return ret, 1 - rejected / num_samples


# ---- bench.py cpu_baseline, config 2: one chain per process on the host cores ------------------------------
def cfg2_worker(args):
    """One single-threaded chain of the config-2 target for about `seconds` of wall clock.
    Returns (trajectories, L, sampling seconds, acceptance, samples[n,3] as a nested list)."""
    import time
    seed, L, eps, seconds, sigma = args
    torch.set_num_threads(1)
    cov = torch.tensor(sigma)

    def lp(w):
        return torch.distributions.MultivariateNormal(torch.zeros(3), cov).log_prob(w).sum()
    init = torch.zeros(3)
    torch.manual_seed(seed)
    t0 = time.time(); port_sample(lp, init, 20, L, eps); dt = time.time() - t0
    n = max(40, int(seconds / (dt / 20)))
    t0 = time.time(); ret, acc = port_sample(lp, init, n, L, eps, burn=-1); dt = time.time() - t0
    return n, L, dt, acc, torch.stack(ret[1:]).tolist()


def notebook_net():
    """notebooks/hamiltorch_split_HMC_BNN_example.ipynb cell 9: Linear(1,100)-ReLU-Linear(100,100)-ReLU-Linear(100,1) as a
    module with the notebook's parameter names (fc1, fc2, fc3): D = 10401."""
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = torch.nn.Linear(1, 100)
            self.fc2 = torch.nn.Linear(100, 100)
            self.fc3 = torch.nn.Linear(100, 1)

        def forward(self, x):
            x = torch.relu(self.fc1(x))
            x = torch.relu(self.fc2(x))
            return self.fc3(x)
    return Net()


if __name__ == "__main__":          # python oracle/torch_port.py cfg2 <seed> <L> <eps> <seconds>  -> one JSON line
    import json
    import sys
    assert sys.argv[1] == "cfg2"
    sigma = [[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]]
    n, L, dt, acc, samples = cfg2_worker((int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]), sigma))
    print(json.dumps({"n": n, "L": L, "dt": dt, "acc": acc, "samples": samples}))

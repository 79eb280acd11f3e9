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
            rho = min(0., float(-new_ham + ham))                                      # S:1000, S:626
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

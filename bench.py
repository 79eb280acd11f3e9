#!/usr/bin/env python
"""Headline benchmark: leapfrog-steps/sec at 1024 chains per GPU (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5]

One "step" = one pass of the hot path over one batch: a single launch of the fused trajectory
kernel that runs `--traj` whole trajectories (momentum draw, H, L leapfrog steps, H, Metropolis,
sample write-out) for every chain of this rank.  Inputs are resident in HBM before the timed
region.  For N > 1 every rank owns its own block of chains (global chain ids, no data-path
collective): weak scaling, value = all ranks' chain-steps / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and, at N=1,
`cpu_baseline` (the reference's per-chain torch/autograd cost structure, oracle/torch_port.py,
timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3       # fp32 vector == fp32 MFMA peak
SIGMA3 = [[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default: the workload's)")
    ap.add_argument("--traj", type=int, default=None, help="trajectories per launch (default: the workload's)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="also print a chain-count sweep (stderr)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------
def _usable_cores():
    """(logical CPUs visible, CPUs this process may actually keep busy: affinity mask and cgroup CPU quota)."""
    avail = os.cpu_count() or 1
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else avail
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())          # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return avail, n


class Cfg2:
    """BASELINE config 2: 3-D correlated Gaussian, HMC, 1024 chains, L=25, eps=0.3, 1000 trajectories."""
    name = "cfg2: 3-D correlated Gaussian HMC, L=25, eps=0.3, identity mass"
    D, L, eps, chains, traj = 3, 25, 0.3, 1024, 1000
    dtype_name = "f32"

    def __init__(self, dev, chains, traj, chain_offset, seed=0):
        import hamiltorch_amd as ht
        from hamiltorch_amd import _abi
        self.abi = _abi
        self.C, self.T = chains or self.chains, traj or self.traj
        self.off, self.seed = chain_offset, seed
        cov = torch.tensor(SIGMA3, dtype=torch.float32, device=dev)
        self.tgt = ht.GaussianTarget(torch.zeros(3, device=dev), covariance=cov)
        g = torch.Generator(device="cpu").manual_seed(1234 + chain_offset)
        self.theta0 = (0.1 * torch.randn(self.C, 3, generator=g)).to(dev)
        self.cur = self.theta0.clone()
        self.samples = torch.empty(self.T + 1, self.C, 3, device=dev)      # burn = -1: every trajectory stored
        self.samples[0].copy_(self.theta0)
        self.rej = torch.zeros(self.C, dtype=torch.int32, device=dev)
        self.ws = torch.empty(_abi.gaussian_workspace_bytes(self.C, 3, self.T, 4), dtype=torch.uint8, device=dev)

    def units_per_step(self):
        return self.C * self.T * self.L

    def bytes_per_unit(self):      # theta and p, fp32, read + written once per chain-step
        return 16 * self.D

    def step(self, k):
        self.abi.hmc_gaussian_sample(self.cur, self.theta0, self.tgt.precision, self.tgt.mean, self.tgt.log_norm,
                                     0, None, None, self.L, self.eps, self.T, 0, -1, self.seed + k, self.off,
                                     self.samples, self.rej, workspace=self.ws)

    def check(self):
        s = self.samples[1:]
        assert torch.isfinite(s).all()
        pooled = s.reshape(-1, 3).double()
        cov = torch.cov(pooled.T).cpu()
        want = torch.tensor(SIGMA3, dtype=torch.float64)
        assert torch.allclose(cov, want, rtol=0.08, atol=0.04), cov
        return 1.0 - float(self.rej.double().mean()) / (self.T * max(1, self._steps_done))

    def cpu_baseline(self, seconds):
        """The reference's CPU cost structure on this host: one single-threaded chain per process (autograd
        callback, torch CPU RNG), one process per usable host core (affinity mask / cgroup CPU quota; HTA_BENCH_CPU_PROCS
        caps it); rate = all leapfrog steps / the slowest process's sampling time."""
        import subprocess
        avail, procs = _usable_cores()
        procs = max(1, min(procs, int(os.environ.get("HTA_BENCH_CPU_PROCS", "64"))))
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        t0 = time.time()
        ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "torch_port.py"), "cfg2", str(1000 + i),
                                str(self.L), repr(self.eps), repr(float(seconds))], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, env=env, text=True) for i in range(procs)]
        res = []
        for p_ in ps:
            out_, _ = p_.communicate(timeout=60 + 20 * seconds)
            r = json.loads(out_.strip().splitlines()[-1])
            res.append((r["n"], r["L"], r["dt"], r["acc"], r["samples"]))
        wall = time.time() - t0
        steps = sum(n * L for n, L, _, _, _ in res)
        dt = max(r[2] for r in res)
        from hamiltorch_amd.ess import ess_min
        ess = sum(ess_min(torch.tensor(r[4]).unsqueeze(1)) for r in res)
        return {"value": steps / dt, "unit": "leapfrog-steps/s", "cores": procs, "kind": "port",
                "sample": "%d processes x 1 chain x ~%d trajectories x L=%d, one thread each (oracle/torch_port.py: per-step "
                          "autograd on a MultivariateNormal.log_prob callback, as the reference), %.1f s sampling, %.1f s with "
                          "process start-up" % (procs, res[0][0], self.L, dt, wall),
                "per_core": steps / dt / procs, "ess_per_sec": ess / dt,
                "acceptance": sum(r[3] for r in res) / procs, "host_cores_available": avail, "host_cores_usable": procs}


class Cfg3:
    """BASELINE config 3: D=100 Gaussian, explicit RMHMC, soft-abs metric, 256 chains."""
    name = "cfg3: D=100 Gaussian explicit RMHMC, softabs alpha=1e6, omega=10, eps=0.1, L=10, jitter=1e-3"
    D, L, eps, chains, traj = 100, 10, 0.1, 256, 400
    omega, alpha, jitter = 10.0, 1e6, 1e-3
    dtype_name = "f32"

    def __init__(self, dev, chains, traj, chain_offset, seed=0):
        import hamiltorch_amd as ht
        from hamiltorch_amd import _abi
        self.abi = _abi
        self.C, self.T = chains or self.chains, traj or self.traj
        self.off, self.seed = chain_offset, seed
        g = torch.Generator().manual_seed(0)
        Q = torch.linalg.qr(torch.randn(self.D, self.D, generator=g, dtype=torch.float64))[0]
        P = (Q * torch.linspace(0.5, 2.0, self.D, dtype=torch.float64)) @ Q.T
        P = 0.5 * (P + P.T)
        self.P64 = P
        self.tgt = ht.GaussianTarget(torch.zeros(self.D, device=dev), precision=P.float().to(dev), normalized=False)
        g2 = torch.Generator().manual_seed(1234 + chain_offset)
        self.theta0 = (0.1 * torch.randn(self.C, self.D, generator=g2)).to(dev)
        self.cur = self.theta0.clone()
        self.samples = torch.empty(self.T + 1, self.C, self.D, device=dev)
        self.samples[0].copy_(self.theta0)
        self.rej = torch.zeros(self.C, dtype=torch.int32, device=dev)
        self.ws = torch.empty(_abi.rmhmc_workspace_bytes(self.C, self.D, 4, self.T), dtype=torch.uint8, device=dev)
        if os.environ.get("HTA_RMHMC_FUSED", "1") == "0":      # reproduce the general (eigendecomposition per evaluation) path
            _abi.set_tuning("rmhmc_fused", 0)
            self.name += " [Jacobi path forced]"
            self.roof_kernel = "metric_eval_kernel<float,2,2>"
            self.flops_per_unit = lambda: 4 * 11.3 * self.D ** 3

    def units_per_step(self):
        return self.C * self.T * self.L

    def flops_per_unit(self):
        # flops the fused kernel executes per explicit step (csrc/rmhmc_fused.hip): 4 half steps x (2 + K) symmetric
        # matrix-vector products (K = 2 refinements at jitter 1e-3) + the trajectory's Cholesky (D^3 / 3 FMAs) spread over
        # its L steps.  (The reference's route - an eigendecomposition per metric evaluation - is 4 x 11.3 D^3 = 4.5e7
        # per step; the soft-abs map is the identity on this spectrum, see DESIGN.md.)
        return 4 * 4 * 2 * self.D ** 2 + (2 * self.D ** 3 / 3) / self.L

    @property
    def roof_kernel(self):      # the trajectory kernel the library picks at this chain count (csrc/rmhmc_fused.hip dispatch)
        if "roof_kernel" in self.__dict__:
            return self.__dict__["roof_kernel"]
        if self.C < 704:
            return "rmhmc_fused_kernel<float,56,1> (one chain per workgroup)"
        if self.C <= 2048:
            return "rmhmc_mfma4_kernel (4 chains per workgroup, v_mfma_f32_4x4x1_16b)"
        return "rmhmc_batch_kernel<25> (16 chains per workgroup, v_mfma_f32_16x16x4) + rmhmc_momentum_wave_kernel<13>"

    @roof_kernel.setter
    def roof_kernel(self, v):
        self.__dict__["roof_kernel"] = v

    def bytes_per_unit(self):
        return 32 * self.D

    def step(self, k):
        self.abi.rmhmc_gaussian_sample(self.cur, self.theta0, self.tgt.precision, self.tgt.mean, self.tgt.log_norm,
                                       self.abi.METRIC_SOFTABS, self.alpha, self.jitter, self.L, self.eps, self.omega,
                                       self.T, 0, -1, self.seed + k, self.off, self.samples, self.rej, self.ws)

    def check(self):
        assert torch.isfinite(self.samples).all()
        return 1.0 - float(self.rej.double().mean()) / (self.T * max(1, self._steps_done))

    def cpu_baseline(self, seconds):
        """Reference cost structure: every dH/dtheta, dH/dp is an autograd pass through hessian + eigh (S:395-422)."""
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch_port as TP
        torch.set_num_threads(1)
        P = self.P64.float()

        def lp(w):
            return -0.5 * torch.dot(w, torch.mv(P, w))
        init = 0.1 * torch.randn(self.D, generator=torch.Generator().manual_seed(0))
        torch.manual_seed(0)
        t0 = time.time(); TP.port_sample_rmhmc(lp, init, 1, 1, self.eps, self.omega, self.alpha, jitter=self.jitter); dt1 = time.time() - t0
        n = max(1, int(seconds / (dt1 * self.L)))
        t0 = time.time()
        _, acc = TP.port_sample_rmhmc(lp, init, n, self.L, self.eps, self.omega, self.alpha, burn=-1, jitter=self.jitter)
        dt = time.time() - t0
        return {"value": n * self.L / dt, "unit": "leapfrog-steps/s", "cores": 1, "kind": "port",
                "sample": "1 chain x %d trajectories x L=%d explicit steps, jitter=%g (oracle/torch_port.py: autograd "
                          "through hessian+eigh per gradient, as the reference), %.1f s" % (n, self.L, self.jitter, dt),
                "acceptance": acc}


class Cfg5(Cfg3):
    """BASELINE config 5: cfg3 sharded over the node, 1024 chains per GPU (8192 on 8 GPUs), 100 trajectories per step; the one
    collective of the path - the gather of samples[S, C/G, D] over RCCL - runs after the timed region and is reported as
    `gather_ms` (excluded from the rate, 8e)."""
    name = "cfg5: cfg3 (D=100 explicit RMHMC, softabs, jitter=1e-3, L=10) sharded, 1024 chains per GPU"
    chains, traj = 1024, 100

    def gather(self, world):
        from hamiltorch_amd.dist import gather_samples
        return gather_samples(self.samples, self.C * world)


class Cfg4:
    """BASELINE config 4: Bayesian MLP 8-100-1 (D=1001), 400 points, symmetric split HMC M=4, 512 chains."""
    name = "cfg4: MLP Linear(8,100)-ReLU-Linear(100,1) regression, split HMC M=4 x 100 points, eps=5e-4, L=10"
    D, L, eps, chains, traj = 1001, 10, 5e-4, 512, 20
    dtype_name = "f32"

    def __init__(self, dev, chains, traj, chain_offset, seed=0):
        from hamiltorch_amd import _abi
        self.abi = _abi
        self.C, self.T = chains or self.chains, traj or self.traj
        self.off, self.seed = chain_offset, seed
        g = torch.Generator().manual_seed(0)
        X = torch.randn(400, 8, generator=g); w = torch.randn(8, 1, generator=g)
        Y = torch.sin(X @ w) + 0.1 * torch.randn(400, 1, generator=g)
        self.X, self.Y = X.to(dev).contiguous(), Y.reshape(-1).to(dev).contiguous()
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(8, 100), torch.nn.ReLU(), torch.nn.Linear(100, 1))
        flat = torch.cat([p.detach().flatten() for p in net.parameters()])
        self.theta0 = flat.repeat(self.C, 1).to(dev).contiguous()
        self.cur = self.theta0.clone()
        self.samples = torch.empty(self.T + 1, self.C, self.D, device=dev)
        self.rej = torch.zeros(self.C, dtype=torch.int32, device=dev)
        self.im = torch.ones(self.D, device=dev); self.mf = torch.ones(self.D, device=dev)

    def units_per_step(self):
        return self.C * self.T * self.L

    def flops_per_unit(self):      # 2M gradient evaluations x 6 flop per (point, weight): 8 x 6 x 100 x 900
        return 2 * 4 * 6 * 100 * 900

    def bytes_per_unit(self):
        return 16 * self.D

    roof_kernel = "mlp_mfma_kernel<2,7,0,512>"

    def step(self, k):
        self.abi.mlp_hmc_sample(self.cur, self.theta0, 8, 100, "relu", self.X, self.Y, 4, 100, [1.0] * 4, 100.0, 4.0,
                                self.abi.MASS_DIAG, self.im, self.mf, self.L, self.eps, self.T, 0, -1, self.seed + k,
                                self.off, self.samples, self.rej)

    def check(self):
        assert torch.isfinite(self.samples[1:]).all()
        return 1.0 - float(self.rej.double().mean()) / (self.T * max(1, self._steps_done))

    def cpu_baseline(self, seconds):
        """Reference cost structure: per-split closure + autograd gradient for every half kick (S:499-540)."""
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch_port as TP
        torch.set_num_threads(1)
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(8, 100), torch.nn.ReLU(), torch.nn.Linear(100, 1))
        X, Y = self.X.cpu(), self.Y.cpu().reshape(-1, 1)
        fl = [TP.port_mlp_closure(net, X[m * 100:(m + 1) * 100], Y[m * 100:(m + 1) * 100], torch.ones(4), 100.0, 4)
              for m in range(4)]
        init = self.theta0[0].cpu()
        im = torch.ones(self.D)
        t0 = time.time(); TP.port_sample_split(fl, init, 2, self.L, self.eps, -1, im); dt2 = time.time() - t0
        n = max(2, int(seconds / (dt2 / 2)))
        t0 = time.time(); _, acc = TP.port_sample_split(fl, init, n, self.L, self.eps, -1, im); dt = time.time() - t0
        return {"value": n * self.L / dt, "unit": "leapfrog-steps/s", "cores": 1, "kind": "port",
                "sample": "1 chain x %d trajectories x L=%d split steps (oracle/torch_port.py: functional model + "
                          "autograd per half kick, as the reference), %.1f s" % (n, self.L, dt), "acceptance": acc}


WORKLOADS = {"cfg2": Cfg2, "cfg3": Cfg3, "cfg4": Cfg4, "cfg5": Cfg5}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one rank per GPU over RCCL ("nccl" on ROCm).  HTA_BENCH_BACKEND=gloo + several ranks on one GPU is only
        # for exercising this code path on a single-GPU box.
        backend = os.environ.get("HTA_BENCH_BACKEND", "nccl")
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if a.workload not in WORKLOADS:
        try:
            import bench_extra
            WORKLOADS.update(bench_extra.WORKLOADS)
        except ImportError:
            pass
    W = WORKLOADS[a.workload]
    w = W(dev, a.chains, a.traj, chain_offset=rank * (a.chains or W.chains))
    w._steps_done = 0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    from hamiltorch_amd import _abi
    for kv in filter(None, os.environ.get("HTA_TUNING", "").split(",")):      # e.g. HTA_TUNING=rmhmc_momwave=0 (A/B runs)
        key, val = kv.split("=")
        _abi.set_tuning(key, int(val))
    for k in range(a.warmup):
        w.step(k)
    w.rej.zero_()
    barrier()
    # HIP event pair around the dominant kernel, on its stream, live in the timed region: every launch for the workloads with
    # several launches per step; every 4th for cfg2, where a pair (two barrier packets) is 3 % of a 0.19 ms step
    _abi.set_tuning("profile", 4 if a.workload == "cfg2" else 1)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    ev[0].record()          # one event pair around the whole timed region (per-step pairs put bubbles between the launches)
    for k in range(a.steps):
        w.step(a.warmup + k)
    ev[1].record()
    barrier()
    dt = time.perf_counter() - t0
    w._steps_done = a.steps
    call_ms = ev[0].elapsed_time(ev[1]) / max(1, a.steps)                 # device time per C-ABI call (all its kernels)
    prof_ms, prof_n = _abi.profile_collect()
    _abi.set_tuning("profile", 0)
    kernel_ms = prof_ms / max(1, prof_n)                                  # the trajectory kernel alone
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    acc = w.check()
    units = w.units_per_step() * a.steps * world
    value = units / dt
    gather_ms = None
    if hasattr(w, "gather"):            # the path's only collective, outside the timed region
        barrier()
        tg = time.perf_counter()
        gathered = w.gather(world)
        barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert gathered.shape[1] == w.C * world
        del gathered

    if rank == 0:
        alg_bytes = w.bytes_per_unit() * w.units_per_step()
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        if hasattr(w, "flops_per_unit"):
            kernel_ms = prof_ms / max(1, a.steps)          # all metric-evaluation launches of one step
            tf = w.flops_per_unit() * w.units_per_step() / (kernel_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / FP32_PEAK_TFLOPS, "traffic": None,      # filled below from profiles/traffic_<workload>.json
                    "kernel": getattr(w, "roof_kernel", "metric_eval_kernel<float>"),
                    "kernel_ms_per_step": kernel_ms, "launches_per_step": prof_n / max(1, a.steps),
                    "algorithmic_flops_per_chain_step": w.flops_per_unit(),
                    "note": "fp32 vector == fp32 MFMA peak (157.3 TF).  cfg3: flops the fused route executes (shared-inverse "
                            "solves + one Cholesky per trajectory; the eigh route of the reference would be 4.5e7 per step); "
                            "kernel time = every profiled launch of a step (trajectory kernels, and the momentum kernel when it "
                            "runs on the same stream); below 704 chains one 4-wave workgroup per chain: latency bound; cfg4: "
                            "2M x 6 flop per (point, weight) per split step (SURVEY 8d)"}
        else:
            roof = None
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % a.workload)
        if os.path.exists(tf) and w.C == W.chains and w.T == W.traj:       # measured at the workload's own shape only
            traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
        if roof is not None:
            roof["traffic"] = traffic                      # per step (all launches), like `achieved` for these workloads
        from hamiltorch_amd.ess import ess_min
        ess = ess_min(w.samples[1:]) if w.T >= 8 else float("nan")
        out = {
            "metric": "leapfrog-steps/sec (whole node) at 1024 chains; ESS/sec vs CPU ref",
            "value": value, "unit": "leapfrog-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": W.dtype_name, "data": "synthetic",
            "config": {"workload": W.name, "chains_per_gpu": w.C, "chains_total": w.C * world,
                       "trajectories_per_step": w.T, "leapfrog_steps_per_trajectory": W.L, "D": W.D,
                       "samples_stored": True, "parallelism": "chains sharded, %d per GPU, no collective" % w.C},
            "roofline": roof or {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "hmc_gauss_quad_kernel<3,false,25>" if w.C <= 65536 else "hmc_gauss_eig_kernel<float,3,false>",
                         "kernel_ms": kernel_ms,
                         "call_ms": call_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "16*D bytes per chain-step (SURVEY 8d); state is register-resident for the whole "
                                 "launch, so real HBM traffic is the draw records and the sample rows only; trajectories "
                                 "are integrated in the eigenbasis of P (identity mass), one eigen-coordinate per lane of "
                                 "a DPP quad: 2 dependent FMAs per step; at 1024 chains the launch is 64 waves on 256 "
                                 "CUs: latency/issue bound, not bandwidth bound"},
            "acceptance_rate": acc,
            "ess_per_sec": ess / (call_ms * 1e-3),
        }
        if gather_ms is not None:
            out["gather_ms"] = gather_ms
            out["config"]["parallelism"] += "; one all_gather of samples[%d, %d, %d] per rank after the timed region" % (
                w.T + 1, w.C, W.D)
        if world == 1 and not a.no_cpu_baseline:
            cb = w.cpu_baseline(a.cpu_seconds)
            if cb is not None:
                out["cpu_baseline"] = cb
                out["cpu_baseline"].setdefault("host_cores_available", os.cpu_count())
                out["speedup_vs_cpu_baseline"] = value / cb["value"]          # against the `cores` the baseline used
                out["speedup_vs_cpu_baseline_1core"] = value / (cb["value"] / max(1, cb.get("cores", 1)))
        print(json.dumps(out), flush=True)
        if a.sweep and a.workload == "cfg2":
            for C in (1024, 4096, 16384, 65536, 262144, 1048576):
                ws = W(dev, C, max(10, min(w.T, (1 << 24) // C)), 0)
                ws._steps_done = 1
                ws.step(0); torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); ws.step(1); e.record(); torch.cuda.synchronize()
                ms = s.elapsed_time(e)
                rate = ws.units_per_step() / (ms * 1e-3)
                print("sweep C=%8d T=%5d  %.3f ms  %.3e chain-steps/s  %.1f GB/s algorithmic (%.2f%% of HBM peak)"
                      % (C, ws.T, ms, rate, rate * ws.bytes_per_unit() / 1e9,
                         rate * ws.bytes_per_unit() / 1e9 / HBM_PEAK_GBS * 100), file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

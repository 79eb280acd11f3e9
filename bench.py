#!/usr/bin/env python
"""Headline benchmark: leapfrog-steps/sec at 1024 chains per GPU (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5]

One "step" = one pass of the hot path over one batch: one C-ABI call that runs `--traj` whole trajectories
(momentum draw, H, L leapfrog steps, H, Metropolis, sample write-out) for every chain of this rank.  Inputs are
resident in HBM before the timed region.  `--gpus N` with N > 1 launches N ranks itself (one per GPU, RCCL) unless it
already runs under torch.distributed.run; every rank owns its own block of chains (global chain ids, no data-path
collective): weak scaling, value = all ranks' chain-steps / max-over-ranks time.

Rank 0 prints ONE JSON line (contract in the task description).  `value` is the primary workload (cfg2: BASELINE
config 2, the 1024-chain headline).  At N = 1 the default run adds `"secondary"`: BASELINE.json's second north-star
target (D=100 explicit RMHMC at 1024 chains) and configs 3 / 4, each with its own `roofline` and `cpu_baseline`.
Every `roofline` carries the SURVEY 8(d) model fraction AND `physical` (counter HBM GB/s, SIMDs occupied, matrix-pipe
busy share: from the committed rocprofv3 PMC passes in profiles/physical.json, collected with tools/physical.sh).
"""
import argparse
import json
import os
import socket
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3       # fp32 vector == fp32 MFMA peak
N_SIMDS = 1024                 # 256 CUs x 4
SIGMA3 = [[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]]
METRIC = "leapfrog-steps/sec (whole node) at 1024 chains; ESS/sec vs CPU ref"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default: the workload's)")
    ap.add_argument("--traj", type=int, default=None, help="trajectories per launch (default: the workload's)")
    ap.add_argument("--cpu-seconds", type=float, default=9.0, help="wall-clock budget of ONE workload's CPU baseline (3 repeats)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="primary workload only")
    ap.add_argument("--no-api", action="store_true", help="skip the timing through hamiltorch_amd.sample()")
    ap.add_argument("--sweep", action="store_true", help="also print a chain-count sweep (stderr)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# helpers
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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _philox_init(abi, C, D, dev, chain_offset, scale=0.1):
    """params_init[c] = scale * N(0, I) from Philox(seed 0, subsequence = global chain id), drawn by the
    library's own generator (hta_momentum_resample with identity mass, draw 0)."""
    z = torch.empty(C, D, device=dev)
    abi.momentum_resample(z, abi.MASS_NONE, None, 0, chain_offset, 0)
    return (scale * z).contiguous()


def _median3(fn):
    """Median-by-value of three runs of fn() -> dict with "value" (repeat >= 3x, report the median)."""
    runs = [fn() for _ in range(3)]
    runs.sort(key=lambda r: r["value"])
    out = runs[1]
    out["repeats"] = [r["value"] for r in runs]
    return out


PINNED = {
    "cfg2": "tests/golden/cfg2.npz (test_torch_port_cfg2_bit_identical: the port reproduces the unmodified reference's sample() bit for bit)",
    "cfg3": "tests/golden/cfg3.npz (test_torch_port_cfg3_with_jitter_matches_reference_run)",
    "cfg4": "tests/golden/cfg4.npz (test_torch_port_cfg4_full_size_matches_reference_run)",
    "nbmlp": "tests/golden/nbmlp.npz (test_torch_port_nbmlp_matches_reference_run)",
    "nbmlp-full": "tests/golden/nbmlp.npz (test_torch_port_nbmlp_matches_reference_run)",
    "funnel-hmc": "tests/golden/funnel_hmc.npz (test_torch_port_funnel_hmc_matches_reference_run)",
    "funnel-rmhmc": "tests/golden/funnel.npz (the oracle's explicit RMHMC on the funnel against the reference's recorded paths)",
}


def cpu_baseline_procs(key, seconds, rounds=1):
    """one single-threaded chain per process, one process per usable host core (affinity mask / cgroup quota;
    HTA_BENCH_CPU_PROCS caps it), rate = all leapfrog steps / the slowest process's sampling time.  Each process is
    oracle/cpu_baseline.py: the UNMODIFIED reference when it is importable on this host (`kind: "reference"`), else the
    per-chain port pinned to the reference's recorded runs (`kind: "port"`).  `rounds` > 1: the median round."""
    import subprocess
    avail, procs = _usable_cores()
    procs = max(1, min(procs, int(os.environ.get("HTA_BENCH_CPU_PROCS", "64"))))
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="",
               PYTHONDONTWRITEBYTECODE="1")
    from hamiltorch_amd.ess import ess_min

    def once(rep):
        t0 = time.time()
        ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), key, str(1000 + 97 * rep + i),
                                repr(float(seconds) / rounds)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
              for i in range(procs)]
        res, failed = [], 0
        for p_ in ps:
            out_, _ = p_.communicate(timeout=300 + 30 * seconds)
            try:
                res.append(json.loads(out_.strip().splitlines()[-1]))
            except (IndexError, ValueError):
                failed += 1                                   # a chain that ended in an exception of the reference's own code path
        if not res:
            raise RuntimeError("cpu_baseline: every worker of %s failed" % key)
        procs_ok = len(res)
        wall = time.time() - t0
        L = res[0]["L"]
        steps = sum(r["n"] * r["L"] for r in res)
        dt = max(r["dt"] for r in res)
        out = {"value": steps / dt, "unit": "leapfrog-steps/s", "cores": procs_ok, "kind": res[0]["kind"], "workers_failed": failed,
               "sample": "%d processes x 1 chain x ~%d trajectories x L=%d, one thread each (%s), %.1f s sampling, %.1f s with "
                         "process start-up%s" % (procs, res[0]["n"], L, res[0]["impl"], dt, wall,
                                                 "; median of %d such rounds" % rounds if rounds > 1 else ""),
               "per_core": steps / dt / procs_ok, "samples_per_s": sum(r["n"] for r in res) / dt,
               "samples_per_s_per_core": sum(r["n"] for r in res) / dt / procs_ok, "acceptance": sum(r["acc"] for r in res) / procs_ok, "host_cores_available": avail, "host_cores_usable": procs}
        if "samples" in res[0]:
            out["ess_per_sec"] = sum(ess_min(torch.tensor(r["samples"]).unsqueeze(1)) for r in res) / dt
        return out
    reps = sorted((once(r) for r in range(rounds)), key=lambda r: r["value"])
    out = reps[len(reps) // 2]
    if rounds > 1:
        out["repeats"] = [r["value"] for r in reps]
    out["pinned_to"] = PINNED.get(key, "")
    return out


_PHYSICAL = None


def _physical(key):
    """Counter-derived utilisation of the workload's dominant kernel from the committed PMC passes (profiles/physical.json;
    tools/physical.sh collects FETCH_SIZE / WRITE_SIZE / SQ_BUSY_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVES in separate
    rocprofv3 --pmc runs of this file's own command line).  None when the profile was taken at another shape."""
    global _PHYSICAL
    if _PHYSICAL is None:
        try:
            _PHYSICAL = json.load(open(os.path.join(ROOT, "profiles", "physical.json")))
        except (OSError, ValueError):
            _PHYSICAL = {}
    return _PHYSICAL.get(key)


# ---------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------
class Cfg2:
    """BASELINE config 2: 3-D correlated Gaussian, HMC, 1024 chains, L=25, eps=0.3, 1000 trajectories."""
    key = "cfg2"
    name = "cfg2: 3-D correlated Gaussian HMC, L=25, eps=0.3, identity mass"
    D, L, eps, chains, traj = 3, 25, 0.3, 1024, 1000
    dtype_name = "f32"

    def __init__(self, dev, chains, traj, chain_offset, seed=1):
        import hamiltorch_amd as ht
        from hamiltorch_amd import _abi
        self.abi, self.ht = _abi, ht
        self.C, self.T = chains or self.chains, traj or self.traj
        self.off, self.seed = chain_offset, seed
        cov = torch.tensor(SIGMA3, dtype=torch.float32, device=dev)
        self.tgt = ht.GaussianTarget(torch.zeros(3, device=dev), covariance=cov)
        self.theta0 = _philox_init(_abi, self.C, 3, dev, chain_offset)
        self.cur = self.theta0.clone()
        self.samples = torch.empty(self.T + 1, self.C, 3, device=dev)      # burn = -1: every trajectory stored
        self.samples[0].copy_(self.theta0)
        self.rej = torch.zeros(self.C, dtype=torch.int32, device=dev)
        self.ws = torch.empty(_abi.gaussian_workspace_bytes(self.C, 3, self.T, 4), dtype=torch.uint8, device=dev)
        # once per target, as hamiltorch_amd.sample() does: the eig block of the workspace (HTA_BENCH_PREPARE=0: every call
        # diagonalises P itself, the behaviour before ABI 8)
        if os.environ.get("HTA_BENCH_PREPARE", "1") != "0":
            _abi.hmc_gaussian_prepare(self.theta0, self.tgt.precision, 0, None, self.C, 3, self.T, self.ws)

    def units_per_step(self):
        return self.C * self.T * self.L

    def bytes_per_unit(self):      # theta and p, fp32, read + written once per chain-step
        return 16 * self.D

    def step(self, k):
        self.abi.hmc_gaussian_sample(self.cur, self.theta0, self.tgt.precision, self.tgt.mean, self.tgt.log_norm,
                                     0, None, None, self.L, self.eps, self.T, 0, -1, self.seed + k, self.off,
                                     self.samples, self.rej, workspace=self.ws)

    def api_call(self, k):
        """The same work through the public API (hamiltorch_amd.sample: route selection, allocation of the sample tensor,
        the list of per-trajectory views)."""
        return self.ht.sample(self.tgt, self.theta0, num_samples=self.T, num_steps_per_sample=self.L, step_size=self.eps,
                              burn=-1, verbose=False, seed=self.seed + k, chain_offset=self.off)

    def check(self):
        s = self.samples[1:]
        assert torch.isfinite(s).all()
        pooled = s.reshape(-1, 3).double()
        cov = torch.cov(pooled.T).cpu()
        want = torch.tensor(SIGMA3, dtype=torch.float64)
        assert torch.allclose(cov, want, rtol=0.08, atol=0.04), cov
        return 1.0 - float(self.rej.double().mean()) / (self.T * max(1, self._steps_done))

    def roofline(self, kernel_ms, call_ms, prof_n, steps):
        alg_bytes = self.bytes_per_unit() * self.units_per_step()
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        quad = self.C <= 65536
        waves = (self.C * 4 + 63) // 64 if quad else (self.C + 63) // 64
        if "fused" in getattr(self, "route", ""):      # round 4: the producers of the draw records run in the same launch
            waves += 4 * self.abi.get_tuning("quad_producers") * ((self.C + 1023) // 1024)
        # the bound that physically applies at 1024 chains: one dependent FMA chain per wave.  2 L dependent v_fma_f32 per
        # trajectory at the 4-cycle dependent-issue latency (MI355X_MICROARCH.md) against the measured cycles per trajectory
        clk_ghz = self.abi.device_info(0)["clock_khz"] / 1e6
        cyc_per_traj = kernel_ms * 1e-3 * clk_ghz * 1e9 / self.T
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": "hmc_gauss_quad_fused_kernel<3,25>" if quad else "hmc_gauss_eig_kernel<float,3,false>",
                "kernel_ms": kernel_ms, "call_ms": call_ms, "algorithmic_bytes_per_launch": alg_bytes,
                "latency_model": {"dependent_fma_per_trajectory": 2 * self.L, "floor_cycles_per_trajectory": 8 * self.L,
                                  "measured_cycles_per_trajectory": cyc_per_traj, "frac_of_latency_floor": 8 * self.L / cyc_per_traj,
                                  "clock_ghz": clk_ghz},
                "waves_per_launch": waves, "simds_occupied_frac": min(1.0, waves / N_SIMDS),
                "note": "frac = SURVEY 8(d)'s streaming MODEL (16*D bytes per chain-step), not a utilisation: the state is "
                        "register-resident for the whole launch, real HBM traffic (`traffic`, PMC) is the draw records and the "
                        "sample rows only.  At 1024 chains the launch is 64 waves on 1024 SIMDs; what bounds it is the serial "
                        "chain of 2 L dependent FMAs per trajectory (eigenbasis of P, one eigen-coordinate per lane of a DPP "
                        "quad): see latency_model and physical"}

    def cpu_baseline(self, seconds):
        """The reference's CPU path on this host: see cpu_baseline_procs.  Median of three rounds."""
        return cpu_baseline_procs("cfg2", seconds, rounds=3)


class Cfg3:
    """BASELINE config 3: D=100 Gaussian, explicit RMHMC, soft-abs metric, 256 chains."""
    key = "cfg3"
    name = "cfg3: D=100 Gaussian explicit RMHMC, softabs alpha=1e6, omega=10, eps=0.1, L=10, jitter=1e-3"
    D, L, eps, chains, traj = 100, 10, 0.1, 256, 400
    omega, alpha, jitter = 10.0, 1e6, 1e-3
    dtype_name = "f32"
    SURVEY_FLOPS = 4 * 11.3 * 100 ** 3          # 4 metric evaluations x (eigh 9 D^3 + assembly 2 D^3 + Cholesky D^3/3): 4.5e7

    def __init__(self, dev, chains, traj, chain_offset, seed=1, jacobi=False):
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
        self.theta0 = _philox_init(_abi, self.C, self.D, dev, chain_offset)
        self.cur = self.theta0.clone()
        self.samples = torch.empty(self.T + 1, self.C, self.D, device=dev)
        self.samples[0].copy_(self.theta0)
        self.rej = torch.zeros(self.C, dtype=torch.int32, device=dev)
        self.ws = torch.empty(_abi.rmhmc_workspace_bytes(self.C, self.D, 4, self.T), dtype=torch.uint8, device=dev)
        self.jacobi = jacobi or os.environ.get("HTA_RMHMC_FUSED", "1") == "0"
        self._prepared = False
        if self.jacobi:      # the general route: an eigendecomposition per metric evaluation (what SURVEY 8d's flop count describes)
            self.name = self.name + " [eigendecomposition route forced: hta_set_tuning('rmhmc_fused', 0)]"

    def units_per_step(self):
        return self.C * self.T * self.L

    def executed_flops_per_unit(self):
        # flops the trajectory kernels execute per explicit step (csrc/rmhmc_fused.hip, csrc/rmhmc_uv.hip, tracked schedule):
        # 4 half steps x K refinement products (K = 2 at jitter 1e-3) + the 4 products after the rotation = 12 symmetric
        # matrix-vector products (round 1: 4 x (2 + K) = 16) + the trajectory's Cholesky (D^3 / 3 FMAs) spread over its L steps.
        return (4 * 2 + 4) * 2 * self.D ** 2 + (2 * self.D ** 3 / 3) / self.L

    @property
    def roof_kernel(self):      # the trajectory kernel the library picks at this chain count (csrc/rmhmc_fused.hip dispatch)
        if self.jacobi:
            return ("metric_traj_mfma_kernel (one launch per trajectory: a chain's workgroup runs its 4 L + 3 metric evaluations - "
                    "eigenvector refinement on v_mfma_f32_16x16x4_f32 - back to back) + mh_select_kernel")
        if self.C <= 256:
            return ("rmhmc_uvc_kernel (one chain per workgroup: state set and copy as columns of v_mfma_f32_4x4x1_16b, one value per "
                    "lane, three product phases per step)")
        if self.C <= 1792:
            return ("rmhmc_uvc2_kernel (two chains per workgroup, two workgroups per CU beyond 512 chains: four columns of "
                    "v_mfma_f32_4x4x1_16b, two values per lane)")
        if self.C <= 2048:
            return "rmhmc_mfma4_kernel (4 chains per two-wave workgroup, v_mfma_f32_4x4x1_16b)"
        return "rmhmc_batch_kernel<25> (16 chains per workgroup, v_mfma_f32_16x16x4) + rmhmc_momentum_wave_kernel<13>"

    def bytes_per_unit(self):
        return 32 * self.D

    def step(self, k):
        if self.jacobi:
            self.abi.set_tuning("rmhmc_fused", 0)
        try:
            if not self._prepared:      # once per target, as hamiltorch_amd.sample() does (rmhmc._prepared_workspace): the cold
                self._prepared = True   # eigendecomposition of P, the fused route's plan, the shared inverse - not per call
                self.abi.rmhmc_gaussian_prepare(self.cur, self.tgt.precision, self.tgt.mean, self.abi.METRIC_SOFTABS, self.alpha,
                                                self.jitter, self.C, self.ws)
            self.abi.rmhmc_gaussian_sample(self.cur, self.theta0, self.tgt.precision, self.tgt.mean, self.tgt.log_norm,
                                           self.abi.METRIC_SOFTABS, self.alpha, self.jitter, self.L, self.eps, self.omega,
                                           self.T, 0, -1, self.seed + k, self.off, self.samples, self.rej, self.ws)
        finally:
            if self.jacobi:
                self.abi.set_tuning("rmhmc_fused", 1)

    def api_call(self, k):
        """The same work through hamiltorch_amd.sample(sampler=RMHMC, integrator=EXPLICIT, metric=SOFTABS)."""
        import hamiltorch_amd as ht
        return ht.sample(self.tgt, self.theta0, num_samples=self.T, num_steps_per_sample=self.L, step_size=self.eps, burn=-1,
                         jitter=self.jitter, softabs_const=self.alpha, explicit_binding_const=self.omega, sampler=ht.Sampler.RMHMC,
                         integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, verbose=False, seed=self.seed + k,
                         chain_offset=self.off)

    def check(self):
        assert torch.isfinite(self.samples).all()
        return 1.0 - float(self.rej.double().mean()) / (self.T * max(1, self._steps_done))

    def roofline(self, kernel_ms, call_ms, prof_n, steps):
        """Two fractions, both against the 157.3 TFLOP/s fp32 peak (vector == matrix): `frac` in SURVEY 8(d)'s terms (4.5e7 flop
        per chain-step: the reference's eigendecomposition per metric evaluation) -- meaningful for the eigendecomposition
        route, and a >1 'model speed-up' for the closed-form routes, which do not do that work; `executed` = the flops the
        timed kernels really execute."""
        units = self.units_per_step()
        tf_survey = self.SURVEY_FLOPS * units / (kernel_ms * 1e-3) / 1e12
        tf_exec = (self.SURVEY_FLOPS if self.jacobi else self.executed_flops_per_unit()) * units / (kernel_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": tf_exec, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf_exec / FP32_PEAK_TFLOPS,
                "traffic": None, "kernel": self.roof_kernel, "kernel_ms_per_step": kernel_ms, "call_ms": call_ms,
                "launches_per_step": prof_n / max(1, steps),
                "executed_flops_per_chain_step": self.SURVEY_FLOPS if self.jacobi else self.executed_flops_per_unit(),
                "survey_8d": {"flops_per_chain_step": self.SURVEY_FLOPS, "achieved": tf_survey, "frac": tf_survey / FP32_PEAK_TFLOPS,
                              "note": "SURVEY 8(d) counts an eigendecomposition per metric evaluation; a frac > 1 here means the "
                                      "timed route does not do that work (closed form for constant curvature, DESIGN.md section 4)"},
                "note": "achieved/frac = flops the timed kernels execute (%s) / fp32 peak 157.3 TF; kernel time = every profiled "
                        "launch of a step" % ("eigendecomposition route: 4 x 11.3 D^3 per step" if self.jacobi else
                                               "shared-inverse solves + one Cholesky per trajectory")}

    def cpu_baseline(self, seconds):
        """Reference cost structure: every dH/dtheta, dH/dp is an autograd pass through hessian + eigh (S:395-422); one chain
        per usable host core."""
        return cpu_baseline_procs("cfg3", seconds)


class Cfg5(Cfg3):
    """BASELINE config 5: cfg3 sharded over the node, 1024 chains per GPU (8192 on 8 GPUs), 100 trajectories per step; the one
    collective of the path - the gather of samples[S, C/G, D] to rank 0 over RCCL - runs after the timed region and is
    reported as `gather_ms` (excluded from the rate, 8e)."""
    key = "cfg5"
    name = "cfg5: cfg3 (D=100 explicit RMHMC, softabs, jitter=1e-3, L=10) sharded, 1024 chains per GPU"
    chains, traj = 1024, 100

    def gather(self, world):
        from hamiltorch_amd.dist import gather_samples
        return gather_samples(self.samples, self.C * world, dst=0)


class Cfg3N(Cfg3):
    """BASELINE.json's second north-star target: the D=100 explicit-RMHMC problem at 1024 chains on one GPU."""
    key = "cfg3@1024"
    name = "north-star RMHMC target: " + Cfg3.name + ", 1024 chains"
    chains, traj = 1024, 100


class Cfg4:
    """BASELINE config 4: Bayesian MLP 8-100-1 (D=1001), 400 points, symmetric split HMC M=4, 512 chains."""
    key = "cfg4"
    name = "cfg4: MLP Linear(8,100)-ReLU-Linear(100,1) regression, split HMC M=4 x 100 points, eps=5e-4, L=10"
    D, L, eps, chains, traj = 1001, 10, 5e-4, 512, 20
    dtype_name = "f32"

    def __init__(self, dev, chains, traj, chain_offset, seed=1):
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
        self.net = net.to(dev)
        flat = torch.cat([p.detach().flatten() for p in net.parameters()])
        self.theta0 = flat.repeat(self.C, 1).to(dev).contiguous()
        self.cur = self.theta0.clone()
        self.samples = torch.empty(self.T + 1, self.C, self.D, device=dev)
        self.rej = torch.zeros(self.C, dtype=torch.int32, device=dev)
        self.im = torch.ones(self.D, device=dev); self.mf = torch.ones(self.D, device=dev)

    def units_per_step(self):
        return self.C * self.T * self.L

    def flops_per_unit(self):
        # EXECUTED flops per split step: 6 flop per (point, weight) and gradient evaluation; SURVEY 8d counts the reference's 2M
        # evaluations per step (8 x 6 x 100 x 900 = 4.32e6), the kernel executes (2M - 2) + 1/L of them - the two kicks at the
        # turning point and at the step boundary share one gradient (csrc/mlp.hpp: split_stage_reuses), same results
        return (2 * 4 - 2 + 1.0 / self.L) * 6 * 100 * 900

    def reference_flops_per_unit(self):
        return 2 * 4 * 6 * 100 * 900

    def bytes_per_unit(self):
        return 16 * self.D

    roof_kernel = "mlp_mfma_kernel"

    def api_call(self, k):
        """The same work through hamiltorch_amd.sample_split_model (the module is recognised by tracing; S:1364-1466)."""
        import hamiltorch_amd as ht
        if not hasattr(self, "_loader"):
            ds = torch.utils.data.TensorDataset(self.X, self.Y.reshape(-1, 1))
            self._loader = torch.utils.data.DataLoader(ds, batch_size=100, shuffle=False)
        return ht.sample_split_model(self.net, self._loader, self.theta0, 4, model_loss="regression", num_samples=self.T,
                                     num_steps_per_sample=self.L, step_size=self.eps, burn=-1, inv_mass=self.im, tau_out=100.0,
                                     tau_list=torch.ones(4), verbose=False, seed=self.seed + k, chain_offset=self.off)

    def step(self, k):
        self.abi.mlp_hmc_sample(self.cur, self.theta0, 8, 100, "relu", self.X, self.Y, 4, 100, [1.0] * 4, 100.0, 4.0,
                                self.abi.MASS_DIAG, self.im, self.mf, self.L, self.eps, self.T, 0, -1, self.seed + k,
                                self.off, self.samples, self.rej)

    def check(self):
        assert torch.isfinite(self.samples[1:]).all()
        return 1.0 - float(self.rej.double().mean()) / (self.T * max(1, self._steps_done))

    def roofline(self, kernel_ms, call_ms, prof_n, steps):
        tf = self.flops_per_unit() * self.units_per_step() / (kernel_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS,
                "traffic": None, "kernel": self.roof_kernel, "kernel_ms_per_step": kernel_ms, "call_ms": call_ms,
                "launches_per_step": prof_n / max(1, steps), "executed_flops_per_chain_step": self.flops_per_unit(),
                "reference_flops_per_chain_step": self.reference_flops_per_unit(),
                "frac_by_reference_flops": self.reference_flops_per_unit() * self.units_per_step() / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                "note": "2M x 6 flop per (point, weight) per split step (SURVEY 8d) against the fp32 matrix peak"}

    def cpu_baseline(self, seconds):
        """Reference cost structure: per-split closure + autograd gradient for every half kick (S:499-540); one chain per core."""
        return cpu_baseline_procs("cfg4", seconds)


class NbMlp:
    """The ONE model the reference publishes a GPU number for: notebooks/hamiltorch_split_HMC_BNN_example.ipynb -
    Linear(1,100)-ReLU-Linear(100,100)-ReLU-Linear(100,1), D = 10401, 400 points, tau = 1, tau_out = 110.44, inv_mass = ones,
    step_size = 5e-4, L = 30 (cell 12); symmetric split HMC with M = 4 splits of 100 points (cell 25: 1.83 samples/s) or full
    HMC (cell 14: 13.47 samples/s), one chain on an RTX 2080 Max-Q.  Here: 1024 chains (the metric's chain count), the
    notebook's data replaced by the synthetic stand-in of oracle/gen_golden.py (no network)."""
    key = "nbmlp"
    name = "nbmlp: Linear(1,100)-ReLU-Linear(100,100)-ReLU-Linear(100,1) regression (D=10401), split HMC M=4 x 100 points, eps=5e-4, L=30"
    D, L, eps, chains, traj = 10401, 30, 5e-4, 1024, 1
    M, Nb = 4, 100
    dtype_name = "f32"
    dims = [1, 100, 100, 1]
    tau_out = 110.4439498986428
    published = {"samples_per_s": 1.83, "hw": "RTX 2080 Max-Q, 1 chain", "src": "split_HMC_BNN nb cell 25"}

    def __init__(self, dev, chains, traj, chain_offset, seed=1):
        from hamiltorch_amd import _abi
        self.abi = _abi
        self.C, self.T = chains or self.chains, traj or self.traj
        self.off, self.seed = chain_offset, seed
        X, Y = self._data()
        self.X, self.Y = X.to(dev).contiguous(), Y.reshape(-1).to(dev).contiguous()
        torch.manual_seed(0)
        net = self._net()
        self.net = net.to(dev)
        flat = torch.cat([p.detach().flatten() for p in net.parameters()])
        self.theta0 = flat.repeat(self.C, 1).to(dev).contiguous()
        self.cur = self.theta0.clone()
        self.samples = torch.empty(self.T + 1, self.C, self.D, device=dev)
        self.rej = torch.zeros(self.C, dtype=torch.int32, device=dev)

    @staticmethod
    def _data(N=400):          # the same stand-in data as tests/golden/nbmlp.npz (oracle/gen_golden.py::nbmlp_data)
        g = torch.Generator().manual_seed(0)
        n3 = N // 3
        x = torch.cat([-7.2 + 2.4 * torch.rand(n3, generator=g), -1.2 + 2.4 * torch.rand(n3, generator=g),
                       4.8 + 2.4 * torch.rand(N - 2 * n3, generator=g)])
        x = x[torch.randperm(N, generator=g)]
        y = 0.3 * x + torch.sin(1.2 * x) * torch.cos(0.4 * x) + 0.25 * torch.randn(N, generator=g)
        X = ((x - x.mean()) / x.std(unbiased=False)).reshape(-1, 1).float()
        Y = ((y - y.mean()) / y.std(unbiased=False)).reshape(-1, 1).float()
        return X, Y

    @staticmethod
    def _net():
        return torch.nn.Sequential(torch.nn.Linear(1, 100), torch.nn.ReLU(), torch.nn.Linear(100, 100), torch.nn.ReLU(),
                                   torch.nn.Linear(100, 1))

    def units_per_step(self):
        return self.C * self.T * self.L

    def flops_per_unit(self):
        # EXECUTED: (2M - 2) + 1/L gradient evaluations per split step x 6 flop per (point, weight), P_w = 10200 weights; the
        # reference's loop differentiates 2M times per step (reference_flops_per_unit), twice at the same point (csrc/mlp.hpp)
        return (2 * self.M - 2 + 1.0 / self.L) * 6 * self.Nb * (100 + 100 * 100 + 100)

    def reference_flops_per_unit(self):      # 2M * 6 * N_b * P_w
        return 2 * self.M * 6 * self.Nb * (100 + 100 * 100 + 100)

    def bytes_per_unit(self):
        return 16 * self.D

    roof_kernel = "mlp3_mfma_kernel<0>"

    def api_call(self, k):
        """The same work through sample_split_model (M = 4) / sample_model (full HMC) on the notebook's module."""
        import hamiltorch_amd as ht
        kw = dict(model_loss="regression", num_samples=self.T, num_steps_per_sample=self.L, step_size=self.eps, burn=-1,
                  tau_out=self.tau_out, tau_list=torch.ones(6), verbose=False, seed=self.seed + k, chain_offset=self.off)
        if self.M == 1:
            return ht.sample_model(self.net, self.X, self.Y.reshape(-1, 1), self.theta0, **kw)
        if not hasattr(self, "_loader"):
            ds = torch.utils.data.TensorDataset(self.X, self.Y.reshape(-1, 1))
            self._loader = torch.utils.data.DataLoader(ds, batch_size=self.Nb, shuffle=False)
        return ht.sample_split_model(self.net, self._loader, self.theta0, self.M, **kw)

    def step(self, k):
        self.abi.netn_hmc_sample(self.cur, self.theta0, self.dims, "relu", self.X, self.Y, self.M, self.Nb, [1.0] * 6, self.tau_out,
                                 float(self.M), self.abi.MASS_NONE, None, None, self.L, self.eps, self.T, 0, -1, self.seed + k, self.off,
                                 self.samples, self.rej, integrator=self.abi.SPLIT_SYMMETRIC if self.M > 1 else 0)

    def check(self):
        assert torch.isfinite(self.samples[1:]).all()
        return 1.0 - float(self.rej.double().mean()) / (self.T * max(1, self._steps_done))

    def roofline(self, kernel_ms, call_ms, prof_n, steps):
        tf = self.flops_per_unit() * self.units_per_step() / (kernel_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS,
                "traffic": None, "kernel": self.roof_kernel, "kernel_ms_per_step": kernel_ms, "call_ms": call_ms,
                "launches_per_step": prof_n / max(1, steps), "executed_flops_per_chain_step": self.flops_per_unit(),
                "reference_flops_per_chain_step": self.reference_flops_per_unit(),
                "frac_by_reference_flops": self.reference_flops_per_unit() * self.units_per_step() / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}

    def cpu_baseline(self, seconds):
        """Reference cost structure: functional model + autograd per half kick (S:499-540) on the notebook's module; one chain per core."""
        return cpu_baseline_procs(self.key, seconds)

    def extras(self):
        """SURVEY 8(f) N2, the step after sampling in every BNN notebook: predict_model over 1000 of the samples just drawn x the
        400 points - natively (one hta_net_forward launch + batched log-probs) and on the torch path it replaces (vmap of the
        closure); the reference loops over the samples (S:1530-1552)."""
        import hamiltorch_amd as ht
        from hamiltorch_amd import bnn
        from hamiltorch_amd.samplelist import SampleList
        S = 1000
        rows = self.samples[1:].reshape(-1, self.D)[:S].contiguous()
        kw = dict(x=self.X, y=self.Y.reshape(-1, 1), model_loss="regression", tau_out=self.tau_out, tau_list=torch.ones(6))

        def timed(reps):
            ht.predict_model(self.net, SampleList(rows), **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                pred, lps = ht.predict_model(self.net, SampleList(rows), **kw)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / reps, bnn.predict_route["last"], tuple(pred.shape)
        ms, route, shape = timed(5)
        keep = bnn._native_forward_ok
        bnn._native_forward_ok = lambda *a, **k: None
        try:
            ms_t, route_t, _ = timed(2)
        finally:
            bnn._native_forward_ok = keep
        return {"predict_route": route, "predict_samples": S, "predict_ms": ms, "predict_ms_torch_path": ms_t,
                "predict_samples_per_s": S / (ms * 1e-3)}


class NbMlpFull(NbMlp):
    """The same model under full HMC (sample_model, notebook cell 14: 13.47 samples/s): plain leapfrog, every gradient over all 400 points."""
    extras = None
    key = "nbmlp-full"
    name = "nbmlp-full: the same model, full HMC (plain leapfrog over all 400 points), eps=5e-4, L=30"
    M, Nb = 1, 400
    published = {"samples_per_s": 13.47, "hw": "RTX 2080 Max-Q, 1 chain", "src": "split_HMC_BNN nb cell 14"}

    def flops_per_unit(self):      # one gradient over all points per step (+ the extra one of the first half kick, amortised over L)
        return 6 * self.Nb * (100 + 100 * 100 + 100) * (self.L + 1) / self.L

    def reference_flops_per_unit(self):
        return self.flops_per_unit()



def funnel_ll_device(w):
    """The funnel of notebooks/hamiltorch_log_prob_examples.ipynb cell 22 (v = w[0] ~ N(0, 3^2), x = w[1:] ~ N(0, exp(-v))) written
    with device-side arithmetic only - the form a HIP graph can replay (examples/funnel.py).  An OPAQUE closure for the library:
    models.probe_gaussian rejects it, every evaluation goes through the callback contract (S:272-274)."""
    v, x = w[0], w[1:]
    hl2p = 0.9189385332046727
    ll_v = -v * v / 18.0 - 1.0986122886681098 - hl2p
    ll_x = -0.5 * torch.exp(v) * (x * x).sum() + 0.5 * x.numel() * v - x.numel() * hl2p
    return ll_v + ll_x


def funnel_ll_notebook(w, dim=10):
    """Cell 22 verbatim (torch.distributions with host scalars: not capturable, evaluated eagerly under vmap)."""
    v_dist = torch.distributions.Normal(0, 3)
    ll = v_dist.log_prob(w[0])
    x_dist = torch.distributions.Normal(0, torch.exp(-w[0]) ** 0.5)
    ll += x_dist.log_prob(w[1:]).sum()
    return ll


class FunnelHMC:
    """The callback contract on the driver's line: the reference's published 11-D funnel run
    (notebook cell 24: HMC, eps = 0.2, L = 25: 56.10 samples/s, one chain) at 1024 chains through hamiltorch_amd.sample() with
    an opaque closure - torch evaluates the callback for all chains (vmap(grad_and_value)), the kicks / drifts / energies /
    Metropolis step are the HIP pieces kernels, a whole trajectory is replayed as one captured HIP graph."""
    key = "funnel-hmc"
    name = "funnel-hmc: 11-D funnel (notebook cell 22-24), HMC eps=0.2 L=25, opaque log_prob_func closure -> generic path"
    D, L, eps, chains, traj = 11, 25, 0.2, 1024, 50
    dtype_name = "f32"
    published = {"samples_per_s": 56.10, "hw": "notebook host, 1 chain", "src": "log_prob_examples nb cell 24 (JSON lines 401-402)"}
    sampler_kw = {}

    def __init__(self, dev, chains, traj, chain_offset, seed=1):
        from hamiltorch_amd import _abi
        self.abi = _abi
        self.C, self.T = chains or self.chains, traj or self.traj
        self.off, self.seed, self.dev = chain_offset, seed, dev
        self.theta0 = torch.ones(self.C, self.D, device=dev)
        self.theta0[:, 0] = 0.0
        self.rej = torch.zeros(self.C, dtype=torch.int32, device=dev)
        self.samples = None
        self._acc = []
        self.fn = funnel_ll_device

    def units_per_step(self):
        return self.C * self.T * self.L

    def bytes_per_unit(self):
        return 16 * self.D

    def _sample(self, fn, k, T):
        import hamiltorch_amd as ht
        return ht.sample(fn, self.theta0, num_samples=T, num_steps_per_sample=self.L, step_size=self.eps, burn=-1, debug=2,
                         verbose=False, seed=self.seed + k, chain_offset=self.off, **self.sampler_kw)

    def step(self, k):
        from hamiltorch_amd.samplelist import as_tensor
        out, acc = self._sample(self.fn, k, self.T)
        self.samples = as_tensor(out)
        self._acc.append(acc)

    def check(self):
        assert self.samples is not None and self.samples.shape[1:] == (self.C, self.D)
        fin = torch.isfinite(self.samples).all(dim=(0, 2))
        assert float(fin.float().mean()) > 0.99
        return float(torch.stack([a.float().mean() if torch.is_tensor(a) else torch.tensor(float(a)) for a in self._acc[-3:]]).mean())

    def _rate(self, fn, T, reps=2, env=None):
        old = {}
        for kk, vv in (env or {}).items():
            old[kk] = os.environ.get(kk); os.environ[kk] = vv
        try:
            self._sample(fn, 100, T)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(reps):
                self._sample(fn, 101 + r, T)
            torch.cuda.synchronize()
            return self.C * T * self.L * reps / (time.perf_counter() - t0)
        finally:
            for kk, vv in old.items():
                if vv is None:
                    os.environ.pop(kk, None)
                else:
                    os.environ[kk] = vv

    def _launches(self):
        """Device launches of one sample() call (torch.profiler, one untimed call); None if the profiler is unavailable."""
        try:
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                self._sample(self.fn, 200, self.T)
                torch.cuda.synchronize()
            return sum(e.count for e in prof.key_averages() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower())
        except Exception:
            return None

    def extras(self):
        """Graph replay on / off and the notebook's verbatim closure, each on a shorter run (not part of `value`)."""
        from hamiltorch_amd import util
        T = max(4, self.T // 5)
        out = {"graph_replay": not any("trajectory" in g_ for g_ in util.graph_log[-8:]),
               "value_graphs_off": self._rate(self.fn, T, env={"HAMILTORCH_AMD_GRAPHS": "0"}),
               "value_notebook_closure": self._rate(funnel_ll_notebook, T),
               "launches_per_step": self._launches(),
               "callback_evaluations_per_step": self.T * (self.L + 1)}
        return out

    def roofline(self, kernel_ms, call_ms, prof_n, steps):
        gbs = self.units_per_step() * self.bytes_per_unit() / (call_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                "kernel": "torch callback + hmc_pieces kernels (HIP graph per trajectory)", "kernel_fixed": True,
                "kernel_ms_per_step": call_ms, "call_ms": call_ms, "launches_per_step": None,
                "note": "SURVEY 8(d) byte model (16 D bytes per chain-step) over the whole call; the path is launch / latency bound "
                        "at this size (D = 11): the fraction is reported, not claimed"}

    def cpu_baseline(self, seconds):
        return cpu_baseline_procs(self.key, seconds)


class FunnelRMHMC(FunnelHMC):
    """SURVEY 8(f) N1 on the driver's line: explicit RMHMC with the soft-abs metric on the same funnel (notebook cell 30:
    eps = 0.14, L = 25, omega = 10, jitter = 1e-3; the reference's progress bar shows < 1 sample/s and its run ends in NaN
    after 14 samples) at 256 chains: per-chain Hessians by torch.func, hta_metric_eval with dmetric_out on the matrix cores."""
    key = "funnel-rmhmc"
    name = "funnel-rmhmc: 11-D funnel, explicit RMHMC softabs alpha=1e6 omega=10 eps=0.14 L=25 jitter=1e-3, opaque closure"
    D, L, eps, chains, traj = 11, 25, 0.14, 256, 2
    published = {"samples_per_s": 0.19, "hw": "notebook host, 1 chain", "src": "log_prob_examples nb cell 30 (JSON lines 637-638)"}

    def __init__(self, dev, chains, traj, chain_offset, seed=1):
        super().__init__(dev, chains, traj, chain_offset, seed)
        import hamiltorch_amd as ht
        self.sampler_kw = dict(sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, softabs_const=1e6,
                               explicit_binding_const=10.0, jitter=1e-3)

    def extras(self):
        T = self.T
        return {"value_graphs_off": self._rate(self.fn, T, reps=1, env={"HAMILTORCH_AMD_GRAPHS": "0"}),
                "metric_evaluations_per_step": self.T * (8 * self.L + 3), "launches_per_step": None}

    def roofline(self, kernel_ms, call_ms, prof_n, steps):
        # SURVEY 8(d)'s count at D = 11: 8 metric evaluations per step (the reference-faithful count: dH/dtheta depends on the metric
        # here) x 11.3 D^3 + the third-derivative contraction D^4 per kick
        flops = 8 * 11.3 * self.D ** 3 + 4 * 2 * self.D ** 4
        tf = flops * self.units_per_step() / (call_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS, "traffic": None,
                "kernel": "torch.func callbacks + metric_warm_mfma_kernel", "kernel_fixed": True,
                "kernel_ms_per_step": call_ms, "call_ms": call_ms, "launches_per_step": None,
                "note": "launch bound: D = 11 systems on kernels sized for D = 100; the fraction is reported, not claimed"}


WORKLOADS = {"funnel-hmc": FunnelHMC, "funnel-rmhmc": FunnelRMHMC, "nbmlp": NbMlp, "nbmlp-full": NbMlpFull, "cfg2": Cfg2, "cfg3": Cfg3, "cfg4": Cfg4, "cfg5": Cfg5, "cfg3@1024": Cfg3N}


# ---------------------------------------------------------------------------------------------------
# one measurement
# ---------------------------------------------------------------------------------------------------
def measure(w, steps, warmup, world, dist, dev, profile_every=1):
    """W untimed warm-up steps, then exactly `steps` timed steps bracketed by barrier + synchronize on both sides; the
    MAX over ranks of the wall time.  Also: device time per call (one event pair around the region) and the dominant
    kernels' time from HIP events recorded inside the library on the launch stream (hta_set_tuning('profile', n))."""
    abi = w.abi

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    w._steps_done = 0
    for k in range(warmup):
        w.step(k)
    w.rej.zero_()
    barrier()
    abi.set_tuning("profile", profile_every)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    ev[0].record()          # one event pair around the whole timed region (per-step pairs put bubbles between the launches)
    for k in range(steps):
        w.step(warmup + k)
    ev[1].record()
    barrier()
    dt = time.perf_counter() - t0
    w._steps_done = steps
    w.route = abi.last_route()                                            # the kernel the library dispatched to (hta_last_route)
    call_ms = ev[0].elapsed_time(ev[1]) / max(1, steps)                   # device time per C-ABI call (all its kernels)
    prof_ms, prof_n = abi.profile_collect()
    abi.set_tuning("profile", 0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    return dt, call_ms, prof_ms, prof_n


def result_of(w, W, dt, call_ms, prof_ms, prof_n, steps, warmup, world):
    """The per-workload record: value, roofline (+ physical), acceptance, ESS/s."""
    acc = w.check()
    units = w.units_per_step() * steps * world
    if isinstance(w, Cfg2):
        kernel_ms = prof_ms / max(1, prof_n)                              # the trajectory kernel alone (sampled launches)
    else:
        kernel_ms = prof_ms / max(1, steps)                               # every profiled launch of one step
    roof = w.roofline(kernel_ms, call_ms, prof_n, steps)
    if getattr(w, "route", "") and not roof.get("kernel_fixed"):
        roof["kernel_expected"], roof["kernel"] = roof.get("kernel"), w.route          # what ran, as the library reports it
    pkey = "%s%s@%d" % (W.key.split("@")[0], "jacobi" if getattr(w, "jacobi", False) else "", w.C)
    phys = _physical(pkey)
    roof["physical"] = phys
    if phys is not None and phys.get("trajectories_per_step") == w.T:      # bytes per step only at the shape they were counted at
        roof["traffic"] = phys.get("hbm_bytes_per_step")
    if phys is not None and roof.get("unit") == "TFLOP/s" and phys.get("mfma_tflops_issued"):
        # matrix-instruction flops ISSUED (PMC) over the useful flops of the roofline: > 1 = padding rows / idle columns of the
        # instruction (cfg3 at 256 chains: one chain per workgroup fills two of the four columns of v_mfma_f32_4x4x1_16b)
        roof["mfma_issued_over_useful"] = phys["mfma_tflops_issued"] / max(roof["achieved"], 1e-9)
    from hamiltorch_amd.ess import ess_min
    ess = ess_min(w.samples[1:]) if w.T >= 8 else float("nan")
    return {"key": W.key + ("-eig" if getattr(w, "jacobi", False) else ""), "workload": W.name, "value": units / dt,
            "unit": "leapfrog-steps/s", "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "dtype": W.dtype_name,
            "config": {"workload": W.name, "chains_per_gpu": w.C, "chains_total": w.C * world,
                       "trajectories_per_step": w.T, "leapfrog_steps_per_trajectory": W.L, "D": W.D,
                       "samples_stored": True, "parallelism": "chains sharded, %d per GPU, no collective" % w.C},
            "roofline": roof, "route": getattr(w, "route", ""), "acceptance_rate": acc, "ess_per_sec": ess / (call_ms * 1e-3)}


def api_timing(w, steps, warmup, reps=5):
    """The same work through hamiltorch_amd.sample(): (pipelined ms per call, synchronised ms per call).
    Pipelined = the headline's own bracket (W untimed calls, K timed calls, one synchronize on either side): what a
    program that keeps calling sample() sees.  Synchronised = median wall time of `reps` single calls each followed by
    a synchronize: the latency of one call (launch path + kernels + wake-up)."""
    for k in range(max(1, warmup)):
        w.api_call(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        out = w.api_call(warmup + k)
    torch.cuda.synchronize()
    pipelined = (time.perf_counter() - t0) * 1e3 / max(1, steps)
    ts = []
    for k in range(reps):
        t0 = time.perf_counter()
        out = w.api_call(1 + k)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    del out
    return pipelined, statistics.median(ts)


# ---------------------------------------------------------------------------------------------------
# the driver's line: the contract keys only, numbers rounded, no prose (the driver keeps an 8 KB stdout tail;
# round 2's 21 KB line could not be parsed).  The complete record goes to bench_detail.json and to an EARLIER line.
# ---------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096


def _r(x, sig=5):
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (sig, x))
    return x


def _short_key(rec):
    """cfg2 | cfg3@1024 | cfg3 | cfg3-eig | cfg4 | nbmlp ... from a full record's workload name."""
    if rec.get("key"):
        return rec["key"]
    name = rec.get("workload") or rec.get("config", {}).get("workload", "")
    key = name.split(":")[0].strip()
    if name.startswith("north-star RMHMC"):
        key = "cfg3@1024"
    if "eigendecomposition route" in name:
        key += "-eig"
    return key


def _compact_roofline(roof):
    phys = roof.get("physical") or {}
    out = {"bound": roof.get("bound"), "achieved": _r(roof.get("achieved")), "peak": roof.get("peak"), "unit": roof.get("unit"),
           "frac": _r(roof.get("frac"), 4), "traffic": _r(roof.get("traffic")),
           "kernel": str(roof.get("kernel", "")).split(" (")[0][:64],
           "kernel_ms": _r(roof.get("kernel_ms", roof.get("kernel_ms_per_step"))),
           "mfma_busy": _r(phys.get("mfma_busy_frac"), 3),
           "simds_occupied_frac": _r(roof.get("simds_occupied_frac", phys.get("simds_occupied_frac")), 3)}
    if roof.get("traffic") is not None:
        out["traffic_src"] = "profiles/physical.json"
    lat = roof.get("latency_model")
    if lat:
        out["frac_of_latency_floor"] = _r(lat.get("frac_of_latency_floor"), 3)
    if roof.get("mfma_issued_over_useful") is not None:
        out["mfma_issued_over_useful"] = _r(roof["mfma_issued_over_useful"], 3)
    return out


def _compact_cpu(cb):
    if not cb:
        return None
    return {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
            "sample": str(cb.get("sample", "")).split(" (")[0][:96], "pinned_to": str(cb.get("pinned_to", "")).split(" (")[0][:40],
            "host_cpu": str(cb.get("host_cpu", ""))[:48], "ess_per_sec": _r(cb.get("ess_per_sec"))}


def compact_line(full, detail_path="bench_detail.json"):
    """The ONE JSON line the driver parses: contract keys + roofline + cpu_baseline + a compact `secondary` list with the
    north-star RMHMC size first.  Always < LINE_LIMIT bytes (secondary entries are dropped from the tail if a future
    workload list would not fit; the complete record is in `detail`)."""
    cfg = full.get("config", {})
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _r(out["value"], 6), _r(out["ms_per_step"], 6)
    out["config"] = {"workload": _short_key(full), "chains_per_gpu": cfg.get("chains_per_gpu"), "chains_total": cfg.get("chains_total"),
                     "trajectories_per_step": cfg.get("trajectories_per_step"), "L": cfg.get("leapfrog_steps_per_trajectory"),
                     "D": cfg.get("D"), "parallelism": str(cfg.get("parallelism", ""))[:48]}
    out["roofline"] = _compact_roofline(full.get("roofline", {}))
    if full.get("cpu_baseline"):
        out["cpu_baseline"] = _compact_cpu(full["cpu_baseline"])
        out["speedup_vs_cpu_baseline"] = _r(full.get("speedup_vs_cpu_baseline"), 4)
    for k in ("api_ms_per_step", "api_value", "api_sync_ms", "acceptance_rate", "ess_per_sec", "ess_per_sec_vs_cpu_baseline", "gather_ms"):
        if full.get(k) is not None:
            out[k] = _r(full[k])
    for k in ("ranks_seen", "rank_devices", "launcher", "collective_backend"):
        if k in full and (full.get("n_gpus", 1) > 1 or k == "ranks_seen"):
            out[k] = full[k]
    sec = []
    for r in full.get("secondary", []) or []:
        if "error" in r:
            sec.append({"key": _short_key(r), "error": r["error"][:80]})
            continue
        roof, cb = _compact_roofline(r.get("roofline", {})), r.get("cpu_baseline") or {}
        # one entry per workload, short: bound / unit / achieved follow from `frac` (bound "mfma": frac x 157.3 TFLOP/s; "hbm":
        # frac x 8000 GB/s) and are spelled out in bench_detail.json
        e = {"key": _short_key(r), "chains": r.get("config", {}).get("chains_per_gpu"), "value": _r(r.get("value")),
             "ms_per_step": _r(r.get("ms_per_step")), "frac": roof["frac"], "bound": roof["bound"], "mfma_busy": roof["mfma_busy"],
             "traffic": roof["traffic"], "kernel": roof["kernel"][:32], "kernel_ms": roof["kernel_ms"],
             "cpu": {"value": _r(cb.get("value"), 4), "cores": cb.get("cores"), "kind": cb.get("kind")}}
        if cb.get("workers_failed"):
            e["cpu"]["failed"] = cb["workers_failed"]
        if roof.get("mfma_issued_over_useful") is not None:
            e["issued_over_useful"] = roof["mfma_issued_over_useful"]
        if r.get("api_ms_per_step") is not None:
            e["api_ms"] = _r(r["api_ms_per_step"], 4)
        for k in ("gather_ms", "n_gpus", "ranks_seen"):
            if r.get(k) is not None:
                e[k] = _r(r[k], 4)
        if r.get("extras"):
            short = {"graph_replay": "graph", "value_graphs_off": "graphs_off", "value_notebook_closure": "nb_closure",
                     "launches_per_step": "launches", "callback_evaluations_per_step": "cb_evals", "metric_evaluations_per_step": "metric_evals",
                     "predict_route": "predict", "predict_samples": "predict_S", "predict_ms_torch_path": "predict_ms_torch"}
            e["extras"] = {short.get(k, k): _r(v, 4) for k, v in r["extras"].items()
                           if not isinstance(v, (dict, list)) and v is not None and k != "predict_samples_per_s"}
        if r.get("published"):
            e["samples_per_s"] = _r(r.get("samples_per_s"), 4)
            e["published_sps"] = r["published"].get("samples_per_s")
            e["cpu"]["samples_per_s"] = _r(cb.get("samples_per_s"), 3)
        sec.append(e)
    if sec:
        out["secondary"] = sec
    out["detail"] = detail_path
    line = json.dumps(out, separators=(",", ":"))
    while len(line) >= LINE_LIMIT and out.get("secondary"):          # never print a line the driver cannot keep
        out["secondary"].pop()
        out["secondary_truncated"] = True
        line = json.dumps(out, separators=(",", ":"))
    return line


def emit(full):
    """Complete record -> bench_detail.json (+ gpurun_out/) and an earlier stdout line; the compact line LAST."""
    detail = json.dumps(full)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
        except OSError:
            pass
    sys.stdout.write("BENCH_DETAIL " + detail + "\n")
    sys.stdout.flush()
    print(compact_line(full), flush=True)


# ---------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _self_launch(n):
    """`python bench.py --gpus N` outside torch.distributed.run: re-exec as N ranks on this node, one per GPU."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        _self_launch(a.gpus)                                              # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or let bench.py launch the "
                         "ranks itself)" % (a.gpus, world, a.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one rank per GPU over RCCL ("nccl" on ROCm).  HTA_BENCH_BACKEND=gloo + several ranks on one GPU is only
        # for exercising this code path on a single-GPU box.
        backend = os.environ.get("HTA_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if backend == "nccl" and ndev < world:
            raise SystemExit("bench.py: %d ranks over RCCL need %d GPUs, this node shows %d" % (world, world, ndev))
        local = local % ndev
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ranks_seen, devices = 1, [[0, local]]
    if world > 1:
        one = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one)                                              # over RCCL: every rank is really there
        ranks_seen = int(one.item())
        ids = [None] * world
        dist.all_gather_object(ids, [rank, local])
        devices = ids
        assert ranks_seen == dist.get_world_size() == a.gpus, (ranks_seen, dist.get_world_size(), a.gpus)
    if a.workload not in WORKLOADS:
        try:
            import bench_extra
            WORKLOADS.update(bench_extra.WORKLOADS)
        except ImportError:
            pass
    from hamiltorch_amd import _abi
    for kv in filter(None, os.environ.get("HTA_TUNING", "").split(",")):      # e.g. HTA_TUNING=rmhmc_momwave=0 (A/B runs)
        key, val = kv.split("=")
        _abi.set_tuning(key, int(val))

    W = WORKLOADS[a.workload]
    w = W(dev, a.chains, a.traj, chain_offset=rank * (a.chains or W.chains))
    # HIP event pair around the dominant kernel, on its stream, live in the timed region: every launch for the workloads with
    # several launches per step; every 4th for cfg2, where a pair (two barrier packets) is 3 % of a 0.19 ms step
    dt, call_ms, prof_ms, prof_n = measure(w, a.steps, a.warmup, world, dist, dev, 4 if a.workload == "cfg2" else 1)
    res = result_of(w, W, dt, call_ms, prof_ms, prof_n, a.steps, a.warmup, world)
    gather_ms = None
    if hasattr(w, "gather"):            # the path's only collective, outside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tg = time.perf_counter()
        gathered = w.gather(world)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert gathered.shape[1] == w.C * world
        del gathered

    # --gpus N > 1 (the driver's scaling runs): the SECOND north-star target on every rank as well - D = 100 explicit RMHMC at
    # 1024 chains per GPU (cfg5) - and the path's only collective, the gather of its samples to rank 0 (after the timed region)
    res5 = None
    if world > 1 and a.workload == "cfg2" and not a.no_secondary and a.chains is None and a.traj is None:
        del w
        torch.cuda.empty_cache()
        w5 = Cfg5(dev, None, None, chain_offset=rank * Cfg5.chains)
        m5 = measure(w5, 10, 2, world, dist, dev, 1)
        res5 = result_of(w5, Cfg5, *m5, 10, 2, world)
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        gathered = w5.gather(world)
        torch.cuda.synchronize()
        dist.barrier()
        res5["gather_ms"] = (time.perf_counter() - tg) * 1e3
        res5["n_gpus"], res5["ranks_seen"] = world, ranks_seen
        if rank == 0:
            assert gathered.shape[1] == w5.C * world
        del gathered
        w = w5

    if rank == 0:
        out = {"key": res["key"], "metric": METRIC, "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": W.dtype_name, "data": "synthetic", "config": res["config"], "roofline": res["roofline"],
               "route": res["route"], "acceptance_rate": res["acceptance_rate"], "ess_per_sec": res["ess_per_sec"],
               "ranks_seen": ranks_seen, "rank_devices": devices,
               "launcher": "torch.distributed.run" if world > 1 else "single process",
               "collective_backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None}
        if gather_ms is not None:
            out["gather_ms"] = gather_ms
            out["config"]["parallelism"] += "; one gather of samples[%d, %d, %d] per rank to rank 0 after the timed region" % (
                w.T + 1, w.C, W.D)
        if hasattr(w, "api_call") and world == 1 and not a.no_api:
            api_ms, api_sync_ms = api_timing(w, a.steps, a.warmup)
            out["api_ms_per_step"] = api_ms
            out["api_value"] = w.units_per_step() / (api_ms * 1e-3)
            out["api_sync_ms"] = api_sync_ms
            out["api_note"] = "api_ms_per_step: the same K steps through hamiltorch_amd.sample() in the headline's bracket (route " \
                              "selection, sample-tensor allocation, the returned list); api_sync_ms: one call + synchronize, median of 5"
        if world == 1 and not a.no_cpu_baseline:
            cb = w.cpu_baseline(a.cpu_seconds)
            if cb is not None:
                cb.setdefault("host_cores_available", os.cpu_count())
                cb["host_cpu"] = _cpu_model()
                out["cpu_baseline"] = cb
                out["speedup_vs_cpu_baseline"] = res["value"] / cb["value"]          # against the `cores` the baseline used
                out["speedup_vs_cpu_baseline_1core"] = res["value"] / (cb["value"] / max(1, cb.get("cores", 1)))
                if cb.get("ess_per_sec"):                                              # the metric's second half: ESS/sec vs the CPU reference
                    out["ess_per_sec_vs_cpu_baseline"] = res["ess_per_sec"] / cb["ess_per_sec"]
        if res5 is not None:
            out["secondary"] = [res5]
        if world == 1 and a.workload == "cfg2" and not a.no_secondary and a.chains is None and a.traj is None:
            del w
            torch.cuda.empty_cache()
            out["secondary"] = secondary(dev, a)
        emit(out)
        if a.sweep and a.workload == "cfg2":
            for C in (1024, 4096, 16384, 65536, 262144, 1048576):
                ws = W(dev, C, max(10, min(1000, (1 << 24) // C)), 0)
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


def secondary(dev, a):
    """BASELINE.json's other targets on the same GPU, each a complete record: the D=100 explicit-RMHMC north-star size
    (1024 chains), config 3 (256 chains) on the default route and on the eigendecomposition route SURVEY 8(d)'s flop
    count describes, config 4 (512 chains)."""
    out = []
    plan = [(Cfg3N, {}, 10, 2, True), (Cfg3, {}, 10, 2, True), (Cfg3, {"jacobi": True, "traj": 20}, 10, 1, False), (Cfg4, {}, 10, 2, True),
            (NbMlp, {}, 10, 1, True), (NbMlpFull, {}, 10, 1, True), (FunnelHMC, {}, 10, 2, True), (FunnelRMHMC, {}, 3, 1, True)]
    cpu_cache = {}
    for W, kw, steps, warmup, want_cpu in plan:
        try:
            w = W(dev, None, kw.get("traj"), 0, **{k: v for k, v in kw.items() if k != "traj"})
            dt, call_ms, prof_ms, prof_n = measure(w, steps, warmup, 1, None, dev, 1)
            r = result_of(w, W, dt, call_ms, prof_ms, prof_n, steps, warmup, 1)
            if getattr(w, "jacobi", False):
                r["workload"] = w.name
                r["config"]["workload"] = w.name
            if not a.no_cpu_baseline:
                ck = "cfg3" if isinstance(w, Cfg3) else W.key
                if ck not in cpu_cache and want_cpu:
                    cpu_cache[ck] = w.cpu_baseline(a.cpu_seconds)
                    cpu_cache[ck]["host_cpu"] = _cpu_model()
                if ck in cpu_cache:
                    r["cpu_baseline"] = cpu_cache[ck]
                    r["speedup_vs_cpu_baseline_1core"] = r["value"] / (cpu_cache[ck]["value"] / max(1, cpu_cache[ck].get("cores", 1)))
            if getattr(W, "published", None):
                r["published"] = W.published
                r["samples_per_s"] = r["value"] / W.L
            if callable(getattr(w, "extras", None)):
                try:
                    r["extras"] = w.extras()
                    if r["extras"].get("launches_per_step") is not None:
                        r["roofline"]["launches_per_step"] = r["extras"]["launches_per_step"]
                except Exception as e:
                    r["extras"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            if hasattr(w, "api_call") and not a.no_api and not getattr(w, "jacobi", False):
                try:        # the same steps through the public API (sample / sample_split_model / sample_model)
                    r["api_ms_per_step"], r["api_sync_ms"] = api_timing(w, steps, warmup, reps=3)
                    r["api_route"] = w.abi.last_route()
                except Exception as e:
                    r["api_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
            out.append(r)
            del w
            torch.cuda.empty_cache()
        except Exception as e:          # a secondary line must never take the headline down with it
            out.append({"workload": W.name, "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
    return out


if __name__ == "__main__":
    main()

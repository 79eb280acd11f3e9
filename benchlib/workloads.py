"""bench.py's workloads: BASELINE.json's configs and the reference's published models, each with its step, its check and its
roofline (the arithmetic lives in benchlib/models.py)."""
import os
import time

import torch

from . import models
from .cpu import cpu_baseline_procs
from .models import FP32_PEAK_TFLOPS, HBM_PEAK_GBS, N_SIMDS

SIGMA3 = [[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]]


def _philox_init(abi, C, D, dev, chain_offset, scale=0.1):
    """params_init[c] = scale * N(0, I) from Philox(seed 0, subsequence = global chain id), drawn by the
    library's own generator (hta_momentum_resample with identity mass, draw 0)."""
    z = torch.empty(C, D, device=dev)
    abi.momentum_resample(z, abi.MASS_NONE, None, 0, chain_offset, 0)
    return (scale * z).contiguous()



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
        quad = self.C <= 65536
        waves = (self.C * 4 + 63) // 64 if quad else (self.C + 63) // 64
        if "fused" in getattr(self, "route", ""):      # round 4: the producers of the draw records run in the same launch
            waves += 4 * self.abi.get_tuning("quad_producers") * ((self.C + 1023) // 1024)
        clk_ghz = self.abi.device_info(0)["clock_khz"] / 1e6
        if quad and self.C * 4 <= 64 * models.N_SIMDS:
            # at most one wave per SIMD: the bound that physically applies is the serial chain of 2 L dependent FMAs per trajectory
            # (eigenbasis of P, one eigen-coordinate per lane of a DPP quad) - benchlib/models.py::cfg2_roofline
            roof = models.cfg2_roofline(self.C, self.T, self.L, self.D, kernel_ms, clk_ghz, waves=waves)
        else:                                          # saturating sizes: several waves per SIMD hide the latency, the VALU rate binds
            units, sec = self.units_per_step(), kernel_ms * 1e-3
            tf = models.hmc_gauss_flops_per_chain_step(self.D) * units / sec / 1e12
            roof = {"bound": "valu", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS,
                    "traffic": None, "kernel_ms": kernel_ms, "waves_per_launch": waves,
                    "simds_occupied_frac": min(1.0, waves / N_SIMDS),
                    "hbm_model_8d": {"achieved_gbs": self.bytes_per_unit() * units / sec / 1e9,
                                     "ratio_to_hbm_peak": self.bytes_per_unit() * units / sec / 1e9 / HBM_PEAK_GBS}}
        roof.update({"kernel": "hmc_gauss_quad_fused_kernel<3,25>" if quad else "hmc_gauss_eig_kernel<float,3,false>", "call_ms": call_ms,
                     "note": "bound = latency: the state is register-resident for the whole launch (HBM traffic = draw records + sample rows "
                             "only: `traffic`, `hbm_counter_frac`), 64 consumer waves on 1024 SIMDs; a trajectory is a serial chain of 2 L "
                             "dependent FMAs.  frac = that chain's floor (4 cycles per dependent FMA) / measured cycles per trajectory.  "
                             "hbm_model_8d = SURVEY 8(d)'s streaming convention, kept as a side field: its ratio exceeds 1 because those "
                             "bytes never move"})
        return roof

    def cpu_baseline(self, seconds):
        """The reference's CPU path on this host: see cpu_baseline_procs.  Median of three rounds."""
        return cpu_baseline_procs("cfg2", seconds)


class Cfg3:
    """BASELINE config 3: D=100 Gaussian, explicit RMHMC, soft-abs metric, 256 chains."""
    key = "cfg3"
    name = "cfg3: D=100 Gaussian explicit RMHMC, softabs alpha=1e6, omega=10, eps=0.1, L=10, jitter=1e-3"
    D, L, eps, chains, traj = 100, 10, 0.1, 256, 400
    omega, alpha, jitter = 10.0, 1e6, 1e-3
    dtype_name = "f32"

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

    K = 2          # refinement products per solve at jitter 1e-3 (rho^(K+1) <= eps / 4: csrc/rmhmc_fused.hip host plan)

    def useful_flops_per_unit(self):
        """benchlib/models.py: (4 K + 4 + (2 K + 5) / L) symmetric matrix-vector products per explicit step on the shared-inverse
        routes (no factorisation in the step: split momentum draw + series log-det); the refinement route's D^3 products on the
        eigendecomposition route."""
        if self.jacobi:
            return models.rmhmc_eig_useful_flops(self.D, self.L)
        return models.rmhmc_closed_form_useful_flops(self.D, self.L, self.K)

    def issued_model_flops_per_unit(self):
        """What the uvc kernels issue (every instruction carries 128 rows x 4 columns); None on routes without a model."""
        if self.jacobi or self.C > 1792:
            return None
        return models.rmhmc_closed_form_issued_flops(self.D, self.L, self.K, 1 if self.C <= 256 else 2)

    @property
    def roof_kernel(self):      # the trajectory kernel the library picks at this chain count (csrc/rmhmc_fused.hip dispatch)
        if self.jacobi:
            return ("metric_traj_mfma_kernel (one launch per trajectory: a chain's workgroup runs its 4 L + 3 metric evaluations - "
                    "eigenvector refinement: formation and second-order product on 3 x v_mfma_f32_16x16x32_bf16 of split operands, "
                    "state resident in LDS in eigen-coordinates - back to back, then its Metropolis selection)")
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
        """`frac` = USEFUL flops (what the algorithm needs at these sizes, no padding rows / idle columns: benchlib/models.py) /
        kernel time / the 157.3 TFLOP/s fp32 matrix peak.  `survey_8d` = the same time priced with SURVEY 8(d)'s count (an
        eigendecomposition per metric evaluation: the reference's work, which no route here mimics product for product) - a
        model speed-up, never a utilisation."""
        units = self.units_per_step()
        sec = kernel_ms * 1e-3
        useful = self.useful_flops_per_unit()
        tf = useful * units / sec / 1e12
        tf_survey = models.rmhmc_survey_flops_per_chain_step(self.D) * units / sec / 1e12
        extra = {}
        if self.jacobi:      # round 6: F E1 runs as three bfloat16 products - the fp32-equivalent fraction is not a pipe utilisation any more
            pt = models.rmhmc_eig_pipe_time_flops(self.D, self.L) * units / sec / 1e12
            extra = {"pipe_time_frac": pt / FP32_PEAK_TFLOPS,
                     "pipe_time_note": "useful work priced in fp32-matrix-pipe time (both D^3 products of an evaluation run as 3 x their flops in bfloat16 at "
                                       "16 x the rate): the figure the pipe-busy counter bounds; `frac` prices every useful flop at the fp32 peak"}
        return {"bound": "mfma", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS,
                "traffic": None, "kernel": self.roof_kernel, "kernel_ms_per_step": kernel_ms, "call_ms": call_ms,
                "launches_per_step": prof_n / max(1, steps), **extra,
                "useful_flops_per_chain_step": useful, "issued_model_flops_per_chain_step": self.issued_model_flops_per_unit(),
                "survey_8d": {"flops_per_chain_step": models.rmhmc_survey_flops_per_chain_step(self.D), "achieved": tf_survey,
                              "ratio_to_peak": tf_survey / FP32_PEAK_TFLOPS,
                              "note": "SURVEY 8(d) counts an eigendecomposition per metric evaluation; a ratio > 1 means the timed "
                                      "route does not do that work (closed form for constant curvature / refinement of a shared basis)"},
                "note": "frac = useful flops (%s) / kernel time / fp32 peak 157.3 TF; kernel time = every profiled launch of a step; "
                        "issued (PMC) and padding = issued / useful are added from profiles/physical.json" % (
                            "refinement route: (4 L + 3) x 3 D^3 per trajectory" if self.jacobi else
                            "(4 K + 4 + (2 K + 5) / L) x 2 D^2 per step, K = 2")}

    def burn_in_states(self, k):
        """The first k chains where the timed steps left them (thousands of trajectories in): the CPU baseline's chains start from these
        posterior draws, so their ESS measures within-chain mixing from the first draw on (benchlib/cpu.py)."""
        torch.cuda.synchronize()
        return self.cur[:k].detach().cpu().clone()

    def cpu_baseline(self, seconds, init_file=None):
        """Reference cost structure: every dH/dtheta, dH/dp is an autograd pass through hessian + eigh (S:395-422); one chain
        per usable host core.  ~1 trajectory per second and core: six times the common budget (three rounds of 18 s at the default) so
        that a chain has more than a handful of draws behind its ESS."""
        return cpu_baseline_procs("cfg3", 6 * seconds, init_file=init_file)


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
        return models.mlp_split_flops_per_chain_step(4, self.L, 100, 900)

    def reference_flops_per_unit(self):
        return models.mlp_split_flops_per_chain_step(4, self.L, 100, 900, reference=True)

    def bytes_per_unit(self):
        return 16 * self.D


    #: untimed steps the chains are advanced by before ESS is measured and before the CPU baseline's workers are started FROM THE
    #: DEVICE'S STATES (every chain of both sides used to start at one point - the ESS of 16 short CPU chains was the
    #: spread they had not yet acquired).  `burn_in_states(k)` returns the first k chains' states after that run.
    burn_in_trajectories = 2000

    def burn_in_states(self, k):
        n = -(-self.burn_in_trajectories // self.T)
        for i in range(n):
            self.step(200000 + i)
        torch.cuda.synchronize()
        return self.cur[:k].detach().cpu().clone()

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
                "launches_per_step": prof_n / max(1, steps), "useful_flops_per_chain_step": self.flops_per_unit(),
                "reference_flops_per_chain_step": self.reference_flops_per_unit(),
                "frac_by_reference_flops": self.reference_flops_per_unit() * self.units_per_step() / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                "note": "2M x 6 flop per (point, weight) per split step (SURVEY 8d) against the fp32 matrix peak"}

    def cpu_baseline(self, seconds, init_file=None):
        """Reference cost structure: per-split closure + autograd gradient for every half kick (S:499-540); one chain per core."""
        return cpu_baseline_procs("cfg4", seconds, init_file=init_file)


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
    ess_extra_steps = 16           # one trajectory per step: ESS from 16 consecutive untimed steps (benchlib/measure.py)
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
        return models.mlp_split_flops_per_chain_step(self.M, self.L, self.Nb, 100 + 100 * 100 + 100)

    def reference_flops_per_unit(self):      # 2M * 6 * N_b * P_w
        return models.mlp_split_flops_per_chain_step(self.M, self.L, self.Nb, 100 + 100 * 100 + 100, reference=True)

    def bytes_per_unit(self):
        return 16 * self.D


    #: untimed steps the chains are advanced by before ESS is measured and before the CPU baseline's workers are started FROM THE
    #: DEVICE'S STATES (every chain of both sides used to start at one point - the ESS of 16 short CPU chains was the
    #: spread they had not yet acquired).  `burn_in_states(k)` returns the first k chains' states after that run.
    burn_in_trajectories = 300

    def burn_in_states(self, k):
        n = -(-self.burn_in_trajectories // self.T)
        for i in range(n):
            self.step(200000 + i)
        torch.cuda.synchronize()
        return self.cur[:k].detach().cpu().clone()

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
                "launches_per_step": prof_n / max(1, steps), "useful_flops_per_chain_step": self.flops_per_unit(),
                "reference_flops_per_chain_step": self.reference_flops_per_unit(),
                "frac_by_reference_flops": self.reference_flops_per_unit() * self.units_per_step() / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}

    def cpu_baseline(self, seconds, init_file=None):
        """Reference cost structure: functional model + autograd per half kick (S:499-540) on the notebook's module; one chain per core."""
        return cpu_baseline_procs(self.key, seconds, init_file=init_file)

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
        return models.mlp_full_flops_per_chain_step(self.L, self.Nb, 100 + 100 * 100 + 100)

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
    """The callback contract on the driver's line: the reference's published 11-D funnel run (notebook cell 24: HMC, eps = 0.2, L = 25:
    56.10 samples/s, one chain) at 1024 chains through hamiltorch_amd.sample() with an OPAQUE closure.  Round 6: the callback compiler
    (hamiltorch_amd/jit/) traces the closure, differentiates and emits it, and hipRTC builds it into the fused trajectory kernel
    (csrc/jit/hmc_callback.hip.in) - one launch per sample() call; rounds 3-5 evaluated the closure with torch (vmap(grad_and_value),
    ~38 launches per leapfrog step) between the pieces kernels: `extras.value_callback_path` is that route today (HAMILTORCH_AMD_JIT=0)."""
    key = "funnel-hmc"
    name = "funnel-hmc: 11-D funnel (notebook cell 22-24), HMC eps=0.2 L=25, opaque log_prob_func closure -> compiled callback kernel"
    D, L, eps, chains, traj = 11, 25, 0.2, 1024, 200
    dtype_name = "f32"
    ess_extra_steps = 5            # ESS from 5 consecutive untimed steps continuing the chains (1000 draws each) instead of one step from the common start
    published = {"samples_per_s": 56.10, "hw": "notebook host, 1 chain", "src": "log_prob_examples nb cell 24 (JSON lines 401-402)"}
    sampler_kw = {}
    kernel_name = "hta_cb_hmc_kernel"

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
        self._start = self.theta0

    def units_per_step(self):
        return self.C * self.T * self.L

    def bytes_per_unit(self):
        return 16 * self.D

    def _sample(self, fn, k, T, start=None):
        import hamiltorch_amd as ht
        return ht.sample(fn, self.theta0 if start is None else start, num_samples=T, num_steps_per_sample=self.L, step_size=self.eps,
                         burn=-1, debug=2, verbose=False, seed=self.seed + k, chain_offset=self.off, **self.sampler_kw)

    def step(self, k):
        # steps >= 100000 are the untimed ESS extension (benchlib/measure.py): the chains CONTINUE from where the previous step ended
        from hamiltorch_amd.samplelist import as_tensor
        cont = k >= 100000 and self.samples is not None
        out, acc = self._sample(self.fn, k, self.T, self.samples[-1].contiguous() if cont else None)
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
            evs = [e for e in prof.key_averages() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
            self._hta_launches = sum(e.count for e in evs if "hta" in str(e.key))
            return sum(e.count for e in evs)
        except Exception:
            return None

    def _graph_flops(self):
        """Useful flops of one leapfrog step from the compiled graph itself: the scalar operations of value + gradient (a transcendental
        counted as one) + 4 D for the kick and the drift."""
        try:
            from hamiltorch_amd import jit
            comp = jit.compile_hmc(self.fn, self.theta0[0], self.theta0.dtype, 0)
            g = comp.traced.graph
            live = g.reachable([comp.traced.value] + comp.traced.grad())
            return sum(1 for i in live if g.nodes[i][0] not in ("const", "bconst", "in", "detach")) + 4 * self.D
        except Exception:
            return None

    def extras(self):
        """The torch-evaluated callback path the compiled kernel replaces, the notebook's verbatim closure (torch.distributions: compiled as
        well), a saturating chain count, the one-off trace / compile cost, launches per sample() call - each on a short run, not part of `value`."""
        from hamiltorch_amd import jit
        T = max(4, self.T // 10)
        out = {"value_callback_path": self._rate(self.fn, T, env={"HAMILTORCH_AMD_JIT": "0"}),
               "value_notebook_closure": self._rate(funnel_ll_notebook, self.T),
               "launches_per_step": self._launches(),
               "callback_evaluations_per_step": self.T * (self.L + 1),
               "trace_compile_s": round(jit.runtime.stats["compile_seconds"], 3), "traces": jit.stats["traced"]}
        out["hta_launches_per_step"] = getattr(self, "_hta_launches", None)
        try:
            big = type(self)(self.dev, 65536, max(4, self.T // 10), 0)
            out["value_65536"] = big._rate(big.fn, big.T)
            del big
        except Exception as e:
            out["value_65536_error"] = str(e)[:80]
        return out

    def roofline(self, kernel_ms, call_ms, prof_n, steps):
        """The kernel is one chain per lane, straight-line VALU code: at 1024 chains 16 waves hold 16 of the 1024 SIMDs, so the fraction of
        the vector peak is small by construction (`simds_occupied_frac`); what binds a lone wave is its instruction issue (~4.5 cycles per
        instruction).  frac = useful flops (the compiled graph's own operations + the 4 D of the leapfrog) / kernel time / fp32 vector peak."""
        sec = kernel_ms * 1e-3
        flops = self._graph_flops()
        tf = (flops or 0) * self.units_per_step() / sec / 1e12
        waves = (self.C + 63) // 64
        clk_ghz = self.abi.device_info(0)["clock_khz"] / 1e6
        return {"bound": "valu", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS, "traffic": None,
                "kernel": self.kernel_name, "kernel_ms_per_step": kernel_ms, "call_ms": call_ms, "launches_per_step": prof_n / max(1, steps),
                "useful_flops_per_chain_step": flops, "simds_occupied_frac": min(1.0, waves / N_SIMDS),
                "cycles_per_leapfrog_step": sec * clk_ghz * 1e9 / (self.T * self.L),
                "hbm_model_8d": {"ratio_to_hbm_peak": self.units_per_step() * self.bytes_per_unit() / sec / 1e9 / HBM_PEAK_GBS},
                "note": "chain per lane, state in registers for the launch; one launch per sample() call; kernel time from HIP events inside "
                        "the library; the wall time of `value` adds the host side of sample() (trace-cache lookup, the run-time check of the "
                        "compiled code against the callable, the sample tensor)"}

    def burn_in_states(self, k):
        torch.cuda.synchronize()
        return self.samples[-1][:k].detach().cpu().clone()

    def cpu_baseline(self, seconds, init_file=None):
        return cpu_baseline_procs(self.key, (2 if self.key == "funnel-rmhmc" else 1) * seconds, init_file=init_file)


class FunnelRMHMC(FunnelHMC):
    """SURVEY 8(f) N1 on the driver's line: explicit RMHMC with the soft-abs metric on the same funnel (notebook cell 30: eps = 0.14, L = 25,
    omega = 10, jitter = 1e-3; the reference's progress bar shows < 1 sample/s and its run ends in NaN after 14 samples).  Round 6: the
    compiled callable (derivatives up to the third order from its traced graph) inside the chain-per-lane trajectory kernel
    csrc/jit/rmhmc_callback.hip.in - Hessian, jitter, Jacobi eigendecomposition, soft-abs map, solves and the derivative of the metric in
    registers, one launch per sample() call; `extras.value_launch_sequence` is the round-5 route (torch.func derivatives + hta_metric_eval)."""
    key = "funnel-rmhmc"
    name = "funnel-rmhmc: 11-D funnel, explicit RMHMC softabs alpha=1e6 omega=10 eps=0.14 L=25 jitter=1e-3, opaque closure -> compiled trajectory kernel"
    D, L, eps, chains, traj = 11, 25, 0.14, 1024, 2
    ess_extra_steps = 60           # 2 trajectories per step: ESS from 60 consecutive untimed steps = 120 draws per chain (benchlib/measure.py)
    published = {"samples_per_s": 0.19, "hw": "notebook host, 1 chain", "src": "log_prob_examples nb cell 30 (JSON lines 637-638)"}
    kernel_name = "hta_cb_rmhmc_kernel"

    def __init__(self, dev, chains, traj, chain_offset, seed=1):
        super().__init__(dev, chains, traj, chain_offset, seed)
        import hamiltorch_amd as ht
        self.sampler_kw = dict(sampler=ht.Sampler.RMHMC, integrator=ht.Integrator.EXPLICIT, metric=ht.Metric.SOFTABS, softabs_const=1e6,
                               explicit_binding_const=10.0, jitter=1e-3)

    def extras(self):
        small = type(self)(self.dev, 256, 1, 0)
        return {"value_launch_sequence_256": small._rate(small.fn, 1, reps=1, env={"HAMILTORCH_AMD_JIT": "0"}),
                "metric_evaluations_per_step": self.T * (8 * self.L + 3), "launches_per_step": self._launches()}

    def _graph_flops(self):
        """Per explicit step: 8 metric evaluations, each ~ 2 D^3 (basis change) + ~2 x 6 D^2 x D(D-1)/2 / ... - counted simply as the
        reference-faithful 8 x 11.3 D^3 of SURVEY 8(d) + the third-derivative contraction 4 x 2 D^2 (the funnel's is sparse)."""
        return 8 * 11.3 * self.D ** 3 + 4 * 2 * self.D ** 2

    def roofline(self, kernel_ms, call_ms, prof_n, steps):
        r = super().roofline(kernel_ms, call_ms, prof_n, steps)
        r["note"] = ("SURVEY 8(d)'s count at D = 11 (8 metric evaluations per step x 11.3 D^3) / kernel time / fp32 vector peak; chain per "
                     "lane, a lone wave per SIMD at 1024 chains: instruction issue binds (profiles/r06b_rmhmc_callback_pmc.json: VALU active "
                     "69 % of the wave's cycles, 29 k VALU instructions per metric evaluation)")
        return r


WORKLOADS = {"funnel-hmc": FunnelHMC, "funnel-rmhmc": FunnelRMHMC, "nbmlp": NbMlp, "nbmlp-full": NbMlpFull, "cfg2": Cfg2, "cfg3": Cfg3, "cfg4": Cfg4, "cfg5": Cfg5, "cfg3@1024": Cfg3N}


"""Roofline models of bench.py's workloads: pure arithmetic, no device, unit-tested (tests/test_bench_models.py).

Every model returns USEFUL work - the arithmetic the algorithm needs at the sizes given, with no padding rows, idle instruction
columns or duplicated evaluations - so that `useful <= issued` (the matrix-instruction flops the PMC pass counts) must hold
for every workload, and every `frac` = useful / (kernel time x peak) is <= 1.  `check_record` asserts exactly that on a
bench record; the test runs it on the committed lines under profiles/.

Peaks: /opt/skills/guides/MI355X_MICROARCH.md (HBM3E 8 TB/s; fp32 matrix = fp32 vector = 157.3 TFLOP/s dense)."""

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E
FP32_PEAK_TFLOPS = 157.3       # fp32 vector == fp32 MFMA peak
N_SIMDS = 1024                 # 256 CUs x 4
MFMA_4X4X1_FLOPS = 512         # v_mfma_f32_4x4x1_16b_f32: 16 blocks x 4 x 4 x 1 x 2; also the unit of SQ_INSTS_VALU_MFMA_MOPS_F32
DEP_FMA_CYCLES = 4             # issue-to-issue latency of two DEPENDENT fp32 VALU instructions of one wave64 (one pass of 16 lanes x 4)


# ---------------------------------------------------------------------------------------------------
# cfg2: Gaussian HMC, register-resident trajectory kernels
# ---------------------------------------------------------------------------------------------------
def hmc_bytes_per_chain_step(D):
    """SURVEY 8(d)'s streaming convention: theta and p (fp32) read + written once per leapfrog step."""
    return 16 * D


def hmc_gauss_flops_per_chain_step(D):
    """Eigenbasis form of S:281-302 for a Gaussian target: per coordinate one drift FMA and one kick FMA per step (4 flop),
    plus the two energies, the momentum draw and the Metropolis test amortised over a trajectory (not counted: a floor)."""
    return 4 * D


def hmc_latency_floor_cycles(L):
    """The serial chain of one trajectory in the eigenbasis: L x (drift FMA -> kick FMA), each waiting for the one before."""
    return 2 * L * DEP_FMA_CYCLES


def cfg2_roofline(C, T, L, D, kernel_ms, clock_ghz, traffic_bytes=None, waves=None):
    """The headline's roofline object.  What binds 1024 chains x D = 3 is neither HBM nor the VALU rate: the state lives in
    registers for the launch and a wave's 16 chains advance through 2 L DEPENDENT FMAs per trajectory.  `bound: latency`,
    `frac` = floor cycles / measured cycles per trajectory (<= 1 by construction).  The SURVEY 8(d) streaming figure (which
    exceeds the HBM peak because those bytes never move), the counter HBM fraction and the VALU fraction are side fields."""
    units = C * T * L
    sec = kernel_ms * 1e-3
    cyc = sec * clock_ghz * 1e9 / T
    floor = hmc_latency_floor_cycles(L)
    alg_bytes = hmc_bytes_per_chain_step(D) * units
    out = {"bound": "latency", "achieved": 2 * L / cyc, "peak": 1.0 / DEP_FMA_CYCLES, "unit": "dependent-FMA/cycle/wave",
           "frac": floor / cyc, "traffic": traffic_bytes,
           "latency_model": {"dependent_fma_per_trajectory": 2 * L, "floor_cycles_per_trajectory": floor,
                             "measured_cycles_per_trajectory": cyc, "frac_of_latency_floor": floor / cyc, "clock_ghz": clock_ghz},
           "hbm_model_8d": {"algorithmic_bytes_per_launch": alg_bytes, "achieved_gbs": alg_bytes / sec / 1e9,
                            "ratio_to_hbm_peak": alg_bytes / sec / 1e9 / HBM_PEAK_GBS,
                            "note": "SURVEY 8(d) convention (16 D bytes per chain-step); these bytes stay in registers: not a utilisation"},
           "valu_frac": hmc_gauss_flops_per_chain_step(D) * units / sec / 1e12 / FP32_PEAK_TFLOPS,
           "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes}
    if traffic_bytes is not None:
        out["hbm_counter_frac"] = traffic_bytes / sec / 1e9 / HBM_PEAK_GBS
    if waves is not None:
        out["waves_per_launch"] = waves
        out["simds_occupied_frac"] = min(1.0, waves / N_SIMDS)
    return out


# ---------------------------------------------------------------------------------------------------
# cfg3: explicit RMHMC, Gaussian target
# ---------------------------------------------------------------------------------------------------
SURVEY_RMHMC_FLOPS_PER_EVAL = 11.3      # x D^3: eigh 9 D^3 + assembly 2 D^3 + Cholesky D^3 / 3


def rmhmc_survey_flops_per_chain_step(D):
    """the reference's 4 distinct metric evaluations per explicit step, an eigendecomposition each."""
    return 4 * SURVEY_RMHMC_FLOPS_PER_EVAL * D ** 3


def rmhmc_closed_form_products(L, K=2):
    """Symmetric matrix-vector products per explicit step on the shared-inverse routes (csrc/rmhmc_uvc.hip, rmhmc_uv.hip;
    S:425-461 with x = (P + E)^-1 g by K refinements from S g):
      per step: 4 solves x K refinement products + the 4 products after the rotation (P d and S g for both state sets) = 4 K + 4;
      per trajectory: ONE closing Hamiltonian pair (S:989 for this trajectory in the U columns, S:971's momentum terms of the next
      one in the V columns): P d, S g, (S.S) e, K refinement products for U; the same without P d for V = 2 K + 5 = 9 at K = 2.
    No factorisation anywhere: the momentum is drawn as chol(P) z1 + sqrt(jitter u) z2 (chol(P) once per target) and the
    log-determinant is the series log|P| + tr(SE) - tr((SE)^2)/2."""
    return 4 * K + 4 + (2 * K + 5) / float(L)


def rmhmc_closed_form_useful_flops(D, L, K=2):
    return rmhmc_closed_form_products(L, K) * 2 * D * D


def rmhmc_closed_form_issued_flops(D, L, K, chains_per_group, deferred=False):
    """Matrix-instruction flops the uvc kernels ISSUE per chain-step (what SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 counts): four
    waves x ceil(D / 2 / 4) * 4 instructions per single product phase (rows padded to 128, contraction to a multiple of 8),
    every instruction carrying all four columns whether a chain fills them or not."""
    kj = 4 * ((D + 7) // 8)                       # instructions of one product per wave (XKJ = 52 at D = 100)
    if chains_per_group == 1:                     # rmhmc_uvc_kernel: 3 phases per step (1 + 2 + 1 products), flush + Hamiltonian (2 + 1 + 1) per trajectory
        per_step, per_traj = 4 * kj, 5 * kj
    elif deferred:                                # rmhmc_uvc2d_kernel (round 5, K = 2): 3 phases per step of 2 + 3 + 2 product chains; flush + Hamiltonian (2 + 1 + 1)
        per_step, per_traj = 7 * kj, 5 * kj
    else:                                         # rmhmc_uvc2_kernel: 2 K solve phases + one double phase per step; Hamiltonian 2 + 1 + K
        per_step, per_traj = (2 * K + 2) * kj, (3 + K) * kj
    return 4 * (per_step + per_traj / float(L)) * MFMA_4X4X1_FLOPS / chains_per_group


def rmhmc_eig_useful_flops(D, L, sqrtdraw=True):
    """The eigendecomposition route as it runs since round 4 (csrc/rmhmc_metric_mfma.hip: refinement of a shared eigenbasis): per
    solve evaluation the formation A = V0^T diag(e) V0 (symmetric: D^3) and the closed-form second pass F E1 (2 D^3); 4 L + 3
    evaluations per trajectory.  Round 6: the momentum draw is solve-shaped as well (p = G^(1/2) z, "metric_sqrtdraw"); with
    sqrtdraw=False it adds Q = V0 X (2 D^3), G = Q diag Q^T (symmetric: D^3) and a Cholesky (D^3 / 3), as rounds 4-5 ran it.
    The vector phases (matrix-vector products, soft-abs, solves: O(D^2)) are not counted."""
    return ((4 * L + 3) * 3 + (0.0 if sqrtdraw else 3 + 1.0 / 3)) * D ** 3 / float(L)


BF16_OVER_FP32_MATRIX_PEAK = 16.0       # MI355X_MICROARCH.md: dense bf16 2.5 PFLOP/s = 16 x the fp32 matrix peak


def rmhmc_eig_pipe_time_flops(D, L, bx3=2):
    """The same work priced in fp32-MATRIX-PIPE TIME (round 6): products that run as three bfloat16 products of split operands take
    3 / 16 of their fp32 time (bf16 at 16 x the fp32 matrix rate) - "metric_bx3" = 1: the second pass's F E1 (2 D^3 -> 0.375 D^3 fp32-pipe
    equivalents) beside the formation's D^3 in fp32; 2 (default): the formation as well (D^3 -> 0.1875 D^3); 0 / False: everything in
    fp32 (= the useful flops).  All 4 L + 3 evaluations of a trajectory are solve-shaped (the draw: "metric_sqrtdraw").  useful / fp32
    peak no longer bounds by the pipe-busy share; this figure / fp32 peak does."""
    r = 3.0 / BF16_OVER_FP32_MATRIX_PEAK
    form = r if int(bx3) >= 2 else 1.0
    second = 2.0 * (r if int(bx3) >= 1 else 1.0)
    return (4 * L + 3) * (form + second) * D ** 3 / float(L)


# ---------------------------------------------------------------------------------------------------
# Bayesian MLPs (cfg4, nbmlp)
# ---------------------------------------------------------------------------------------------------
def mlp_split_flops_per_chain_step(M, L, n_batch, n_weights, reference=False):
    """6 flop per (point, weight) and gradient evaluation (forward 2, backward 4).  The reference differentiates 2 M times per split
    step (S:499-540), twice at the same point at the turning point and at the step boundary; the kernels evaluate those once:
    2 M - 2 + 1 / L (csrc/mlp.hpp: split_stage_reuses, same results)."""
    evals = 2 * M if reference else 2 * M - 2 + 1.0 / L
    return evals * 6 * n_batch * n_weights


def mlp_full_flops_per_chain_step(L, n_points, n_weights):
    """Plain leapfrog (S:281-302): one gradient over all points per step + the first half kick's, amortised."""
    return 6 * n_points * n_weights * (L + 1) / float(L)


# ---------------------------------------------------------------------------------------------------
# the invariants
# ---------------------------------------------------------------------------------------------------
def roofline_problems(key, roof, tol=1e-9):
    """Violations of the physical invariants in one roofline object (list of strings, empty = fine):
    frac <= 1; frac = achieved / peak; useful flops <= matrix flops issued (counter) and <= the issue model."""
    bad = []
    frac, ach, peak = roof.get("frac"), roof.get("achieved"), roof.get("peak")
    if frac is None or not (0 <= frac <= 1 + tol):
        bad.append("%s: frac %r not in [0, 1]" % (key, frac))
    if None not in (frac, ach, peak) and peak and abs(ach / peak - frac) > 2e-3 * max(frac, 1e-12) + 1e-12:
        bad.append("%s: frac %r != achieved / peak = %r" % (key, frac, ach / peak))
    useful, issued = roof.get("useful_flops_per_chain_step"), roof.get("issued_flops_per_chain_step")
    if useful is not None and issued is not None and useful > issued * (1 + 1e-6):
        bad.append("%s: useful %.4g flop per chain-step > issued %.4g" % (key, useful, issued))
    model = roof.get("issued_model_flops_per_chain_step")
    if useful is not None and model is not None and useful > model * (1 + 1e-6):
        bad.append("%s: useful %.4g flop per chain-step > issue model %.4g" % (key, useful, model))
    pad = roof.get("padding")
    if pad is not None and pad < 1 - 1e-6:
        bad.append("%s: padding (issued / useful) %.3f < 1" % (key, pad))
    for side in ("hbm_counter_frac", "valu_frac"):
        if roof.get(side) is not None and roof[side] > 1 + tol:
            bad.append("%s: %s %r > 1" % (key, side, roof[side]))
    return bad


def check_record(rec):
    """All violations in a complete bench record or a compact line (primary + secondaries)."""
    bad = roofline_problems(rec.get("key") or rec.get("config", {}).get("workload", "primary"), rec.get("roofline", {}))
    for s in rec.get("secondary", []) or []:
        if "error" in s:
            continue
        roof = s.get("roofline")
        if roof is None:                          # compact secondary entry: the roofline keys are inlined
            roof = {"frac": s.get("frac"), "padding": s.get("padding")}
        bad += roofline_problems(s.get("key", "?"), roof)
    return bad

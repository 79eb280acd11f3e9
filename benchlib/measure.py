"""One measurement of a workload (the bench.py contract's timed region) and its record."""
import json
import os
import statistics
import time

import torch

from . import models
from .workloads import Cfg2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ESS_DIMS = 128          # = oracle/cpu_baseline.py::ESS_DIMS

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
        if roof.get("bound") == "latency" and roof["traffic"]:             # cfg2: the counter traffic against the HBM rate
            roof["hbm_counter_frac"] = roof["traffic"] / (kernel_ms * 1e-3) / 1e9 / models.HBM_PEAK_GBS
    if phys is not None and roof.get("unit") == "TFLOP/s" and phys.get("mfma_tflops_issued") and roof.get("useful_flops_per_chain_step"):
        # matrix-instruction flops ISSUED per chain-step (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 of the dominant kernel's launches, from the
        # committed counter pass) over the USEFUL flops of the roofline: padding rows / idle columns of the instruction, duplicated
        # products.  issued >= useful is asserted by tests/test_bench_models.py on every committed line
        dom = phys["kernels"][phys["dominant_kernel"]]
        chain_steps = w.C * (phys.get("trajectories_per_step") or w.T) * W.L
        issued = phys["mfma_tflops_issued"] * 1e12 * dom["ms_per_step"] * 1e-3 / chain_steps
        if phys.get("mfma_bf16_tflops_issued"):      # round 6 (cfg3-eig): a product run as THREE bfloat16 products counts as one fp32 product's flops
            bf = phys["mfma_bf16_tflops_issued"] * 1e12 * dom["ms_per_step"] * 1e-3 / chain_steps
            roof["issued_bf16_flops_per_chain_step"] = bf
            issued += bf / 3.0
        roof["issued_flops_per_chain_step"] = issued
        roof["padding"] = issued / roof["useful_flops_per_chain_step"]
        roof["mfma_issued_over_useful"] = roof["padding"]
    from hamiltorch_amd.ess import ess_min, rhat_max
    # ESS over the first ESS_DIMS coordinates - the ones the CPU baseline's samples travel with (oracle/cpu_baseline.py), so that
    # both sides of `ess_per_sec_vs_cpu_baseline` are the same estimator on the same coordinates
    ess_seconds = call_ms * 1e-3
    ess_draws = rhat = None
    if w.T >= 8 and not getattr(W, "ess_extra_steps", 0):
        used = w.samples[1:, :, :ESS_DIMS]
        ess, ess_draws, rhat = ess_min(used), [int(used.shape[0]), int(used.shape[1])], rhat_max(used)
    elif getattr(W, "ess_extra_steps", 0) and w.samples is not None and os.environ.get("HTA_BENCH_ESS_EXTRA", "1") != "0":
        # workloads whose step is one or two trajectories (the published-model runs, funnel-rmhmc): an UNTIMED run of consecutive steps
        # after the measurement - the chain state travels from step to step - supplies the draws; ESS / s = ESS of those draws / (their
        # steps x the measured time per step)
        draws = []
        for k in range(W.ess_extra_steps):
            w.step(100000 + k)
            draws.append(w.samples[1:, :, :ESS_DIMS].clone())
        torch.cuda.synchronize()
        used = torch.cat(draws)
        ess, ess_draws, rhat = ess_min(used), [int(used.shape[0]), int(used.shape[1])], rhat_max(used)
        ess_seconds = W.ess_extra_steps * call_ms * 1e-3
    else:
        ess = float("nan")
    return {"key": W.key + ("-eig" if getattr(w, "jacobi", False) else ""), "workload": W.name, "value": units / dt,
            "unit": "leapfrog-steps/s", "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "dtype": W.dtype_name,
            "config": {"workload": W.name, "chains_per_gpu": w.C, "chains_total": w.C * world,
                       "trajectories_per_step": w.T, "leapfrog_steps_per_trajectory": W.L, "D": W.D,
                       "samples_stored": True, "parallelism": "chains sharded, %d per GPU, no collective" % w.C},
            "roofline": roof, "route": getattr(w, "route", ""), "acceptance_rate": acc, "ess_per_sec": ess / ess_seconds,
            "ess_dims": min(W.D, ESS_DIMS), "ess_draws": ess_draws, "rhat": rhat}


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


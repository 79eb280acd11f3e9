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

The parts live in benchlib/: models.py (the roofline arithmetic, unit-tested: tests/test_bench_models.py), workloads.py, cpu.py (the CPU
baseline leg), measure.py (the timed region), line.py (the compact record).
"""
import argparse
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib import line as _line                                                   # noqa: E402
from benchlib.cpu import _cpu_model                                                  # noqa: E402
from benchlib.line import LINE_LIMIT, compact_line                                   # noqa: E402,F401  (tests import them from here)
from benchlib.measure import api_timing, measure, result_of                          # noqa: E402
from benchlib.workloads import (WORKLOADS, Cfg3, Cfg3N, Cfg4, Cfg5, FunnelHMC,       # noqa: E402
                                FunnelRMHMC, NbMlp, NbMlpFull)

METRIC = "leapfrog-steps/sec (whole node) at 1024 chains; ESS/sec vs CPU ref"
HBM_PEAK_GBS = 8000.0


def emit(full):
    """Complete record -> bench_detail.json (+ gpurun_out/) and an earlier stdout line; the compact line LAST."""
    return _line.emit(full, ROOT)


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
    res5 = res5s = None
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
        # "also report 1/2/4 GPUs at 8192 total" - the STRONG-scaling form next to the weak one: 8192 chains in all,
        # 8192 / N per GPU (N = 8: the same 1024 per GPU as the weak line), so the driver's N = 1, 2, 4, 8 runs yield both curves
        if 8192 % world == 0:
            del w5
            torch.cuda.empty_cache()
            per = 8192 // world
            w5s = Cfg5(dev, per, 20 if per > 2048 else None, chain_offset=rank * per)
            m5s = measure(w5s, 5, 1, world, dist, dev, 1)
            res5s = result_of(w5s, Cfg5, *m5s, 5, 1, world)
            res5s["key"], res5s["scaling"] = "cfg5-strong", "strong"
            res5s["n_gpus"], res5s["ranks_seen"] = world, ranks_seen
            w = w5s

    if rank == 0:
        out = {"key": res["key"], "metric": METRIC, "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": W.dtype_name, "data": "synthetic", "config": res["config"], "roofline": res["roofline"],
               "route": res["route"], "acceptance_rate": res["acceptance_rate"], "ess_per_sec": res["ess_per_sec"],
               "ess_draws": res.get("ess_draws"), "rhat": res.get("rhat"), "ranks_seen": ranks_seen, "rank_devices": devices,
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
            out["secondary"] = [res5] + ([res5s] if res5s is not None else [])
        if world == 1 and a.workload == "cfg2" and not a.no_secondary and a.chains is None and a.traj is None:
            del w
            torch.cuda.empty_cache()
            out["secondary"] = secondary(dev, a)
        if a.sweep and a.workload == "cfg2":
            # "also report a saturating sweep" - the headline's 1024 chains hold one wave on 6 % of the SIMDs; the same
            # kernels at 2^10 ... 2^22 chains (median of 5 calls each, whole call on the stream; route as the library reports it)
            from benchlib import models as M_
            sweep = []
            for lg in (10, 12, 14, 16, 18, 20, 22):
                C = 1 << lg
                ws = W(dev, C, max(10, min(1000, (1 << 24) // C)), 0)
                ws._steps_done = 1
                ws.step(0); torch.cuda.synchronize()
                ts = []
                for r in range(5):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record(); ws.step(1 + r); e.record(); torch.cuda.synchronize()
                    ts.append(s.elapsed_time(e))
                ms = sorted(ts)[2]
                rate = ws.units_per_step() / (ms * 1e-3)
                rec = {"chains": C, "trajectories_per_call": ws.T, "ms_per_call": ms, "value": rate, "route": ws.abi.last_route(),
                       "valu_frac": M_.hmc_gauss_flops_per_chain_step(3) * rate / 1e12 / M_.FP32_PEAK_TFLOPS,
                       "hbm_model_8d_ratio": rate * ws.bytes_per_unit() / 1e9 / M_.HBM_PEAK_GBS,
                       "stored_sample_gbs": rate / ws.L * 12 / 1e9}
                sweep.append(rec)
                print("sweep C=%8d T=%5d  %.3f ms  %.3e chain-steps/s  %s  valu %.3f  8(d)-model %.2f x HBM peak  stored rows %.0f GB/s"
                      % (C, ws.T, ms, rate, rec["route"], rec["valu_frac"], rec["hbm_model_8d_ratio"], rec["stored_sample_gbs"]),
                      file=sys.stderr, flush=True)
                del ws
                torch.cuda.empty_cache()
            out["sweep"] = sweep
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def secondary(dev, a):
    """BASELINE.json's other targets on the same GPU, each a complete record: the D=100 explicit-RMHMC north-star size
    (1024 chains), config 3 (256 chains) on the default route and on the eigendecomposition route SURVEY 8(d)'s flop
    count describes, config 4 (512 chains)."""
    out = []
    # (the north-star RMHMC size gets the headline's own bracket, 20 timed steps after 5: with 2 warm-up steps of 4 ms the first timed
    #  steps still run on ramping clocks - 2.40 ... 2.43e8 against a steady 2.47e8, tools/history/r05n.sh)
    # (the callback contract - the round's subject - right behind the north-star RMHMC size: the compact line drops entries from the TAIL
    #  if it would exceed its 4 KB)
    plan = [(Cfg3N, {}, 20, 5, True), (FunnelHMC, {}, 10, 2, True), (FunnelRMHMC, {}, 5, 1, True), (Cfg3, {}, 10, 2, True),
            (Cfg3, {"jacobi": True, "traj": 20}, 10, 1, False), (Cfg4, {}, 10, 2, True), (NbMlp, {}, 10, 1, True), (NbMlpFull, {}, 10, 1, True)]
    cpu_cache = {}
    for W, kw, steps, warmup, want_cpu in plan:
        t_w = time.perf_counter()
        try:
            w = W(dev, None, kw.get("traj"), 0, **{k: v for k, v in kw.items() if k != "traj"})
            dt, call_ms, prof_ms, prof_n = measure(w, steps, warmup, 1, None, dev, 1)
            r = result_of(w, W, dt, call_ms, prof_ms, prof_n, steps, warmup, 1)
            t_parts = {"measure+ess": time.perf_counter() - t_w}
            if getattr(w, "jacobi", False):
                r["workload"] = w.name
                r["config"]["workload"] = w.name
                try:        # the eigendecomposition route at the north-star chain count as well: a short run
                    del w
                    torch.cuda.empty_cache()
                    w = W(dev, 1024, 5, 0, jacobi=True)
                    # (a 30 ms window: one host-side stall of the launching thread triples it - profiles/r06y had 26 ms of wall per step
                    #  around 9.7 ms of kernel time.  Two windows, the better one reported, both kept in the detail record.)
                    runs = []
                    for _ in range(2):
                        m1k = measure(w, 3, 1, 1, None, dev, 1)
                        runs.append(result_of(w, W, *m1k, 3, 1, 1))
                    r1k = max(runs, key=lambda q: q["value"])
                    r["extras"] = {"value_1024": r1k["value"], "kernel_ms_1024": r1k["roofline"].get("kernel_ms_per_step"),
                                   "frac_1024": r1k["roofline"]["frac"], "trajectories_per_step_1024": 5,
                                   "runs_1024": [q["value"] for q in runs]}
                except Exception as e:
                    r["extras"] = {"error_1024": "%s: %s" % (type(e).__name__, str(e)[:120])}
            if not a.no_cpu_baseline:
                ck = "cfg3" if isinstance(w, Cfg3) else W.key
                if ck not in cpu_cache and want_cpu:
                    if callable(getattr(w, "burn_in_states", None)) and getattr(w, "cpu_takes_start", True) and os.environ.get("HTA_BENCH_BURNED_IN", "1") != "0":
                        # both sides of the ESS comparison start from the device's burned-in chains
                        import tempfile
                        with tempfile.NamedTemporaryFile(suffix=".pt", delete=False) as tf_:
                            torch.save(w.burn_in_states(64), tf_.name)
                        try:
                            cpu_cache[ck] = w.cpu_baseline(a.cpu_seconds, init_file=tf_.name)
                        finally:
                            os.unlink(tf_.name)
                    else:
                        cpu_cache[ck] = w.cpu_baseline(a.cpu_seconds)
                    cpu_cache[ck]["host_cpu"] = _cpu_model()
                if ck in cpu_cache:
                    r["cpu_baseline"] = cpu_cache[ck]
                    r["speedup_vs_cpu_baseline_1core"] = r["value"] / (cpu_cache[ck]["value"] / max(1, cpu_cache[ck].get("cores", 1)))
                    if cpu_cache[ck].get("ess_per_sec") and r.get("ess_per_sec") == r.get("ess_per_sec"):
                        # a ratio of sampling efficiencies only where both sides' chains have mixed (split R-hat <= 1.1); otherwise the
                        # two ESS numbers measure how far the chains still are from each other and the ratio is withheld
                        rh = [v for v in (r.get("rhat"), cpu_cache[ck].get("rhat")) if v is not None and v == v]
                        if rh and max(rh) > 1.1:
                            r["ess_ratio_withheld"] = "split R-hat %.2f (device) / %.2f (cpu) > 1.1" % (r.get("rhat") or float("nan"), cpu_cache[ck].get("rhat") or float("nan"))
                        else:
                            r["ess_per_sec_vs_cpu_baseline"] = r["ess_per_sec"] / cpu_cache[ck]["ess_per_sec"]
            t_parts["cpu"] = time.perf_counter() - t_w - sum(t_parts.values())
            if getattr(W, "published", None):
                r["published"] = W.published
                r["samples_per_s"] = r["value"] / W.L
            if callable(getattr(w, "extras", None)):
                try:
                    r["extras"] = dict(r.get("extras") or {}, **w.extras())
                    if r["extras"].get("launches_per_step") is not None:
                        r["roofline"]["launches_per_step"] = r["extras"]["launches_per_step"]
                except Exception as e:
                    r["extras"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            t_parts["extras"] = time.perf_counter() - t_w - sum(t_parts.values())
            if hasattr(w, "api_call") and not a.no_api and not getattr(w, "jacobi", False):
                try:        # the same steps through the public API (sample / sample_split_model / sample_model)
                    r["api_ms_per_step"], r["api_sync_ms"] = api_timing(w, steps, warmup, reps=3)
                    r["api_route"] = w.abi.last_route()
                except Exception as e:
                    r["api_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
            out.append(r)
            del w
            torch.cuda.empty_cache()
            print("bench: %-14s %6.1f s  %s" % (r["key"], time.perf_counter() - t_w, {k: round(v, 1) for k, v in t_parts.items()}), file=sys.stderr, flush=True)
        except Exception as e:          # a secondary line must never take the headline down with it
            out.append({"workload": W.name, "error": "%s: %s" % (type(e).__name__, str(e)[:300])})
    return out


if __name__ == "__main__":
    main()


"""bench.py's `cpu_baseline` leg: one single-threaded chain of the reference (or its pinned port) per usable host core."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def cpu_baseline_procs(key, seconds, rounds=3, init_file=None):
    """one single-threaded chain per process, one process per usable host core (affinity mask / cgroup quota;
    HTA_BENCH_CPU_PROCS caps it), rate = all leapfrog steps / the slowest process's sampling time.  Each process is
    oracle/cpu_baseline.py: the UNMODIFIED reference when it is importable on this host (`kind: "reference"`), else the
    per-chain port pinned to the reference's recorded runs (`kind: "port"`).  Three rounds of seconds / 3 each (BASELINE.md 3.3:
    repeat >= 3 x, report the median): `value` is the median round's; ESS is computed ONCE on the chains of all rounds pooled
    (rounds x cores chains, truncated to the shortest) and divided by the rounds' summed sampling time, with the draw count
    (`ess_draws` = [draws per chain, chains]) and the split R-hat of those chains beside it.  `init_file`: a torch-saved [k, D] tensor
    of start states (worker i of a round starts from row i mod k) - the device's burned-in chains, so that both sides of an ESS ratio
    measure mixing from the same place rather than the distance from a common starting point."""
    import subprocess
    avail, procs = _usable_cores()
    procs = max(1, min(procs, int(os.environ.get("HTA_BENCH_CPU_PROCS", "64"))))
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="",
               PYTHONDONTWRITEBYTECODE="1")
    from hamiltorch_amd.ess import ess_min, rhat_max
    pooled_chains = []

    # ONE process per core runs all the rounds (a fresh chain each: own seed, own start state): the rounds share the interpreter's and
    # torch's start-up, ~4 s per process that three separate launches of 16 processes paid three times
    t_all = time.time()
    # (the workers write to temporary FILES: a round's samples are more than a pipe holds, and a worker blocked on a full pipe would not
    #  start its next round before the parent has read every worker in front of it - the rounds would run one process after the other)
    import tempfile
    ps = []
    for i in range(procs):
        fo, fe = tempfile.TemporaryFile(mode="w+"), tempfile.TemporaryFile(mode="w+")
        ps.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), key, str(1000 + i),
                                     repr(float(seconds) / rounds), init_file or "-", str(i), str(rounds)], stdout=fo, stderr=fe, env=env, text=True), fo, fe))
    per_round, causes = [[] for _ in range(rounds)], {}
    for p_, fo, fe in ps:
        p_.wait(timeout=300 + 30 * seconds)
        fo.seek(0); fe.seek(0)
        out_, err_ = fo.read(), fe.read()
        fo.close(); fe.close()
        got = 0
        for ln in out_.strip().splitlines():
            try:
                rec = json.loads(ln)
                per_round[int(rec.get("round", 0))].append(rec)
                got += 1
            except (ValueError, IndexError):
                pass
        if got < rounds:
            # a worker that ended in an exception: the last line of its traceback, counted per cause (a
            # baseline over the survivors is survivorship - the port rejects where the reference rejects, S:1045, so this stays 0)
            last = (err_.strip().splitlines() or ["no output"])[-1][:120]
            causes[last] = causes.get(last, 0) + 1
    wall_all = time.time() - t_all

    def once(rep):
        res = per_round[rep]
        failed = sum(causes.values())
        if not res:
            raise RuntimeError("cpu_baseline: every worker of %s failed: %r" % (key, causes))
        procs_ok = len(res)
        wall = wall_all / rounds
        L = res[0]["L"]
        steps = sum(r["n"] * r["L"] for r in res)
        dt = max(r["dt"] for r in res)
        out = {"value": steps / dt, "unit": "leapfrog-steps/s", "cores": procs_ok, "kind": res[0]["kind"], "workers_failed": failed,
               "sample": "%d processes x 1 chain x ~%d trajectories x L=%d, one thread each (%s), %.1f s sampling, %.1f s with "
                         "process start-up%s" % (procs, res[0]["n"], L, res[0]["impl"], dt, wall,
                                                 "; median of %d such rounds" % rounds if rounds > 1 else ""),
               "per_core": steps / dt / procs_ok, "samples_per_s": sum(r["n"] for r in res) / dt,
               "samples_per_s_per_core": sum(r["n"] for r in res) / dt / procs_ok, "acceptance": sum(r["acc"] for r in res) / procs_ok, "host_cores_available": avail, "host_cores_usable": procs}
        if failed:
            out["workers_failed_causes"] = causes
        if "samples" in res[0]:
            # ESS / s with the device side's estimator: the workers' chains pooled as [S, chains, dims] (truncated to the shortest
            # chain), min over the coordinates that travelled (`ess_dims`: the first 128 at most)
            S = min(len(r["samples"]) for r in res)
            out["ess_dims"] = res[0].get("ess_dims")
            pooled_chains.extend(r["samples"] for r in res)
            out["_dt"] = dt
            if S >= 4:
                pooled = torch.tensor([r["samples"][:S] for r in res], dtype=torch.float64).permute(1, 0, 2)
                ess = ess_min(pooled)
                out["ess"] = ess
                out["ess_draws"] = [S, procs_ok]
                out["ess_per_sec"] = ess / dt
        return out
    reps = [once(r) for r in range(rounds)]
    out = pick_round(reps)
    if rounds > 1 and pooled_chains:
        # one estimate from every chain of every round: more chains for the between-chain variance than any single round has
        S = min(len(c) for c in pooled_chains)
        if S >= 4:
            pooled = torch.tensor([c[:S] for c in pooled_chains], dtype=torch.float64).permute(1, 0, 2)
            total_dt = sum(r.get("_dt", 0.0) for r in reps)
            out["ess"], out["ess_draws"], out["rhat"] = ess_min(pooled), [S, len(pooled_chains)], rhat_max(pooled)
            out["ess_per_sec"] = out["ess"] / total_dt
            out["ess_note"] = "all %d chains of the %d rounds pooled, %d draws each, over the rounds' summed sampling time" % (len(pooled_chains), rounds, S)
    for r in reps:
        r.pop("_dt", None)
    out["started_from"] = "the device's burned-in chains" if init_file else "the workload's initial point"
    out["pinned_to"] = PINNED.get(key, "")
    return out


def pick_round(reps):
    """The record of several rounds of one baseline: the median round by `value`, with `repeats`; ESS / s - the noisier
    number (short chains: one chain that sits in a slow region decides the minimum over dimensions; the notebook funnel moved between 3.8
    and 35 from run to run with one round) - is the median of the rounds' own values, listed in `ess_per_sec_rounds`."""
    reps = sorted(reps, key=lambda r: r["value"])
    out = reps[len(reps) // 2]
    if len(reps) > 1:
        out["repeats"] = [r["value"] for r in reps]
        es = sorted(r["ess_per_sec"] for r in reps if r.get("ess_per_sec") is not None and r["ess_per_sec"] == r["ess_per_sec"])
        if len(es) == len(reps):
            out["ess_per_sec_rounds"] = es
            out["ess_per_sec"] = es[len(es) // 2]
    return out


"""The driver's line: contract keys only, numbers rounded, no prose (the driver keeps an 8 KB stdout tail; round 2's 21 KB
line could not be parsed).  The complete record goes to bench_detail.json and to an EARLIER stdout line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LINE_LIMIT = 4096


def _r(x, sig=5):
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        y = float("%.*g" % (sig, x))
        return int(y) if abs(y) >= 1e4 and y == int(y) else y          # 240410000 rather than 240410000.0: the line has a size limit
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
    out = {"bound": roof.get("bound"), "achieved": _r(roof.get("achieved")), "peak": _r(roof.get("peak")), "unit": roof.get("unit"),
           "frac": _r(roof.get("frac"), 4), "traffic": _r(roof.get("traffic")),
           "kernel": str(roof.get("kernel", "")).split(" (")[0][:64],
           "kernel_ms": _r(roof.get("kernel_ms", roof.get("kernel_ms_per_step"))),
           "mfma_busy": _r(phys.get("mfma_busy_frac"), 3),
           "simds_occupied_frac": _r(roof.get("simds_occupied_frac", phys.get("simds_occupied_frac")), 3)}
    lat = roof.get("latency_model")
    if lat:            # cfg2: the bound is the dependent-FMA chain; the other candidate bounds as named side fields
        out["cycles_per_trajectory"] = _r(lat.get("measured_cycles_per_trajectory"), 4)      # (floor: 2 L x 4 = frac x this)
    if roof.get("hbm_model_8d"):
        out["hbm_model_8d_ratio"] = _r(roof["hbm_model_8d"].get("ratio_to_hbm_peak"), 4)      # SURVEY 8(d) streaming convention: not a utilisation
    for k in ("hbm_counter_frac", "valu_frac"):
        if roof.get(k) is not None:
            out[k] = _r(roof[k], 3)
    if roof.get("useful_flops_per_chain_step") is not None:
        out["useful"] = _r(roof["useful_flops_per_chain_step"], 4)
    if roof.get("issued_flops_per_chain_step") is not None:
        out["issued"] = _r(roof["issued_flops_per_chain_step"], 4)
        out["padding"] = _r(roof.get("padding"), 3)
    return out


def _compact_cpu(cb):
    if not cb:
        return None
    return {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
            "sample": str(cb.get("sample", "")).split(" (")[0].replace("processes", "proc").replace("trajectories", "traj")[:72], "pinned_to": str(cb.get("pinned_to", "")).split(" (")[0][:40],
            "host_cpu": str(cb.get("host_cpu", ""))[:48], "ess_per_sec": _r(cb.get("ess_per_sec")), "ess_draws": cb.get("ess_draws"),
            "rhat": _r(cb.get("rhat"), 3), "rounds": len(cb.get("repeats") or [1])}


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
                     "D": cfg.get("D"), "parallelism": str(cfg.get("parallelism", "")).replace(" per GPU", "/GPU")[:44]}
    out["roofline"] = _compact_roofline(full.get("roofline", {}))
    if full.get("cpu_baseline"):
        out["cpu_baseline"] = _compact_cpu(full["cpu_baseline"])
        out["speedup_vs_cpu_baseline"] = _r(full.get("speedup_vs_cpu_baseline"), 4)
    for k in ("api_ms_per_step", "acceptance_rate", "ess_per_sec", "ess_per_sec_vs_cpu_baseline", "gather_ms", "ess_draws", "rhat"):
        if full.get(k) is not None:
            out[k] = _r(full[k]) if not isinstance(full[k], list) else full[k]
    for k in ("ranks_seen", "rank_devices", "launcher", "collective_backend"):
        if k in full and full.get("n_gpus", 1) > 1:
            out[k] = full[k]
    sec, withheld = [], []
    for r in full.get("secondary", []) or []:
        if "error" in r:
            sec.append({"key": _short_key(r), "error": r["error"][:80]})
            continue
        roof, cb = _compact_roofline(r.get("roofline", {})), r.get("cpu_baseline") or {}
        # one entry per workload, short: bound / unit / achieved follow from `frac` (bound "mfma": frac x 157.3 TFLOP/s; "hbm":
        # frac x 8000 GB/s) and are spelled out in bench_detail.json
        # (`bound` is spelled out only where it is not "mfma" = useful flops / kernel time / 157.3 TFLOP/s)
        # (sizes: the line must stay < LINE_LIMIT with nine workloads - wall ms per step = chains x T x L / value and padding = issued /
        # useful follow from the fields kept; `cores` / `kind` of a CPU baseline are spelled out only where they differ from the primary's)
        pcb = full.get("cpu_baseline") or {}
        e = {"key": _short_key(r), "chains": r.get("config", {}).get("chains_per_gpu"), "value": _r(r.get("value")),
             "frac": roof["frac"], "bound": roof["bound"] if roof["bound"] != "mfma" else None, "mfma_busy": roof["mfma_busy"],
             "traffic": _r(roof["traffic"], 3), "kernel": roof["kernel"].split("<")[0].replace("_kernel", "")[:20], "kernel_ms": _r(roof["kernel_ms"], 4),
             "cpu": {"value": _r(cb.get("value"), 4)}}
        if cb.get("cores") != pcb.get("cores"):
            e["cpu"]["cores"] = cb.get("cores")
        if cb.get("kind") != pcb.get("kind"):
            e["cpu"]["kind"] = cb.get("kind")
        if cb.get("workers_failed"):
            e["cpu"]["failed"] = cb["workers_failed"]
        if roof.get("padding") is not None:     # issued (PMC) / useful flops per chain-step; both spelled out in bench_detail.json
            e["padding"] = roof["padding"]
        # BASELINE.json's metric, second half: ESS / s on the device and against the CPU reference (same estimator, same coordinates)
        if r.get("ess_per_sec") is not None and r["ess_per_sec"] == r["ess_per_sec"]:
            e["ess_per_sec"] = _r(r["ess_per_sec"], 4)
        if r.get("ess_draws"):                       # [draws per chain, chains] behind the device's ESS, and their split R-hat
            e["ess_draws"], e["rhat"] = r["ess_draws"], _r(r.get("rhat"), 3)
        if cb.get("ess_per_sec"):
            e["cpu"]["ess_per_sec"] = _r(cb["ess_per_sec"], 4)
            if cb.get("ess_draws"):
                e["cpu"]["ess_draws"], e["cpu"]["rhat"] = cb["ess_draws"], _r(cb.get("rhat"), 3)
        if r.get("ess_per_sec_vs_cpu_baseline") is not None:
            e["ess_per_sec_vs_cpu"] = _r(r["ess_per_sec_vs_cpu_baseline"], 4)
        elif r.get("ess_ratio_withheld"):
            withheld.append(e["key"])
        for k in ("gather_ms", "n_gpus", "ranks_seen"):
            if r.get(k) is not None:
                e[k] = _r(r[k], 4)
        if r.get("scaling"):                         # "strong": a fixed total (cfg5-strong: 8192 chains over the node)
            e["scaling"], e["chains_total"] = r["scaling"], r.get("config", {}).get("chains_total")
        if r.get("extras"):
            short = {"graph_replay": "graph", "value_graphs_off": "graphs_off", "value_notebook_closure": "nb_closure", "value_callback_path": "cb_path", "value_launch_sequence_256": "launch_seq_256", "trace_compile_s": "compile_s",
                     "launches_per_step": "launches", "hta_launches_per_step": "hta_launches", "callback_evaluations_per_step": "cb_evals", "metric_evaluations_per_step": "metric_evals",
                     "predict_route": "predict", "predict_ms_torch_path": "predict_ms_torch"}
            e["extras"] = {short.get(k, k): _r(v, 4) for k, v in r["extras"].items()
                           if not isinstance(v, (dict, list)) and v is not None and k not in ("predict_samples_per_s", "predict_samples", "predict_ms_torch_path", "trajectories_per_step_1024", "kernel_ms_1024", "callback_evaluations_per_step", "metric_evaluations_per_step", "hta_launches_per_step", "traces", "predict_route", "frac_1024")}
        if r.get("published"):          # the reference's published samples / s (one chain); ours = value / L, spelled out in bench_detail.json
            e["published_sps"] = r["published"].get("samples_per_s")
        if e.get("bound") == "valu":                 # (no matrix instructions, a few MB of HBM per launch: both are in bench_detail.json)
            e.pop("mfma_busy", None); e.pop("traffic", None)
        sec.append({k: v for k, v in e.items() if v is not None or k in ("value", "frac", "chains")})
    if sec:
        out["secondary"] = sec
    if withheld:        # ESS ratios are printed only where both sides' chains have mixed (split R-hat <= 1.1); `rhat` is beside every ESS
        out["ess_ratio_withheld_rhat_gt_1.1"] = withheld
    out["detail"] = detail_path
    line = json.dumps(out, separators=(",", ":"))
    while len(line) >= LINE_LIMIT and out.get("secondary"):          # never print a line the driver cannot keep
        out["secondary"].pop()
        out["secondary_truncated"] = True
        line = json.dumps(out, separators=(",", ":"))
    return line


def emit(full, root=None):
    """Complete record -> bench_detail.json (+ gpurun_out/) and an earlier stdout line; the compact line LAST."""
    root = root or ROOT
    detail = json.dumps(full)
    for d in (root, os.path.join(root, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
        except OSError:
            pass
    sys.stdout.write("BENCH_DETAIL " + detail + "\n")
    sys.stdout.flush()
    print(compact_line(full), flush=True)


"""Builds profiles/physical.json (read by bench.py: `roofline.physical`) from rocprofv3 output directories.

    python tools/physical.py <key>=<dir> ... > profiles/physical.json

Each <dir> holds, for ONE bench.py command line, the csv files of three separate passes (tools/physical.sh):
  stats/   --kernel-trace --stats                      -> *kernel_trace.csv (durations, grid, workgroup size)
  pmc_rd/  --pmc FETCH_SIZE                            -> *counter_collection.csv
  pmc_wr/  --pmc WRITE_SIZE
  pmc_sq/  --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE
Counters are per launch, summed over the chip.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-byte read
requests at 64 bytes); WRITE_SIZE is taken as reported.  Both are KiB."""
import collections
import csv
import glob
import json
import os
import sys

N_SIMDS, HBM_PEAK = 1024, 8000.0


def rows(d, pat):
    for p in glob.glob(os.path.join(d, "**", pat), recursive=True):
        with open(p, newline="") as f:
            yield from csv.DictReader(f)


def short(k):
    return k.split("(")[0].replace("void ", "").strip()


def main():
    out = {}
    for arg in sys.argv[1:]:
        key, d = arg.split("=", 1)
        meta = json.load(open(os.path.join(d, "meta.json")))
        dur, geom = collections.defaultdict(list), {}
        for r in rows(os.path.join(d, "stats"), "*kernel_trace.csv"):
            k = short(r["Kernel_Name"])
            if "hta::" not in k and not k.startswith("hta_cb"):          # (the library's kernels and the run-time compiled callback kernels)
                continue
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
            wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
            gr = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            geom[k] = (gr // max(1, wg), wg)
        ctr = collections.defaultdict(lambda: collections.defaultdict(list))
        for sub in ("pmc_rd", "pmc_wr", "pmc_sq", "pmc_valu"):
            for r in rows(os.path.join(d, sub), "*counter_collection.csv"):
                k = short(r["Kernel_Name"])
                if "hta::" in k or k.startswith("hta_cb"):
                    ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        steps = meta["steps"] + meta["warmup"]
        kernels = {}
        tot_ms = tot_bytes = 0.0
        for k, ds in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            ms_per_step = sum(ds) / steps
            c = {n: sum(v) / len(v) for n, v in ctr[k].items()}
            launches = len(ds) / steps
            nwg, wg = geom[k]
            waves = nwg * ((wg + 63) // 64)
            rec = {"ms_per_launch": sum(ds) / len(ds), "launches_per_step": launches, "ms_per_step": ms_per_step,
                   "workgroups": nwg, "workgroup_size": wg, "waves_per_launch": waves,
                   "simds_occupied_frac": min(1.0, waves / N_SIMDS)}
            if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
                b = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
                rec["hbm_bytes_per_launch"] = b
                rec["hbm_gbs"] = b / (rec["ms_per_launch"] * 1e-3) / 1e9
                tot_bytes += b * launches
            if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
                simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMDS           # the counter sums the 8 XCDs
                rec["mfma_busy_frac_of_all_simd_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
                rec["mfma_tflops_issued"] = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0 / (rec["ms_per_launch"] * 1e-3) / 1e12
                if c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) > 0:      # round 6: the metric kernel's second-order product
                    rec["mfma_bf16_tflops_issued"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0 / (rec["ms_per_launch"] * 1e-3) / 1e12
            if "SQ_INSTS_VALU" in c and c.get("SQ_WAVE_CYCLES", 0) > 0:      # the VALU kernels (compiled callbacks): optional fourth pass
                rec["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / max(1.0, c.get("SQ_WAVES", waves))
                rec["valu_active_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
            kernels[k] = rec
            tot_ms += ms_per_step
        dom = next(iter(kernels))
        out[key] = {"command": meta["command"], "trajectories_per_step": meta.get("traj"), "dominant_kernel": dom,
                    "kernel_ms_per_step": tot_ms, "hbm_bytes_per_step": tot_bytes or None,
                    "hbm_gbs": (tot_bytes / (tot_ms * 1e-3) / 1e9) if tot_bytes else None,
                    "hbm_frac_of_peak": (tot_bytes / (tot_ms * 1e-3) / 1e9 / HBM_PEAK) if tot_bytes else None,
                    "simds_occupied_frac": kernels[dom]["simds_occupied_frac"],
                    "mfma_busy_frac": kernels[dom].get("mfma_busy_frac_of_all_simd_cycles"),
                    "mfma_tflops_issued": kernels[dom].get("mfma_tflops_issued"),
                    "mfma_bf16_tflops_issued": kernels[dom].get("mfma_bf16_tflops_issued"),
                    "valu_active_frac": kernels[dom].get("valu_active_frac_of_wave_cycles"),
                    "valu_insts_per_wave": kernels[dom].get("valu_insts_per_wave"),
                    "kernels": kernels, "source": meta.get("source", "tools/physical.sh")}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

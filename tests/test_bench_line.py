"""The driver keeps an 8 KB tail of bench.py's stdout and parses its LAST line (round 2's 21 KB line came back as
`parsed: null`).  These tests pin the shape and size of that line on a complete record of a real run
(profiles/r02zz_bench.json, 21 KB) and on an inflated one."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOF = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "mfma_busy", "simds_occupied_frac")
CPU = ("value", "unit", "cores", "kind", "sample", "pinned_to", "host_cpu")


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r02zz_bench.json")))


def _driver_view(stdout, tail=8000):
    """What the driver does: keep the tail, parse the last line."""
    return json.loads(stdout[-tail:].splitlines()[-1])


def test_compact_line_is_small_and_complete():
    full = _full()
    assert len(json.dumps(full)) > 16000                  # the record that could not be parsed in round 2
    line = bench.compact_line(full)
    assert len(line.encode()) < bench.LINE_LIMIT == 4096
    assert "\n" not in line
    rec = json.loads(line)
    for k in CONTRACT:
        assert k in rec, k
    for k in ROOF:
        assert k in rec["roofline"], k
    for k in CPU:
        assert k in rec["cpu_baseline"], k
    assert rec["value"] == pytest.approx(full["value"], rel=1e-5)
    assert rec["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3)
    assert rec["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    assert rec["config"]["workload"] == "cfg2" and rec["config"]["chains_per_gpu"] == 1024
    sec = rec["secondary"]
    assert [s["key"] for s in sec] == ["cfg3@1024", "cfg3", "cfg3-eig", "cfg4"]          # the north-star RMHMC size first
    for s, f in zip(sec, full["secondary"]):
        assert s["value"] == pytest.approx(f["value"], rel=1e-4)
        assert s["frac"] == pytest.approx(f["roofline"]["frac"], rel=1e-3)
        assert s["cpu"]["cores"] == 1 and s["cpu"]["value"] > 0
        assert s["chains"] == f["config"]["chains_per_gpu"]
    assert "note" not in line and "physical" not in rec["roofline"]                       # no prose, no inlined profile


def test_last_line_survives_the_drivers_tail(tmp_path, monkeypatch, capsys):
    full = _full()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(full)
    out = capsys.readouterr().out
    assert out.count("\n") == 2 and out.startswith("BENCH_DETAIL ")
    rec = _driver_view(out)
    assert rec["value"] == pytest.approx(full["value"], rel=1e-5)
    assert rec["secondary"][0]["key"] == "cfg3@1024"
    assert rec["roofline"]["kernel"].startswith("hmc_gauss_quad_kernel")
    # also with a stderr suffix interleaved before it (warnings printed earlier cannot cut the last line)
    rec2 = _driver_view("x" * 5000 + "\n" + out)
    assert rec2 == rec
    # the complete record went to the side file
    side = json.load(open(tmp_path / "bench_detail.json"))
    assert side["secondary"][0]["roofline"]["physical"] is not None
    assert json.loads(out.splitlines()[0][len("BENCH_DETAIL "):]) == side


def test_line_never_exceeds_the_limit_even_with_many_workloads():
    full = _full()
    extra = copy.deepcopy(full["secondary"])
    for i in range(6):
        for e in copy.deepcopy(extra):
            e["workload"] = "extra%d: " % i + "x" * 200
            e["roofline"]["kernel"] = "k" * 300
            full["secondary"].append(e)
    line = bench.compact_line(full)
    assert len(line.encode()) < 4096
    rec = json.loads(line)
    assert rec["secondary"][0]["key"] == "cfg3@1024" and rec["secondary_truncated"] is True
    for k in CONTRACT:
        assert k in rec


def test_multi_gpu_keys_are_in_the_line():
    full = _full()
    full.update({"n_gpus": 8, "ranks_seen": 8, "rank_devices": [[r, r] for r in range(8)], "collective_backend": "rccl",
                 "launcher": "torch.distributed.run", "gather_ms": 1.25})
    full.pop("secondary"); full.pop("cpu_baseline")
    rec = json.loads(bench.compact_line(full))
    assert rec["n_gpus"] == 8 and rec["ranks_seen"] == 8 and rec["collective_backend"] == "rccl"
    assert rec["gather_ms"] == 1.25 and len(rec["rank_devices"]) == 8


def test_nonfinite_numbers_do_not_break_the_json():
    full = _full()
    full["ess_per_sec"] = float("nan")
    full["secondary"][0]["value"] = float("inf")
    rec = json.loads(bench.compact_line(full))          # strict JSON: NaN / Infinity would not parse elsewhere
    assert "NaN" not in bench.compact_line(full) and "Infinity" not in bench.compact_line(full)
    assert rec["secondary"][0]["value"] is None


def test_multi_gpu_line_carries_the_second_north_star_target_with_its_gather():
    """under --gpus N > 1 the line's `secondary[0]` is cfg5 (D = 100 explicit RMHMC, 1024 chains per
    GPU, measured on every rank) with gather_ms / n_gpus / ranks_seen - one scaling run yields both north-star curves."""
    full = _full()
    cfg5 = copy.deepcopy(full["secondary"][0])
    cfg5.update({"key": "cfg5", "workload": "cfg5: cfg3 sharded, 1024 chains per GPU", "gather_ms": 3.4567, "n_gpus": 8, "ranks_seen": 8})
    cfg5["config"]["chains_per_gpu"] = 1024
    full.update({"n_gpus": 8, "ranks_seen": 8, "rank_devices": [[r, r] for r in range(8)], "collective_backend": "rccl",
                 "launcher": "torch.distributed.run", "secondary": [cfg5]})
    full.pop("cpu_baseline")
    line = bench.compact_line(full)
    assert len(line.encode()) < bench.LINE_LIMIT
    rec = json.loads(line)
    s5 = rec["secondary"][0]
    assert s5["key"] == "cfg5" and s5["gather_ms"] == pytest.approx(3.457, rel=1e-3) and s5["n_gpus"] == 8 and s5["ranks_seen"] == 8
    assert s5["chains"] == 1024 and s5["value"] > 0 and s5["frac"] > 0


def test_funnel_entries_carry_their_extras_and_stay_inside_the_limit():
    """The callback-contract entries (funnel-hmc / funnel-rmhmc): graph replay on / off, the notebook's closure, launches per
    step and the published samples/s travel in the compact entry; with eight secondary workloads the line still fits."""
    full = _full()
    base = copy.deepcopy(full["secondary"][0])
    while len(full["secondary"]) < 6:
        e = copy.deepcopy(base); e["key"] = "w%d" % len(full["secondary"]); full["secondary"].append(e)
    for key, pub in (("funnel-hmc", 56.10), ("funnel-rmhmc", 0.19)):
        e = copy.deepcopy(base)
        e.update({"key": key, "workload": key + ": 11-D funnel", "samples_per_s": 1234.5,
                  "published": {"samples_per_s": pub, "hw": "notebook host, 1 chain", "src": "nb"},
                  "extras": {"graph_replay": True, "value_graphs_off": 1.23456e6, "value_notebook_closure": 2.5e5, "launches_per_step": 1300,
                             "callback_evaluations_per_step": 1300}})
        full["secondary"].append(e)
    line = bench.compact_line(full)
    assert len(line.encode()) < bench.LINE_LIMIT
    rec = json.loads(line)
    keys = [s["key"] for s in rec["secondary"]]
    assert "funnel-hmc" in keys and "funnel-rmhmc" in keys and not rec.get("secondary_truncated")
    fh = rec["secondary"][keys.index("funnel-hmc")]
    assert fh["extras"]["graph"] is True and fh["extras"]["graphs_off"] == pytest.approx(1.235e6, rel=1e-3)
    assert fh["published_sps"] == 56.10


def test_cpu_baseline_rounds_report_the_median_of_each_number():
    """benchlib/cpu.py: `value` is the median round's, ESS / s the median of the rounds' own (the funnel's moved 9 x between single rounds)."""
    from benchlib.cpu import pick_round
    out = pick_round([{"value": 3.0, "ess_per_sec": 34.5}, {"value": 1.0, "ess_per_sec": 6.7}, {"value": 2.0, "ess_per_sec": 3.8}])
    assert out["value"] == 2.0 and out["repeats"] == [1.0, 2.0, 3.0]
    assert out["ess_per_sec"] == 6.7 and out["ess_per_sec_rounds"] == [3.8, 6.7, 34.5]
    one = pick_round([{"value": 5.0, "ess_per_sec": 1.5}])
    assert one == {"value": 5.0, "ess_per_sec": 1.5}
    nan = float("nan")
    out = pick_round([{"value": 1.0, "ess_per_sec": nan}, {"value": 2.0, "ess_per_sec": 4.0}, {"value": 3.0, "ess_per_sec": 5.0}])
    assert out["value"] == 2.0 and out["ess_per_sec"] == 4.0 and "ess_per_sec_rounds" not in out      # a round without an estimate: the median round's own

#!/usr/bin/env python
"""ISA of a compiled callback kernel (needs no GPU): traces the notebook funnel (or `module:function` D), builds it with hipRTC
under each option set and prints register counts and the instruction mix of the leapfrog loop (the innermost backward branch).
python tools/jit_isa.py [--dump out.s]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hamiltorch_amd.jit import runtime  # noqa: E402
from hamiltorch_amd.jit.trace import trace_callback  # noqa: E402
from benchlib.workloads import funnel_ll_device, funnel_ll_notebook  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def loop_of(lines):
    """Instructions of the innermost loop: the shortest span closed by a backward branch."""
    labels, best = {}, None
    for i, ln in enumerate(lines):
        m = re.match(r"^\s*[0-9a-f]+ <(L\d+)>:|^(L\d+|\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1) or m.group(2)] = i
    for i, ln in enumerate(lines):
        m = re.search(r"s_cbranch_\w+\s+(\S+)", ln) or re.search(r"s_branch\s+(\S+)", ln)
        if m and m.group(1).strip("<>") in labels:
            j = labels[m.group(1).strip("<>")]
            if j < i and (best is None or i - j < best[1] - best[0]):
                best = (j, i)
    return lines[best[0]:best[1] + 1] if best else []


def report(fn, opts, dump=None):
    tr = trace_callback(fn, torch.ones(11))
    src = runtime.hmc_generated_source(tr, torch.float32, 0)
    old = runtime.OPTIONS
    runtime.OPTIONS = tuple(opts)
    try:
        key, blob = runtime.compile_source(src, runtime.SKELETON_HMC)
    finally:
        runtime.OPTIONS = old
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(blob)
    notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    regs = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", notes)}
    asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--symbolize-operands", f.name], capture_output=True, text=True).stdout
    if dump:
        open(dump, "w").write(asm)
    lines = asm.splitlines()
    body = [ln for ln in loop_of(lines) if re.match(r"^\s+[sv]_|^\s+(global|ds|buffer|flat)_", ln)]
    mix = collections.Counter(re.match(r"^\s+(\S+)", ln).group(1) for ln in body)
    valu = sum(v for k, v in mix.items() if k.startswith("v_"))
    pk = sum(v for k, v in mix.items() if k.startswith("v_pk_"))
    total = sum(1 for ln in lines if re.match(r"^\s+[sv]_|^\s+(global|ds|buffer|flat)_", ln))
    print("%-20s %-40s nodes %4d  kernel %5d instr  loop %4d (VALU %4d, packed %3d)  %s" % (fn.__name__, " ".join(o for o in opts if "slp" in o or "fast" in o or "contract" in o), len(tr.graph.reachable([tr.value] + tr.grad())), total, len(body), valu, pk, regs))
    os.unlink(f.name)
    return mix


if __name__ == "__main__":
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    base = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast"]
    for fn in (funnel_ll_device, funnel_ll_notebook):
        report(fn, base + ["-fno-slp-vectorize"])
        mix = report(fn, base, dump if fn is funnel_ll_device else None)
    print(dict(mix.most_common(14)))

"""Where a kernel's scratch (spill) traffic sits: per basic block of the gfx950 assembly, the number of scratch_load / scratch_store
instructions next to the number of matrix instructions.

    python tools/scratch_in_loops.py hamiltorch_amd/csrc/mlp_mfma.hip mlp_mfma_kernelILi2ELi7ELi0ELi512 -DHTA_MLP_SINGLE

Prints every block that holds scratch or MFMA instructions and a one-line verdict: blocks with MFMAs (the hot chunk code) must
hold no scratch access for the spill count of the code object's metadata to be harmless."""
import collections
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-ffp-contract=on", "-fno-slp-vectorize", "-x", "hip", "-S",
         "--cuda-device-only"]


def blocks(src, kernel, extra=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([HIPCC] + FLAGS + list(extra) + [src, "-o", out], check=True, capture_output=True)
        lines = open(out).read().splitlines()
    start = next(i for i, l in enumerate(lines) if kernel in l and re.match(r"^_ZN3hta\S+:", l))
    end = next(i for i in range(start + 1, len(lines)) if ".Lfunc_end" in lines[i])
    res, cur = collections.OrderedDict(), "entry"
    res[cur] = dict(comment="", n=0, scratch=0, mfma=0)
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?", l)
        if m:
            cur = m.group(1)
            res[cur] = dict(comment=(m.group(2) or "").strip("; "), n=0, scratch=0, mfma=0)
            continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        res[cur]["n"] += 1
        if t[0].startswith("scratch_"):
            res[cur]["scratch"] += 1
        if t[0].startswith("v_mfma"):
            res[cur]["mfma"] += 1
    return res


def main():
    src, kernel, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    res = blocks(src, kernel, extra)
    hot = [b for b, v in res.items() if v["mfma"]]
    for b, v in res.items():
        if v["scratch"] or v["mfma"]:
            print("%-12s %-44s instructions %4d  scratch %3d  mfma %3d" % (b, v["comment"][:44], v["n"], v["scratch"], v["mfma"]))
    tot = sum(v["scratch"] for v in res.values())
    inhot = sum(res[b]["scratch"] for b in hot)
    print("scratch instructions: %d in the kernel, %d inside the %d blocks that hold its %d matrix instructions"
          % (tot, inhot, len(hot), sum(res[b]["mfma"] for b in hot)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Disassembly of one gfx950 kernel out of a built object / shared library (CPU only: llvm-objdump).

    python tools/isa_of.py hamiltorch_amd/csrc/build/hmc_gaussian.o 'hmc_gauss_quad_kernelILi3ELb0ELi25ELi0E' [--loops]

Prints the kernel's instructions (addresses and encodings stripped).  --loops: for every backward branch the number of
instructions of the loop it closes, split into the dependent FMA chain (v_fmac / v_fma / v_pk_fma), other vector ALU, scalar,
memory and wait / nop instructions - the static cost model of a kernel that runs ONE wave per SIMD (every instruction of a
lone wave takes an issue slot: DESIGN.md section 4, cfg2)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_lines(path, pattern):
    d = tempfile.mkdtemp(prefix="isa_of_")
    try:
        shutil.copy(path, os.path.join(d, "in.bin"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "in.bin"], cwd=d, check=True, capture_output=True)
        rx = re.compile(r"^[0-9a-f]+ <(\S*%s\S*)>:" % pattern)
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f], cwd=d, check=True, capture_output=True, text=True).stdout
            out, on, name = [], False, None
            for ln in txt.splitlines():
                m = rx.match(ln)
                if m and not on:
                    on, name = True, m.group(1)
                    continue
                if on:
                    if re.match(r"^[0-9a-f]+ <", ln):
                        break
                    mm = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-F]+):", ln)
                    if mm:
                        out.append((int(mm.group(2), 16), mm.group(1).strip()))
                        if mm.group(1).startswith("s_endpgm"):
                            break
            if out:
                return name, out
    finally:
        shutil.rmtree(d, ignore_errors=True)
    raise SystemExit("no kernel matching %r in %s" % (pattern, path))


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(("v_accvgpr", "v_accvgpr_read", "v_accvgpr_write")):
        return "acc"
    if op.startswith(("v_fmac_f32", "v_fma_f32", "v_pk_fma_f32")) and "dpp" not in ins:
        return "fma"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "mem"
    if op.startswith("v_"):
        return "dpp" if ("dpp" in ins or "quad_perm" in ins or "row_" in ins) else "valu"
    return "salu"


def loops(lines):
    """(start index, end index) of every loop closed by a backward branch."""
    addr = {a: i for i, (a, _) in enumerate(lines)}
    out = []
    for i, (a, ins) in enumerate(lines):
        m = re.match(r"s_c?branch\S*\s+(\d+)", ins)
        if not m:
            continue
        off = int(m.group(1))
        if off >= 32768:
            tgt = a + 4 + (off - 65536) * 4
            if tgt in addr:
                out.append((addr[tgt], i))
    return out


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    name, lines = kernel_lines(sys.argv[1], sys.argv[2])
    if "--loops" in sys.argv:
        print(name, "-", len(lines), "instructions")
        for s, e in loops(lines):
            cnt = {}
            for _, ins in lines[s:e + 1]:
                k = classify(ins)
                cnt[k] = cnt.get(k, 0) + 1
            print("loop [%d, %d]: %d instructions  %s" % (s, e, e - s + 1, "  ".join("%s %d" % kv for kv in sorted(cnt.items()))))
    else:
        print("//", name)
        for _, ins in lines:
            print(ins)


if __name__ == "__main__":
    main()

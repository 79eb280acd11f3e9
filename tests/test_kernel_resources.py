"""CPU: the built gfx950 code objects of libhamiltorch_amd.so - no kernel of the hot paths spills to scratch, and the
register budgets the occupancy arguments of DESIGN.md rest on hold (read from the code objects' metadata)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    lib = os.path.join(ROOT, "hamiltorch_amd", "libhamiltorch_amd.so")
    if not (os.path.exists(lib) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("needs the built library and the ROCm llvm tools")
    d = tmp_path_factory.mktemp("co")
    shutil.copy(lib, d / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
    out = {}
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f], cwd=d, check=True, capture_output=True,
                               text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))      # noqa: E731
            out[name.group(1)] = dict(agpr=int(re.match(r"\s*(\d+)", blk).group(1)), vgpr=g("vgpr_count"),
                                      scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"),
                                      spill=g("vgpr_spill_count"))
    assert len(out) > 20
    return out


def _find(kernels, *parts):
    hits = [k for k in kernels if all(p in k for p in parts)]
    assert hits, parts
    return hits


def test_spill_budget_of_the_hot_kernels(kernels):
    """No scratch at all in the cfg2 kernels, in the many-chain RMHMC kernels and in the instance BASELINE config 3 runs on
    (rmhmc_fused_kernel_wide<56>: one workgroup per CU, 288 registers; 16 bytes of scratch are the compiler's emergency slot,
    no register lives there).  Still bounded, not zero: the two-workgroups-per-CU instance of the one-chain kernel (256-register
    cap, used from 257 to 703 chains) and the MLP MFMA kernel (128-register cap from two 448-thread workgroups per CU)."""
    clean = ["hmc_gauss_quad_kernel", "hmc_gauss_eig_kernel", "hmc_gauss_wave_eig_kernel", "rmhmc_batch_kernel", "rmhmc_mfma4_kernel",
             "rmhmc_mfma4x4_kernel", "rmhmc_uv_kernel", "rmhmc_momentum_wave_kernel", "rmhmc_momentum_kernel"]
    for h in clean:
        for k in _find(kernels, h):
            if "rmhmc_uv_kernelILi" in k and "ELb1ELb1E" in k:        # the co-resident instances: next test
                continue
            assert kernels[k]["scratch"] == 0 and kernels[k]["spill"] == 0, (k, kernels[k])
    for k in _find(kernels, "rmhmc_fused_kernel_wide"):                 # BASELINE config 3's instance (KH = 56) and KH = 64
        assert kernels[k]["spill"] == 0 and kernels[k]["scratch"] <= 16, (k, kernels[k])
    for k in _find(kernels, "rmhmc_fused_kernelIfLi56ELi1E"):
        assert kernels[k]["scratch"] <= 128 and kernels[k]["spill"] <= 32, (k, kernels[k])
    for k in _find(kernels, "mlp_mfma_kernelILi2ELi7ELi0ELi512E"):      # BASELINE config 4's instance
        assert kernels[k]["scratch"] <= 256 and kernels[k]["spill"] <= 64, (k, kernels[k])
    # the matrix-core metric kernel: its GEMM / Cholesky / element-wise phases are separate functions without scratch; the
    # kernel body (calls only) parks a few values around the calls
    # (round 3: per-lane addresses are derived from an opaque index at the point of use - vgpr_spill 20 -> 1, scratch 112 -> 32)
    # (round 6: the kernel's body - the fast solve's call sequence - parks nothing; the general sequence runs out of line and re-reads its
    # arguments from the kernel-argument segment: the private segment is that function's stack frame)
    for k in _find(kernels, "metric_warm_mfma_kernel"):
        assert kernels[k]["spill"] == 0 and kernels[k]["scratch"] <= 128, (k, kernels[k])
    # round 4: the trajectory kernel of the eigendecomposition route (the same evaluation body inside a loop over the trajectory's
    # 4 L + 3 evaluations): nothing of the loop's state lives in scratch
    # (round 6: the kernel's own body parks nothing; the 128 bytes are the stack frame of traj_general_eval, the out-of-line general
    # evaluations - momentum draw, the rare way out of the resident loop -, which the resident loop never touches:
    # test_fast_solve_phase_functions_stay_inside_the_caller_saved_registers checks the body's instructions)
    hits = _find(kernels, "metric_traj_mfma_kernel")
    assert hits
    for k in hits:
        assert kernels[k]["spill"] == 0 and kernels[k]["scratch"] <= 128, (k, kernels[k])
    # the notebook-model kernel (round 3): its loads / stores / momentum draws run out of line and the momentum's W2 share lives
    # in a workspace, so the code object's scratch is a stack for those calls (three 160-byte state vectors and change), not a
    # home for the gradient pass's values: < 1 KB per lane (the first version: 2.4 KB and six serial reloads in every kick)
    for k in _find(kernels, "mlp3_mfma_kernel"):
        assert kernels[k]["scratch"] <= 1024 and kernels[k]["vgpr"] <= 256, (k, kernels[k])


def test_coresident_uv_instances_fit_two_workgroups_and_keep_scratch_out_of_the_step_loop(kernels):
    """Round 4: rmhmc_uv_kernel<G, lean, CO = true, NACC> is capped at 256 registers (two four-wave workgroups per CU).  The
    cap parks <= 24 values in scratch - all of them per-launch / per-trajectory values: no scratch access lies inside the leapfrog
    step loop (the loop with the step's 208 static matrix instructions and four barriers), no accumulation registers at all."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_of
    obj = os.path.join(ROOT, "hamiltorch_amd", "csrc", "build", "rmhmc_uv.o")
    if not os.path.exists(obj):
        pytest.skip("needs the object file of rmhmc_uv.hip")
    hits = [k for k in _find(kernels, "rmhmc_uv_kernelILi") if "ELb1ELb1E" in k]
    assert len(hits) == 4
    for k in hits:
        assert kernels[k]["vgpr"] <= 256 and kernels[k]["agpr"] == 0 and kernels[k]["spill"] <= 24 and kernels[k]["scratch"] <= 96, (k, kernels[k])
        _, lines = isa_of.kernel_lines(obj, re.escape(k))
        step_loops = []
        for s_, e_ in isa_of.loops(lines):
            ops = [isa_of.classify(i) for _, i in lines[s_:e_ + 1]]
            if ops.count("mfma") == 208 and ops.count("barrier") == 4:
                step_loops.append((s_, e_))
        assert step_loops, k
        for s_, e_ in step_loops:
            assert not any(i.startswith("scratch_") for _, i in lines[s_:e_ + 1]), k


def test_round4_rmhmc_kernels_fit_two_workgroups_per_cu_and_keep_scratch_out_of_the_step_loop(kernels):
    """rmhmc_uvc_kernel<co> (BASELINE config 3) and rmhmc_uvc2_kernel<co> (257 ... 1792 chains: the north-star size) are capped at
    256 registers so that two four-wave workgroups share a CU; the one-chain kernel spills nothing, the two-chain kernel parks
    <= 32 per-launch / per-trajectory values in scratch and none inside the leapfrog step loop; no accumulation registers."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_of
    obj = os.path.join(ROOT, "hamiltorch_amd", "csrc", "build", "rmhmc_uvc.o")
    if not os.path.exists(obj):
        pytest.skip("needs the object file of rmhmc_uvc.hip")
    for k in _find(kernels, "rmhmc_uvc_kernelILb1E"):
        assert kernels[k]["vgpr"] <= 256 and kernels[k]["agpr"] == 0 and kernels[k]["spill"] == 0 and kernels[k]["scratch"] == 0, (k, kernels[k])
    for k in _find(kernels, "rmhmc_uvc2_kernelILb1E"):
        assert kernels[k]["vgpr"] <= 256 and kernels[k]["agpr"] == 0 and kernels[k]["spill"] <= 32 and kernels[k]["scratch"] <= 128, (k, kernels[k])
        _, lines = isa_of.kernel_lines(obj, re.escape(k))
        hot = []
        for s_, e_ in isa_of.loops(lines):
            ops = [isa_of.classify(i) for _, i in lines[s_:e_ + 1]]
            if ops.count("mfma") >= 156 and ops.count("barrier") >= 3 and e_ - s_ < 1200:      # the step loop and its rotated forms
                hot.append((s_, e_))
        assert hot, k
        for s_, e_ in hot:
            assert not any(i.startswith("scratch_") for _, i in lines[s_:e_ + 1]), (k, s_, e_)


def test_register_budgets_behind_the_occupancy_claims(kernels):
    # 512 registers per SIMD lane; .vgpr_count is the unified count (architectural + accumulation registers)
    def waves(k):
        return 512 // (-(-kernels[k]["vgpr"] // 8) * 8)
    for k in _find(kernels, "rmhmc_momentum_wave_kernel", "Li13E"):
        assert waves(k) >= 2, (k, kernels[k])                      # DESIGN: 2 waves per SIMD at D = 100
    for k in _find(kernels, "rmhmc_batch_kernel"):
        assert waves(k) >= 2, (k, kernels[k])                      # 7 waves of a workgroup on 4 SIMDs
    for k in _find(kernels, "rmhmc_mfma4_kernel") + _find(kernels, "rmhmc_mfma4x4_kernel"):
        assert waves(k) >= 1 and kernels[k]["lds"] == 0, (k, kernels[k])   # one wave per SIMD, dynamic LDS only
    for k in _find(kernels, "rmhmc_fused_kernelIfLi56ELi1E"):
        assert waves(k) >= 2, (k, kernels[k])                      # __launch_bounds__(256, 2)
    # the momentum draws of the next block can run UNDER the two-workgroup instance of the one-chain kernel (side stream): both
    # fit one SIMD.  (The one-workgroup-per-CU instance takes 288 registers and runs alone: with the wave-per-draw momentum
    # kernel the overlap no longer pays at 256 chains - profiles/r02f_cfg3_wide_ab.txt.)
    fused = max(kernels[k]["vgpr"] for k in _find(kernels, "rmhmc_fused_kernelIfLi56ELi1E"))
    mom = max(kernels[k]["vgpr"] for k in _find(kernels, "rmhmc_momentum_wave_kernel", "Li13E"))
    assert fused + mom <= 512, (fused, mom)
    for k in _find(kernels, "hmc_gauss_quad_kernelILi3ELb0ELi25E"):
        assert kernels[k]["vgpr"] <= 96, (k, kernels[k])           # cfg2: the whole state of a chain in registers (66 with the 16 lane offsets of the unrolled pass)


def test_cfg4_kernel_keeps_its_scratch_traffic_off_the_matrix_blocks():
    """The MLP MFMA kernel of BASELINE config 4 carries 55 spilled registers in its metadata (128-register cap from two 448-thread
    workgroups per CU).  What matters is WHERE they are touched: none of the basic blocks that hold its matrix instructions -
    the per-chunk forward / backward code, > 95 % of a gradient evaluation - contains a scratch access; the 135 scratch
    instructions sit in the stage bookkeeping between them (tools/scratch_in_loops.py; one assembly-only compile, ~40 s)."""
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scratch_in_loops as SL
    res = SL.blocks(os.path.join(ROOT, "hamiltorch_amd", "csrc", "mlp_mfma.hip"), "mlp_mfma_kernelILi2ELi7ELi0ELi512", ["-DHTA_MLP_SINGLE"])
    hot = [v for v in res.values() if v["mfma"]]
    assert sum(v["mfma"] for v in hot) >= 100
    assert all(v["scratch"] == 0 for v in hot), [v for v in hot if v["scratch"]]
    assert sum(v["scratch"] for v in res.values()) <= 200


def test_notebook_model_kernel_keeps_scratch_off_its_matrix_blocks():
    """csrc/mlp3_mfma.hip: the basic blocks that hold the 560 matrix instructions of a gradient pass (twice: the stage loop's call
    site and the log-p / evaluation one) touch scratch at most a handful of times; what the allocator spills sits in the
    per-trajectory code (tools/scratch_in_loops.py; one assembly-only compile)."""
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scratch_in_loops as SL
    res = SL.blocks(os.path.join(ROOT, "hamiltorch_amd", "csrc", "mlp3_mfma.hip"), "mlp3_mfma_kernelILi0E", [])
    hot = [v for v in res.values() if v["mfma"]]
    assert sum(v["mfma"] for v in hot) == 2 * 560
    # (8 before the stage loop merged the two kicks that share a gradient - a second run-time kick coefficient; 9 with it.  The
    #  guard is against the first version's hundreds: its kick reloaded the momentum's 28 registers from scratch every pass.)
    assert sum(v["scratch"] for v in hot) <= 12, [v for v in hot if v["scratch"]]


def test_fast_solve_phase_functions_stay_inside_the_caller_saved_registers():
    """Round 6: the phases of the metric kernel's fast solve (csrc/rmhmc_metric_mfma.hip: ph_fast_vt / ph_fast_form / ph_fast_second /
    ph_fast_chain) are out-of-line functions.  A function that needs more than the 80 caller-saved VGPRs saves and restores the others
    through scratch memory in its prologue / epilogue - measured at 4-7 k cycles per call with 16 waves on the CU (the first bfloat16
    instance of ph_fast_second did: the phase got slower than the fp32 product it replaced).  The instances BASELINE config 3 runs
    (leading dimension 116 as a compile-time constant) must not touch scratch at all."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_of
    obj = os.path.join(ROOT, "hamiltorch_amd", "csrc", "build", "rmhmc_metric_mfma.o")
    if not os.path.exists(obj):
        pytest.skip("needs the object file of rmhmc_metric_mfma.hip")
    for pat in ["ph_fast_vt", "ph_fast_formILi116ELb1ELb1E", "ph_fast_formILi116ELb1ELb0E", "ph_fast_formILi116ELb0ELb0E", "ph_fast_secondILi116EE", "ph_fast_second_stripILi116E", "ph_fast_chainE",
                "ph_fast_form_r", "ph_fast_second_strip_r", "ph_fast_chain_r", "metric_traj_mfma_kernel"]:      # (_r: the packed-argument entries of the resident evaluations)
        name, lines = isa_of.kernel_lines(obj, pat)
        assert not any(i.startswith("scratch_") for _, i in lines), name
    # the bfloat16 instance really is one: three v_mfma_f32_16x16x32_bf16 per tile and 32 indices in the four macro-tile shapes (12 + 6 + 6 + 3 static instructions), none of the fp32 form
    _, lines = isa_of.kernel_lines(obj, "ph_fast_second_stripILi116E")
    ops = [i.split()[0] for _, i in lines]
    assert ops.count("v_mfma_f32_16x16x32_bf16") >= 27 and ops.count("v_mfma_f32_16x16x4_f32") == 0
    # and so is the formation of "metric_bx3" = 2: planes against planes, no fp32 matrix instruction
    _, lines = isa_of.kernel_lines(obj, "ph_fast_formILi116ELb1ELb1E")
    ops = [i.split()[0] for _, i in lines]
    assert ops.count("v_mfma_f32_16x16x32_bf16") >= 9 and ops.count("v_mfma_f32_16x16x4_f32") == 0

"""CPU: the boundary rules of include/hamiltorch_amd.h hold in the sources - "every call only ENQUEUES work on `stream`; it never
synchronises, never allocates persistent memory" (SURVEY 8b).  Every allocation / synchronisation / stream-creation call in csrc/ must
be on the allow-list below, each entry a documented preparation or measurement step outside the launch path."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hamiltorch_amd", "csrc")

PATTERN = re.compile(r"\b(hipMalloc\w*|hipFree\w*|hipHostMalloc|hipStreamSynchronize|hipDeviceSynchronize|hipEventSynchronize|hipStreamCreate\w*|"
                     r"hipMemcpy(?!Async|DtoH)\w*|hipMemcpyDtoH|malloc|free)\s*\(")

#: (file, call) -> why it is allowed
ALLOWED = {
    ("abi.cpp", "hipEventSynchronize"): "hta_profile_collect: measurement (waits for the event pairs it recorded itself)",
    ("rmhmc_explicit.hip", "hipStreamSynchronize"): "hta_rmhmc_gaussian_prepare: once per target - the ONE read-back of the spectrum, from which the host "
                                                    "chooses the fused route's plan (refinement count, series log-det); chol(P) is factorised on the device",
    ("rmhmc_fused.hip", "hipStreamCreateWithFlags"): "the side stream of tuning key rmhmc_overlap = 1 (off by default), created on first use per device",
    ("jit_runtime.cpp", "malloc"): "hta_jit_compile: the code object handed to the caller (host work, released by hta_jit_free)",
    ("jit_runtime.cpp", "free"): "hta_jit_free / the error paths of hta_jit_compile",
    ("jit_runtime.cpp", "hipMemcpyDtoH"): "hta_jit_load: once per module - the 32-byte info block",
}


#: calls allowed wherever they appear
ALLOWED_ANYWHERE = {"hipMemcpyFromSymbol": "the read-out of the per-phase tick counters in developer timing builds (-DHTA_*_TIMING: not compiled by the Makefile)"}


def _calls():
    found = []
    for dp, _, fs in os.walk(CSRC):
        if os.path.basename(dp) == "build":
            continue
        for f in fs:
            if not f.endswith((".hip", ".cpp", ".hpp", ".h", ".in")):
                continue
            text = open(os.path.join(dp, f)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = "\n".join(ln.split("//")[0] for ln in text.splitlines())
            for m in PATTERN.finditer(text):
                if m.group(1) not in ALLOWED_ANYWHERE:
                    found.append((f, m.group(1)))
    return found


def test_every_allocation_and_synchronisation_in_csrc_is_on_the_allow_list():
    calls = _calls()
    unknown = sorted({c for c in calls if c not in ALLOWED})
    assert not unknown, "calls that allocate / synchronise behind the C ABI and are not documented: %r" % (unknown,)
    stale = sorted(k for k in ALLOWED if k not in calls)
    assert not stale, "allow-list entries with no call left (delete them): %r" % (stale,)


def test_integration_md_describes_the_current_abi():
    from hamiltorch_amd import _abi
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "hipMallocAsync" not in text and "three documented exceptions" not in text          # the ABI 9 paragraph
    assert "`hta_abi_version` (%d)" % _abi.ABI_VERSION in text
    for name in ("hta_jit_compile", "hta_jit_hmc_sample", "hta_jit_derivs", "hta_jit_rmhmc_sample", "hta_metric_eval_workspace_bytes"):
        assert name in text, name

"""CPU: the C-ABI library loads and exports every symbol include/hamiltorch_amd.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "hamiltorch_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hta_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from hamiltorch_amd import _abi
    lib = _abi.load()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert lib.hta_abi_version() == _abi.ABI_VERSION


def test_binding_covers_header():
    from hamiltorch_amd import _abi
    bound = set(_abi.PLAIN_SYMBOLS)
    for n in _abi.TYPED_SYMBOLS:
        bound |= {n + "_f32", n + "_f64"}
    assert bound == set(declared_symbols())


def test_error_channel_without_gpu():
    """bad arguments are rejected before any launch (no GPU needed)."""
    from hamiltorch_amd import _abi
    lib = _abi.load()
    rc = lib.hta_momentum_resample_f32(None, 0, None, 0, 0, 0, 0, 0, None)
    assert rc == -1 and b"bad shape" in lib.hta_last_error()
    assert lib.hta_set_tuning(b"nope", 1) == -1


def test_product_does_not_import_oracle():
    """the shipped package never touches oracle/ (only tests, smoke and bench's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "hamiltorch_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert "hmc_oracle" not in txt and "import oracle" not in txt, f


def test_every_tuning_key_is_documented_and_accepted():
    """every key hta_set_tuning accepts (the table in csrc/abi.cpp) is described in the header; hta_get_tuning reads a key
    back, hta_reset_tuning restores the table's defaults (no GPU needed: the keys only select routes / launch shapes)."""
    import ctypes
    from hamiltorch_amd import _abi
    src = open(os.path.join(ROOT, "hamiltorch_amd", "csrc", "abi.cpp")).read()
    table = re.findall(r'\{"([a-z0-9_]+)", &hta::g_[a-z0-9_]+, (\d+)\}', src)
    keys = [k for k, _ in table]
    assert len(keys) >= 18 and len(set(keys)) == len(keys)
    header = open(os.path.join(ROOT, "include", "hamiltorch_amd.h")).read()
    for k in keys + ["profile"]:
        assert '"%s"' % k in header, "tuning key %s is not documented in include/hamiltorch_amd.h" % k
    lib = _abi.load()
    assert lib.hta_reset_tuning() == 0
    for k, d in table:
        assert _abi.get_tuning(k) == int(d), k
        _abi.set_tuning(k, int(d) + 1)
        assert _abi.get_tuning(k) == int(d) + 1
    _abi.reset_tuning()
    for k, d in table:
        assert _abi.get_tuning(k) == int(d), k
    v = ctypes.c_int(0)
    assert lib.hta_get_tuning(b"nope", ctypes.byref(v)) == -1
    assert _abi.last_route() == ""                       # nothing dispatched yet in this thread (no GPU here)


def test_tuning_defaults_from_the_environment():
    """HTA_TUNING_DEFAULTS moves the default of the named route keys for the process: in force when the library is loaded and
    restored by hta_reset_tuning (A/B runs of unmodified test files under another route); unknown names are ignored."""
    import subprocess
    import sys
    code = ("from hamiltorch_amd import _abi\n"
            "g = _abi.get_tuning\n"
            "a = (g('quad_variant'), g('rmhmc_uv'), g('gauss_eig'))\n"
            "_abi.set_tuning('quad_variant', 0); _abi.set_tuning('rmhmc_uv', 1); _abi.reset_tuning()\n"
            "print(a, (g('quad_variant'), g('rmhmc_uv'), g('gauss_eig')))\n")
    env = dict(os.environ, HTA_TUNING_DEFAULTS="quad_variant=3,nope=3,rmhmc_uv=2", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    assert out.strip() == "(3, 2, 1) (3, 2, 1)", out


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """without the built HIP extension the package raises (no CPU path, no silent fallback)."""
    import pytest
    from hamiltorch_amd import _abi
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", str(tmp_path / "libhamiltorch_amd.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _abi.load()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _abi.set_tuning("rmhmc_batch", 1)

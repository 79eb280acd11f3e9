"""ctypes binding of libhamiltorch_amd.so (the C ABI declared in include/hamiltorch_amd.h).

The library is the product: there is no CPU or torch fallback for the kernels it exports.
If it is missing, or asked to run on a non-ROCm tensor, the calls below raise.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhamiltorch_amd.so")
ABI_VERSION = 11

MASS_NONE, MASS_DIAG, MASS_FULL = 0, 1, 2

_lib = None
_lock = threading.Lock()

c_i64, c_int, c_u64, c_u32, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p
c_f32, c_f64 = ctypes.c_float, ctypes.c_double


class HtaDeviceInfo(ctypes.Structure):
    _fields_ = [("abi_version", c_int), ("device", c_int), ("compute_units", c_int), ("wavefront_size", c_int),
                ("lds_bytes_per_cu", c_int), ("clock_khz", c_int), ("hbm_bytes", c_i64), ("arch", ctypes.c_char * 64)]


class HtaMetricArgs(ctypes.Structure):
    """include/hamiltorch_amd.h: HtaMetricArgs."""
    _fields_ = [("B", c_i64), ("D", ctypes.c_int32), ("metric", ctypes.c_int32), ("Hs", c_vp), ("hs_stride", c_i64),
                ("alpha", c_f64), ("has_jitter", ctypes.c_int32), ("max_sweeps", ctypes.c_int32), ("jitter", c_f64),
                ("seed", c_u64), ("chain_offset", c_u64), ("draw", c_u32), ("sub", c_u32), ("X", c_vp), ("Pm", c_vp),
                ("mu", c_vp), ("log_norm", c_f64), ("m", c_vp), ("p_out", c_vp), ("x_out", c_vp), ("G_out", c_vp),
                ("lam_out", c_vp), ("V_out", c_vp), ("L_out", c_vp), ("logdet_out", c_vp), ("quad_out", c_vp),
                ("H_out", c_vp), ("logp_out", c_vp), ("upd_x", c_vp), ("cx", c_f64), ("upd_g", c_vp), ("cg", c_f64),
                ("V0", c_vp), ("lam0", c_vp), ("lamraw_out", c_vp), ("dmetric_out", c_vp), ("v0_stride", c_i64),
                ("workspace", c_vp), ("workspace_bytes", c_i64)]


class HtaCbHmcArgs(ctypes.Structure):
    """csrc/jit/jit_args.h: HtaCbHmcArgs."""
    _fields_ = [("cur", c_vp), ("init", c_vp), ("inv_mass", c_vp), ("mass_factor", c_vp), ("samples", c_vp),
                ("reject_count", c_vp), ("H_old", c_vp), ("H_new", c_vp), ("accept", c_vp), ("gcur", c_vp), ("lp_out", c_vp),
                ("C", ctypes.c_longlong), ("eps", c_f64), ("seed", c_u64), ("chain_offset", c_u64), ("L", c_int),
                ("n_traj", c_int), ("traj_offset", c_int), ("burn", c_int), ("resume", c_int), ("reserved", c_int), ("pre", c_vp),
                ("pre_bytes", ctypes.c_longlong)]


class HtaCbRmhmcArgs(ctypes.Structure):
    """csrc/jit/jit_args.h: HtaCbRmhmcArgs."""
    _fields_ = [("cur", c_vp), ("init", c_vp), ("samples", c_vp), ("reject_count", c_vp), ("H_old", c_vp), ("H_new", c_vp),
                ("accept", c_vp), ("lp_out", c_vp), ("C", ctypes.c_longlong), ("eps", c_f64), ("alpha", c_f64), ("jitter", c_f64),
                ("omega", c_f64), ("seed", c_u64), ("chain_offset", c_u64), ("L", c_int), ("n_traj", c_int), ("traj_offset", c_int),
                ("burn", c_int)]


class HtaCbDerivArgs(ctypes.Structure):
    """csrc/jit/jit_args.h: HtaCbDerivArgs."""
    _fields_ = [("theta", c_vp), ("logp", c_vp), ("grad", c_vp), ("neg_hess", c_vp), ("M", c_vp), ("contract", c_vp),
                ("upd", c_vp), ("grad_in", c_vp), ("coef", c_f64), ("C", ctypes.c_longlong)]


METRIC_HESSIAN, METRIC_SOFTABS = 0, 1
SPLIT_SYMMETRIC, SPLIT_RAND, SPLIT_KMID = 0, 1, 2


def _sig(scalar):
    """argtypes of the dtype-suffixed entry points (scalar = c_float | c_double)."""
    return {
        "hta_momentum_resample": [c_vp, c_int, c_vp, c_i64, c_int, c_u64, c_u64, c_u32, c_vp],
        "hta_kick_drift": [c_vp, c_vp, c_vp, scalar, scalar, c_int, c_vp, c_i64, c_int, c_vp],
        "hta_hamiltonian": [c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_int, c_vp],
        "hta_mh_select": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_u64,
                          c_u64, c_vp],
        "hta_momentum_resample_at": [c_vp, c_int, c_vp, c_i64, c_int, c_u64, c_u64, c_vp, c_vp],
        "hta_mh_select_at": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_int, c_u64, c_u64,
                             c_vp],
        "hta_hmc_gaussian_sample": [c_vp, c_vp, c_vp, c_vp, scalar, c_int, c_vp, c_vp, c_i64, c_int, c_int, scalar,
                                    c_int, c_int, c_int, c_u64, c_u64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
        "hta_hmc_gaussian_leapfrog": [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_int, c_int, scalar, c_vp, c_vp,
                                      c_vp],
        "hta_hmc_gaussian_prepare": [c_vp, c_int, c_vp, c_i64, c_int, c_int, c_vp, c_i64, c_vp],
        "hta_metric_eval": [ctypes.POINTER(HtaMetricArgs), c_vp],
        "hta_mlp_hmc_sample": [c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int,
                               ctypes.POINTER(scalar), scalar, scalar, c_int, c_vp, c_vp, c_int, c_int, scalar, c_int,
                               c_int, c_int, c_u64, c_u64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
        "hta_mlp_logp_grad": [c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int,
                              ctypes.POINTER(scalar), scalar, scalar, c_vp, c_vp, c_vp],
        "hta_netn_hmc_sample": [c_vp, c_vp, c_i64, c_int, ctypes.POINTER(c_int), c_int, c_int, c_vp, c_vp, c_int, c_int, c_int,
                                ctypes.POINTER(scalar), scalar, scalar, c_int, c_vp, c_vp, c_int, c_int, scalar, c_int,
                                c_int, c_int, c_u64, c_u64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp],
        "hta_netn_logp_grad": [c_vp, c_i64, c_int, ctypes.POINTER(c_int), c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int,
                               ctypes.POINTER(scalar), scalar, scalar, c_vp, c_vp, c_vp, c_i64, c_vp],
        "hta_net_forward": [c_vp, c_i64, c_int, ctypes.POINTER(c_int), c_int, c_vp, c_int, c_vp, c_vp],
        "hta_rmhmc_gaussian_leapfrog": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_f64, c_int, c_f64, c_u64, c_u64,
                                        c_u32, c_i64, c_int, c_int, c_f64, c_f64, c_vp, c_vp, c_vp, c_i64, c_vp],
        "hta_rmhmc_binding_rotation": [c_vp, c_vp, c_vp, c_vp, c_i64, c_f64, c_f64, c_vp],
        "hta_rmhmc_gaussian_sample": [c_vp, c_vp, c_vp, c_vp, c_f64, c_int, c_f64, c_int, c_f64, c_i64, c_int, c_int,
                                      c_f64, c_f64, c_int, c_int, c_int, c_u64, c_u64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                      c_vp, c_i64, c_vp],
        "hta_rmhmc_gaussian_prepare": [c_vp, c_vp, c_int, c_f64, c_int, c_f64, c_i64, c_int, c_vp, c_i64, c_vp],
    }


#: every symbol include/hamiltorch_amd.h declares (checked by tests/test_abi_symbols.py)
PLAIN_SYMBOLS = ["hta_abi_version", "hta_last_error", "hta_device_info", "hta_set_tuning", "hta_get_tuning", "hta_reset_tuning",
                 "hta_last_route", "hta_profile_collect", "hta_counter_add", "hta_run_begin", "hta_rmhmc_gaussian_forget", "hta_hmc_gaussian_forget",
                 "hta_hmc_gaussian_workspace_bytes", "hta_rmhmc_workspace_bytes", "hta_hmc_gaussian_status_offset",
                 "hta_metric_eval_workspace_bytes", "hta_netn_hmc_workspace_bytes",
                 "hta_jit_available", "hta_jit_last_log", "hta_jit_note_fallback", "hta_jit_compile", "hta_jit_free", "hta_jit_load", "hta_jit_unload",
                 "hta_jit_module_info", "hta_jit_hmc_workspace_bytes", "hta_jit_hmc_predraw_bytes", "hta_jit_hmc_sample", "hta_jit_derivs",
                 "hta_jit_rmhmc_workspace_bytes", "hta_jit_rmhmc_sample"]
TYPED_SYMBOLS = sorted(_sig(c_f32).keys())


def load():
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "hamiltorch_amd: %s is missing -- build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C hamiltorch_amd/csrc). "
                "There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        lib.hta_abi_version.restype = c_int
        lib.hta_last_error.restype = ctypes.c_char_p
        lib.hta_device_info.argtypes = [c_int, ctypes.POINTER(HtaDeviceInfo)]
        lib.hta_set_tuning.argtypes = [ctypes.c_char_p, c_int]
        lib.hta_get_tuning.argtypes = [ctypes.c_char_p, ctypes.POINTER(c_int)]
        lib.hta_reset_tuning.argtypes = []
        lib.hta_last_route.argtypes = []
        lib.hta_last_route.restype = ctypes.c_char_p
        lib.hta_profile_collect.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]
        lib.hta_hmc_gaussian_workspace_bytes.argtypes = [c_i64, c_int, c_int, c_int]
        lib.hta_hmc_gaussian_workspace_bytes.restype = c_i64
        lib.hta_hmc_gaussian_status_offset.argtypes = [c_i64, c_int, c_int, c_int]
        lib.hta_hmc_gaussian_status_offset.restype = c_i64
        lib.hta_metric_eval_workspace_bytes.argtypes = [c_i64, c_int, c_int]
        lib.hta_metric_eval_workspace_bytes.restype = c_i64
        lib.hta_netn_hmc_workspace_bytes.argtypes = [c_i64, c_int, ctypes.POINTER(c_int), c_int]
        lib.hta_netn_hmc_workspace_bytes.restype = c_i64
        lib.hta_counter_add.argtypes = [c_vp, c_int, c_vp]
        lib.hta_counter_add.restype = c_int
        lib.hta_run_begin.argtypes = [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]
        lib.hta_run_begin.restype = c_int
        lib.hta_rmhmc_workspace_bytes.argtypes = [c_i64, c_int, c_int]
        lib.hta_rmhmc_workspace_bytes.restype = c_i64
        lib.hta_rmhmc_gaussian_forget.argtypes = [c_vp]
        lib.hta_rmhmc_gaussian_forget.restype = c_int
        lib.hta_hmc_gaussian_forget.argtypes = [c_vp]
        lib.hta_hmc_gaussian_forget.restype = c_int
        cpp = ctypes.POINTER(ctypes.c_char_p)
        lib.hta_jit_available.argtypes = []
        lib.hta_jit_note_fallback.argtypes = [ctypes.c_char_p]
        lib.hta_jit_last_log.argtypes = []
        lib.hta_jit_last_log.restype = ctypes.c_char_p
        lib.hta_jit_compile.argtypes = [ctypes.c_char_p, ctypes.c_char_p, c_int, cpp, cpp, c_int, cpp, ctypes.POINTER(c_vp),
                                        ctypes.POINTER(c_i64)]
        lib.hta_jit_free.argtypes = [c_vp]
        lib.hta_jit_free.restype = None
        lib.hta_jit_load.argtypes = [c_vp, c_i64, ctypes.POINTER(c_vp)]
        lib.hta_jit_unload.argtypes = [c_vp]
        lib.hta_jit_module_info.argtypes = [c_vp, ctypes.POINTER(c_int)]
        lib.hta_jit_hmc_workspace_bytes.argtypes = [c_i64, c_int, c_int]
        lib.hta_jit_hmc_workspace_bytes.restype = c_i64
        lib.hta_jit_hmc_predraw_bytes.argtypes = [c_i64, c_int, c_int, c_int]
        lib.hta_jit_hmc_predraw_bytes.restype = c_i64
        lib.hta_jit_hmc_sample.argtypes = [c_vp, ctypes.POINTER(HtaCbHmcArgs), c_int, c_int, c_int, c_vp, c_i64, c_vp]
        lib.hta_jit_derivs.argtypes = [c_vp, ctypes.POINTER(HtaCbDerivArgs), c_int, c_int, c_int, c_vp]
        lib.hta_jit_rmhmc_workspace_bytes.argtypes = [c_i64, c_int, c_int]
        lib.hta_jit_rmhmc_workspace_bytes.restype = c_i64
        lib.hta_jit_rmhmc_sample.argtypes = [c_vp, ctypes.POINTER(HtaCbRmhmcArgs), c_int, c_int, c_int, c_vp, c_i64, c_vp]
        for suf, scalar in (("f32", c_f32), ("f64", c_f64)):
            for name, args in _sig(scalar).items():
                fn = getattr(lib, "%s_%s" % (name, suf))
                fn.argtypes = args
                fn.restype = c_int
        if lib.hta_abi_version() != ABI_VERSION:
            raise RuntimeError("hamiltorch_amd: ABI mismatch (library %d, binding %d) -- rebuild"
                               % (lib.hta_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def last_error() -> str:
    return load().hta_last_error().decode("utf-8", "replace")


class InvalidArguments(RuntimeError):
    """HTA_ERR_INVALID (-1): the library refused the arguments before launching anything."""


def _check(rc, what):
    if rc != 0:
        raise (InvalidArguments if rc == -1 else RuntimeError)("hamiltorch_amd: %s failed (%d): %s" % (what, rc, last_error()))


def _suffix(t: torch.Tensor) -> str:
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError("hamiltorch_amd kernels compute in float32 or float64, got %s" % t.dtype)


def require_device(t: torch.Tensor, what="tensor"):
    """The kernels only exist for gfx950; anything else is an error, never a silent fallback."""
    if not t.is_cuda:
        raise RuntimeError(
            "hamiltorch_amd: %s lives on '%s'; this engine runs on an AMD Instinct GPU only "
            "(move params_init and the tensors your log_prob_func closes over to 'cuda'). "
            "There is no CPU fallback." % (what, t.device))


def _p(t, like=None):
    """device pointer of a contiguous tensor (or NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("hamiltorch_amd: non-contiguous tensor passed to the C ABI")
    if like is not None and (t.dtype != like.dtype or t.device != like.device):
        raise ValueError("hamiltorch_amd: dtype/device mismatch between kernel operands (%s/%s vs %s/%s)"
                         % (t.dtype, t.device, like.dtype, like.device))
    return c_vp(t.data_ptr())


def _stream(t):
    return c_vp(torch.cuda.current_stream(t.device).cuda_stream)


def device_info(device=0) -> dict:
    info = HtaDeviceInfo()
    _check(load().hta_device_info(int(device), ctypes.byref(info)), "hta_device_info")
    return {k: (getattr(info, k).decode() if k == "arch" else getattr(info, k)) for k, _ in HtaDeviceInfo._fields_}


def set_tuning(key: str, value: int):
    _check(load().hta_set_tuning(key.encode(), int(value)), "hta_set_tuning")


def get_tuning(key: str) -> int:
    v = c_int(0)
    _check(load().hta_get_tuning(key.encode(), ctypes.byref(v)), "hta_get_tuning")
    return v.value


def reset_tuning():
    """Every route key back to its default (the keys are process-global: test fixtures call this between tests)."""
    _check(load().hta_reset_tuning(), "hta_reset_tuning")


def last_route() -> str:
    """The dominant kernel this thread's last sampling / evaluation call dispatched to, with its template arguments."""
    return load().hta_last_route().decode()


# ---- thin typed wrappers ------------------------------------------------------------------------
def momentum_resample(p, mass_kind, mass_factor, seed, chain_offset, draw):
    require_device(p, "momentum")
    C, D = p.shape
    fn = getattr(load(), "hta_momentum_resample_" + _suffix(p))
    with torch.cuda.device(p.device):
        _check(fn(_p(p), mass_kind, _p(mass_factor, p), C, D, seed, chain_offset, draw & 0xFFFFFFFF, _stream(p)),
               "hta_momentum_resample")


def kick_drift(theta, p, grad, kick, drift, mass_kind, inv_mass):
    require_device(theta, "params")
    C, D = theta.shape
    fn = getattr(load(), "hta_kick_drift_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), _p(p, theta), _p(grad, theta), float(kick), float(drift), mass_kind,
                  _p(inv_mass, theta), C, D, _stream(theta)), "hta_kick_drift")


def hamiltonian(p, logp, mass_kind, inv_mass, out):
    require_device(p, "momentum")
    C, D = p.shape
    fn = getattr(load(), "hta_hamiltonian_" + _suffix(p))
    with torch.cuda.device(p.device):
        _check(fn(_p(p), _p(logp, p), mass_kind, _p(inv_mass, p), _p(out, p), C, D, _stream(p)), "hta_hamiltonian")


def momentum_resample_at(p, mass_kind, mass_factor, seed, chain_offset, n_dev):
    """gibbs() with the trajectory index in device memory (int32 tensor): capturable in a HIP graph."""
    require_device(p, "momentum")
    C, D = p.shape
    fn = getattr(load(), "hta_momentum_resample_at_" + _suffix(p))
    with torch.cuda.device(p.device):
        _check(fn(_p(p), mass_kind, _p(mass_factor, p), C, D, seed, chain_offset, c_vp(n_dev.data_ptr()), _stream(p)),
               "hta_momentum_resample_at")


def mh_select_at(cur, prop, init, H_old, H_new, logp_new, samples_base, reject_count, accept, n_dev, burn, seed, chain_offset):
    require_device(cur, "params")
    C, D = cur.shape
    fn = getattr(load(), "hta_mh_select_at_" + _suffix(cur))
    with torch.cuda.device(cur.device):
        _check(fn(_p(cur), _p(prop, cur), _p(init, cur), _p(H_old, cur), _p(H_new, cur), _p(logp_new, cur),
                  _p(samples_base, cur), _p(reject_count), _p(accept), C, D, c_vp(n_dev.data_ptr()), int(burn), seed,
                  chain_offset, _stream(cur)), "hta_mh_select_at")


def counter_add(n_dev, delta=1):
    with torch.cuda.device(n_dev.device):
        _check(load().hta_counter_add(c_vp(n_dev.data_ptr()), int(delta), _stream(n_dev)), "hta_counter_add")


def mh_select(cur, prop, init, H_old, H_new, logp_new, row, reject_count, accept, n, burn, seed, chain_offset):
    require_device(cur, "params")
    C, D = cur.shape
    fn = getattr(load(), "hta_mh_select_" + _suffix(cur))
    with torch.cuda.device(cur.device):
        _check(fn(_p(cur), _p(prop, cur), _p(init, cur), _p(H_old, cur), _p(H_new, cur), _p(logp_new, cur),
                  _p(row, cur), _p(reject_count), _p(accept), C, D, int(n), int(burn), seed, chain_offset,
                  _stream(cur)), "hta_mh_select")


def profile_collect():
    """(summed ms, launches) of the kernels bracketed since set_tuning("profile", 1)."""
    ms, n = ctypes.c_double(0), c_int(0)
    _check(load().hta_profile_collect(ctypes.byref(ms), ctypes.byref(n)), "hta_profile_collect")
    return ms.value, n.value


def gaussian_workspace_bytes(C, D, n_traj, itemsize):
    return int(load().hta_hmc_gaussian_workspace_bytes(int(C), int(D), int(n_traj), int(itemsize)))


def hmc_gaussian_sample(theta, theta_init, P, mu, log_norm, mass_kind, inv_mass, mass_factor, L, eps, n_traj,
                        traj_offset, burn, seed, chain_offset, samples, reject_count, H_old=None, H_new=None,
                        accept=None, workspace=None):
    require_device(theta, "params")
    C, D = theta.shape
    fn = getattr(load(), "hta_hmc_gaussian_sample_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), _p(theta_init, theta), _p(P, theta), _p(mu, theta), float(log_norm), mass_kind,
                  _p(inv_mass, theta), _p(mass_factor, theta), C, D, int(L), float(eps), int(n_traj),
                  int(traj_offset), int(burn), seed, chain_offset, _p(samples, theta), _p(reject_count),
                  _p(H_old, theta), _p(H_new, theta), _p(accept),
                  None if workspace is None else c_vp(workspace.data_ptr()),
                  0 if workspace is None else workspace.numel() * workspace.element_size(),
                  _stream(theta)), "hta_hmc_gaussian_sample")


def run_begin(init, cur, row0, reject_count):
    """hta_run_begin: cur <- init, row0 <- init, reject_count <- 0 in one launch (the first lines of every run)."""
    require_device(init, "params_init")
    C, D = init.shape
    assert cur.is_contiguous() and init.is_contiguous() and (row0 is None or row0.is_contiguous())
    with torch.cuda.device(init.device):
        _check(load().hta_run_begin(_p(init), _p(cur, init), _p(row0, init), _p(reject_count), C, D, init.element_size(),
                                    _stream(init)), "hta_run_begin")


def hmc_gaussian_prepare(like, P, mass_kind, mass_factor, C, D, n_traj, workspace):
    """hta_hmc_gaussian_prepare: the eig block of `workspace` filled once for sample calls of `n_traj` trajectories on this
    target (contract in include/hamiltorch_amd.h: P / mass_factor unchanged until the next prepare or forget)."""
    require_device(like, "params")
    fn = getattr(load(), "hta_hmc_gaussian_prepare_" + _suffix(like))
    with torch.cuda.device(like.device):
        _check(fn(_p(P, like), int(mass_kind), _p(mass_factor, like), int(C), int(D), int(n_traj), c_vp(workspace.data_ptr()),
                  workspace.numel() * workspace.element_size(), _stream(like)), "hta_hmc_gaussian_prepare")


def hmc_gaussian_forget(workspace):
    # (the library keys its plans by (current device, pointer): forget on the workspace's device, whatever is current - a
    #  finaliser may run while another GPU is selected)
    with torch.cuda.device(workspace.device):
        _check(load().hta_hmc_gaussian_forget(c_vp(workspace.data_ptr())), "hta_hmc_gaussian_forget")


def hmc_gaussian_leapfrog(theta, p, P, mu, mass_kind, inv_mass, steps, eps, path_theta=None, path_p=None):
    require_device(theta, "params")
    C, D = theta.shape
    fn = getattr(load(), "hta_hmc_gaussian_leapfrog_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), _p(p, theta), _p(P, theta), _p(mu, theta), mass_kind, _p(inv_mass, theta), C, D,
                  int(steps), float(eps), _p(path_theta, theta), _p(path_p, theta), _stream(theta)),
               "hta_hmc_gaussian_leapfrog")


# ---- RMHMC ---------------------------------------------------------------------------------------
def metric_eval(like, B, D, metric, Hs, hs_stride, alpha, jitter=None, seed=0, chain_offset=0, draw=0, sub=0, X=None,
                Pm=None, mu=None, log_norm=0.0, m=None, p_out=None, x_out=None, G_out=None, lam_out=None, V_out=None,
                L_out=None, logdet_out=None, quad_out=None, H_out=None, logp_out=None, upd_x=None, cx=0.0, upd_g=None,
                cg=0.0, max_sweeps=0, V0=None, lam0=None, lamraw_out=None, dmetric_out=None, v0_stride=0, workspace=None):
    """One batched metric evaluation (see HtaMetricArgs in include/hamiltorch_amd.h).  `like` fixes dtype/device."""
    require_device(like, "params")
    a = HtaMetricArgs()
    a.B, a.D, a.metric, a.hs_stride, a.alpha = int(B), int(D), int(metric), int(hs_stride), float(alpha if alpha is not None else 0.0)
    a.has_jitter, a.jitter, a.max_sweeps = (0, 0.0, int(max_sweeps)) if jitter is None else (1, float(jitter), int(max_sweeps))
    a.seed, a.chain_offset, a.draw, a.sub = int(seed), int(chain_offset), int(draw) & 0xFFFFFFFF, int(sub)
    a.log_norm, a.cx, a.cg = float(log_norm), float(cx), float(cg)
    a.v0_stride = int(v0_stride)
    keep = []
    for name, t in (("Hs", Hs), ("X", X), ("Pm", Pm), ("mu", mu), ("m", m), ("p_out", p_out), ("x_out", x_out),
                    ("G_out", G_out), ("lam_out", lam_out), ("V_out", V_out), ("L_out", L_out),
                    ("logdet_out", logdet_out), ("quad_out", quad_out), ("H_out", H_out), ("logp_out", logp_out),
                    ("upd_x", upd_x), ("upd_g", upd_g), ("V0", V0), ("lam0", lam0), ("lamraw_out", lamraw_out),
                    ("dmetric_out", dmetric_out)):
        setattr(a, name, None if t is None else _p(t, like).value)
        keep.append(t)
    with torch.cuda.device(like.device):       # (the size query reads the CURRENT device's geometry: ask on the tensors' device)
        ws = scratch(like, metric_eval_workspace_bytes(B, D, like.element_size()), "metric") if workspace is None else workspace
    a.workspace, a.workspace_bytes = (None, 0) if ws is None else (ws.data_ptr(), ws.numel() * ws.element_size())
    fn = getattr(load(), "hta_metric_eval_" + _suffix(like))
    with torch.cuda.device(like.device):
        _check(fn(ctypes.byref(a), _stream(like)), "hta_metric_eval")


def metric_eval_workspace_bytes(B, D, itemsize):
    """Bytes of HtaMetricArgs::workspace for B systems of size D (0 while both matrices of a system fit one CU's LDS)."""
    return int(load().hta_metric_eval_workspace_bytes(int(B), int(D), int(itemsize)))


_scratch = {}


def scratch(like, nbytes, tag):
    """Caller-side scratch for the ABI's `*_workspace_bytes` contracts (ABI 10: the library allocates nothing): one uint8 tensor per
    (device, stream, purpose), grown on demand, owned by torch's allocator - visible in its accounting, capturable in a HIP graph.
    Calls on one stream are ordered, so they may share it; another stream gets another buffer."""
    if nbytes <= 0:
        return None
    key = (like.device, torch.cuda.current_stream(like.device).cuda_stream, tag)
    t = _scratch.get(key)
    if t is None or t.numel() < nbytes:
        t = _scratch[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=like.device)
    return t


def free_scratch(min_bytes=0):
    """Drop the cached scratch buffers of at least `min_bytes` bytes (all of them by default): torch's allocator gets the memory back.
    sample() calls this with 256 MiB at its end - a metric evaluation at D = 1024 in fp64 needs 8.6 GB once, not for the life of the process."""
    for k in [k for k, t in _scratch.items() if t.numel() >= min_bytes]:
        del _scratch[k]


def rmhmc_workspace_bytes(C, D, itemsize, n_traj=0, cap_bytes=256 << 20):
    """Base layout plus room for the pre-drawn momenta of up to `n_traj` trajectories (capped at `cap_bytes`)."""
    base = int(load().hta_rmhmc_workspace_bytes(int(C), int(D), int(itemsize)))
    per = int(C) * int(D) * int(itemsize)
    extra = min(int(n_traj), max(0, cap_bytes // max(per, 1))) * per
    return base + extra


def rmhmc_gaussian_leapfrog(theta, p, theta_c, p_c, P, mu, metric, alpha, jitter, seed, chain_offset, draw, steps, eps,
                            omega, path_theta=None, path_p=None):
    require_device(theta, "params")
    C, D = theta.shape
    fn = getattr(load(), "hta_rmhmc_gaussian_leapfrog_" + _suffix(theta))
    ws = scratch(theta, metric_eval_workspace_bytes(C, D, theta.element_size()), "metric")
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), _p(p, theta), _p(theta_c, theta), _p(p_c, theta), _p(P, theta), _p(mu, theta), int(metric),
                  float(alpha if alpha is not None else 0.0), 0 if jitter is None else 1,
                  0.0 if jitter is None else float(jitter), int(seed), int(chain_offset), int(draw) & 0xFFFFFFFF, C, D,
                  int(steps), float(eps), float(omega), _p(path_theta, theta), _p(path_p, theta),
                  None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), _stream(theta)),
               "hta_rmhmc_gaussian_leapfrog")


def rmhmc_binding_rotation(theta, p, theta_c, p_c, eps, omega):
    """phi_C of the explicit integrator (S:435-450), in place on the augmented state."""
    require_device(theta, "params")
    fn = getattr(load(), "hta_rmhmc_binding_rotation_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), _p(p, theta), _p(theta_c, theta), _p(p_c, theta), theta.numel(), float(eps), float(omega),
                  _stream(theta)), "hta_rmhmc_binding_rotation")


def rmhmc_gaussian_sample(theta, theta_init, P, mu, log_norm, metric, alpha, jitter, L, eps, omega, n_traj, traj_offset,
                          burn, seed, chain_offset, samples, reject_count, workspace, H_old=None, H_new=None, accept=None):
    require_device(theta, "params")
    C, D = theta.shape
    fn = getattr(load(), "hta_rmhmc_gaussian_sample_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), _p(theta_init, theta), _p(P, theta), _p(mu, theta), float(log_norm), int(metric),
                  float(alpha if alpha is not None else 0.0), 0 if jitter is None else 1,
                  0.0 if jitter is None else float(jitter), C, D, int(L), float(eps), float(omega), int(n_traj),
                  int(traj_offset), int(burn), int(seed), int(chain_offset), _p(samples, theta), _p(reject_count),
                  _p(H_old, theta), _p(H_new, theta), _p(accept), c_vp(workspace.data_ptr()),
                  workspace.numel() * workspace.element_size(), _stream(theta)), "hta_rmhmc_gaussian_sample")


def rmhmc_gaussian_prepare(like, P, mu, metric, alpha, jitter, C, workspace):
    """The once-per-target setup of rmhmc_gaussian_sample (eigenbasis of P, the fused route's plan, the shared inverse) into
    `workspace`; later sample calls on that workspace with the same P / metric / alpha / jitter skip it."""
    require_device(like, "params")
    D = P.shape[0]
    fn = getattr(load(), "hta_rmhmc_gaussian_prepare_" + _suffix(like))
    with torch.cuda.device(like.device):
        _check(fn(_p(P, like), _p(mu, like), int(metric), float(alpha if alpha is not None else 0.0), 0 if jitter is None else 1,
                  0.0 if jitter is None else float(jitter), int(C), int(D), c_vp(workspace.data_ptr()),
                  workspace.numel() * workspace.element_size(), _stream(like)), "hta_rmhmc_gaussian_prepare")


def rmhmc_gaussian_forget(workspace):
    with torch.cuda.device(workspace.device):
        _check(load().hta_rmhmc_gaussian_forget(c_vp(workspace.data_ptr())), "hta_rmhmc_gaussian_forget")


# ---- Bayesian MLP (regression) -----------------------------------------------------------------------
ACTS = {"relu": 0, "tanh": 1, "sigmoid": 2}
LOSSES = {"regression": 0, "binary_class_linear_output": 1}       # HTA_LOSS_REGRESSION / HTA_LOSS_BINARY_LOGITS
NET_LOSSES = dict(LOSSES, multi_class_linear_output=2)             # + HTA_LOSS_SOFTMAX_CE (hta_netn_* only)
NETN_MAX_LAYERS, NETN_MAX_WIDTH, NETN_MAX_PARAMS, NETN_MAX_BLOCKS = 4, 64, 512, 96     # csrc/netn_hmc.hip (blocks of 4 x 4 weights)
METRIC_MFMA_MAX_D = 112                                                                # csrc/rmhmc_metric_mfma.hip (metric_warm_mfma_eligible)
MLP3_MAX_IN, MLP3_MAX_WIDTH = 4, 104                                                   # csrc/mlp3_mfma.hip (M3_NIN, M3_HMAX)


def _tau4(like, tau):
    ct = c_f32 if like.dtype == torch.float32 else c_f64
    return (ct * 4)(*[float(t) for t in tau])


def mlp_hmc_sample(theta, theta_init, n_in, H, act, X, Y, M, Nb, tau, tau_out, prior_scale, mass_kind, inv_mass,
                   mass_factor, L, eps, n_traj, traj_offset, burn, seed, chain_offset, samples, reject_count,
                   H_old=None, H_new=None, accept=None, integrator=0, loss="regression"):
    require_device(theta, "params")
    C = theta.shape[0]
    fn = getattr(load(), "hta_mlp_hmc_sample_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), _p(theta_init, theta), C, int(n_in), int(H), ACTS[act], LOSSES[loss], _p(X, theta), _p(Y, theta),
                  X.shape[0], int(M), int(Nb), _tau4(theta, tau), float(tau_out), float(prior_scale), mass_kind,
                  _p(inv_mass, theta), _p(mass_factor, theta), int(integrator), int(L), float(eps), int(n_traj), int(traj_offset),
                  int(burn), int(seed), int(chain_offset), _p(samples, theta), _p(reject_count), _p(H_old, theta),
                  _p(H_new, theta), _p(accept), _stream(theta)), "hta_mlp_hmc_sample")


def mlp_logp_grad(theta, n_in, H, act, X, Y, M, Nb, split, tau, tau_out, prior_scale, grad_out, logp_out, loss="regression"):
    require_device(theta, "params")
    C = theta.shape[0]
    fn = getattr(load(), "hta_mlp_logp_grad_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), C, int(n_in), int(H), ACTS[act], LOSSES[loss], _p(X, theta), _p(Y, theta), X.shape[0], int(M), int(Nb),
                  int(split), _tau4(theta, tau), float(tau_out), float(prior_scale), _p(grad_out, theta),
                  _p(logp_out, theta), _stream(theta)), "hta_mlp_logp_grad")


# ---- Bayesian networks of any small shape (csrc/netn_hmc.hip) ----------------------------------------
def _net_operands(like, dims, taus):
    ct = c_f32 if like.dtype == torch.float32 else c_f64
    dims = [int(d) for d in dims]
    taus = [float(t) for t in taus]
    if len(taus) != 2 * (len(dims) - 1):
        raise InvalidArguments("hta_netn: %d precisions for %d Linear layers (one per weight and per bias)" % (len(taus), len(dims) - 1))
    return len(dims) - 1, (c_int * len(dims))(*dims), (ct * len(taus))(*taus)


def netn_hmc_sample(theta, theta_init, dims, act, X, Y, M, Nb, taus, tau_out, prior_scale, mass_kind, inv_mass, mass_factor,
                    L, eps, n_traj, traj_offset, burn, seed, chain_offset, samples, reject_count, H_old=None, H_new=None,
                    accept=None, integrator=0, loss="regression", workspace=None):
    require_device(theta, "params")
    C = theta.shape[0]
    nl, cd, ct = _net_operands(theta, dims, taus)
    fn = getattr(load(), "hta_netn_hmc_sample_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        ws = scratch(theta, netn_hmc_workspace_bytes(C, dims, theta.element_size()), "netn") if workspace is None else workspace
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), _p(theta_init, theta), C, nl, cd, ACTS[act], NET_LOSSES[loss], _p(X, theta), _p(Y, theta),
                  X.shape[0], int(M), int(Nb), ct, float(tau_out), float(prior_scale), mass_kind, _p(inv_mass, theta),
                  _p(mass_factor, theta), int(integrator), int(L), float(eps), int(n_traj), int(traj_offset), int(burn),
                  int(seed), int(chain_offset), _p(samples, theta), _p(reject_count), _p(H_old, theta), _p(H_new, theta),
                  _p(accept), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), _stream(theta)),
               "hta_netn_hmc_sample")


def net_forward(theta, dims, act, X, out):
    """out[S, N, O] = f(x_p; theta_s) for every row of theta [S, D] (include/hamiltorch_amd.h: hta_net_forward)."""
    require_device(theta, "samples")
    S, N = theta.shape[0], X.shape[0]
    cd = (c_int * len(dims))(*[int(v) for v in dims])
    fn = getattr(load(), "hta_net_forward_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), S, len(dims) - 1, cd, ACTS[act], _p(X, theta), N, _p(out, theta), _stream(theta)), "hta_net_forward")


def netn_logp_grad(theta, dims, act, X, Y, M, Nb, split, taus, tau_out, prior_scale, grad_out, logp_out, loss="regression"):
    require_device(theta, "params")
    C = theta.shape[0]
    nl, cd, ct = _net_operands(theta, dims, taus)
    fn = getattr(load(), "hta_netn_logp_grad_" + _suffix(theta))
    with torch.cuda.device(theta.device):
        ws = scratch(theta, netn_hmc_workspace_bytes(C, dims, theta.element_size()), "netn")
    with torch.cuda.device(theta.device):
        _check(fn(_p(theta), C, nl, cd, ACTS[act], NET_LOSSES[loss], _p(X, theta), _p(Y, theta), X.shape[0], int(M), int(Nb),
                  int(split), ct, float(tau_out), float(prior_scale), _p(grad_out, theta), _p(logp_out, theta),
                  None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), _stream(theta)), "hta_netn_logp_grad")


def netn_hmc_workspace_bytes(C, dims, itemsize):
    """Bytes of the `workspace` argument of hta_netn_hmc_sample / hta_netn_logp_grad (non-zero on the matrix-core route's shapes only)."""
    cd = (c_int * len(dims))(*[int(v) for v in dims])
    return int(load().hta_netn_hmc_workspace_bytes(int(C), len(dims) - 1, cd, int(itemsize)))


def hmc_gaussian_status_word(workspace, C, D, n_traj, itemsize):
    """The sticky status word of a prepared Gaussian-HMC workspace as a 1-element int32 view of it (include/hamiltorch_amd.h:
    hta_hmc_gaussian_status_offset), or None for shapes without the fused route.  Non-zero after a synchronise = a fused launch
    gave up waiting for its draw records: the samples since the preparation are invalid (and NaN)."""
    off = int(load().hta_hmc_gaussian_status_offset(int(C), int(D), int(n_traj), int(itemsize)))
    if off < 0 or workspace is None or off + 4 > workspace.numel():
        return None
    return workspace[off:off + 4].view(torch.int32)

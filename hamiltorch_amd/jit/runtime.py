"""Compile traced callbacks with hipRTC (through the C ABI) and keep the results.

``compile_source`` turns generated text + the hand-written skeleton under ``csrc/jit/`` into a gfx950 code object
(``hta_jit_compile``; host work, needs no GPU - the CPU tests compile every example this way); ``Module`` loads one on a
device (``hta_jit_load``).  Code objects are cached by the hash of everything that went into them, modules by (hash, device).
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import threading

import torch

from .. import _abi
from . import emit
from .ir import Unsupported

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
_HEADERS = (("cb_math.hpp", "jit/cb_math.hpp"), ("philox.hpp", "philox.hpp"), ("mh_rules.hpp", "mh_rules.hpp"),
            ("jit_args.h", "jit/jit_args.h"))
SKELETON_HMC = "jit/hmc_callback.hip.in"
SKELETON_DERIVS = "jit/derivs_callback.hip.in"
SKELETON_RMHMC = "jit/rmhmc_callback.hip.in"
# (SLP vectorisation ON: the straight-line callback code packs into v_pk_mul / v_pk_fma pairs - 67 -> 59 instructions per
#  leapfrog step of the notebook funnel, tools/jit_isa.py; -ffp-contract=fast fuses across the generated statements)
OPTIONS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast")

_lock = threading.Lock()
_text = {}
_code = {}          # sha1 -> bytes (gfx950 code object)
_modules = {}       # (sha1, device index) -> Module
stats = {"compiled": 0, "code_hits": 0, "loaded": 0, "compile_seconds": 0.0}


class CompileError(RuntimeError):
    """hipRTC rejected generated code (a bug of the emitter, or a device function hipRTC lacks); `log` has its messages."""

    def __init__(self, msg, log):
        super().__init__(msg)
        self.log = log


def _read(rel):
    t = _text.get(rel)
    if t is None:
        with open(os.path.join(_CSRC, rel)) as f:
            t = _text[rel] = f.read()
    return t


def compile_source(generated, skeleton):
    """(sha1, code-object bytes) of `skeleton` (a file under csrc/) compiled with `generated` as "hta_cb_generated.inc"."""
    import time
    main = _read(skeleton)
    headers = [(n, _read(rel)) for n, rel in _HEADERS] + [("hta_cb_generated.inc", generated)]
    h = hashlib.sha1()
    for part in (main,) + tuple(t for _, t in headers) + OPTIONS:
        h.update(part.encode()); h.update(b"\0")
    key = h.hexdigest()
    with _lock:
        hit = _code.get(key)
    if hit is not None:
        stats["code_hits"] += 1
        return key, hit
    lib = _abi.load()
    n = len(headers)
    names = (ctypes.c_char_p * n)(*[a.encode() for a, _ in headers])
    srcs = (ctypes.c_char_p * n)(*[b.encode() for _, b in headers])
    opts = (ctypes.c_char_p * len(OPTIONS))(*[o.encode() for o in OPTIONS])
    code, size = ctypes.c_void_p(), ctypes.c_int64(0)
    t0 = time.perf_counter()
    rc = lib.hta_jit_compile(main.encode(), os.path.basename(skeleton).encode(), n, names, srcs, len(OPTIONS), opts,
                             ctypes.byref(code), ctypes.byref(size))
    stats["compile_seconds"] += time.perf_counter() - t0
    if rc != 0:
        log = lib.hta_jit_last_log().decode("utf-8", "replace")
        msg = "hamiltorch_amd: hta_jit_compile failed (%d): %s" % (rc, _abi.last_error())
        if rc == -3:
            raise Unsupported(msg)
        raise CompileError(msg + "\n" + log[-4000:], log)
    try:
        blob = ctypes.string_at(code.value, size.value)
    finally:
        lib.hta_jit_free(code)
    with _lock:
        _code[key] = blob
    stats["compiled"] += 1
    return key, blob


class Module:
    """A code object loaded on one device (hta_jit_load); unloaded with the object."""

    def __init__(self, blob, device):
        self.device = torch.device(device)
        self._blob = blob                       # (hipModuleLoadData reads it during the call only; kept for reloads / debugging)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _abi._check(_abi.load().hta_jit_load(blob, len(blob), ctypes.byref(h)), "hta_jit_load")
        self.handle = h
        info = (ctypes.c_int * 8)()
        _abi._check(_abi.load().hta_jit_module_info(h, info), "hta_jit_module_info")
        self.info = list(info)
        stats["loaded"] += 1

    def __del__(self):
        try:
            if self.handle:
                _abi.load().hta_jit_unload(self.handle)
                self.handle = None
        except Exception:       # interpreter shutdown
            pass


def module_for(key, blob, device):
    device = torch.device(device)
    k = (key, device.index if device.index is not None else torch.cuda.current_device())
    with _lock:
        m = _modules.get(k)
    if m is None:
        m = Module(blob, device)
        with _lock:
            if len(_modules) >= 64:
                _modules.pop(next(iter(_modules)))
            _modules[k] = m
    return m


def dtype_name(dtype):
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise Unsupported("compiled callbacks compute in float32 or float64, got %s" % dtype)


# ---- HMC ------------------------------------------------------------------------------------------------------------
MAX_HMC_DIM = 64                # value + gradient + state in one lane's registers
MAX_HMC_NODES = 6000            # live scalar operations of value + gradient (straight-line code in every lane)


def hmc_generated_source(traced, dtype, mass_kind):
    D = traced.D
    if D > MAX_HMC_DIM:
        raise Unsupported("D = %d: the chain-per-lane kernel holds theta, p and the gradient in registers (D <= %d)" % (D, MAX_HMC_DIM))
    grads = traced.grad()
    live = len(traced.graph.reachable([traced.value] + grads))
    if live > MAX_HMC_NODES:
        raise Unsupported("value + gradient are %d scalar operations (limit %d)" % (live, MAX_HMC_NODES))
    return emit.value_grad_source(traced.graph, traced.value, grads, dtype_name(dtype), mass_kind)


def hmc_predraw_bytes(C, D, n_traj, itemsize):
    return int(_abi.load().hta_jit_hmc_predraw_bytes(int(C), int(D), int(n_traj), int(itemsize)))


def hmc_workspace_bytes(C, D, itemsize):
    return int(_abi.load().hta_jit_hmc_workspace_bytes(int(C), int(D), int(itemsize)))


def hmc_sample(module, cur, init, mass_kind, inv_mass, mass_factor, L, eps, n_traj, traj_offset, burn, seed, chain_offset,
               samples, reject_count, workspace, H_old=None, H_new=None, accept=None, resume=False, pre=None):
    """hta_jit_hmc_sample: trajectories [traj_offset, traj_offset + n_traj) on the compiled callback, one launch."""
    _abi.require_device(cur, "params")
    C, D = cur.shape
    a = _abi.HtaCbHmcArgs()
    a.cur, a.init = cur.data_ptr(), _abi._p(init, cur).value
    a.inv_mass = None if inv_mass is None else _abi._p(inv_mass, cur).value
    a.mass_factor = None if mass_factor is None else _abi._p(mass_factor, cur).value
    a.samples = None if samples is None else _abi._p(samples, cur).value
    a.reject_count = reject_count.data_ptr()
    a.H_old = None if H_old is None else _abi._p(H_old, cur).value
    a.H_new = None if H_new is None else _abi._p(H_new, cur).value
    a.accept = None if accept is None else accept.data_ptr()
    a.C, a.eps, a.seed, a.chain_offset = C, float(eps), int(seed) & 0xFFFFFFFFFFFFFFFF, int(chain_offset)
    a.L, a.n_traj, a.traj_offset, a.burn = int(L), int(n_traj), int(traj_offset), int(burn)
    a.resume = 1 if resume else 0
    if pre is not None:
        a.pre, a.pre_bytes = pre.data_ptr(), pre.numel() * pre.element_size()
    with torch.cuda.device(cur.device):
        _abi._check(_abi.load().hta_jit_hmc_sample(module.handle, ctypes.byref(a), D, cur.element_size(), int(mass_kind),
                                                   workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                                   _abi._stream(cur)), "hta_jit_hmc_sample")


def hmc_final_logp(workspace, C, D, dtype):
    """log p at the state the last launch ended in ([C] view of the workspace's second block)."""
    item = torch.empty((), dtype=dtype).element_size()
    return workspace[C * D * item:C * D * item + C * item].view(dtype)


# ---- derivatives for the Riemannian samplers ---------------------------------------------------------------------------
MAX_DERIV_DIM = 32
MAX_DERIV_NODES = 20000         # live scalar operations of each generated function


MAX_RMHMC_DIM = 16              # a chain's matrices in one lane's registers (csrc/jit/rmhmc_callback.hip.in)


def derivs_generated_source(traced, dtype, jitter=False):
    """Value, gradient, Hessian (D reverse passes over the gradient's graph) and the third derivatives (one pass per Hessian
    entry of the lower triangle) of a traced callable, as the generated include of csrc/jit/derivs_callback.hip.in."""
    D = traced.D
    if D > MAX_DERIV_DIM:
        raise Unsupported("D = %d: third derivatives are generated entry by entry (D <= %d)" % (D, MAX_DERIV_DIM))
    g = traced.graph
    grads = traced.grad()
    hess = [g.grad(gi) for gi in grads]
    if len(g.reachable([traced.value] + grads + [hess[i][j] for i in range(D) for j in range(i + 1)])) > MAX_DERIV_NODES:
        raise Unsupported("value + gradient + Hessian exceed %d scalar operations" % MAX_DERIV_NODES)
    third = {}
    for i in range(D):
        for j in range(i + 1):
            third[(i, j)] = g.grad(hess[i][j])
            if len(g.nodes) > 40 * MAX_DERIV_NODES:
                raise Unsupported("the third derivatives exceed the graph size limit")
    if len(g.reachable([t for v in third.values() for t in v])) > MAX_DERIV_NODES:
        raise Unsupported("the third derivatives exceed %d scalar operations" % MAX_DERIV_NODES)
    return emit.derivs_source(g, traced.value, grads, hess, third, dtype_name(dtype), jitter)


def rmhmc_workspace_bytes(C, D, itemsize):
    return int(_abi.load().hta_jit_rmhmc_workspace_bytes(int(C), int(D), int(itemsize)))


def rmhmc_sample(module, cur, init, L, eps, alpha, jitter, omega, n_traj, traj_offset, burn, seed, chain_offset, samples,
                 reject_count, workspace):
    """hta_jit_rmhmc_sample: explicit soft-abs RMHMC trajectories [traj_offset, traj_offset + n_traj) on the compiled callable."""
    _abi.require_device(cur, "params")
    C, D = cur.shape
    a = _abi.HtaCbRmhmcArgs()
    a.cur, a.init = cur.data_ptr(), _abi._p(init, cur).value
    a.samples = None if samples is None else _abi._p(samples, cur).value
    a.reject_count = reject_count.data_ptr()
    a.C, a.eps, a.alpha, a.omega = C, float(eps), float(alpha), float(omega)
    a.jitter = 0.0 if jitter is None else float(jitter)
    a.seed, a.chain_offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(chain_offset)
    a.L, a.n_traj, a.traj_offset, a.burn = int(L), int(n_traj), int(traj_offset), int(burn)
    with torch.cuda.device(cur.device):
        _abi._check(_abi.load().hta_jit_rmhmc_sample(module.handle, ctypes.byref(a), D, cur.element_size(), 0 if jitter is None else 1,
                                                     workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                                     _abi._stream(cur)), "hta_jit_rmhmc_sample")


def _deriv_call(module, a, which, like):
    C, D = like.shape
    with torch.cuda.device(like.device):
        _abi._check(_abi.load().hta_jit_derivs(module.handle, ctypes.byref(a), int(which), D, like.element_size(), _abi._stream(like)),
                    "hta_jit_derivs")


def derivs(module, theta, logp, grad, neg_hess):
    """One launch: logp[C], grad[C, D], neg_hess[C, D, D] of the compiled callable at theta[C, D]."""
    _abi.require_device(theta, "params")
    a = _abi.HtaCbDerivArgs()
    a.theta, a.C = _abi._p(theta).value, theta.shape[0]
    a.logp = None if logp is None else _abi._p(logp, theta).value
    a.grad = None if grad is None else _abi._p(grad, theta).value
    a.neg_hess = _abi._p(neg_hess, theta).value
    _deriv_call(module, a, 0, theta)


def contract(module, theta, M, out=None, upd=None, grad_in=None, coef=0.0):
    """One launch: c = d < Hess log p (theta), M >|_(M fixed) per chain, to `out` and / or as upd += coef (grad_in + c)."""
    _abi.require_device(theta, "params")
    a = _abi.HtaCbDerivArgs()
    a.theta, a.C, a.M = _abi._p(theta).value, theta.shape[0], _abi._p(M, theta).value
    a.contract = None if out is None else _abi._p(out, theta).value
    a.upd = None if upd is None else _abi._p(upd, theta).value
    a.grad_in = None if grad_in is None else _abi._p(grad_in, theta).value
    a.coef = float(coef)
    _deriv_call(module, a, 1, theta)

"""Host-side utilities mirroring ``hamiltorch/util.py`` (reference lines cited as U:n).

Only what the sampling path needs is here: seeding (U:11-23), the NaN/Inf guard and
``LogProbError`` (U:92-104), the flat parameter layout helpers (U:121-141) and the
multi-chain adapters (U:385-405).  Seeding also keys the device-side Philox streams.
"""
from __future__ import annotations

import concurrent.futures
import os
import random
import sys
import threading
import weakref
import time

import numpy as np
import torch

_random_seed = 0
_call_counter = 0
_seed_lock = threading.Lock()


def set_random_seed(seed=None):
    """Seed python / numpy / torch (as U:11-20) and the Philox key used by the HIP kernels."""
    global _random_seed, _call_counter
    if seed is None:
        seed = int((time.time() * 1e6) % 1e8)
    with _seed_lock:
        _random_seed = int(seed)
        _call_counter = 0
    random.seed(seed)
    np.random.seed(int(seed) % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


set_random_seed()  # the reference seeds from the clock at import time (U:23)


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def next_stream_seed():
    """64-bit Philox key for the next ``sample`` call: a hash of (seed, number of calls since
    the seed was set), the counterpart of successive calls consuming one global generator."""
    global _call_counter
    with _seed_lock:
        k = _splitmix64((_random_seed & 0xFFFFFFFFFFFFFFFF) ^ _splitmix64(_call_counter))
        _call_counter += 1
    return k


# ---- progress reporting (U:33-89 prints a live bar; the fused kernels have no per-sample host
# ---- loop, so this is a start/finish line with the same closing message) ---------------------
class _Progress:
    def __init__(self, message, num_iters, enabled):
        self.enabled, self.n, self.t0 = enabled, num_iters, time.time()
        if enabled:
            print(message)
            sys.stdout.flush()

    def update(self, i):
        if self.enabled and self.n and (i == self.n - 1 or (i & 63) == 0):
            dt = max(time.time() - self.t0, 1e-9)
            print("%6d/%d | %.2f Samples/sec      " % (i + 1, self.n, (i + 1) / dt), end="\r")

    def end(self, message=None):
        if self.enabled:
            dt = max(time.time() - self.t0, 1e-9)
            print("%d/%d | %.2f Samples/sec" % (self.n, self.n, self.n / dt))
            if message is not None:
                print(message)


def has_nan_or_inf(value):
    """U:92-100: tensors are summed first, so inf + (-inf) also trips."""
    if torch.is_tensor(value):
        v = torch.sum(value)
        return bool(torch.isnan(v)) or bool(torch.isinf(v))
    v = float(value)
    return v != v or v in (float("inf"), float("-inf"))


class DeviceStatusError(RuntimeError):
    """A kernel reported a failure through a sticky status word in its workspace (include/hamiltorch_amd.h:
    hta_hmc_gaussian_status_offset): the samples drawn since the workspace was prepared are invalid (and were overwritten with NaN)."""


class _StatusWatch:
    """Surfaces a device-side sticky status word without synchronising.  The library never synchronises and neither does
    ``sample(verbose=False)``: after every run that used the workspace the word is copied to a pinned host int (non-blocking); the
    copy that has landed by the NEXT entry into the library is looked at there.  A word is sticky and only ever goes 0 -> non-zero, so a
    copy that has not landed yet reads as 0 and is seen one call later; ``check_device_status()`` synchronises and looks at all of them;
    ``sample(verbose=True)`` / ``debug=2`` synchronise anyway (the acceptance rate) and check on the spot."""

    def __init__(self, word, what, on_error=None):
        self.word, self.what, self.on_error = word, what, on_error
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._event = None

    def refresh(self):
        self.host.copy_(self.word, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record(torch.cuda.current_stream(self.word.device))      # the copy is complete when this event is

    def landed(self):
        """The word as of the last COMPLETED copy (0 while a copy is still in flight: the pinned int is not read under a write)."""
        if self._event is not None and not self._event.query():
            return 0
        return int(self.host[0])

    def error(self, value):
        return DeviceStatusError("hamiltorch_amd: %s reported status %d: a fused HMC launch gave up waiting for its draw records (its "
                                 "producer blocks were not scheduled - a shared or preempted GPU); the samples of that run are invalid "
                                 "and were overwritten with NaN.  hta_set_tuning('quad_fused', 0) selects the two-launch form, which "
                                 "needs no co-residency." % (self.what, value))


# WEAK references: a watch keeps a view of its (up to 128 MiB) workspace; it lives exactly as long as the workspace's cache entry on
# the target (samplers._prepared_hmc_workspace keeps the watch on the workspace's handle) - a dropped target frees both
_watches = weakref.WeakSet()
_watch_lock = threading.Lock()


def _watch_status(word, what, on_error=None):
    """`on_error()` runs when the word is reported (the owner drops its prepared workspace: the next run prepares a fresh one, which
    zeroes the word).  The caller owns the returned watch; this module only holds it weakly."""
    w = _StatusWatch(word, what, on_error)
    with _watch_lock:
        _watches.add(w)
    return w


def _poll_status():
    """Raise for any live watched word whose last completed copy is non-zero (no synchronisation)."""
    with _watch_lock:
        live = list(_watches)
    for w in live:
        v = w.landed()
        if v:
            with _watch_lock:
                _watches.discard(w)
            if w.on_error is not None:
                w.on_error()
            raise w.error(v)


def check_device_status(device=None):
    """Synchronise and raise DeviceStatusError if any kernel reported a failure since its workspace was prepared."""
    if torch.cuda.is_available():
        with _watch_lock:
            live = list(_watches)
        for w in live:
            w.refresh()
        torch.cuda.synchronize(device)
    _poll_status()


class LogProbError(Exception):
    pass


def flatten(model):
    """U:121-122: parameters() order, each flattened."""
    return torch.cat([p.flatten() for p in model.parameters()])


def unflatten(model, flattened_params):
    """U:125-136."""
    if flattened_params.dim() != 1:
        raise ValueError("Expecting a 1d flattened_params")
    out, i = [], 0
    for val in model.parameters():
        n = val.nelement()
        out.append(flattened_params[i:i + n].view_as(val))
        i += n
    return out


def update_model_params_in_place(model, params):
    """U:139-141."""
    for weights, new_w in zip(model.parameters(), params):
        weights.data = new_w


# ---- many chains (U:385-405).  The reference runs one sample() per seed, serially or in a
# ---- thread pool; here the same adapters exist, plus the on-device batched form. -------------
_chain_lock = threading.Lock()


def _accepts_seed(fn):
    import inspect
    try:
        ps = inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False
    return "seed" in ps          # an explicit parameter only: a **kwargs wrapper may forward to a function without one


def setup_chain(sampler, prior, kwargs):
    """U:385-390.  The reference's chain seeds the GLOBAL generator and then samples from it, which is racy under
    multi_chain(parallel=True) (threads interleave seeding and drawing).  Here seeding, the draw of params_init and the
    derivation of the chain's Philox key happen under one lock and the key is handed to the sampler explicitly (when `seed`
    is an explicit parameter of the sampler, as it is for this package's `sample*` functions), so a chain's result depends
    on its seed alone for serial and threaded runs.  `multi_chain(batched=True)` is a different stream layout: every
    chain's `params_init` comes from its own seed, but the device streams are keyed (seeds[0], chain index), so a batched
    run is reproducible from `seeds` as a whole and does not equal the serial per-seed results."""
    takes_seed = _accepts_seed(sampler)

    def chain(seed):
        kw = dict(kwargs)
        with _chain_lock:
            set_random_seed(seed)          # torch.manual_seed(seed) in the reference (U:387)
            params_init = prior()
            if takes_seed and kw.get("seed") is None:
                kw["seed"] = next_stream_seed()
        return sampler(params_init=params_init, **kw)
    chain._hta = (sampler, prior, kwargs)
    return chain


def multi_chain(chain, num_workers, seeds, parallel=False, batched=False):
    """``batched=True`` (extension): draw every chain's ``params_init`` from ``prior`` under its
    own seed, stack them to [C, D] and run ONE on-device batched ``sample`` call; the result has
    the reference's shape (list over chains of lists of (D,) samples)."""
    if batched:
        sampler, prior, kwargs = chain._hta
        inits = []
        for s in seeds:
            torch.manual_seed(s)
            inits.append(prior())
        set_random_seed(seeds[0])
        out = sampler(params_init=torch.stack(inits), **kwargs)
        extra = None
        if isinstance(out, tuple):
            out, extra = out
        per_chain = [[row[c] for row in out] for c in range(len(seeds))]
        return per_chain if extra is None else (per_chain, extra)
    if parallel:
        with concurrent.futures.ThreadPoolExecutor(max_workers=num_workers) as ex:
            return list(ex.map(chain, seeds))
    return [chain(s) for s in seeds]


# ---- subset order of Integrator.SPLITTING_RAND -----------------------------------------------------------
def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on Python ints: the same function as csrc/philox.hpp (host copy for host-side control flow)."""
    M32 = 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def split_permutation(seed, draw, M):
    """The order in which SPLITTING_RAND visits the M subsets during trajectory `draw` (the reference takes
    torch.randperm(M) once per leapfrog call, S:549).  Fisher-Yates on integer Philox draws keyed (seed, draw),
    purpose 4, chain key 0xFFFFFFFF: one order per trajectory for the whole batch; identical in
    csrc/philox.hpp:split_permutation (native kernels); the CPU checker under oracle/ restates it."""
    perm = list(range(M))
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    for i in range(M - 1, 0, -1):
        r = _philox4x32_10(i >> 2, int(draw) & 0xFFFFFFFF, 0xFFFFFFFF, 4, seed & 0xFFFFFFFF, seed >> 32)
        j = (r[i & 3] * (i + 1)) >> 32
        perm[i], perm[j] = perm[j], perm[i]
    return perm


# ---- HIP-graph replay of torch-side callbacks ----------------------------------------------------------------
class GraphedCallable:
    """Replays a pure tensor function (a user's log_prob_func under vmap / grad / hessian) as a captured HIP graph.

    A generic-callback trajectory calls the user's function hundreds of times on tensors of one fixed shape; in eager
    mode every call is dozens to hundreds of tiny launches and the run is bound by launch overhead, not by the GPU.
    The function is captured once per input signature (``torch.cuda.CUDAGraph``) into static buffers and replayed on
    the current stream, between the native kernels.  The outputs are static buffers: they are valid until the next
    call with the same signature, which is how the samplers use them.

    A function is *not capturable* -- and is then evaluated eagerly for good, silently (``graph_log`` keeps the reason) --
    when the capture raises (``.item()``, host tensors, data-dependent control flow: "operation not permitted when stream
    is capturing"), when it records no device work at all (an empty graph: its "static output" would never be
    refreshed), or when a replay on a perturbed input does not reproduce the eager result (a value baked in at capture
    time).  The last check is what makes a stale-output replay impossible: a graph is only used after it has been
    seen to follow its input.  ``HAMILTORCH_AMD_GRAPHS=0`` disables capturing altogether."""

    def __init__(self, fn):
        self.fn = fn
        self.cache = {}
        self.enabled = os.environ.get("HAMILTORCH_AMD_GRAPHS", "1") != "0"

    def capturable(self):
        """False once a capture of this function has failed (or graphs are disabled)."""
        return self.enabled and all(v is not False for v in self.cache.values())

    def __call__(self, *args):
        if not self.enabled or not args[0].is_cuda or torch.cuda.is_current_stream_capturing():
            return self.fn(*args)          # (inside an enclosing capture the ops are recorded inline)
        key = tuple((tuple(a.shape), a.dtype, a.device) for a in args)
        ent = self.cache.get(key)
        if ent is None:
            ent = self.cache[key] = self._capture(args)
        if ent is False:
            return self.fn(*args)
        static_in, graph, static_out = ent
        for s_, a in zip(static_in, args):
            s_.copy_(a)
        graph.replay()
        return static_out

    def _give_up(self, why):
        graph_log.append("%s: %s" % (getattr(self.fn, "__name__", type(self.fn).__name__), why))
        del graph_log[:-64]
        self.enabled = False
        return False

    def _capture(self, args):
        import warnings
        dev = args[0].device
        static_in = [a.detach().clone() for a in args]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):                      # warm-up (lazy initialisation, allocator) - errors propagate to the caller
                self.fn(*static_in)
        torch.cuda.current_stream(dev).wait_stream(side)
        try:
            graph = torch.cuda.CUDAGraph()
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                with torch.cuda.graph(graph):
                    out = self.fn(*static_in)
        except Exception as e:  # not capturable: stay eager
            if isinstance(e, torch.OutOfMemoryError):
                raise
            torch.cuda.synchronize(dev)
            return self._give_up("%s: %s" % (type(e).__name__, str(e).split("\n")[0][:160]))
        if any("graph is empty" in str(w.message).lower() for w in caught):
            return self._give_up("the capture recorded no device work (empty graph)")
        # the graph must follow its input: replay on perturbed inputs and compare with the eager evaluation
        try:
            for s_ in static_in:
                if s_.is_floating_point():
                    s_.mul_(1.0 + 2.0 ** -6).add_(2.0 ** -7)
            graph.replay()
            got = _flat_outputs(out)
            want = _flat_outputs(self.fn(*static_in))
            ok = len(got) == len(want) and all(
                g.shape == w_.shape and bool(torch.isclose(g.double(), w_.double(), rtol=1e-4, atol=1e-6, equal_nan=True).all())
                for g, w_ in zip(got, want))
        except Exception as e:
            if isinstance(e, torch.OutOfMemoryError):
                raise
            ok = False
        if not ok:
            return self._give_up("a replay on perturbed inputs does not reproduce the eager result")
        return static_in, graph, out


#: reasons of the most recent capture refusals (diagnostics; nothing is printed or warned)
graph_log = []


def _flat_outputs(out):
    if torch.is_tensor(out):
        return [out]
    if isinstance(out, (tuple, list)):
        return [t for o in out for t in _flat_outputs(o)]
    return []

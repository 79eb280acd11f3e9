"""fx trace of a user ``log_prob_func`` -> scalar graph (``ir.Graph``).

The callback contract (hamiltorch/samplers.py:272-274): a callable taking one ``(D,)`` tensor and returning a 0-d or
1-element tensor, differentiable in its argument.  ``trace_callback`` runs it ONCE under
``torch.fx.experimental.proxy_tensor.make_fx`` on an example point, which yields the flat list of aten operations the
closure performs (tensors it closes over appear as constants), and lowers that list element by element: a tensor of
the trace is an integer array of node ids of its shape, views / indexing / broadcasting are numpy operations on those
arrays, arithmetic creates graph nodes.  What cannot be lowered - an operation outside the table below, control flow
that depends on the argument's VALUES (make_fx refuses to hand out a Python bool of a traced tensor), the tuple /
``pass_grad`` protocols, graphs beyond the size limits - raises ``ir.Unsupported`` with the reason; the sampler then
stays on the torch-evaluated callback path and reports that reason in its route.

No inductor, no Triton: tracing is the only part of torch's compiler stack used, the rest is this package.
"""
from __future__ import annotations

import contextlib
import math
import operator

import numpy as np
import torch

from .ir import Graph, Unsupported

MAX_ELEMENTS = 1 << 16          # elements of one traced tensor
MAX_NODES = 200_000             # scalar nodes of the lowered value graph

aten = torch.ops.aten


class TV:
    """A traced tensor: `ids` (int64 array of the tensor's shape) of graph nodes; kind 'f' float, 'b' bool."""
    __slots__ = ("ids", "kind", "concrete")

    def __init__(self, ids, kind="f", concrete=None):
        self.ids = np.asarray(ids, dtype=np.int64)
        self.kind = kind
        self.concrete = concrete        # the torch tensor itself when the value does not depend on the argument (see _fold_constant)

    @property
    def shape(self):
        return self.ids.shape


class IV:
    """A constant integer tensor of the trace (index tensors, aranges)."""
    __slots__ = ("arr", "concrete")

    def __init__(self, arr, concrete=None):
        self.arr = np.asarray(arr, dtype=np.int64)
        self.concrete = concrete

    @property
    def shape(self):
        return self.arr.shape


@contextlib.contextmanager
def _no_distribution_validation():
    """torch.distributions validates arguments with ``if not valid.all(): raise`` - control flow on values, which a trace cannot
    record.  Off for the duration of the trace: an invalid value then gives NaN (a rejected proposal) instead of a ValueError."""
    import torch.distributions as td
    old_default = td.Distribution._validate_args
    old_sample = td.Distribution._validate_sample
    td.Distribution.set_default_validate_args(False)
    td.Distribution._validate_sample = lambda self, value: None
    try:
        yield
    finally:
        td.Distribution.set_default_validate_args(old_default)
        td.Distribution._validate_sample = old_sample


class _Lowering:
    def __init__(self, D):
        self.g = Graph(D)
        self.D = D

    # ---- helpers ---------------------------------------------------------------------------------------------
    def from_tensor(self, t):
        """A concrete tensor of the trace (a closed-over constant) -> TV / IV."""
        if t.numel() > MAX_ELEMENTS:
            raise Unsupported("a closed-over tensor of %d elements (limit %d)" % (t.numel(), MAX_ELEMENTS))
        a = t.detach().cpu()
        if a.dtype == torch.bool:
            return TV(np.vectorize(self.g.bconst, otypes=[np.int64])(a.numpy()) if a.numel() else np.zeros(a.shape, np.int64), "b", t)
        if not a.dtype.is_floating_point:
            return IV(a.numpy(), t)
        arr = a.double().numpy()
        return TV(np.vectorize(self.g.const, otypes=[np.int64])(arr) if arr.size else np.zeros(arr.shape, np.int64), "f", t)

    def scalar(self, v):
        if isinstance(v, bool):
            return TV(np.asarray(self.g.bconst(v)), "b")
        return TV(np.asarray(self.g.const(float(v))))

    def f(self, v):
        """anything -> float TV."""
        if isinstance(v, TV):
            if v.kind == "b":
                return TV(self._map1(self.g.to_float, v.ids))
            return v
        if isinstance(v, IV):
            return TV(np.vectorize(self.g.const, otypes=[np.int64])(v.arr.astype(np.float64)) if v.arr.size else np.zeros(v.shape, np.int64))
        if isinstance(v, (int, float, bool)):
            return TV(np.asarray(self.g.const(float(v))))
        raise Unsupported("operand of type %s" % type(v).__name__)

    def b(self, v):
        if isinstance(v, TV):
            return v if v.kind == "b" else TV(self._map1(self.g.to_bool, v.ids), "b")
        if isinstance(v, bool):
            return TV(np.asarray(self.g.bconst(v)), "b")
        if isinstance(v, (int, float)):
            return TV(np.asarray(self.g.bconst(v != 0)), "b")
        if isinstance(v, IV):
            return TV(np.vectorize(lambda x: self.g.bconst(x != 0), otypes=[np.int64])(v.arr), "b")
        raise Unsupported("operand of type %s" % type(v).__name__)

    @staticmethod
    def _map1(fn, ids):
        if ids.size == 0:
            return np.zeros(ids.shape, np.int64)
        return np.vectorize(fn, otypes=[np.int64])(ids)

    @staticmethod
    def _map2(fn, a, b):
        a, b = np.broadcast_arrays(a, b)
        if a.size == 0:
            return np.zeros(a.shape, np.int64)
        return np.vectorize(fn, otypes=[np.int64])(a, b)

    def un(self, op, x):
        return TV(self._map1(lambda i: self.g.unary(op, int(i)), self.f(x).ids))

    def bi(self, op, x, y):
        return TV(self._map2(lambda i, j: self.g.binary(op, int(i), int(j)), self.f(x).ids, self.f(y).ids))

    def cmp(self, op, x, y):
        return TV(self._map2(lambda i, j: self.g.compare(op, int(i), int(j)), self.f(x).ids, self.f(y).ids), "b")

    def where(self, c, x, y):
        c = self.b(c)
        if isinstance(x, TV) and x.kind == "b" and isinstance(y, TV) and y.kind == "b":
            xx, yy, kind = x, y, "b"
        else:
            xx, yy, kind = self.f(x), self.f(y), "f"
        ci, xi, yi = np.broadcast_arrays(c.ids, xx.ids, yy.ids)
        if ci.size == 0:
            return TV(np.zeros(ci.shape, np.int64), kind)
        return TV(np.vectorize(lambda a, p, q: self.g.select(int(a), int(p), int(q)), otypes=[np.int64])(ci, xi, yi), kind)

    def fold(self, op, ids):
        """Reduce a 1-D array of node ids with a binary op, pairwise (shallow expression trees: more instruction-level
        parallelism for the in-order wave than a serial chain, and the rounding of a tree sum)."""
        ids = [int(i) for i in ids]
        if not ids:
            return self.g.const(0.0 if op == "add" else (1.0 if op == "mul" else float("nan")))
        while len(ids) > 1:
            nxt = [self.g.binary(op, ids[k], ids[k + 1]) for k in range(0, len(ids) - 1, 2)]
            if len(ids) & 1:
                nxt.append(ids[-1])
            ids = nxt
        return ids[0]

    def reduce(self, op, x, dims, keepdim=False):
        x = self.f(x)
        nd = x.ids.ndim
        if dims is None or (isinstance(dims, (list, tuple)) and len(dims) == 0):
            dims = list(range(nd))
        if isinstance(dims, int):
            dims = [dims]
        dims = sorted(set(d % nd for d in dims)) if nd else []
        if nd == 0:
            return x
        keep = [d for d in range(nd) if d not in dims]
        moved = np.transpose(x.ids, keep + dims)
        kshape = moved.shape[:len(keep)]
        flat = moved.reshape(int(np.prod(kshape, dtype=np.int64)), -1)
        out = np.array([self.fold(op, row) for row in flat], dtype=np.int64).reshape(kshape)
        if keepdim:
            for d in dims:
                out = np.expand_dims(out, d)
        return TV(out)

    def matmul(self, a, b):
        a, b = self.f(a), self.f(b)
        A, B = a.ids, b.ids
        va, vb = A.ndim == 1, B.ndim == 1
        if va:
            A = A[None, :]
        if vb:
            B = B[:, None]
        if A.shape[-1] != B.shape[-2]:
            raise Unsupported("matmul of shapes %s x %s" % (a.shape, b.shape))
        batch = np.broadcast_shapes(A.shape[:-2], B.shape[:-2])
        A = np.broadcast_to(A, batch + A.shape[-2:])
        B = np.broadcast_to(B, batch + B.shape[-2:])
        n, k, m = A.shape[-2], A.shape[-1], B.shape[-1]
        if int(np.prod(batch, dtype=np.int64)) * n * k * m > MAX_NODES:
            raise Unsupported("a matrix product of %d multiply-adds" % (int(np.prod(batch, dtype=np.int64)) * n * k * m))
        out = np.zeros(batch + (n, m), np.int64)
        for idx in np.ndindex(*batch):
            for i in range(n):
                for j in range(m):
                    out[idx + (i, j)] = self.fold("add", [self.g.mul(int(A[idx + (i, t)]), int(B[idx + (t, j)])) for t in range(k)])
        if va:
            out = out[..., 0, :]
        if vb:
            out = out[..., 0] if not va else out[..., 0]
        return TV(out)

    def solve_triangular(self, A, B, upper, left=True, unitriangular=False):
        A, B = self.f(A), self.f(B)
        if not left:
            raise Unsupported("linalg_solve_triangular(left=False)")
        Ai, Bi = A.ids, B.ids
        batch = np.broadcast_shapes(Ai.shape[:-2], Bi.shape[:-2])
        Ai = np.broadcast_to(Ai, batch + Ai.shape[-2:])
        Bi = np.broadcast_to(Bi, batch + Bi.shape[-2:])
        n, m = Bi.shape[-2], Bi.shape[-1]
        X = np.zeros(batch + (n, m), np.int64)
        order = range(n - 1, -1, -1) if upper else range(n)
        for idx in np.ndindex(*batch):
            for j in range(m):
                for i in order:
                    ks = range(i + 1, n) if upper else range(i)
                    acc = int(Bi[idx + (i, j)])
                    if len(ks):
                        acc = self.g.sub(acc, self.fold("add", [self.g.mul(int(Ai[idx + (i, t)]), int(X[idx + (t, j)])) for t in ks]))
                    X[idx + (i, j)] = acc if unitriangular else self.g.div(acc, int(Ai[idx + (i, i)]))
        return TV(X)


def _const_int(v, what):
    if isinstance(v, IV):
        return v.arr
    if isinstance(v, (int, np.integer)):
        return np.asarray(v, np.int64)
    raise Unsupported("%s must be a constant integer tensor (an index computed from the argument's values is data-dependent)" % what)


def _shape_arg(s):
    return tuple(int(v) for v in s)


def _build_table():
    T = {}

    def reg(*ops):
        def deco(fn):
            for o in ops:
                T[o] = fn
            return fn
        return deco

    def _ov(name, *overloads):
        pk = getattr(aten, name, None)
        out = []
        if pk is None:
            return out
        for o in overloads or pk.overloads():
            ov = getattr(pk, o, None)
            if ov is not None:
                out.append(ov)
        return out

    # ---- identities / dtype / copies ----------------------------------------------------------------------------
    @reg(*_ov("clone"), *_ov("alias"), *_ov("lift_fresh_copy"), *_ov("lift_fresh"), *_ov("contiguous"),
         *_ov("positive"), *_ov("view_as_real"), *_ov("alias_copy"), *_ov("_conj"), *_ov("conj"),
         *_ov("resolve_conj"), *_ov("resolve_neg"))
    def _identity(L, x, *a, **k):
        return x

    @reg(*_ov("detach"), *_ov("detach_"), *_ov("detach_copy"))
    def _detach(L, x):
        """`w.detach()` inside the callable: the value flows on, the derivative does not (ir.py's 'detach' node)."""
        if isinstance(x, TV) and x.kind == "f" and x.concrete is None:
            return L.un("detach", x)
        return x

    @reg(*_ov("_to_copy"), *_ov("to", "dtype", "dtype_layout", "device", "other"), *_ov("type_as"))
    def _to_copy(L, x, *a, dtype=None, **k):
        for v in a:
            if isinstance(v, torch.dtype):
                dtype = v
        if dtype is None or isinstance(x, (int, float, bool)):
            return x
        if dtype == torch.bool:
            return L.b(x)
        if dtype.is_floating_point:
            return L.f(x)
        if isinstance(x, IV):
            return x
        raise Unsupported("a cast of a traced value to %s" % dtype)

    @reg(*_ov("copy", "default"))
    def _copy(L, dst, src, *a, **k):
        if isinstance(dst, TV) and dst.kind == "b":
            return TV(np.broadcast_to(L.b(src).ids, dst.shape).copy(), "b")
        return TV(np.broadcast_to(L.f(src).ids, dst.shape).copy())

    @reg(*_ov("_assert_tensor_metadata"), *_ov("_assert_async"))
    def _nothing(L, *a, **k):
        return None

    # ---- creation -------------------------------------------------------------------------------------------
    def _full(L, shape, v, dtype=None):
        shape = _shape_arg(shape)
        if dtype is not None and dtype != torch.bool and not dtype.is_floating_point:
            return IV(np.full(shape, int(v), np.int64))
        if dtype == torch.bool or (dtype is None and isinstance(v, bool)):
            return TV(np.full(shape, L.g.bconst(bool(v)), np.int64), "b")
        if isinstance(v, TV):
            return TV(np.broadcast_to(v.ids.reshape(()), shape).copy(), v.kind)
        return TV(np.full(shape, L.g.const(float(v)), np.int64))

    @reg(*_ov("scalar_tensor"))
    def _scalar_tensor(L, v, dtype=None, **k):
        return _full(L, (), v, dtype)

    @reg(*_ov("full", "default"))
    def _full_op(L, shape, v, dtype=None, **k):
        return _full(L, shape, v, dtype if dtype is not None else (torch.float32 if isinstance(v, float) else (torch.bool if isinstance(v, bool) else torch.int64)))

    @reg(*_ov("zeros", "default"), *_ov("empty", "memory_format"))
    def _zeros(L, shape, dtype=None, **k):
        return _full(L, shape, 0.0, dtype or torch.float32)

    @reg(*_ov("ones", "default"))
    def _ones(L, shape, dtype=None, **k):
        return _full(L, shape, 1.0, dtype or torch.float32)

    def _like_dtype(x, dtype):
        if dtype is not None:
            return dtype
        if isinstance(x, IV):
            return torch.int64
        return torch.bool if x.kind == "b" else torch.float32

    @reg(*_ov("zeros_like"), *_ov("empty_like"))
    def _zeros_like(L, x, dtype=None, **k):
        return _full(L, x.shape, 0.0, _like_dtype(x, dtype))

    @reg(*_ov("ones_like"))
    def _ones_like(L, x, dtype=None, **k):
        return _full(L, x.shape, 1.0, _like_dtype(x, dtype))

    @reg(*_ov("full_like"))
    def _full_like(L, x, v, dtype=None, **k):
        return _full(L, x.shape, v, _like_dtype(x, dtype))

    @reg(*_ov("new_zeros"), *_ov("new_empty"))
    def _new_zeros(L, x, shape, dtype=None, **k):
        return _full(L, shape, 0.0, _like_dtype(x, dtype))

    @reg(*_ov("new_ones"))
    def _new_ones(L, x, shape, dtype=None, **k):
        return _full(L, shape, 1.0, _like_dtype(x, dtype))

    @reg(*_ov("new_full"))
    def _new_full(L, x, shape, v, dtype=None, **k):
        return _full(L, shape, v, _like_dtype(x, dtype))

    @reg(*_ov("arange", "default", "start", "start_step"))
    def _arange(L, *a, dtype=None, **k):
        vals = np.arange(*a)
        if (dtype is not None and dtype.is_floating_point) or any(isinstance(v, float) for v in a):
            return TV(L._map1(lambda v: L.g.const(float(v)), vals.astype(np.float64)))
        return IV(vals)

    @reg(*_ov("eye", "default", "m"))
    def _eye(L, n, m=None, dtype=None, **k):
        return L.f(IV(np.eye(n, m if isinstance(m, int) else n, dtype=np.int64)))

    # ---- views ------------------------------------------------------------------------------------------------
    def _arr(x):
        return x.arr if isinstance(x, IV) else x.ids

    def _like(x, arr):
        return IV(arr) if isinstance(x, IV) else TV(arr, x.kind)

    @reg(*_ov("view", "default"), *_ov("_unsafe_view"), *_ov("reshape"), *_ov("view_copy", "default"))
    def _view(L, x, shape):
        return _like(x, _arr(x).reshape(_shape_arg(shape)))

    @reg(*_ov("flatten", "using_ints"))
    def _flatten(L, x, start=0, end=-1):
        a = _arr(x)
        nd = max(a.ndim, 1)
        s, e = start % nd, end % nd
        return _like(x, a.reshape(a.shape[:s] + (-1,) + a.shape[e + 1:]))

    @reg(*_ov("permute"), *_ov("permute_copy"))
    def _permute(L, x, dims):
        return _like(x, np.transpose(_arr(x), [int(d) for d in dims]))

    @reg(*_ov("transpose", "int"), *_ov("transpose_copy", "int"))
    def _transpose(L, x, d0, d1):
        return _like(x, np.swapaxes(_arr(x), d0, d1)) if _arr(x).ndim else x

    @reg(*_ov("t"), *_ov("t_copy"), *_ov("adjoint"), *_ov("mT"), *_ov("mH"))
    def _t(L, x):
        a = _arr(x)
        return _like(x, a.T if a.ndim <= 2 else np.swapaxes(a, -1, -2))

    @reg(*_ov("unsqueeze"), *_ov("unsqueeze_copy"))
    def _unsqueeze(L, x, d):
        a = _arr(x)
        return _like(x, np.expand_dims(a, d if d >= 0 else d + a.ndim + 1))

    @reg(*_ov("squeeze", "default"), *_ov("squeeze_copy", "default"))
    def _squeeze(L, x):
        return _like(x, np.squeeze(_arr(x)))

    @reg(*_ov("squeeze", "dim", "dims"), *_ov("squeeze_copy", "dim", "dims"))
    def _squeeze_dims(L, x, dims):
        a = _arr(x)
        dims = [dims] if isinstance(dims, int) else list(dims)
        dims = tuple(d % a.ndim for d in dims if a.ndim and a.shape[d % a.ndim] == 1)
        return _like(x, np.squeeze(a, dims) if dims else a)

    @reg(*_ov("expand"), *_ov("expand_copy"), *_ov("broadcast_to"))
    def _expand(L, x, shape, implicit=False):
        a = _arr(x)
        shape = list(_shape_arg(shape))
        off = len(shape) - a.ndim
        for k in range(len(shape)):
            if shape[k] == -1:
                shape[k] = a.shape[k - off]
        return _like(x, np.broadcast_to(a, shape).copy())

    @reg(*_ov("expand_as"))
    def _expand_as(L, x, y):
        return _like(x, np.broadcast_to(_arr(x), y.shape).copy())

    @reg(*_ov("select", "int"), *_ov("select_copy", "int"))
    def _select(L, x, dim, index):
        return _like(x, np.take(_arr(x), int(index), axis=dim))

    @reg(*_ov("slice", "Tensor"), *_ov("slice_copy", "Tensor"))
    def _slice(L, x, dim=0, start=None, end=None, step=1):
        a = _arr(x)
        sl = [slice(None)] * a.ndim
        n = a.shape[dim]
        end = None if (end is None or end >= n) else end
        sl[dim] = slice(start, end, step)
        return _like(x, a[tuple(sl)])

    @reg(*_ov("narrow", "default"))
    def _narrow(L, x, dim, start, length):
        return _slice(L, x, dim, start, start + length)

    @reg(*_ov("diagonal", "default"), *_ov("diagonal_copy", "default"))
    def _diagonal(L, x, offset=0, dim1=0, dim2=1):
        return _like(x, np.diagonal(_arr(x), offset, dim1, dim2).copy())

    @reg(*_ov("diag_embed"))
    def _diag_embed(L, x, offset=0, dim1=-2, dim2=-1):
        if offset != 0 or (dim1, dim2) not in ((-2, -1),):
            raise Unsupported("diag_embed with offset / dims")
        a = L.f(x).ids
        n = a.shape[-1]
        out = np.full(a.shape + (n,), L.g.const(0.0), np.int64)
        for i in range(n):
            out[..., i, i] = a[..., i]
        return TV(out)

    @reg(*_ov("diag", "default"))
    def _diag(L, x, diagonal=0):
        a = _arr(L.f(x))
        if a.ndim == 1:
            n = a.shape[0] + abs(diagonal)
            out = np.full((n, n), L.g.const(0.0), np.int64)
            for i in range(a.shape[0]):
                out[i + max(-diagonal, 0), i + max(diagonal, 0)] = a[i]
            return TV(out)
        return TV(np.diagonal(a, diagonal).copy())

    def _tri(upper):
        def fn(L, x, k=0):
            a = L.f(x).ids.copy()
            n, m = a.shape[-2:]
            z = L.g.const(0.0)
            for i in range(n):
                for j in range(m):
                    if (j - i < k) if upper else (j - i > k):
                        a[..., i, j] = z
            return TV(a)
        return fn
    for o in _ov("tril", "default"):
        T[o] = _tri(False)
    for o in _ov("triu", "default"):
        T[o] = _tri(True)

    @reg(*_ov("flip"))
    def _flip(L, x, dims):
        return _like(x, np.flip(_arr(x), tuple(dims)).copy())

    @reg(*_ov("repeat"))
    def _repeat(L, x, reps):
        a = _arr(x)
        reps = _shape_arg(reps)
        a = a.reshape((1,) * (len(reps) - a.ndim) + a.shape)
        return _like(x, np.tile(a, reps))

    @reg(*_ov("cat", "default"), *_ov("concat", "default"))
    def _cat(L, xs, dim=0):
        xs = [x for x in xs if not (isinstance(x, TV) and x.ids.ndim == 1 and x.ids.size == 0 and len(xs) > 1)]
        if all(isinstance(x, IV) for x in xs):
            return IV(np.concatenate([x.arr for x in xs], dim))
        kind = "b" if all(isinstance(x, TV) and x.kind == "b" for x in xs) else "f"
        return TV(np.concatenate([(x.ids if kind == "b" else L.f(x).ids) for x in xs], dim), kind)

    @reg(*_ov("stack", "default"))
    def _stack(L, xs, dim=0):
        if all(isinstance(x, IV) for x in xs):
            return IV(np.stack([x.arr for x in xs], dim))
        return TV(np.stack([L.f(x).ids for x in xs], dim))

    @reg(*_ov("unbind", "int"), *_ov("unbind_copy", "int"))
    def _unbind(L, x, dim=0):
        a = _arr(x)
        return [_like(x, np.take(a, i, axis=dim)) for i in range(a.shape[dim])]

    @reg(*_ov("split", "Tensor"), *_ov("split_copy", "Tensor"))
    def _split(L, x, size, dim=0):
        a = _arr(x)
        n = a.shape[dim]
        return [_slice(L, x, dim, s, min(s + size, n)) for s in range(0, n, size)]

    @reg(*_ov("split_with_sizes", "default"), *_ov("split_with_sizes_copy", "default"))
    def _split_sizes(L, x, sizes, dim=0):
        out, s = [], 0
        for k in sizes:
            out.append(_slice(L, x, dim, s, s + k)); s += k
        return out

    @reg(*_ov("chunk", "default"))
    def _chunk(L, x, chunks, dim=0):
        n = _arr(x).shape[dim]
        return _split(L, x, -(-n // chunks), dim)

    @reg(*_ov("index", "Tensor"))
    def _index(L, x, indices):
        a = _arr(x)
        key = tuple(slice(None) if i is None else _const_int(i, "an index tensor") for i in indices)
        return _like(x, a[key])

    @reg(*_ov("index_select", "default"))
    def _index_select(L, x, dim, index):
        return _like(x, np.take(_arr(x), _const_int(index, "index_select's index"), axis=dim))

    @reg(*_ov("gather", "default"))
    def _gather(L, x, dim, index, sparse_grad=False):
        return _like(x, np.take_along_axis(_arr(x), _const_int(index, "gather's index"), axis=dim))

    @reg(*_ov("slice_scatter", "default"))
    def _slice_scatter(L, x, src, dim=0, start=None, end=None, step=1):
        a = L.f(x).ids.copy()
        sl = [slice(None)] * a.ndim
        n = a.shape[dim]
        sl[dim] = slice(start, None if (end is None or end >= n) else end, step)
        a[tuple(sl)] = L.f(src).ids
        return TV(a)

    @reg(*_ov("select_scatter", "default"))
    def _select_scatter(L, x, src, dim, index):
        a = L.f(x).ids.copy()
        sl = [slice(None)] * a.ndim
        sl[dim] = int(index)
        a[tuple(sl)] = L.f(src).ids
        return TV(a)

    @reg(*_ov("index_put", "default"))
    def _index_put(L, x, indices, values, accumulate=False):
        a = L.f(x).ids.copy()
        key = tuple(slice(None) if i is None else _const_int(i, "an index tensor") for i in indices)
        v = L.f(values).ids
        if accumulate:
            cur = a[key]
            a[key] = L._map2(lambda p, q: L.g.add(int(p), int(q)), cur, np.broadcast_to(v, cur.shape))
        else:
            a[key] = v
        return TV(a)

    # ---- element-wise arithmetic ---------------------------------------------------------------------------
    def _both_int(x, y):
        return isinstance(x, (IV, int)) and not isinstance(x, bool) and isinstance(y, (IV, int)) and not isinstance(y, bool)

    def _int_arr(v):
        return v.arr if isinstance(v, IV) else np.asarray(v, np.int64)

    @reg(*_ov("add", "Tensor", "Scalar"))
    def _add(L, x, y, alpha=1):
        if _both_int(x, y) and isinstance(alpha, int):
            return IV(_int_arr(x) + alpha * _int_arr(y))
        return L.bi("add", x, y if alpha == 1 else L.bi("mul", y, alpha))

    @reg(*_ov("sub", "Tensor", "Scalar"))
    def _sub(L, x, y, alpha=1):
        if _both_int(x, y) and isinstance(alpha, int):
            return IV(_int_arr(x) - alpha * _int_arr(y))
        return L.bi("sub", x, y if alpha == 1 else L.bi("mul", y, alpha))

    @reg(*_ov("rsub", "Tensor", "Scalar"))
    def _rsub(L, x, y, alpha=1):
        return L.bi("sub", y, x if alpha == 1 else L.bi("mul", x, alpha))

    @reg(*_ov("mul", "Tensor", "Scalar"))
    def _mul(L, x, y):
        if _both_int(x, y):
            return IV(_int_arr(x) * _int_arr(y))
        if isinstance(x, TV) and x.kind == "b" and isinstance(y, TV) and y.kind == "b":
            return TV(L._map2(lambda i, j: L.g.boolean("and", int(i), int(j)), x.ids, y.ids), "b")
        return L.bi("mul", x, y)

    @reg(*_ov("div", "Tensor", "Scalar"), *_ov("true_divide", "Tensor", "Scalar"))
    def _div(L, x, y):
        return L.bi("div", x, y)

    @reg(*_ov("div", "Tensor_mode", "Scalar_mode"))
    def _div_mode(L, x, y, rounding_mode=None):
        q = L.bi("div", x, y)
        return q if rounding_mode is None else L.un("floor" if rounding_mode == "floor" else "trunc", q)

    @reg(*_ov("floor_divide", "default", "Scalar"))
    def _floor_divide(L, x, y):
        if _both_int(x, y):
            return IV(_int_arr(x) // _int_arr(y))
        return L.un("floor", L.bi("div", x, y))

    @reg(*_ov("pow", "Tensor_Scalar", "Tensor_Tensor", "Scalar"))
    def _pow(L, x, y):
        return L.bi("pow", x, y)

    @reg(*_ov("square", "default"))
    def _square(L, x):
        return L.bi("mul", x, x)

    @reg(*_ov("maximum", "default"), *_ov("fmax", "default"), *_ov("max", "other"))
    def _maximum(L, x, y):
        return L.bi("max", x, y)

    @reg(*_ov("minimum", "default"), *_ov("fmin", "default"), *_ov("min", "other"))
    def _minimum(L, x, y):
        return L.bi("min", x, y)

    for name, op in (("neg", "neg"), ("negative", "neg"), ("exp", "exp"), ("log", "log"), ("sqrt", "sqrt"), ("rsqrt", "rsqrt"),
                     ("tanh", "tanh"), ("sigmoid", "sigmoid"), ("log1p", "log1p"), ("expm1", "expm1"), ("sin", "sin"),
                     ("cos", "cos"), ("abs", "abs"), ("absolute", "abs"), ("sign", "sign"), ("sgn", "sign"), ("erf", "erf"),
                     ("reciprocal", "recip"), ("floor", "floor"), ("ceil", "ceil"), ("round", "round"), ("trunc", "trunc"),
                     ("atan", "atan"), ("arctan", "atan"), ("lgamma", "lgamma"), ("digamma", "digamma")):
        for o in _ov(name, "default"):
            T[o] = (lambda op: (lambda L, x: L.un(op, x)))(op)

    @reg(*_ov("special_erf", "default"))
    def _serf(L, x):
        return L.un("erf", x)

    @reg(*_ov("erfc", "default"), *_ov("special_erfc", "default"))
    def _erfc(L, x):
        return L.bi("sub", 1.0, L.un("erf", x))

    @reg(*_ov("log2", "default"))
    def _log2(L, x):
        return L.bi("mul", L.un("log", x), 1.0 / math.log(2.0))

    @reg(*_ov("log10", "default"))
    def _log10(L, x):
        return L.bi("mul", L.un("log", x), 1.0 / math.log(10.0))

    @reg(*_ov("exp2", "default"))
    def _exp2(L, x):
        return L.un("exp", L.bi("mul", x, math.log(2.0)))

    @reg(*_ov("tan", "default"))
    def _tan(L, x):
        return L.bi("div", L.un("sin", x), L.un("cos", x))

    @reg(*_ov("cosh", "default"))
    def _cosh(L, x):
        return L.bi("mul", L.bi("add", L.un("exp", x), L.un("exp", L.un("neg", x))), 0.5)

    @reg(*_ov("sinh", "default"))
    def _sinh(L, x):
        return L.bi("mul", L.bi("sub", L.un("exp", x), L.un("exp", L.un("neg", x))), 0.5)

    @reg(*_ov("softplus", "default"))
    def _softplus(L, x, beta=1.0, threshold=20.0):
        # (beta x > threshold -> x in torch: a difference below exp(-threshold) = 2e-9, and the derivative stays finite here)
        if beta == 1.0:
            return L.un("softplus", x)
        return L.bi("div", L.un("softplus", L.bi("mul", x, beta)), beta)

    @reg(*_ov("log_sigmoid", "default"))
    def _log_sigmoid(L, x):
        return L.un("neg", L.un("softplus", L.un("neg", x)))

    @reg(*_ov("log_sigmoid_forward", "default"))
    def _log_sigmoid_fwd(L, x):
        return [_log_sigmoid(L, x), None]

    @reg(*_ov("relu", "default"))
    def _relu(L, x):
        return L.bi("max", x, 0.0)

    @reg(*_ov("leaky_relu", "default"))
    def _leaky(L, x, slope=0.01):
        return L.where(L.cmp("gt", x, 0.0), x, L.bi("mul", x, slope))

    @reg(*_ov("elu", "default"))
    def _elu(L, x, alpha=1.0, scale=1.0, input_scale=1.0):
        neg = L.bi("mul", L.un("expm1", L.bi("mul", x, input_scale)), alpha)
        return L.bi("mul", L.where(L.cmp("gt", x, 0.0), x, neg), scale)

    @reg(*_ov("silu", "default"))
    def _silu(L, x):
        return L.bi("mul", x, L.un("sigmoid", x))

    @reg(*_ov("gelu", "default"))
    def _gelu(L, x, approximate="none"):
        if approximate != "none":
            inner = L.bi("mul", L.bi("add", x, L.bi("mul", L.bi("pow", x, 3.0), 0.044715)), math.sqrt(2.0 / math.pi))
            return L.bi("mul", L.bi("mul", x, 0.5), L.bi("add", L.un("tanh", inner), 1.0))
        return L.bi("mul", L.bi("mul", x, 0.5), L.bi("add", L.un("erf", L.bi("mul", x, 1.0 / math.sqrt(2.0))), 1.0))

    @reg(*_ov("hardtanh", "default"))
    def _hardtanh(L, x, lo=-1.0, hi=1.0):
        return L.bi("min", L.bi("max", x, lo), hi)

    @reg(*_ov("clamp", "default", "Tensor"), *_ov("clip", "default"))
    def _clamp(L, x, lo=None, hi=None):
        if lo is not None:
            x = L.bi("max", x, lo)
        if hi is not None:
            x = L.bi("min", x, hi)
        return x

    @reg(*_ov("clamp_min", "default", "Tensor"))
    def _clamp_min(L, x, lo):
        return L.bi("max", x, lo)

    @reg(*_ov("clamp_max", "default", "Tensor"))
    def _clamp_max(L, x, hi):
        return L.bi("min", x, hi)

    @reg(*_ov("xlogy", "Tensor", "Scalar_Self", "Scalar_Other"))
    def _xlogy(L, x, y):
        return L.where(L.cmp("eq", x, 0.0), 0.0, L.bi("mul", x, L.un("log", y)))

    @reg(*_ov("logaddexp", "default"))
    def _logaddexp(L, x, y):
        m = L.bi("max", x, y)
        return L.bi("add", m, L.un("log1p", L.un("exp", L.un("neg", L.un("abs", L.bi("sub", x, y))))))

    @reg(*_ov("addcmul", "default"))
    def _addcmul(L, x, a, b, value=1):
        return L.bi("add", x, L.bi("mul", L.bi("mul", a, b), value))

    @reg(*_ov("addcdiv", "default"))
    def _addcdiv(L, x, a, b, value=1):
        return L.bi("add", x, L.bi("mul", L.bi("div", a, b), value))

    @reg(*_ov("lerp", "Scalar", "Tensor"))
    def _lerp(L, a, b, w):
        return L.bi("add", a, L.bi("mul", L.bi("sub", b, a), w))

    # ---- comparisons / logic -------------------------------------------------------------------------------------
    for name in ("gt", "ge", "lt", "le", "eq", "ne"):
        for o in _ov(name, "Scalar", "Tensor"):
            T[o] = (lambda op: (lambda L, x, y: L.cmp(op, x, y)))(name)
    for name, op in (("greater", "gt"), ("greater_equal", "ge"), ("less", "lt"), ("less_equal", "le"), ("not_equal", "ne")):
        for o in _ov(name, "Scalar", "Tensor"):
            T[o] = (lambda op: (lambda L, x, y: L.cmp(op, x, y)))(op)

    @reg(*_ov("logical_and", "default"), *_ov("bitwise_and", "Tensor"), *_ov("__and__", "Tensor"))
    def _and(L, x, y):
        return TV(L._map2(lambda i, j: L.g.boolean("and", int(i), int(j)), L.b(x).ids, L.b(y).ids), "b")

    @reg(*_ov("logical_or", "default"), *_ov("bitwise_or", "Tensor"), *_ov("__or__", "Tensor"))
    def _or(L, x, y):
        return TV(L._map2(lambda i, j: L.g.boolean("or", int(i), int(j)), L.b(x).ids, L.b(y).ids), "b")

    @reg(*_ov("logical_not", "default"), *_ov("bitwise_not", "default"))
    def _not(L, x):
        return TV(L._map1(lambda i: L.g.boolean("not", int(i)), L.b(x).ids), "b")

    @reg(*_ov("isnan", "default"))
    def _isnan(L, x):
        return TV(L._map1(lambda i: L.g.boolean("isnan", int(i)), L.f(x).ids), "b")

    @reg(*_ov("isinf", "default"))
    def _isinf(L, x):
        return TV(L._map1(lambda i: L.g.boolean("isinf", int(i)), L.f(x).ids), "b")

    @reg(*_ov("isfinite", "default"))
    def _isfinite(L, x):
        return _not(L, _or(L, _isnan(L, x), _isinf(L, x)))

    @reg(*_ov("where", "self", "ScalarSelf", "ScalarOther", "Scalar"))
    def _where(L, c, x, y):
        return L.where(c, x, y)

    @reg(*_ov("masked_fill", "Scalar", "Tensor"))
    def _masked_fill(L, x, mask, v):
        return L.where(mask, v, x)

    @reg(*_ov("nan_to_num", "default"))
    def _nan_to_num(L, x, nan=0.0, posinf=None, neginf=None):
        big = 3.4028234663852886e38
        y = L.where(_isnan(L, x), 0.0 if nan is None else nan, x)
        y = L.bi("min", y, big if posinf is None else posinf)
        return L.bi("max", y, -big if neginf is None else neginf)

    # ---- reductions --------------------------------------------------------------------------------------------
    @reg(*_ov("sum", "dim_IntList", "default"))
    def _sum(L, x, dims=None, keepdim=False, dtype=None):
        return L.reduce("add", x, dims, keepdim)

    @reg(*_ov("mean", "dim", "default"))
    def _mean(L, x, dims=None, keepdim=False, dtype=None):
        s = L.reduce("add", x, dims, keepdim)
        n = max(1, int(np.prod(x.shape, dtype=np.int64)) // max(1, int(np.prod(s.shape, dtype=np.int64))))
        return L.bi("div", s, float(n))

    @reg(*_ov("prod", "default", "dim_int"))
    def _prod(L, x, dim=None, keepdim=False, dtype=None):
        return L.reduce("mul", x, dim, keepdim)

    @reg(*_ov("amax", "default"))
    def _amax(L, x, dims=(), keepdim=False):
        return L.reduce("max", x, list(dims), keepdim)

    @reg(*_ov("amin", "default"))
    def _amin(L, x, dims=(), keepdim=False):
        return L.reduce("min", x, list(dims), keepdim)

    @reg(*_ov("max", "default"))
    def _max_all(L, x):
        return L.reduce("max", x, None)

    @reg(*_ov("min", "default"))
    def _min_all(L, x):
        return L.reduce("min", x, None)

    @reg(*_ov("max", "dim"))
    def _max_dim(L, x, dim, keepdim=False):
        return [L.reduce("max", x, [dim], keepdim), _DataDependent("the index output of max(dim)")]

    @reg(*_ov("min", "dim"))
    def _min_dim(L, x, dim, keepdim=False):
        return [L.reduce("min", x, [dim], keepdim), _DataDependent("the index output of min(dim)")]

    @reg(*_ov("logsumexp", "default"))
    def _logsumexp(L, x, dims, keepdim=False):
        dims = [dims] if isinstance(dims, int) else list(dims)
        m = L.reduce("max", x, dims, True)
        m = L.un("detach", L.where(_isinf(L, m), 0.0, m))     # the shift: its derivative cancels exactly, so it is not differentiated
        s = L.reduce("add", L.un("exp", L.bi("sub", x, m)), dims, True)
        out = L.bi("add", L.un("log", s), m)
        return out if keepdim else _squeeze_dims(L, out, dims)

    @reg(*_ov("_log_softmax", "default"), *_ov("log_softmax", "int"))
    def _log_softmax(L, x, dim, *a, **k):
        return L.bi("sub", x, _logsumexp(L, x, [dim], True))

    @reg(*_ov("_softmax", "default"), *_ov("softmax", "int"))
    def _softmax(L, x, dim, *a, **k):
        return L.un("exp", _log_softmax(L, x, dim))

    def _var_impl(std):
        def fn(L, x, dim=None, *a, correction=None, keepdim=False, unbiased=None):
            for v in a:
                if isinstance(v, bool) and unbiased is None and correction is None:
                    unbiased = v
                elif isinstance(v, bool):
                    keepdim = v
            if correction is None:
                correction = 1 if (unbiased is None or unbiased) else 0
            dims = None if dim is None else ([dim] if isinstance(dim, int) else list(dim))
            m = _mean(L, x, dims, True)
            d = L.bi("sub", x, m)
            s = L.reduce("add", L.bi("mul", d, d), dims, keepdim)
            n = int(np.prod(x.shape, dtype=np.int64)) // max(1, int(np.prod(s.shape, dtype=np.int64)))
            v = L.bi("div", s, float(max(n - correction, 0)) if n - correction > 0 else float("nan"))
            return L.un("sqrt", v) if std else v
        return fn
    for o in _ov("var", "correction", "dim", "default"):
        T[o] = _var_impl(False)
    for o in _ov("std", "correction", "dim", "default"):
        T[o] = _var_impl(True)

    @reg(*_ov("cumsum", "default"))
    def _cumsum(L, x, dim, dtype=None):
        a = np.moveaxis(L.f(x).ids, dim, -1).copy()
        for k in range(1, a.shape[-1]):
            a[..., k] = L._map2(lambda p, q: L.g.add(int(p), int(q)), a[..., k - 1], a[..., k])
        return TV(np.moveaxis(a, -1, dim))

    @reg(*_ov("linalg_vector_norm", "default"), *_ov("norm", "Scalar", "ScalarOpt_dim"))
    def _norm(L, x, ord=2, dim=None, keepdim=False, dtype=None):
        ord = 2 if ord is None else ord
        if ord == 2:
            return L.un("sqrt", L.reduce("add", L.bi("mul", x, x), dim, keepdim))
        if ord == 1:
            return L.reduce("add", L.un("abs", x), dim, keepdim)
        if ord == float("inf"):
            return L.reduce("max", L.un("abs", x), dim, keepdim)
        return L.bi("pow", L.reduce("add", L.bi("pow", L.un("abs", x), float(ord)), dim, keepdim), 1.0 / float(ord))

    @reg(*_ov("trace", "default"))
    def _trace(L, x):
        return L.reduce("add", TV(np.diagonal(L.f(x).ids).copy()), None)

    # ---- products --------------------------------------------------------------------------------------------------
    @reg(*_ov("dot", "default"), *_ov("vdot", "default"), *_ov("inner", "default"))
    def _dot(L, x, y):
        return L.reduce("add", L.bi("mul", x, y), [-1])

    @reg(*_ov("mv", "default"), *_ov("mm", "default"), *_ov("bmm", "default"), *_ov("matmul", "default"))
    def _mm(L, a, b):
        return L.matmul(a, b)

    @reg(*_ov("addmm", "default"), *_ov("addmv", "default"), *_ov("baddbmm", "default"))
    def _addmm(L, c, a, b, beta=1, alpha=1):
        prod = L.matmul(a, b)
        if alpha != 1:
            prod = L.bi("mul", prod, alpha)
        return L.bi("add", c if beta == 1 else L.bi("mul", c, beta), prod)

    @reg(*_ov("linear", "default"))
    def _linear(L, x, w, b=None):
        out = L.matmul(x, _t(L, w))
        return out if b is None else L.bi("add", out, b)

    @reg(*_ov("outer", "default"), *_ov("ger", "default"))
    def _outer(L, x, y):
        return L.bi("mul", _unsqueeze(L, L.f(x), 1), _unsqueeze(L, L.f(y), 0))

    @reg(*_ov("linalg_solve_triangular", "default"))
    def _solve_tri(L, A, B, upper, left=True, unitriangular=False):
        return L.solve_triangular(A, B, upper, left, unitriangular)

    @reg(*_ov("triangular_solve", "default"))
    def _tri_solve_old(L, B, A, upper=True, transpose=False, unitriangular=False):
        if transpose:
            A, upper = _t(L, A), not upper
        return [L.solve_triangular(A, B, upper, True, unitriangular), A]

    return T


class _DataDependent:
    """Placeholder for an output whose VALUE would be needed as an index (argmax ...): fine while nobody consumes it."""

    def __init__(self, what):
        self.what = what


_TABLE = None


def lowering_table():
    global _TABLE
    if _TABLE is None:
        _TABLE = _build_table()
    return _TABLE


_INPLACE_TO_FUNCTIONAL = {}


def _functional_of(target):
    """aten.add_.Tensor -> aten.add.Tensor (plain make_fx keeps the in-place spelling of ``ll += ...`` on a fresh tensor)."""
    name = target._schema.name.split("::")[1]
    if not name.endswith("_") or name.endswith("__"):
        return None
    pk = getattr(aten, name[:-1], None)
    return getattr(pk, target._overloadname, None) if pk is not None else None


class Traced:
    """The lowered callback: `graph`, the node of the value (`value`) and the example point's dimension."""

    def __init__(self, graph, value, n_fx_nodes, source_name):
        self.graph, self.value, self.n_fx_nodes, self.source_name = graph, value, n_fx_nodes, source_name
        self._grad = None

    @property
    def D(self):
        return self.graph.n_inputs

    def grad(self):
        if self._grad is None:
            self._grad = self.graph.grad(self.value)
        return self._grad


def trace_callback(fn, example):
    """Trace ``fn`` at the (D,) tensor ``example`` and lower it.  Raises ``ir.Unsupported`` (with the reason) when the callback
    cannot be compiled."""
    from torch.fx.experimental.proxy_tensor import make_fx
    if example.dim() != 1:
        raise Unsupported("the example point must be one (D,) vector")
    D = int(example.numel())
    table = lowering_table()

    def wrapped(w):
        r = fn(w)
        if isinstance(r, tuple):
            raise Unsupported("the (log_prob, params) tuple protocol (samplers.py:54-58)")
        if not torch.is_tensor(r):
            raise Unsupported("log_prob_func returned %s, not a tensor" % type(r).__name__)
        return r.sum() if r.dim() else r

    from torch._decomp import core_aten_decompositions
    decomp = {op: d for op, d in core_aten_decompositions().items() if op not in table}
    try:
        with _no_distribution_validation(), torch.enable_grad():
            gm = make_fx(torch.func.functionalize(wrapped), tracing_mode="real", decomposition_table=decomp)(example.detach().clone())
    except Unsupported:
        raise
    except Exception as e:
        msg = str(e).split("\n")[0][:200]
        if "get value out of a tracing tensor" in str(e) or "data-dependent" in str(e).lower():
            raise Unsupported("control flow that depends on the argument's values (%s)" % msg) from None
        raise Unsupported("trace failed: %s: %s" % (type(e).__name__, msg)) from None

    L = _Lowering(D)
    env = {}
    n_fx = 0
    out_val = None
    placeholders = 0
    for node in gm.graph.nodes:
        n_fx += 1
        if node.op == "placeholder":
            placeholders += 1
            if placeholders > 1:
                raise Unsupported("more than one traced argument")
            env[node] = TV(np.array(L.g.inputs, dtype=np.int64))
        elif node.op == "get_attr":
            t = getattr(gm, node.target)
            env[node] = L.from_tensor(t) if torch.is_tensor(t) else t
        elif node.op == "call_function":
            args = torch.fx.node.map_arg(node.args, lambda n: _use(env[n]))
            kwargs = dict(torch.fx.node.map_arg(node.kwargs, lambda n: _use(env[n])))
            if node.target is operator.getitem:
                env[node] = args[0][args[1]]
                continue
            fnl = table.get(node.target)
            if fnl is None and isinstance(node.target, torch._ops.OpOverload):
                alt = _functional_of(node.target)
                fnl = table.get(alt) if alt is not None else None
            if fnl is None:
                raise Unsupported("operation %s is not in the lowering table" % (node.target,))
            try:
                folded = _fold_constant(L, node.target, args, kwargs)
            except Unsupported:
                raise
            except Exception:
                folded = _MISSING
            if folded is not _MISSING:
                env[node] = folded
                continue
            for k in ("layout", "device", "pin_memory", "memory_format", "non_blocking", "copy"):
                kwargs.pop(k, None)
            try:
                env[node] = fnl(L, *args, **kwargs)
            except Unsupported:
                raise
            except Exception as e:
                raise Unsupported("lowering of %s failed: %s: %s" % (node.target, type(e).__name__, str(e)[:160])) from None
            if len(L.g.nodes) > MAX_NODES:
                raise Unsupported("the lowered graph exceeds %d scalar operations" % MAX_NODES)
        elif node.op == "output":
            o = node.args[0]
            o = o[0] if isinstance(o, (tuple, list)) else o
            out_val = _use(env[o])
        else:
            raise Unsupported("fx node kind %s" % node.op)
    if not isinstance(out_val, TV) or out_val.ids.size != 1:
        raise Unsupported("log_prob_func must return a scalar")
    value = int(L.f(out_val).ids.reshape(()))
    return Traced(L.g, value, n_fx, getattr(fn, "__name__", type(fn).__name__))


_MISSING = object()


def _concrete(v):
    """The torch-level value of a lowered argument if it does not depend on the traced input, else _MISSING."""
    if isinstance(v, (TV, IV)):
        return _MISSING if v.concrete is None else v.concrete
    if isinstance(v, (list, tuple)):
        out = [_concrete(x) for x in v]
        return _MISSING if any(x is _MISSING for x in out) else type(v)(out)
    if isinstance(v, _DataDependent):
        return _MISSING
    return v


def _fold_constant(L, target, args, kwargs):
    """An operation on constants only (tensors the callable closes over, literals): evaluated by torch ITSELF, in the dtype torch uses -
    `Normal(0, 3).log_prob` takes log(3) in float32 whatever the argument's dtype, and so does the compiled code.  _MISSING if an
    operand depends on the argument."""
    if not any(isinstance(a, (TV, IV)) or (isinstance(a, (list, tuple)) and any(isinstance(x, (TV, IV)) for x in a)) for a in args):
        return _MISSING                                    # creation ops (zeros, arange ...): the table handles them
    cargs = _concrete(list(args))
    ckw = _concrete(list(kwargs.values()))
    if cargs is _MISSING or ckw is _MISSING:
        return _MISSING
    with torch.no_grad():
        res = target(*cargs, **dict(zip(kwargs.keys(), ckw)))

    def wrap(r):
        if torch.is_tensor(r):
            return L.from_tensor(r)
        if isinstance(r, (list, tuple)):
            return [wrap(x) for x in r]
        return r
    return wrap(res)


def _use(v):
    if isinstance(v, _DataDependent):
        raise Unsupported(v.what + " (an index computed from the argument's values)")
    return v

"""The return value of ``sample()``: the reference's Python ``list`` of per-trajectory tensors (S:957-959, S:1009-1026,
S:1084-1091) without paying for S tensor-view objects on the host before the caller asks for them.

At BASELINE config 2 (1024 chains, 1000 trajectories per call) the kernels finish a call in 0.19 ms; building the 1001
views of ``samples.unbind(0)`` alone takes 0.25 ms of host time.  ``SampleList`` IS a ``list`` (``isinstance`` holds,
``torch.stack`` / ``torch.cat`` accept it) whose rows exist as ONE ``[S, C, D]`` tensor until something needs the
individual objects:

* ``len(x)``, ``x[i]`` (also negative), ``torch.stack(x)``, ``torch.cat(x)`` and ``predict_model(samples=x)`` work on the
  backing tensor directly (no per-row objects);
* anything else -- iteration, slicing, comparison, mutation, ``repr``, pickling, concatenation, and any torch function
  other than stack / cat that receives the list -- first MATERIALISES it (one ``unbind``), after which the object behaves
  as the plain list it then is.

How the C level stays honest: until materialised, the underlying list storage holds S references to one placeholder
object that carries ``__torch_function__``, so (i) ``PyList_GET_SIZE`` is S for every C consumer, (ii) torch's argument
parser (which reads list items directly, bypassing ``__getitem__``) dispatches to the placeholder, which materialises or
answers from the backing tensor.  CPython's own fast paths (``list(x)``, ``tuple(x)``, ``sorted``, ``*x``) use
``PyList_CheckExact`` and therefore go through ``__iter__`` for a subclass.
"""
from __future__ import annotations

import torch

#: below this many rows the plain list is built right away (the views cost less than the bookkeeping)
LAZY_MIN_ROWS = 65


class _Row:
    """Placeholder held S times by an un-materialised SampleList; routes torch functions that receive the list."""
    __slots__ = ()

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        lst = args[0] if args else None
        if isinstance(lst, SampleList) and not lst._done and "out" not in kwargs:
            name = getattr(func, "__name__", "")
            dim = kwargs.get("dim", args[1] if len(args) > 1 else 0)
            if name == "stack" and len(args) <= 2 and dim in (0, -lst._t.dim()) and set(kwargs) <= {"dim"}:
                return lst._t.clone()                                        # torch.stack copies; so does this
            if name in ("cat", "concat", "concatenate") and len(args) <= 2 and set(kwargs) <= {"dim"} and lst._t.dim() >= 2 \
                    and dim in (0, -(lst._t.dim() - 1)):
                return lst._t.reshape((-1,) + tuple(lst._t.shape[2:])).clone()
        found = [False]
        args = _materialise_in(args, found)
        kwargs = {k: _materialise_in(v, found) for k, v in kwargs.items()}
        if not found[0]:
            # a placeholder reached torch outside its SampleList (a C-level copy of the list's slots): calling `func` again
            # would dispatch straight back here
            raise TypeError("hamiltorch_amd.SampleList placeholder outside its list (the list's storage was copied at the C "
                            "level); use list(samples) or samples.tensor")
        return func(*args, **kwargs)


_ROW = _Row()


def _materialise_in(obj, found=None):
    if isinstance(obj, SampleList):
        if found is not None and not obj._done:
            found[0] = True
        obj._materialise()
        return obj
    if isinstance(obj, tuple):
        return tuple(_materialise_in(o, found) for o in obj)
    if type(obj) is list:
        for o in obj:
            _materialise_in(o, found)
    return obj


def _mat(name):
    base = getattr(list, name)

    def method(self, *a, **k):
        if not self._done:
            self._materialise()
        return base(self, *a, **k)
    method.__name__ = name
    return method


class SampleList(list):
    """Lazy list over the rows of ``samples`` ([S, C, D] for a batch of chains, [S, D] for one chain)."""
    __slots__ = ("_t", "_done")

    def __init__(self, samples):
        list.__init__(self)
        self._t = samples
        self._done = False
        list.extend(self, (_ROW,) * samples.shape[0])

    # ---- answered from the backing tensor -------------------------------------------------------------------------
    @property
    def tensor(self):
        """The rows as one tensor ([S, ...]; zero-copy while un-materialised, a stack afterwards)."""
        return self._t if not self._done else torch.stack(list(self))

    def __getitem__(self, i):
        if self._done or not isinstance(i, int):
            if not self._done:
                self._materialise()
            return list.__getitem__(self, i)
        n = self._t.shape[0]
        if i < -n or i >= n:
            raise IndexError("list index out of range")
        return self._t[i]

    def _materialise(self):
        if not self._done:
            self._done = True
            list.__setitem__(self, slice(None), self._t.unbind(0))
            self._t = None
        return self

    # ---- everything else sees the real rows -----------------------------------------------------------------------
    def __reduce_ex__(self, protocol):
        self._materialise()
        return (list, (list(self),))

    def __copy__(self):
        self._materialise()
        return list(self)

    def __radd__(self, other):
        # `plain_list + samples`: without this CPython's list_concat accepts the subclass and copies its raw slots (the
        # placeholders); a subclass's reflected method is tried first
        self._materialise()
        return other + list(self)

    def __deepcopy__(self, memo):
        import copy
        self._materialise()
        return [copy.deepcopy(t, memo) for t in list.__iter__(self)]


for _n in ("__iter__", "__reversed__", "__contains__", "__add__", "__iadd__", "__mul__", "__rmul__", "__imul__", "__eq__", "__ne__",
           "__lt__", "__le__", "__gt__", "__ge__", "__setitem__", "__delitem__", "__repr__", "append", "extend", "insert", "pop",
           "remove", "clear", "index", "count", "sort", "reverse", "copy"):
    setattr(SampleList, _n, _mat(_n))
SampleList.__str__ = SampleList.__repr__
SampleList.__hash__ = None


def rows_of(samples, one):
    """The reference's return list for ``samples[S, C, D]``: rows (C, D), or (D,) for a single chain (`one`)."""
    t = samples[:, 0] if one else samples
    if t.shape[0] < LAZY_MIN_ROWS or not t.is_cuda:
        return list(t.unbind(0))
    return SampleList(t)


def as_tensor(samples):
    """[S, ...] tensor of a list of samples (zero-copy for an un-materialised SampleList)."""
    if isinstance(samples, SampleList):
        return samples.tensor
    return torch.stack(list(samples))

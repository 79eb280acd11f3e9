"""Scalar expression graph of a traced ``log_prob_func`` - the middle of the callback compiler.

A user callback (the contract of hamiltorch/samplers.py:272-274: a (D,) tensor in, a scalar out) is traced once
(``trace.py``), every tensor of the trace is taken apart into its ELEMENTS, and each element becomes a node of the
graph below: a hash-consed DAG of scalar operations on the D inputs.  Reverse-mode differentiation (``grad``) is a
graph-to-graph map, so the value, the gradient the leapfrog needs (what ``params_grad`` gets from autograd,
samplers.py:270-278, :33-66), the Hessian of ``fisher`` (samplers.py:108) and the third derivatives behind
``dH/dtheta`` of the Riemannian sampler (samplers.py:397-398) all come out of the same structure, simplified as they are
built (constant folding, x * 0, x + 0, common subexpressions) and emitted as straight-line device code by ``emit.py``.

Node ids are topologically ordered (operands have smaller ids).  ``evaluate`` runs a set of nodes for many points at
once in numpy - the CPU check of the lowering (tests/test_jit_ir.py) and nothing the product path calls.
"""
from __future__ import annotations

import math

import numpy as np

# op -> arity.  'f' ops produce floats, 'b' ops booleans.
UNARY = ("neg", "exp", "log", "sqrt", "rsqrt", "tanh", "sigmoid", "log1p", "expm1", "sin", "cos", "abs", "sign", "erf",
         "recip", "softplus", "floor", "ceil", "round", "trunc", "atan", "lgamma", "digamma", "detach")
BINARY = ("add", "sub", "mul", "div", "pow", "max", "min")
COMPARE = ("gt", "ge", "lt", "le", "eq", "ne")
BOOL = ("and", "or", "not", "isnan", "isinf")

_TWO_OVER_SQRT_PI = 2.0 / math.sqrt(math.pi)


def _softplus(x):
    return np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))


def _erf(x):
    try:
        from scipy.special import erf
        return erf(x)
    except Exception:       # pragma: no cover
        return np.vectorize(math.erf)(x)


def _sp(name):
    def f(x):
        import scipy.special as sp
        return getattr(sp, name)(x)
    return f


_NP_UNARY = {
    "neg": np.negative, "exp": np.exp, "log": np.log, "sqrt": np.sqrt, "rsqrt": lambda x: 1.0 / np.sqrt(x), "tanh": np.tanh,
    "sigmoid": lambda x: 1.0 / (1.0 + np.exp(-x)), "log1p": np.log1p, "expm1": np.expm1, "sin": np.sin, "cos": np.cos,
    "abs": np.abs, "sign": np.sign, "erf": _erf, "recip": lambda x: 1.0 / x, "softplus": _softplus, "floor": np.floor,
    "ceil": np.ceil, "round": np.round, "trunc": np.trunc, "atan": np.arctan, "lgamma": _sp("gammaln"), "digamma": _sp("digamma"),
    "detach": lambda x: x,
}
_NP_BINARY = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "pow": np.power, "max": np.maximum,
              "min": np.minimum}
_NP_COMPARE = {"gt": np.greater, "ge": np.greater_equal, "lt": np.less, "le": np.less_equal, "eq": np.equal, "ne": np.not_equal}


class Graph:
    """Hash-consed scalar DAG.  Build with the methods below (they simplify); never append to ``nodes`` directly."""

    def __init__(self, n_inputs):
        self.n_inputs = int(n_inputs)
        self.nodes = []          # tuples (op, *operands); operands are node ids except for 'const' / 'bconst' / 'in'
        self._memo = {}
        self.inputs = [self._new(("in", i)) for i in range(self.n_inputs)]

    # ---- construction ------------------------------------------------------------------------------------------
    def _new(self, key):
        i = self._memo.get(key)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(key)
            self._memo[key] = i
        return i

    def const(self, v):
        v = float(v)
        if v == 0.0:
            v = 0.0              # one zero (-0.0 would be another key)
        if v != v:
            return self._new(("const", "nan"))
        return self._new(("const", v))

    def bconst(self, v):
        return self._new(("bconst", bool(v)))

    def is_const(self, i):
        return self.nodes[i][0] == "const"

    def cval(self, i):
        v = self.nodes[i][1]
        return float("nan") if v == "nan" else v

    def is_bool(self, i):
        op = self.nodes[i][0]
        return op in COMPARE or op in BOOL or op == "bconst" or (op == "sel" and self.is_bool(self.nodes[i][2]))

    def _is(self, i, v):
        return self.nodes[i][0] == "const" and self.nodes[i][1] == v

    def unary(self, op, a):
        n = self.nodes[a]
        if n[0] == "const":
            with np.errstate(all="ignore"):
                return self.const(float(_NP_UNARY[op](np.float64(self.cval(a)))))
        if op == "neg":
            if n[0] == "neg":
                return n[1]
            if n[0] == "sub":
                return self.binary("sub", n[2], n[1])
        if op == "recip" and n[0] == "recip":
            return n[1]
        if op in ("abs",) and n[0] in ("abs", "exp", "sqrt", "softplus"):
            return a
        if op == "log":
            if n[0] == "exp":
                return n[1]
            if n[0] == "sqrt":                       # log sqrt y = 1/2 log y
                return self.binary("mul", self.const(0.5), self.unary("log", n[1]))
            if n[0] == "rsqrt":
                return self.binary("mul", self.const(-0.5), self.unary("log", n[1]))
            if n[0] == "recip":
                return self.unary("neg", self.unary("log", n[1]))
        if op == "sqrt" and n[0] == "exp":           # sqrt exp x = exp(x / 2)  (and no overflow of the inner exp)
            return self.unary("exp", self.binary("mul", self.const(0.5), n[1]))
        if op == "rsqrt" and n[0] == "exp":
            return self.unary("exp", self.binary("mul", self.const(-0.5), n[1]))
        if op == "recip" and n[0] == "exp":
            return self.unary("exp", self.unary("neg", n[1]))
        if op == "exp" and n[0] == "log":
            return n[1]
        return self._new((op, a))

    def binary(self, op, a, b):
        na, nb = self.nodes[a], self.nodes[b]
        ca, cb = na[0] == "const", nb[0] == "const"
        if ca and cb:
            with np.errstate(all="ignore"):
                return self.const(float(_NP_BINARY[op](np.float64(self.cval(a)), np.float64(self.cval(b)))))
        if (ca and na[1] == "nan") or (cb and nb[1] == "nan"):
            return self.const(float("nan"))
        if op == "add":
            if ca and na[1] == 0.0:
                return b
            if cb and nb[1] == 0.0:
                return a
            if nb[0] == "neg":
                return self.binary("sub", a, nb[1])
            if na[0] == "neg":
                return self.binary("sub", b, na[1])
            if a == b:
                return self.binary("mul", self.const(2.0), a)
            if a > b:
                a, b = b, a
        elif op == "sub":
            if cb and nb[1] == 0.0:
                return a
            if ca and na[1] == 0.0:
                return self.unary("neg", b)
            if a == b:
                return self.const(0.0)
            if nb[0] == "neg":
                return self.binary("add", a, nb[1])
        elif op == "mul":
            if (ca and na[1] == 0.0) or (cb and nb[1] == 0.0):
                return self.const(0.0)
            if ca and na[1] == 1.0:
                return b
            if cb and nb[1] == 1.0:
                return a
            if ca and na[1] == -1.0:
                return self.unary("neg", b)
            if cb and nb[1] == -1.0:
                return self.unary("neg", a)
            if na[0] == "neg" and nb[0] == "neg":
                return self.binary("mul", na[1], nb[1])
            if na[0] == "neg":
                return self.unary("neg", self.binary("mul", na[1], b))
            if nb[0] == "neg":
                return self.unary("neg", self.binary("mul", a, nb[1]))
            # c1 * (c2 * x) -> (c1 c2) * x
            if ca and nb[0] == "mul" and self.is_const(nb[1]):
                return self.binary("mul", self.const(na[1] * self.cval(nb[1])), nb[2])
            if cb and na[0] == "mul" and self.is_const(na[1]):
                return self.binary("mul", self.const(nb[1] * self.cval(na[1])), na[2])
            if a == b and na[0] == "sqrt":
                return na[1]
            if a == b and na[0] == "exp":            # exp(x)^2 = exp(2 x)
                return self.unary("exp", self.binary("mul", self.const(2.0), na[1]))
            # x * (c * y) -> c * (x * y): constants float to the outside of a product, where they meet and fold
            if not ca and nb[0] == "mul" and self.is_const(nb[1]):
                return self.binary("mul", nb[1], self.binary("mul", a, nb[2]))
            if not cb and na[0] == "mul" and self.is_const(na[1]):
                return self.binary("mul", na[1], self.binary("mul", na[2], b))
            if a > b:
                a, b = b, a
            if self.is_const(b):        # constants first: the (c1, (c2, x)) rule above looks there
                a, b = b, a
        elif op == "div":
            if ca and na[1] == 0.0:
                return self.const(0.0)
            if cb and nb[1] == 1.0:
                return a
            if cb and nb[1] == -1.0:
                return self.unary("neg", a)
            if cb and nb[1] != 0.0 and nb[1] != "nan" and math.isfinite(nb[1]):
                r = 1.0 / nb[1]
                if r * nb[1] == 1.0 and math.isfinite(r):        # exact reciprocals only (powers of two ...): x / c stays a division otherwise
                    return self.binary("mul", a, self.const(r))
            if na[0] == "neg":
                return self.unary("neg", self.binary("div", na[1], b))
            if not cb:
                # x / y = x * (1 / y): the reciprocal is a shared node (ten x_i / (2 var) of a Normal.log_prob cost one division),
                # at <= 1.5 ulp instead of 0.5
                return self.binary("mul", a, self.unary("recip", b))
        elif op == "pow":
            if cb:
                e = nb[1]
                if e == 0.0:
                    return self.const(1.0)
                if e == 1.0:
                    return a
                if e == 2.0:
                    return self.binary("mul", a, a)
                if e == 0.5:
                    return self.unary("sqrt", a)
                if e == -1.0:
                    return self.unary("recip", a)
                if e == -0.5:
                    return self.unary("rsqrt", a)
                if e == -2.0:
                    return self.unary("recip", self.binary("mul", a, a))
                if e == "nan":
                    return self.const(float("nan"))
                if float(e).is_integer() and 2.0 < abs(e) <= 8.0:
                    k, acc, base = int(abs(e)), None, a
                    while k:
                        if k & 1:
                            acc = base if acc is None else self.binary("mul", acc, base)
                        k >>= 1
                        if k:
                            base = self.binary("mul", base, base)
                    return acc if e > 0 else self.unary("recip", acc)
        elif op in ("max", "min"):
            if a == b:
                return a
            if a > b:
                a, b = b, a
        return self._new((op, a, b))

    def add(self, a, b): return self.binary("add", a, b)
    def sub(self, a, b): return self.binary("sub", a, b)
    def mul(self, a, b): return self.binary("mul", a, b)
    def div(self, a, b): return self.binary("div", a, b)
    def neg(self, a): return self.unary("neg", a)

    def compare(self, op, a, b):
        if self.is_const(a) and self.is_const(b):
            with np.errstate(all="ignore"):
                return self.bconst(bool(_NP_COMPARE[op](self.cval(a), self.cval(b))))
        return self._new((op, a, b))

    def boolean(self, op, a, b=None):
        na = self.nodes[a]
        if op == "not":
            if na[0] == "bconst":
                return self.bconst(not na[1])
            if na[0] == "not":
                return na[1]
            return self._new(("not", a))
        if op in ("isnan", "isinf"):
            if na[0] == "const":
                v = self.cval(a)
                return self.bconst(v != v if op == "isnan" else math.isinf(v))
            return self._new((op, a))
        nb = self.nodes[b]
        if na[0] == "bconst":
            return (b if na[1] else a) if op == "and" else (a if na[1] else b)
        if nb[0] == "bconst":
            return (a if nb[1] else b) if op == "and" else (b if nb[1] else a)
        if a == b:
            return a
        if a > b:
            a, b = b, a
        return self._new((op, a, b))

    def select(self, c, a, b):
        nc = self.nodes[c]
        if nc[0] == "bconst":
            return a if nc[1] else b
        if a == b:
            return a
        if nc[0] == "not":
            return self.select(nc[1], b, a)
        return self._new(("sel", c, a, b))

    def to_float(self, i):
        """bool node -> 0.0 / 1.0 (a float node passes)."""
        return self.select(i, self.const(1.0), self.const(0.0)) if self.is_bool(i) else i

    def to_bool(self, i):
        return i if self.is_bool(i) else self.compare("ne", i, self.const(0.0))

    # ---- analysis ----------------------------------------------------------------------------------------------
    def reachable(self, outs):
        """Sorted ids of every node the given outputs depend on (operands first)."""
        seen = set()
        stack = [int(o) for o in outs]
        while stack:
            i = stack.pop()
            if i in seen:
                continue
            seen.add(i)
            n = self.nodes[i]
            if n[0] in ("const", "bconst", "in"):
                continue
            stack.extend(n[1:])
        return sorted(seen)

    def depends_on_input(self, i):
        return any(self.nodes[j][0] == "in" for j in self.reachable([i]))

    # ---- reverse-mode differentiation ------------------------------------------------------------------------------
    def grad(self, out, wrt=None):
        """d out / d wrt[k] as node ids (wrt defaults to every input).  Non-differentiable operations (comparisons,
        floor, sign ...) contribute zero, as in autograd."""
        wrt = self.inputs if wrt is None else list(wrt)
        live = self.reachable([out])
        adj = {int(out): self.const(1.0)}
        zero = self.const(0.0)
        for i in reversed(live):
            a = adj.get(i)
            if a is None or a == zero:
                continue
            n = self.nodes[i]
            op = n[0]
            if op in ("const", "bconst", "in") or op in COMPARE or op in BOOL:
                continue

            def acc(x, v):
                if v != zero:
                    adj[x] = self.add(adj[x], v) if x in adj else v
            if op == "add":
                acc(n[1], a); acc(n[2], a)
            elif op == "sub":
                acc(n[1], a); acc(n[2], self.neg(a))
            elif op == "mul":
                acc(n[1], self.mul(a, n[2])); acc(n[2], self.mul(a, n[1]))
            elif op == "div":
                q = self.div(a, n[2])
                acc(n[1], q); acc(n[2], self.neg(self.mul(q, i)))
            elif op == "neg":
                acc(n[1], self.neg(a))
            elif op == "exp":
                acc(n[1], self.mul(a, i))
            elif op == "log":
                acc(n[1], self.div(a, n[1]))
            elif op == "sqrt":
                acc(n[1], self.div(self.mul(a, self.const(0.5)), i))
            elif op == "rsqrt":
                acc(n[1], self.mul(a, self.mul(self.const(-0.5), self.div(i, n[1]))))
            elif op == "recip":
                acc(n[1], self.neg(self.mul(a, self.mul(i, i))))
            elif op == "tanh":
                acc(n[1], self.mul(a, self.sub(self.const(1.0), self.mul(i, i))))
            elif op == "sigmoid":
                acc(n[1], self.mul(a, self.mul(i, self.sub(self.const(1.0), i))))
            elif op == "softplus":
                acc(n[1], self.mul(a, self.unary("sigmoid", n[1])))
            elif op == "log1p":
                acc(n[1], self.div(a, self.add(n[1], self.const(1.0))))
            elif op == "expm1":
                acc(n[1], self.mul(a, self.add(i, self.const(1.0))))
            elif op == "sin":
                acc(n[1], self.mul(a, self.unary("cos", n[1])))
            elif op == "cos":
                acc(n[1], self.neg(self.mul(a, self.unary("sin", n[1]))))
            elif op == "atan":
                acc(n[1], self.div(a, self.add(self.const(1.0), self.mul(n[1], n[1]))))
            elif op == "abs":
                acc(n[1], self.mul(a, self.unary("sign", n[1])))
            elif op == "erf":
                acc(n[1], self.mul(a, self.mul(self.const(_TWO_OVER_SQRT_PI), self.unary("exp", self.neg(self.mul(n[1], n[1]))))))
            elif op == "lgamma":
                acc(n[1], self.mul(a, self.unary("digamma", n[1])))
            elif op == "digamma":
                raise Unsupported("derivative of digamma (a second derivative of lgamma)")
            elif op in ("sign", "floor", "ceil", "round", "trunc", "detach"):
                pass
            elif op == "pow":
                x, y = n[1], n[2]
                acc(x, self.mul(a, self.mul(y, self.binary("pow", x, self.sub(y, self.const(1.0))))))
                if not self.is_const(y):
                    acc(y, self.mul(a, self.mul(i, self.unary("log", x))))
            elif op in ("max", "min"):
                c = self.compare("ge" if op == "max" else "le", n[1], n[2])
                acc(n[1], self.select(c, a, zero)); acc(n[2], self.select(c, zero, a))
            elif op == "sel":
                acc(n[2], self.select(n[1], a, zero)); acc(n[3], self.select(n[1], zero, a))
            else:       # pragma: no cover
                raise Unsupported("no derivative rule for %r" % (op,))
        return [adj.get(int(w), zero) for w in wrt]

    # ---- numpy evaluation (tests / verification) ---------------------------------------------------------------
    def evaluate(self, outs, theta, dtype=None):
        """Values of the nodes `outs` at the points theta[..., D]: array [..., len(outs)] in `dtype` (the dtype every operation
        is carried out in: float32 shows what the device computes up to the rounding of fused multiply-adds)."""
        theta = np.asarray(theta)
        dt = np.dtype(dtype or theta.dtype)
        val = {}
        with np.errstate(all="ignore"):
            for i in self.reachable(outs):
                n = self.nodes[i]
                op = n[0]
                if op == "const":
                    v = np.asarray(self.cval(i), dt)
                elif op == "bconst":
                    v = np.asarray(n[1])
                elif op == "in":
                    v = theta[..., n[1]].astype(dt)
                elif op in _NP_UNARY:
                    v = np.asarray(_NP_UNARY[op](val[n[1]]), dt)
                elif op in _NP_BINARY:
                    v = np.asarray(_NP_BINARY[op](val[n[1]], val[n[2]]), dt)
                elif op in _NP_COMPARE:
                    v = _NP_COMPARE[op](val[n[1]], val[n[2]])
                elif op == "and":
                    v = np.logical_and(val[n[1]], val[n[2]])
                elif op == "or":
                    v = np.logical_or(val[n[1]], val[n[2]])
                elif op == "not":
                    v = np.logical_not(val[n[1]])
                elif op == "isnan":
                    v = np.isnan(val[n[1]])
                elif op == "isinf":
                    v = np.isinf(val[n[1]])
                elif op == "sel":
                    v = np.where(val[n[1]], val[n[2]], val[n[3]])
                else:       # pragma: no cover
                    raise Unsupported("evaluate: %r" % (op,))
                val[i] = v
        shape = theta.shape[:-1]
        out = np.empty(shape + (len(outs),), dt)
        for k, o in enumerate(outs):
            out[..., k] = np.broadcast_to(val[int(o)], shape)
        return out


class Unsupported(Exception):
    """The callback uses something the compiler does not lower; the caller keeps the torch-evaluated path and reports the reason."""


def hessian(g: Graph, grads):
    """Rows of d grads[i] / d theta_j as node ids (symmetric up to the order of operations)."""
    return [g.grad(gi) for gi in grads]

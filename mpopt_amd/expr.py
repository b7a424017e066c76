"""Operator-overloading expression tracer, symbolic differentiation and HIP code emission.

The reference traces the user's OCP callables over CasADi ``SX`` symbols
(mpopt.py:196-206, 277-298) and lets ``ca.nlpsol`` differentiate the resulting graph
(mpopt.py:757).  This module plays that role for the MI355X build: the same Python
callables are traced once into a hash-consed expression DAG, differentiated symbolically
(first and second order, structural zeros detected exactly as ``SX`` would by dropping
``0*x``), and emitted as straight-line ``__device__`` code that is inlined into the
hand-written collocation kernels (mpopt_amd/csrc/mpx_kernels.hip).

Nothing here runs on the hot path: tracing happens once per problem, like ``nlpsol``
construction in the reference ("costly operation", mpopt.py:756).
"""
import math
import numbers

import numpy as np

_UNARY_C = {
    "neg": "-({0})",
    "sqrt": "sqrt({0})",
    "exp": "exp({0})",
    "log": "log({0})",
    "sin": "sin({0})",
    "cos": "cos({0})",
    "tan": "tan({0})",
    "asin": "asin({0})",
    "acos": "acos({0})",
    "atan": "atan({0})",
    "sinh": "sinh({0})",
    "cosh": "cosh({0})",
    "tanh": "tanh({0})",
    "abs": "fabs({0})",
    "sign": "(({0}) > 0.0 ? 1.0 : (({0}) < 0.0 ? -1.0 : 0.0))",
}
_UNARY_PY = {
    "neg": lambda v: -v,
    "sqrt": math.sqrt,
    "exp": math.exp,
    "log": math.log,
    "sin": math.sin,
    "cos": math.cos,
    "tan": math.tan,
    "asin": math.asin,
    "acos": math.acos,
    "atan": math.atan,
    "sinh": math.sinh,
    "cosh": math.cosh,
    "tanh": math.tanh,
    "abs": abs,
    "sign": lambda v: (v > 0) - (v < 0),
}
_BINARY_C = {"add": "{0} + {1}", "sub": "{0} - {1}", "mul": "{0} * {1}", "div": "{0} / {1}", "pow": "pow({0}, {1})"}


class Tracer:
    """Owns one hash-consed DAG."""

    def __init__(self):
        self._table = {}
        self._nodes = []
        self.zero = self.const(0.0)
        self.one = self.const(1.0)

    # -- node construction -------------------------------------------------------------
    def _mk(self, op, args=(), val=None):
        key = (op, tuple(a.id for a in args), val)
        e = self._table.get(key)
        if e is None:
            e = Expr(self, op, tuple(args), val, len(self._nodes))
            self._table[key] = e
            self._nodes.append(e)
        return e

    def const(self, v):
        v = float(v)
        if v == 0.0:
            v = 0.0  # merge -0.0
        return self._mk("const", (), v)

    def var(self, name):
        return self._mk("var", (), name)

    def wrap(self, v):
        if isinstance(v, Expr):
            assert v.tr is self, "expression from a different tracer"
            return v
        if isinstance(v, (numbers.Real, np.floating, np.integer)):
            return self.const(v)
        if isinstance(v, np.ndarray) and v.size == 1:
            return self.const(v.reshape(-1)[0])
        raise TypeError(f"cannot trace object of type {type(v).__name__}")

    # -- algebra with on-the-fly simplification (mirrors SX's 0*x -> 0, x+0 -> x) ------
    def add(self, a, b):
        if a.is_const and b.is_const:
            return self.const(a.val + b.val)
        if a.is_zero:
            return b
        if b.is_zero:
            return a
        if b.op == "neg":
            return self.sub(a, b.args[0])
        if a.op == "neg":
            return self.sub(b, a.args[0])
        if a.id > b.id:
            a, b = b, a
        return self._mk("add", (a, b))

    def sub(self, a, b):
        if a.is_const and b.is_const:
            return self.const(a.val - b.val)
        if b.is_zero:
            return a
        if a.is_zero:
            return self.neg(b)
        if a is b:
            return self.zero
        if b.op == "neg":
            return self.add(a, b.args[0])
        return self._mk("sub", (a, b))

    def mul(self, a, b):
        if a.is_const and b.is_const:
            return self.const(a.val * b.val)
        if a.is_zero or b.is_zero:
            return self.zero
        if a.is_const and a.val == 1.0:
            return b
        if b.is_const and b.val == 1.0:
            return a
        if a.is_const and a.val == -1.0:
            return self.neg(b)
        if b.is_const and b.val == -1.0:
            return self.neg(a)
        if a.op == "neg" and b.op == "neg":
            return self.mul(a.args[0], b.args[0])
        if a.op == "neg":
            return self.neg(self.mul(a.args[0], b))
        if b.op == "neg":
            return self.neg(self.mul(a, b.args[0]))
        if a.id > b.id:
            a, b = b, a
        return self._mk("mul", (a, b))

    def div(self, a, b):
        if a.is_const and b.is_const:
            return self.const(a.val / b.val)
        if a.is_zero:
            return self.zero
        if b.is_const and b.val == 1.0:
            return a
        if b.is_const and b.val == -1.0:
            return self.neg(a)
        if a.op == "neg":
            return self.neg(self.div(a.args[0], b))
        return self._mk("div", (a, b))

    def neg(self, a):
        if a.is_const:
            return self.const(-a.val)
        if a.op == "neg":
            return a.args[0]
        return self._mk("neg", (a,))

    def pow(self, a, b):
        if b.is_const:
            n = b.val
            if n == 0.0:
                return self.one
            if n == 1.0:
                return a
            if n == 0.5:
                return self.unary("sqrt", a)
            if n == int(n) and abs(n) <= 16:
                k = int(abs(n))
                r, base = None, a
                while k:
                    if k & 1:
                        r = base if r is None else self.mul(r, base)
                    base = self.mul(base, base)
                    k >>= 1
                return r if n > 0 else self.div(self.one, r)
        if a.is_const and b.is_const:
            return self.const(a.val ** b.val)
        return self._mk("pow", (a, b))

    def unary(self, op, a):
        if op == "neg":
            return self.neg(a)
        if a.is_const:
            return self.const(_UNARY_PY[op](a.val))
        return self._mk(op, (a,))

    # -- differentiation ---------------------------------------------------------------
    def diff(self, e, x, memo=None):
        """d e / d x for a ``var`` node x (forward symbolic, memoised per x)."""
        assert x.op == "var"
        if memo is None:
            memo = {}
        return self._diff(e, x, memo)

    def _diff(self, e, x, memo):
        r = memo.get(e.id)
        if r is not None:
            return r
        # iterative post-order to stay clear of the recursion limit on long chains
        stack = [e]
        while stack:
            n = stack[-1]
            if n.id in memo:
                stack.pop()
                continue
            pend = [a for a in n.args if a.id not in memo]
            if pend:
                stack.extend(pend)
                continue
            stack.pop()
            memo[n.id] = self._diff_node(n, x, [memo[a.id] for a in n.args])
        return memo[e.id]

    def _diff_node(self, n, x, da):
        op = n.op
        if op == "const":
            return self.zero
        if op == "var":
            return self.one if n is x else self.zero
        if all(d.is_zero for d in da):
            return self.zero
        a = n.args
        if op == "add":
            return self.add(da[0], da[1])
        if op == "sub":
            return self.sub(da[0], da[1])
        if op == "mul":
            return self.add(self.mul(da[0], a[1]), self.mul(a[0], da[1]))
        if op == "div":
            # (a/b)' = a'/b - (a/b) * b'/b
            t1 = self.div(da[0], a[1])
            if da[1].is_zero:
                return t1
            return self.sub(t1, self.mul(n, self.div(da[1], a[1])))
        if op == "neg":
            return self.neg(da[0])
        if op == "pow":
            # general x**y = exp(y log x)
            t = self.zero
            if not da[0].is_zero:
                t = self.add(t, self.mul(self.mul(a[1], self.pow(a[0], self.sub(a[1], self.one))), da[0]))
            if not da[1].is_zero:
                t = self.add(t, self.mul(self.mul(n, self.unary("log", a[0])), da[1]))
            return t
        u, du = a[0], da[0]
        if op == "sqrt":
            return self.div(du, self.mul(self.const(2.0), n))
        if op == "exp":
            return self.mul(n, du)
        if op == "log":
            return self.div(du, u)
        if op == "sin":
            return self.mul(self.unary("cos", u), du)
        if op == "cos":
            return self.neg(self.mul(self.unary("sin", u), du))
        if op == "tan":
            return self.mul(self.add(self.one, self.mul(n, n)), du)
        if op == "asin":
            return self.div(du, self.unary("sqrt", self.sub(self.one, self.mul(u, u))))
        if op == "acos":
            return self.neg(self.div(du, self.unary("sqrt", self.sub(self.one, self.mul(u, u)))))
        if op == "atan":
            return self.div(du, self.add(self.one, self.mul(u, u)))
        if op == "sinh":
            return self.mul(self.unary("cosh", u), du)
        if op == "cosh":
            return self.mul(self.unary("sinh", u), du)
        if op == "tanh":
            return self.mul(self.sub(self.one, self.mul(n, n)), du)
        if op == "abs":
            return self.mul(self.unary("sign", u), du)
        if op == "sign":
            return self.zero
        raise NotImplementedError(op)

    # -- evaluation / emission ---------------------------------------------------------
    @staticmethod
    def toposort(outputs):
        order, seen = [], set()
        for root in outputs:
            stack = [(root, False)]
            while stack:
                n, done = stack.pop()
                if done:
                    order.append(n)
                    continue
                if n.id in seen:
                    continue
                seen.add(n.id)
                stack.append((n, True))
                for a in n.args:
                    if a.id not in seen:
                        stack.append((a, False))
        return order

    def evaluate(self, outputs, env):
        """Numerically evaluate ``outputs`` with ``env`` = {var name: float}."""
        vals = {}
        for n in self.toposort(outputs):
            if n.op == "const":
                vals[n.id] = n.val
            elif n.op == "var":
                vals[n.id] = env[n.val]
            elif n.op in _UNARY_PY:
                vals[n.id] = _UNARY_PY[n.op](vals[n.args[0].id])
            else:
                x, y = vals[n.args[0].id], vals[n.args[1].id]
                vals[n.id] = {"add": x + y, "sub": x - y, "mul": x * y,
                              "div": x / y if n.op == "div" else 0.0,
                              "pow": x ** y if n.op == "pow" else 0.0}[n.op]
        return [vals[o.id] for o in outputs]

    def emit(self, assignments, var_names, indent="  "):
        """Straight-line C for ``assignments`` = [(lhs string, Expr)], sharing sub-expressions.
        ``var_names`` maps var name -> C expression."""
        outs = [e for _, e in assignments]
        order = self.toposort(outs)
        uses = {}
        for n in order:
            for a in n.args:
                uses[a.id] = uses.get(a.id, 0) + 1
        for e in outs:
            uses[e.id] = uses.get(e.id, 0) + 1
        name = {}
        lines = []

        def ref(n):
            return name[n.id]

        for n in order:
            if n.op == "const":
                name[n.id] = _cfloat(n.val)
                continue
            if n.op == "var":
                name[n.id] = var_names[n.val]
                continue
            if n.op in _UNARY_C:
                rhs = _UNARY_C[n.op].format(ref(n.args[0]))
            else:
                rhs = _BINARY_C[n.op].format(ref(n.args[0]), ref(n.args[1]))
            tmp = f"v{n.id}"
            lines.append(f"{indent}const double {tmp} = {rhs};")
            name[n.id] = tmp
        for lhs, e in assignments:
            lines.append(f"{indent}{lhs} = {ref(e)};")
        return lines


def _cfloat(v):
    if math.isinf(v):
        return "(1.0/0.0)" if v > 0 else "(-1.0/0.0)"
    if math.isnan(v):
        return "(0.0/0.0)"
    s = repr(float(v))
    if "e" not in s and "." not in s:
        s += ".0"
    return f"({s})" if v < 0 else s


class Expr:
    __slots__ = ("tr", "op", "args", "val", "id")
    __array_ufunc__ = None

    def __init__(self, tr, op, args, val, id_):
        self.tr, self.op, self.args, self.val, self.id = tr, op, args, val, id_

    @property
    def is_const(self):
        return self.op == "const"

    @property
    def is_zero(self):
        return self.op == "const" and self.val == 0.0

    def depends_on(self, names):
        return any(n.op == "var" and n.val in names for n in Tracer.toposort([self]))

    def __add__(self, o):
        return self.tr.add(self, self.tr.wrap(o))

    def __radd__(self, o):
        return self.tr.add(self.tr.wrap(o), self)

    def __sub__(self, o):
        return self.tr.sub(self, self.tr.wrap(o))

    def __rsub__(self, o):
        return self.tr.sub(self.tr.wrap(o), self)

    def __mul__(self, o):
        return self.tr.mul(self, self.tr.wrap(o))

    def __rmul__(self, o):
        return self.tr.mul(self.tr.wrap(o), self)

    def __truediv__(self, o):
        return self.tr.div(self, self.tr.wrap(o))

    def __rtruediv__(self, o):
        return self.tr.div(self.tr.wrap(o), self)

    def __pow__(self, o):
        return self.tr.pow(self, self.tr.wrap(o))

    def __rpow__(self, o):
        return self.tr.pow(self.tr.wrap(o), self)

    def __neg__(self):
        return self.tr.neg(self)

    def __pos__(self):
        return self

    def __abs__(self):
        return self.tr.unary("abs", self)

    def __bool__(self):
        raise TypeError("traced expressions have no truth value (data-dependent branches are not supported)")

    def __repr__(self):
        if self.op == "const":
            return repr(self.val)
        if self.op == "var":
            return str(self.val)
        return f"{self.op}({', '.join(map(repr, self.args))})"


def _math_fn(op, npfn):
    def f(x):
        if isinstance(x, Expr):
            return x.tr.unary(op, x)
        try:
            import sympy as sp

            if isinstance(x, sp.Expr):
                return getattr(sp, {"abs": "Abs"}.get(op, op))(x)
        except ImportError:  # pragma: no cover
            pass
        return npfn(x)

    f.__name__ = op
    return f


class _Math:
    """Math namespace for OCP callables (the reference's examples call ``ca.sqrt/ca.exp/...``,
    e.g. examples/Multi-phase/multistage_launch_vehicle.py).  Dispatches on the argument type so
    one problem statement serves the tracer, numpy and sympy."""

    sqrt = staticmethod(_math_fn("sqrt", np.sqrt))
    exp = staticmethod(_math_fn("exp", np.exp))
    log = staticmethod(_math_fn("log", np.log))
    sin = staticmethod(_math_fn("sin", np.sin))
    cos = staticmethod(_math_fn("cos", np.cos))
    tan = staticmethod(_math_fn("tan", np.tan))
    asin = staticmethod(_math_fn("asin", np.arcsin))
    acos = staticmethod(_math_fn("acos", np.arccos))
    atan = staticmethod(_math_fn("atan", np.arctan))
    sinh = staticmethod(_math_fn("sinh", np.sinh))
    cosh = staticmethod(_math_fn("cosh", np.cosh))
    tanh = staticmethod(_math_fn("tanh", np.tanh))
    fabs = staticmethod(_math_fn("abs", np.abs))
    pi = math.pi
    inf = math.inf


math_ns = _Math()

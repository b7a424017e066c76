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
import operator
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
_BINARY_C = {"add": "{0} + {1}", "sub": "{0} - {1}", "mul": "{0} * {1}", "div": "{0} / {1}", "pow": "pow({0}, {1})",
             "atan2": "atan2({0}, {1})", "fmax": "fmax({0}, {1})", "fmin": "fmin({0}, {1})",
             "ge": "(({0}) >= ({1}) ? 1.0 : 0.0)"}  # ge: indicator used by the derivatives of fmax / fmin
_BINARY_PY = {"add": lambda x, y: x + y, "sub": lambda x, y: x - y, "mul": lambda x, y: x * y, "div": lambda x, y: x / y,
              "pow": lambda x, y: x ** y, "atan2": math.atan2, "fmax": max, "fmin": min, "ge": lambda x, y: 1.0 if x >= y else 0.0}


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

    def binary(self, op, a, b):
        """atan2 / fmax / fmin / ge (the arithmetic operators have their own simplifying constructors)."""
        if a.is_const and b.is_const:
            return self.const(_BINARY_PY[op](a.val, b.val))
        return self._mk(op, (a, b))

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
        if op == "atan2":  # atan2(y, x): (x y' - y x') / (x^2 + y^2)
            den = self.add(self.mul(a[0], a[0]), self.mul(a[1], a[1]))
            return self.div(self.sub(self.mul(a[1], da[0]), self.mul(a[0], da[1])), den)
        if op in ("fmax", "fmin"):  # derivative of the active argument (ties: the first one)
            first = self.binary("ge", a[0], a[1]) if op == "fmax" else self.binary("ge", a[1], a[0])
            return self.add(self.mul(first, da[0]), self.mul(self.sub(self.one, first), da[1]))
        if op == "ge":
            return self.zero
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
                vals[n.id] = _BINARY_PY[n.op](vals[n.args[0].id], vals[n.args[1].id])
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

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        """numpy functions on traced values (the reference's examples write np.sqrt / np.cos / np.sin / np.dot on
        CasADi symbols): element-wise ufuncs map to tracer operations; arrays broadcast to object arrays."""
        if method != "__call__" or kwargs.get("out") is not None:
            return NotImplemented
        if any(isinstance(a, np.ndarray) and a.ndim > 0 for a in inputs):
            boxed = []  # plain object arrays, so that the element-wise helper is not dispatched back to this method
            for a in inputs:
                if isinstance(a, Expr):
                    cell = np.empty((), dtype=object)
                    cell[()] = a
                    a = cell
                boxed.append(a)
            return np.frompyfunc(lambda *xs: ufunc(*xs), len(inputs), 1)(*boxed)
        args = [a.item() if isinstance(a, np.ndarray) else a for a in inputs]
        tr = self.tr
        if ufunc in _UFUNC_UNARY:
            return tr.unary(_UFUNC_UNARY[ufunc], tr.wrap(args[0]))
        if ufunc in _UFUNC_OPERATOR:
            return _UFUNC_OPERATOR[ufunc](*[a if isinstance(a, Expr) else float(a) for a in args])
        if ufunc in _UFUNC_BINARY:
            return tr.binary(_UFUNC_BINARY[ufunc], tr.wrap(args[0]), tr.wrap(args[1]))
        if ufunc is np.square:
            return args[0] * args[0]
        if ufunc is np.reciprocal:
            return 1.0 / args[0]
        return NotImplemented

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
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.add(self, self.tr.wrap(o))

    def __radd__(self, o):
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.add(self.tr.wrap(o), self)

    def __sub__(self, o):
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.sub(self, self.tr.wrap(o))

    def __rsub__(self, o):
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.sub(self.tr.wrap(o), self)

    def __mul__(self, o):
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.mul(self, self.tr.wrap(o))

    def __rmul__(self, o):
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.mul(self.tr.wrap(o), self)

    def __truediv__(self, o):
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.div(self, self.tr.wrap(o))

    def __rtruediv__(self, o):
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.div(self.tr.wrap(o), self)

    def __pow__(self, o):
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.pow(self, self.tr.wrap(o))

    def __rpow__(self, o):
        if isinstance(o, np.ndarray) and o.ndim > 0:
            return NotImplemented  # numpy broadcasts (through __array_ufunc__)
        return self.tr.pow(self.tr.wrap(o), self)

    def __neg__(self):
        return self.tr.neg(self)

    def __pos__(self):
        return self

    def __abs__(self):
        return self.tr.unary("abs", self)

    # numpy's object-dtype loops (np.sin(array_of_exprs), ...) call a method named like the ufunc
    def _u(op):  # noqa: N805
        return lambda self: self.tr.unary(op, self)

    sin, cos, tan, exp, log, sqrt = _u("sin"), _u("cos"), _u("tan"), _u("exp"), _u("log"), _u("sqrt")
    arcsin, arccos, arctan, sinh, cosh, tanh = _u("asin"), _u("acos"), _u("atan"), _u("sinh"), _u("cosh"), _u("tanh")
    fabs, absolute, sign = _u("abs"), _u("abs"), _u("sign")
    del _u

    def arctan2(self, o):
        return self.tr.binary("atan2", self, self.tr.wrap(o))

    def __bool__(self):
        raise TypeError("traced expressions have no truth value (data-dependent branches are not supported)")

    def __repr__(self):
        if self.op == "const":
            return repr(self.val)
        if self.op == "var":
            return str(self.val)
        return f"{self.op}({', '.join(map(repr, self.args))})"


_UFUNC_UNARY = {np.sin: "sin", np.cos: "cos", np.tan: "tan", np.exp: "exp", np.log: "log", np.sqrt: "sqrt", np.arcsin: "asin",
                np.arccos: "acos", np.arctan: "atan", np.sinh: "sinh", np.cosh: "cosh", np.tanh: "tanh", np.absolute: "abs",
                np.fabs: "abs", np.sign: "sign", np.negative: "neg"}
_UFUNC_OPERATOR = {np.add: operator.add, np.subtract: operator.sub, np.multiply: operator.mul, np.true_divide: operator.truediv,
                   np.power: operator.pow}
_UFUNC_BINARY = {np.arctan2: "atan2", np.maximum: "fmax", np.minimum: "fmin", np.fmax: "fmax", np.fmin: "fmin"}


def symvec(items):
    """Vector of traced values handed to the user callables: a 1-D numpy object array, so that the statements the
    reference's examples apply to CasADi column vectors work unchanged -- slices (``x[:3]``), ``scalar * x[3:6]``,
    ``x[-1]``, ``len(x)``, numpy functions."""
    out = np.empty(len(items), dtype=object)
    for k, v in enumerate(items):
        out[k] = v
    return out


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
    sign = staticmethod(_math_fn("sign", np.sign))
    pi = math.pi
    inf = math.inf

    # two-argument functions and the CasADi spellings the reference's examples use on symbols
    @staticmethod
    def _binary(op, npfn, spname):
        def f(a, b):
            for x in (a, b):
                if isinstance(x, Expr):
                    return x.tr.binary(op, x.tr.wrap(a), x.tr.wrap(b))
            try:
                import sympy as sp

                if isinstance(a, sp.Expr) or isinstance(b, sp.Expr):
                    return getattr(sp, spname)(a, b)
            except ImportError:  # pragma: no cover
                pass
            return npfn(a, b)

        return f

    @staticmethod
    def vertcat(*xs):
        """ca.vertcat of scalars / vectors -> a flat numpy vector (float, or object for traced / symbolic entries), so that
        ``scalar * vertcat(...)`` and indexing behave as on a CasADi column."""
        out = []
        for x in xs:
            out += list(np.asarray(x, dtype=object).reshape(-1)) if isinstance(x, (list, tuple, np.ndarray)) else [x]
        if all(isinstance(v, (numbers.Real, np.floating, np.integer)) for v in out):
            return np.array(out, dtype=float)
        return symvec(out)  # traced (or sympy) entries: an object vector, arithmetic is element-wise

    @staticmethod
    def sumsqr(xs):
        xs = _Math.vertcat(xs)
        r = xs[0] * xs[0]
        for v in xs[1:]:
            r = r + v * v
        return r

    @staticmethod
    def dot(a, b):
        a, b = _Math.vertcat(a), _Math.vertcat(b)
        r = a[0] * b[0]
        for x, y in zip(a[1:], b[1:]):
            r = r + x * y
        return r


_Math.atan2 = staticmethod(_Math._binary("atan2", np.arctan2, "atan2"))
_Math.fmax = staticmethod(_Math._binary("fmax", np.maximum, "Max"))
_Math.fmin = staticmethod(_Math._binary("fmin", np.minimum, "Min"))
_Math.norm_2 = staticmethod(lambda xs: _Math.sqrt(_Math.sumsqr(xs)))
# (numpy spellings; `_Math.asin` read through the class is the bare function: wrap again, or an instance would bind it as a method)
_Math.arcsin, _Math.arccos, _Math.arctan, _Math.arctan2 = (staticmethod(f) for f in (_Math.asin, _Math.acos, _Math.atan, _Math.atan2))
_Math.power = staticmethod(lambda a, b: a ** b)


math_ns = _Math()

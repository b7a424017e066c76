"""Exact first and second derivatives of a restated VALUE code by operator overloading (TEST INFRASTRUCTURE: only tests/, bench.py's
cpu leg and __graft_entry__.smoke() may import anything under oracle/).

Why: the oracle of the widths-as-variables NLP (``OracleAdaptiveNLP``, reference mpopt.py:2927-2979, 3034-3136) restates f and g line by
line and is generic in the number type; its derivatives came from sympy on the WHOLE NLP, which stops being practical above a few
segments, so at the bench size (20 x 5) and beyond the GPU kernels were checked against finite differences of their own outputs
(missing 5).  ``SD`` is a sparse hyper-dual number: a value, a sparse gradient {variable: d/dz} and -- in second-order mode
-- a sparse upper-triangular Hessian {(i, j): d2/dz_i dz_j}.  Running the SAME value code on SD numbers gives jac_g, grad_f and the
Hessian of the Lagrangian exactly (to rounding; no step size, no symbolic expression swell), by a route that shares nothing with the
product's per-point AD + chain-rule expansion (mpopt_amd/assembly.py).  Pinned to the sympy derivatives and to the reference's goldens
on the five adaptive golden cases (tests/test_oracle.py).

The user callables of tests/problems.py reach numpy ufuncs through ``mpopt_amd.math`` (np.cos(x) ...): ``__array_ufunc__`` maps them here.
"""
import math


class SD:
    __slots__ = ("v", "g", "h")
    ORDER = 2  # 1: gradients only (the Hessian dictionaries are never formed)

    def __init__(self, v, g=None, h=None):
        self.v, self.g, self.h = float(v), (g if g is not None else {}), (h if h is not None else {})

    @staticmethod
    def var(v, index):
        return SD(v, {int(index): 1.0})

    # -- helpers ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _lin(a, ca, b, cb):
        """ca * a + cb * b for dictionaries."""
        if not b or cb == 0.0:
            return {k: ca * x for k, x in a.items()} if ca != 1.0 else dict(a)
        out = {k: ca * x for k, x in a.items()} if ca != 1.0 else dict(a)
        for k, x in b.items():
            out[k] = out.get(k, 0.0) + cb * x
        return out

    @staticmethod
    def _outer_into(h, ga, gb, scale):
        """h += scale * (ga gb^T + gb ga^T), upper triangle."""
        if scale == 0.0:
            return
        for i, x in ga.items():
            sx = scale * x
            for j, y in gb.items():
                if i < j:
                    k = (i, j)
                    h[k] = h.get(k, 0.0) + sx * y
                elif i > j:
                    k = (j, i)
                    h[k] = h.get(k, 0.0) + sx * y
                else:
                    k = (i, i)
                    h[k] = h.get(k, 0.0) + 2.0 * sx * y

    def _unary(self, f0, f1, f2):
        """phi(self) from phi, phi', phi'' at self.v."""
        h = {}
        if SD.ORDER >= 2:
            h = {k: f1 * x for k, x in self.h.items()} if self.h else {}
            if f2 != 0.0 and self.g:
                SD._outer_into(h, self.g, self.g, 0.5 * f2)  # (ga ga^T + ga ga^T) / 2
        return SD(f0, {k: f1 * x for k, x in self.g.items()}, h)

    # -- arithmetic ---------------------------------------------------------------------------------------------------------------
    def __add__(self, o):
        if isinstance(o, SD):
            return SD(self.v + o.v, SD._lin(self.g, 1.0, o.g, 1.0), SD._lin(self.h, 1.0, o.h, 1.0) if SD.ORDER >= 2 else {})
        return SD(self.v + float(o), self.g, self.h)

    __radd__ = __add__

    def __neg__(self):
        return SD(-self.v, {k: -x for k, x in self.g.items()}, {k: -x for k, x in self.h.items()})

    def __sub__(self, o):
        if isinstance(o, SD):
            return SD(self.v - o.v, SD._lin(self.g, 1.0, o.g, -1.0), SD._lin(self.h, 1.0, o.h, -1.0) if SD.ORDER >= 2 else {})
        return SD(self.v - float(o), self.g, self.h)

    def __rsub__(self, o):
        return (-self) + float(o)

    def __mul__(self, o):
        if isinstance(o, SD):
            h = {}
            if SD.ORDER >= 2:
                h = SD._lin(self.h, o.v, o.h, self.v)
                SD._outer_into(h, self.g, o.g, 1.0)
            return SD(self.v * o.v, SD._lin(self.g, o.v, o.g, self.v), h)
        c = float(o)
        if c == 0.0:
            return 0.0  # (structural zeros of the dense composite matrices)
        if c == 1.0:
            return self
        return SD(self.v * c, {k: c * x for k, x in self.g.items()}, {k: c * x for k, x in self.h.items()})

    __rmul__ = __mul__

    def reciprocal(self):
        r = 1.0 / self.v
        return self._unary(r, -r * r, 2.0 * r * r * r)

    def __truediv__(self, o):
        if isinstance(o, SD):
            return self * o.reciprocal()
        return self * (1.0 / float(o))

    def __rtruediv__(self, o):
        return self.reciprocal() * float(o)

    def __pow__(self, p):
        if isinstance(p, SD):
            return (p * self.log()).exp()
        p = float(p)
        if p == 2.0:
            return self * self
        if p == int(p) and 0 <= p <= 4:
            out = 1.0
            for _ in range(int(p)):
                out = self * out
            return out
        v = self.v ** p
        return self._unary(v, p * self.v ** (p - 1.0), p * (p - 1.0) * self.v ** (p - 2.0))

    def __rpow__(self, b):
        return (self * math.log(float(b))).exp()

    def __float__(self):
        return self.v

    # -- functions (numpy ufuncs land here through __array_ufunc__) ------------------------------------------------------------------
    def sqrt(self):
        s = math.sqrt(self.v)
        return self._unary(s, 0.5 / s, -0.25 / (s * self.v))

    def exp(self):
        e = math.exp(self.v)
        return self._unary(e, e, e)

    def log(self):
        return self._unary(math.log(self.v), 1.0 / self.v, -1.0 / (self.v * self.v))

    def sin(self):
        s, c = math.sin(self.v), math.cos(self.v)
        return self._unary(s, c, -s)

    def cos(self):
        s, c = math.sin(self.v), math.cos(self.v)
        return self._unary(c, -s, -c)

    def tan(self):
        t = math.tan(self.v)
        return self._unary(t, 1.0 + t * t, 2.0 * t * (1.0 + t * t))

    def arcsin(self):
        d = 1.0 - self.v * self.v
        return self._unary(math.asin(self.v), 1.0 / math.sqrt(d), self.v / (d * math.sqrt(d)))

    def arccos(self):
        d = 1.0 - self.v * self.v
        return self._unary(math.acos(self.v), -1.0 / math.sqrt(d), -self.v / (d * math.sqrt(d)))

    def arctan(self):
        d = 1.0 + self.v * self.v
        return self._unary(math.atan(self.v), 1.0 / d, -2.0 * self.v / (d * d))

    def sinh(self):
        return self._unary(math.sinh(self.v), math.cosh(self.v), math.sinh(self.v))

    def cosh(self):
        return self._unary(math.cosh(self.v), math.sinh(self.v), math.cosh(self.v))

    def tanh(self):
        t = math.tanh(self.v)
        return self._unary(t, 1.0 - t * t, -2.0 * t * (1.0 - t * t))

    # piecewise functions, as AD takes them away from the kinks (ties: the first argument, as mpopt_amd/expr.py and CasADi's fmax / fmin at a
    # generic point): d|x| = sign x, d sign = 0, d max = the active argument's
    def absolute(self):
        return self._unary(abs(self.v), math.copysign(1.0, self.v) if self.v != 0.0 else 0.0, 0.0)

    def sign(self):
        return SD(math.copysign(1.0, self.v) if self.v != 0.0 else 0.0)

    def maximum(self, o):
        return self if self.v >= value(o) else o

    def minimum(self, o):
        return self if self.v <= value(o) else o

    def _rmaximum(self, o):  # max(o, self), o a plain number
        return o if float(o) >= self.v else self

    def _rminimum(self, o):
        return o if float(o) <= self.v else self

    def arctan2(self, x):
        """atan2(self, x), x != 0: the derivatives of atan(self / x), the value on the right branch."""
        r = (self / x).arctan()
        return SD(math.atan2(self.v, value(x)), r.g, r.h)

    def _rarctan2(self, y):
        r = (float(y) / self).arctan()
        return SD(math.atan2(float(y), self.v), r.g, r.h)

    _UFUNCS = {"absolute": "absolute", "fabs": "absolute", "sign": "sign", "sqrt": "sqrt", "exp": "exp", "log": "log", "sin": "sin", "cos": "cos", "tan": "tan", "arcsin": "arcsin", "arccos": "arccos",
               "arctan": "arctan", "sinh": "sinh", "cosh": "cosh", "tanh": "tanh", "negative": "__neg__", "reciprocal": "reciprocal"}
    _BINARY = {"add": "__add__", "subtract": "__sub__", "multiply": "__mul__", "true_divide": "__truediv__", "divide": "__truediv__", "power": "__pow__",
               "maximum": "maximum", "fmax": "maximum", "minimum": "minimum", "fmin": "minimum", "arctan2": "arctan2"}

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs:
            return NotImplemented
        name = ufunc.__name__
        if len(inputs) == 1 and name in SD._UFUNCS:
            return getattr(self, SD._UFUNCS[name])()
        if len(inputs) == 2 and name in SD._BINARY:
            a, b = inputs
            if isinstance(a, SD):
                return getattr(a, SD._BINARY[name])(b)
            r = {"add": "__radd__", "subtract": "__rsub__", "multiply": "__rmul__", "true_divide": "__rtruediv__", "divide": "__rtruediv__",
                 "power": "__rpow__", "maximum": "_rmaximum", "fmax": "_rmaximum", "minimum": "_rminimum", "fmin": "_rminimum", "arctan2": "_rarctan2"}[name]
            return getattr(b, r)(a)
        return NotImplemented


def value(x):
    return x.v if isinstance(x, SD) else float(x)


def gradient(x):
    return x.g if isinstance(x, SD) else {}


def hessian(x):
    return x.h if isinstance(x, SD) else {}

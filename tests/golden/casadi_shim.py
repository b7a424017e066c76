"""A sympy-backed stand-in for the small part of the CasADi Python API that the reference's
transcription code touches (mpopt.py call sites listed in SURVEY.md section 2).

WHY: CasADi (the reference's pinned third-party dependency, casadi==3.6.0, requirements.txt:4)
is not installable in the build container.  With this module registered as ``casadi`` the
reference's own ``mpopt.create_nlp()`` (mpopt.py:574-639) runs unmodified and returns the NLP
``{"f","x","g","p"}`` as sympy expressions; ``make_golden.py`` differentiates them with sympy
and stores numbers.  It is test infrastructure: used ONLY by ``tests/golden/make_golden.py``,
never shipped in the product, never present on the GPU box.  What it is NOT: it is not CasADi's
AD/VM, so the goldens pin the reference's *transcription* (row order, scaling, quirks, values)
but not CasADi's floating-point evaluation order (differences are O(1e-16) relative).

Semantics mirrored from CasADi: matrices are 2-D, column-major linearisation for ``m[:]`` and
single-index access, 1x1 matrices broadcast in element-wise arithmetic, ``vertcat`` skips empty
operands, products with a structural/numerical zero are dropped.
"""
import numbers

import numpy as np
import sympy as sp


def _is_scalar(v):
    return isinstance(v, (numbers.Number, sp.Expr, np.generic))


def _clean(v):
    if isinstance(v, np.generic):
        return v.item()
    return v


class M:
    """Dense 2-D matrix of python floats / sympy expressions."""

    __array_ufunc__ = None  # make numpy scalars defer to our reflected operators

    def __init__(self, data=None):
        if data is None:
            data = np.zeros((0, 0), dtype=object)
        if isinstance(data, M):
            data = data.a.copy()
        elif _is_scalar(data):
            data = np.array([[_clean(data)]], dtype=object)
        else:
            if isinstance(data, (list, tuple)):  # e.g. ca.DM([expr, expr, ...]) with 1x1 entries
                data = [e.scalar() if isinstance(e, M) and e.a.size == 1 else _clean(e) for e in data]
            data = np.array(data, dtype=object)
            if data.ndim == 0:
                data = data.reshape(1, 1)
            elif data.ndim == 1:
                data = data.reshape(-1, 1)
        self.a = data

    # -- shape -------------------------------------------------------------------------
    @property
    def shape(self):
        return self.a.shape

    def size1(self):
        return self.a.shape[0]

    def size2(self):
        return self.a.shape[1]

    def numel(self):
        return self.a.size

    def size(self):
        return self.a.shape

    def is_empty(self):
        return self.a.size == 0

    @property
    def T(self):
        return type(self)(self.a.T.copy())

    def full(self):
        return np.array(self.a, dtype=float)

    def __array__(self, dtype=None, copy=None):
        return np.array(self.a, dtype=float if dtype is None else dtype)

    def __float__(self):
        assert self.a.size == 1
        return float(self.a.flat[0])

    def __len__(self):
        return self.a.shape[0]

    def scalar(self):
        assert self.a.size == 1, self.a.shape
        return self.a[0, 0]

    # -- indexing ----------------------------------------------------------------------
    def _lin(self):
        return self.a.reshape(-1, order="F")

    def __getitem__(self, key):
        if isinstance(key, tuple):
            r, c = key
            sub = self.a[_as_index(r, self.a.shape[0]), :][:, _as_index(c, self.a.shape[1])]
            return type(self)(sub)
        lin = self._lin()
        sub = lin[_as_index(key, lin.size)]
        return type(self)(sub.reshape(-1, 1))

    def __setitem__(self, key, value):
        if isinstance(value, M):
            v = value.a
        elif _is_scalar(value):
            v = _clean(value)
        else:
            v = np.array(value, dtype=object)
        if isinstance(key, tuple):
            r, c = key
            ri, ci = _as_index(r, self.a.shape[0]), _as_index(c, self.a.shape[1])
            if isinstance(v, np.ndarray):
                v = v.reshape(len(ri), len(ci))
                for ii, rr in enumerate(ri):
                    for jj, cc in enumerate(ci):
                        self.a[rr, cc] = v[ii, jj]
            else:
                for rr in ri:
                    for cc in ci:
                        self.a[rr, cc] = v
        else:
            n0 = self.a.shape[0]
            idx = _as_index(key, self.a.size)
            if isinstance(v, np.ndarray):
                v = v.reshape(-1, order="F")
                if v.size == 1:
                    v = [v[0]] * len(idx)
            else:
                v = [v] * len(idx)
            for k, val in zip(idx, v):
                self.a[k % n0, k // n0] = val

    def __iter__(self):
        raise TypeError("casadi matrices are not iterable in the shim")

    # -- arithmetic --------------------------------------------------------------------
    def _bin(self, other, op, reflect=False):
        if isinstance(other, (list, tuple)) and len(other) == 0:
            other = 0
        o = other if isinstance(other, M) else M(other)
        a, b = (o.a, self.a) if reflect else (self.a, o.a)
        if a.shape != b.shape:
            if a.size == 1:
                a = np.broadcast_to(a, b.shape)
            elif b.size == 1:
                b = np.broadcast_to(b, a.shape)
            else:
                raise ValueError(f"shape mismatch {a.shape} vs {b.shape}")
        out = np.empty(a.shape, dtype=object)
        for idx in np.ndindex(a.shape):
            out[idx] = op(a[idx], b[idx])
        cls = SX if (isinstance(self, SX) or isinstance(o, SX)) else DM
        return cls(out)

    def __add__(self, o):
        return self._bin(o, lambda x, y: x + y)

    def __radd__(self, o):
        return self._bin(o, lambda x, y: x + y, True)

    def __sub__(self, o):
        return self._bin(o, lambda x, y: x - y)

    def __rsub__(self, o):
        return self._bin(o, lambda x, y: x - y, True)

    def __mul__(self, o):
        return self._bin(o, _mul)

    def __rmul__(self, o):
        return self._bin(o, _mul, True)

    def __truediv__(self, o):
        return self._bin(o, lambda x, y: x / y)

    def __rtruediv__(self, o):
        return self._bin(o, lambda x, y: x / y, True)

    def __pow__(self, o):
        return self._bin(o, lambda x, y: x ** y)

    def __rpow__(self, o):
        return self._bin(o, lambda x, y: x ** y, True)

    def __neg__(self):
        return type(self)(-self.a)

    def __pos__(self):
        return self

    def __repr__(self):
        return f"{type(self).__name__}({self.a.tolist()})"


def _mul(x, y):
    # CasADi's SX simplifies products with an exact zero on the fly
    if _is_zero(x) or _is_zero(y):
        return 0.0
    return x * y


def _is_zero(v):
    if isinstance(v, sp.Expr):
        return v == 0
    return v == 0


def _as_index(key, n):
    if isinstance(key, slice):
        return list(range(*key.indices(n)))
    if isinstance(key, M):
        key = int(float(key))
    if isinstance(key, (int, np.integer)):
        k = int(key)
        if k < 0:
            k += n
        return [k]
    return [int(k) if k >= 0 else int(k) + n for k in key]


class DM(M):
    @staticmethod
    def zeros(*shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        if len(shape) == 1:
            shape = (shape[0], 1)
        out = np.empty(shape, dtype=object)
        out[...] = 0.0
        return DM(out)

    @staticmethod
    def eye(n):
        out = np.empty((n, n), dtype=object)
        out[...] = 0.0
        for i in range(n):
            out[i, i] = 1.0
        return DM(out)


class SX(M):
    @staticmethod
    def sym(name, *dims):
        dims = [int(d) for d in dims]
        if len(dims) == 0:
            return SX(sp.Symbol(name, real=True))
        if len(dims) == 1:
            dims = [dims[0], 1]
        if len(dims) == 2:
            n, m = dims
            arr = np.empty((n, m), dtype=object)
            for j in range(m):
                for i in range(n):
                    arr[i, j] = sp.Symbol(f"{name}_{i}_{j}", real=True)
            return SX(arr)
        n, m, k = dims
        return [SX.sym(f"{name}{q}", n, m) for q in range(k)]


# -- free functions ---------------------------------------------------------------------
def _to_m(x):
    if isinstance(x, M):
        return x
    if isinstance(x, (list, tuple)) and len(x) == 0:
        return M()
    return DM(np.array(x, dtype=object)) if not _is_scalar(x) else DM(x)


def vertcat(*xs):
    parts, sym = [], False
    for x in xs:
        m = _to_m(x)
        if m.a.size == 0:
            continue
        sym = sym or isinstance(m, SX) or any(isinstance(v, sp.Expr) for v in m.a.flat)
        parts.append(m.a)
    if not parts:
        return DM(np.zeros((0, 1), dtype=object))
    out = np.vstack(parts)
    return SX(out) if sym else DM(out)


def horzcat(*xs):
    return vertcat(*[_to_m(x).T for x in xs]).T


def mtimes(a, b):
    a, b = _to_m(a), _to_m(b)
    if a.a.size == 1 or b.a.size == 1:
        return a * b
    assert a.shape[1] == b.shape[0], (a.shape, b.shape)
    out = np.empty((a.shape[0], b.shape[1]), dtype=object)
    for i in range(a.shape[0]):
        nz = [k for k in range(a.shape[1]) if not _is_zero(a.a[i, k])]
        for j in range(b.shape[1]):
            acc = 0.0
            first = True
            for k in nz:
                if _is_zero(b.a[k, j]):
                    continue
                term = a.a[i, k] * b.a[k, j]
                acc = term if first else acc + term
                first = False
            out[i, j] = acc
    cls = SX if (isinstance(a, SX) or isinstance(b, SX)) else DM
    return cls(out)


def diag(v):
    v = _to_m(v)
    n = v.a.size
    out = DM.zeros((n, n))
    flat = v.a.reshape(-1, order="F")
    for i in range(n):
        out.a[i, i] = flat[i]
    return out


def solve(a, b):
    a = _to_m(a)
    return DM(np.linalg.solve(np.array(a.a, dtype=float), np.array(b, dtype=float)))


def kron(a, b):
    a, b = _to_m(a), _to_m(b)
    n1, m1 = a.shape
    n2, m2 = b.shape
    out = np.empty((n1 * n2, m1 * m2), dtype=object)
    out[...] = 0.0
    for i in range(n1):
        for j in range(m1):
            if _is_zero(a.a[i, j]):
                continue
            for p in range(n2):
                for q in range(m2):
                    out[i * n2 + p, j * m2 + q] = _mul(a.a[i, j], b.a[p, q])
    return DM(out)


def sum1(x):
    x = _to_m(x)
    out = np.empty((1, x.shape[1]), dtype=object)
    for j in range(x.shape[1]):
        out[0, j] = sum(x.a[:, j])
    return type(x)(out)


def gradient(expr, var):
    e, v = _to_m(expr).scalar(), _to_m(var).scalar()
    return SX(sp.diff(e, v) if isinstance(e, sp.Expr) else 0.0)


class Function:
    def __init__(self, name, args, outs, *rest):
        self._args = [_to_m(a) for a in args]
        self._outs = [_to_m(o) for o in outs]
        syms = [s for a in self._args for s in a.a.reshape(-1, order="F")]
        self._f = sp.lambdify(syms, [list(o.a.reshape(-1, order="F")) for o in self._outs], "math")

    def __call__(self, *vals):
        flat = []
        for v in vals:
            flat.extend(np.array(_to_m(v).a, dtype=float).reshape(-1, order="F"))
        res = self._f(*flat)
        outs = [DM(np.array(r, dtype=object).reshape(o.shape, order="F")) for r, o in zip(res, self._outs)]
        return outs[0] if len(outs) == 1 else outs


def integrator(name, plugin, dae, opts):
    """Exact polynomial quadrature in place of IDAS (mpopt.py:3869-3877 integrates basis
    polynomials; the real integrator is tolerance-limited, this one is exact)."""
    t = _to_m(dae["t"]).scalar()
    ode = _to_m(dae["ode"]).scalar()
    val = sp.integrate(sp.expand(ode), (t, opts["t0"], opts["tf"]))

    def run(x0=0, **kw):
        return {"xf": DM(float(val) + float(x0))}

    return run


def _unary(fn_sym, fn_num):
    def f(x):
        if isinstance(x, M):
            out = np.empty(x.shape, dtype=object)
            for idx in np.ndindex(x.shape):
                v = x.a[idx]
                out[idx] = fn_sym(v) if isinstance(v, sp.Expr) else fn_num(v)
            return type(x)(out)
        if isinstance(x, sp.Expr):
            return fn_sym(x)
        return fn_num(x)

    return f


sqrt = _unary(sp.sqrt, np.sqrt)
exp = _unary(sp.exp, np.exp)
log = _unary(sp.log, np.log)
sin = _unary(sp.sin, np.sin)
cos = _unary(sp.cos, np.cos)
tan = _unary(sp.tan, np.tan)
acos = _unary(sp.acos, np.arccos)
asin = _unary(sp.asin, np.arcsin)
atan = _unary(sp.atan, np.arctan)
tanh = _unary(sp.tanh, np.tanh)
fabs = _unary(sp.Abs, np.abs)
pi = np.pi
inf = np.inf


def nlpsol(*a, **k):
    raise RuntimeError("the shim has no NLP solver; use create_nlp() only")


class Callback:
    pass

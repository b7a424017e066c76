"""Host-side mirror of the reference's public interface for the collocation hot path.

Same names, argument meaning and defaults as ``/root/reference/mpopt/mpopt.py`` for

    OCP                 (mpopt.py:3378-3703)   problem container -- pure data + arity adapters
    CollocationRoots    (mpopt.py:4134-4276)   node sets
    Collocation         (mpopt.py:3706-4131)   D / W / interpolation tables and composites
    mpopt               (mpopt.py:31-855)      transcription OCP -> NLP and the solve entry
    solve               (mpopt.py:4279-4308)

but nothing here builds a CasADi graph: the tables come from libmpx (mpx_colloc_*), and the NLP
functions f, g, grad_f, jac_g, hess_l are ``NlpFunctions`` (GPU kernels behind the C ABI).
Post-solve analysis, plotting and the adaptive subclasses are out of scope.
"""
import copy
import time

import numpy as np

from . import _lib
from .nlp import NlpFunctions


# ---------------------------------------------------------------------------------------------
class OCP:
    """Optimal control problem in Bolza form; attribute-for-attribute the reference's ``OCP``."""

    LB_DYNAMICS = 0
    UB_DYNAMICS = 0
    LB_PATH_CONSTRAINTS = -np.inf
    UB_PATH_CONSTRAINTS = 0
    LB_TERMINAL_CONSTRAINTS = 0
    UB_TERMINAL_CONSTRAINTS = 0

    def __init__(self, n_states=1, n_controls=1, n_phases=1, n_params=0, **kwargs):
        self.nx, self.nu, self.na, self.n_phases = n_states, n_controls, n_params, n_phases
        P = n_phases
        self.dynamics = [lambda x, u, t, a=None: [0] * self.nx] * P
        self.path_constraints = [lambda x, u, t, a=None: None] * P
        self.terminal_costs = [lambda xf, tf, x0, t0, a=None: 0] * P
        self.running_costs = [lambda x, u, t, a=None: 0] * P
        self.terminal_constraints = [lambda xf, tf, x0, t0, a=None: None] * P
        self.phase_links = [(i, i + 1) for i in range(P - 1)]
        # scaling
        self.scale_x, self.scale_u, self.scale_a = np.ones(self.nx), np.ones(self.nu), np.ones(self.na)
        self.scale_t = 1.0
        # initial guess
        self.x00, self.xf0 = np.zeros((P, self.nx)), np.zeros((P, self.nx))
        self.u00, self.uf0 = np.zeros((P, self.nu)), np.zeros((P, self.nu))
        self.t00, self.tf0 = np.zeros((P, 1)), np.ones((P, 1))
        self.a0 = np.zeros((P, self.na))
        # bounds
        self.lbx, self.ubx = np.full((P, self.nx), -np.inf), np.full((P, self.nx), np.inf)
        self.lbu, self.ubu = np.full((P, self.nu), -np.inf), np.full((P, self.nu), np.inf)
        self.lba, self.uba = np.full((P, self.na), -np.inf), np.full((P, self.na), np.inf)
        self.lbt0, self.ubt0 = np.zeros((P, 1)), np.full((P, 1), np.inf)
        self.ubt0[0] = 0.0  # the first phase starts at t = 0
        self.lbtf, self.ubtf = np.zeros((P, 1)), np.full((P, 1), np.inf)
        self.lbe, self.ube = np.zeros((P - 1, self.nx)), np.zeros((P - 1, self.nx))
        # optional row blocks
        self.diff_u = np.zeros(P, dtype=int)
        self.lbdu, self.ubdu = np.full(P, -15), np.full(P, 15)
        self.midu = np.ones(P, dtype=int)
        self.du_continuity = np.zeros(P, dtype=int)
        # post-processing defaults (kept for interface compatibility)
        self.n_figures = 1
        self.phases_to_plot = [tuple(range(P))]
        self.plot_type = 1
        self.plot_interpolation_level = 3

    # arity adapters: user functions omit `a` when the problem has no parameters
    def _node_fn(self, table, phase):
        fn = table[phase]
        return (lambda x, u, t, a: fn(x, u, t)) if self.na == 0 else fn

    def _term_fn(self, table, phase):
        fn = table[phase]
        return (lambda xf, tf, x0, t0, a: fn(xf, tf, x0, t0)) if self.na == 0 else fn

    def get_dynamics(self, phase=0):
        return self._node_fn(self.dynamics, phase)

    def get_path_constraints(self, phase=0):
        return self._node_fn(self.path_constraints, phase)

    def get_running_costs(self, phase=0):
        return self._node_fn(self.running_costs, phase)

    def get_terminal_constraints(self, phase=0):
        return self._term_fn(self.terminal_constraints, phase)

    def get_terminal_costs(self, phase=0):
        return self._term_fn(self.terminal_costs, phase)

    def has_path_constraints(self, phase=0):
        args = (self.x00[phase], self.u00[phase], self.t00[phase]) + ((self.a0[phase],) if self.na else ())
        return self.path_constraints[phase](*args) is not None

    def has_terminal_constraints(self, phase=0):
        args = (self.xf0[phase], self.tf0[phase], self.x00[phase], self.t00[phase]) + ((self.a0[phase],) if self.na else ())
        return self.terminal_constraints[phase](*args) is not None

    def validate(self):
        P = self.n_phases
        assert P > 0
        for name in ("dynamics", "running_costs", "terminal_costs", "path_constraints", "terminal_constraints"):
            assert len(getattr(self, name)) == P, name
        for ph in range(P):
            x, u, t, a = self.x00[ph], self.u00[ph], self.t00[ph], self.a0[ph]
            assert len(self.get_dynamics(ph)(x, u, t, a)) == self.nx
            assert self.get_terminal_costs(ph)(x, t, x, t, a) is not None
            assert self.get_running_costs(ph)(x, u, t, a) is not None
            pc = self.get_path_constraints(ph)(x, u, t, a)
            tc = self.get_terminal_constraints(ph)(x, t, x, t, a)
            assert pc is None or len(pc) > 0
            assert tc is None or len(tc) > 0
        assert len(self.scale_x) == self.nx and len(self.scale_u) == self.nu and len(self.scale_a) == self.na
        for name, n in (("x00", self.nx), ("xf0", self.nx), ("u00", self.nu), ("uf0", self.nu), ("a0", self.na),
                        ("t00", 1), ("tf0", 1), ("lbx", self.nx), ("ubx", self.nx), ("lbu", self.nu), ("ubu", self.nu),
                        ("lba", self.na), ("uba", self.na), ("lbt0", 1), ("ubt0", 1), ("lbtf", 1), ("ubtf", 1)):
            assert np.shape(getattr(self, name)) == (P, n), name
        assert np.shape(self.lbe)[0] == P - 1 and np.shape(self.ube)[0] == P - 1
        if P > 1:
            assert np.shape(self.lbe)[1] == self.nx and np.shape(self.ube)[1] == self.nx
        for ph in range(P):
            assert (np.asarray(self.lbx[ph]) <= np.asarray(self.ubx[ph])).all()
            assert (np.asarray(self.lbu[ph]) <= np.asarray(self.ubu[ph])).all()
            assert (np.asarray(self.lba[ph]) <= np.asarray(self.uba[ph])).all()
            assert self.lbt0[ph] <= self.ubt0[ph] and self.lbtf[ph] <= self.ubtf[ph]
            if ph < P - 1:
                assert (np.asarray(self.lbe[ph]) <= np.asarray(self.ube[ph])).all()


# ---------------------------------------------------------------------------------------------
class _Size(int):
    """ndarray.size stays an int, but is also callable like CasADi's ``DM.size()`` -> (rows, cols)."""

    def __new__(cls, n, shape):
        obj = int.__new__(cls, n)
        obj._shape = tuple(shape)
        return obj

    def __call__(self):
        return self._shape


class _Mat(np.ndarray):
    """ndarray with CasADi-DM's ``full()`` / ``size()`` so callers written against the reference keep working."""

    def full(self):
        return np.asarray(self)

    @property
    def size(self):
        return _Size(int(np.prod(self.shape)), self.shape if self.ndim == 2 else (int(np.prod(self.shape)), 1))

    def size1(self):
        return self.shape[0]

    def size2(self):
        return self.shape[1] if self.ndim > 1 else 1


def _mat(a):
    return np.asarray(a, dtype=float).view(_Mat)


class CollocationRoots:
    """Node sets LG / LGR / LGL / CGL on [_TAU_MIN, _TAU_MAX] (mpopt.py:4134-4276) via libmpx."""

    _TAU_MIN = -1
    _TAU_MAX = 1

    def __init__(self, scheme="LGR"):
        self.scheme = scheme
        self._taus_fn = self.get_collocation_points(scheme)

    @classmethod
    def get_collocation_points(cls, scheme):
        return cls._make(_lib.SCHEMES.get(scheme, _lib.SCHEME_EQUI), cls._TAU_MIN, cls._TAU_MAX)

    @staticmethod
    def _make(scheme_id, tau_min, tau_max):
        def taus(deg):
            L = _lib.lib()
            n = L.mpx_colloc_n_nodes(scheme_id, int(deg))
            if n < 0:
                raise ValueError(f"degree {deg} is invalid for this scheme")
            out = np.empty(n)
            _lib.check(L.mpx_colloc_roots(scheme_id, int(deg), float(tau_min), float(tau_max), _lib.dptr(out)))
            return out

        return taus

    @staticmethod
    def roots_legendre_gauss(tau_min=-1, tau_max=1):
        return CollocationRoots._make(_lib.SCHEMES["LG"], tau_min, tau_max)

    @staticmethod
    def roots_legendre_gauss_radau(tau_min=-1, tau_max=1):
        return CollocationRoots._make(_lib.SCHEMES["LGR"], tau_min, tau_max)

    @staticmethod
    def roots_legendre_gauss_lobatto(tau_min=-1, tau_max=1):
        return CollocationRoots._make(_lib.SCHEMES["LGL"], tau_min, tau_max)

    @staticmethod
    def roots_chebyshev_gauss_lobatto(tau_min=-1, tau_max=1):
        return CollocationRoots._make(_lib.SCHEMES["CGL"], tau_min, tau_max)


class _Basis:
    """Callable Lagrange basis polynomial l_j on a node set (what ``polys[key][j]`` is used for)."""

    def __init__(self, nodes, j):
        self.nodes, self.j = np.ascontiguousarray(nodes, dtype=float), j

    def __call__(self, tau):
        taus = np.ascontiguousarray(np.atleast_1d(tau), dtype=float)
        C = np.empty((len(taus), len(self.nodes)))
        _lib.check(_lib.lib().mpx_colloc_interp_matrix(_lib.dptr(self.nodes), len(self.nodes), _lib.dptr(taus), len(taus), _lib.dptr(C)))
        return C[:, self.j] if np.ndim(tau) else float(C[0, self.j])


class Collocation:
    """Differentiation / quadrature / interpolation tables (mpopt.py:3706-4131).

    ``D_MATRIX_METHOD`` is accepted for compatibility; both values select the same native
    implementation (barycentric D, exact interpolatory quadrature), which agrees with the
    reference's "numerical" back-end to rounding and is more accurate at high degree."""

    D_MATRIX_METHOD = "symbolic"

    def __init__(self, poly_orders=[], scheme="LGR", polynomial_type="lagrange"):
        self.poly_orders = poly_orders
        colloc_roots = CollocationRoots(scheme)
        self._taus_fn = colloc_roots._taus_fn
        self.tau0, self.tau1 = colloc_roots._TAU_MIN, colloc_roots._TAU_MAX
        self.poly_fn = self.get_polynomial_function(polynomial_type)
        self.roots, self.polys = {}, {}
        self.unique_polys = set(self.poly_orders)
        self.init_polynomials(self.unique_polys)
        self.diff_matrix_fn = self.get_diff_matrix_fn(polynomial_type)
        self.quad_matrix_fn = self.get_quadrature_weights_fn(polynomial_type)

    @classmethod
    def get_diff_matrix_fn(cls, polynomial_type="lagrange"):
        return cls.get_diff_matrix

    @classmethod
    def get_quadrature_weights_fn(cls, polynomial_type="lagrange"):
        return cls.get_quadrature_weights

    @classmethod
    def get_polynomial_function(cls, polynomial_type="lagrange"):
        return cls.get_lagrange_polynomials if polynomial_type == "lagrange" else 0

    @classmethod
    def get_lagrange_polynomials(cls, roots):
        return [_Basis(roots, j) for j in range(len(roots))]

    def init_polynomials(self, poly_orders):
        for degree in poly_orders:
            self.roots[degree] = self._taus_fn(degree)
            self.polys[degree] = self.poly_fn(self.roots[degree])

    def init_polynomials_with_customized_roots(self, roots_dict=None):
        for key in roots_dict:
            self.roots[key] = np.asarray(roots_dict[key], dtype=float)
            self.polys[key] = self.poly_fn(self.roots[key])

    def _nodes(self, key):
        if key not in self.roots or key not in self.polys:
            self.init_polynomials([key])
        return np.ascontiguousarray(self.roots[key], dtype=float)

    def get_diff_matrix(self, key, taus=None, order=1):
        x = self._nodes(key)
        n = len(x)
        if taus is None:
            D = np.empty((n, n))
            _lib.check(_lib.lib().mpx_colloc_diff_matrix(_lib.dptr(x), n, None, 0, int(order), _lib.dptr(D)))
        else:
            t = np.ascontiguousarray(taus, dtype=float)
            D = np.empty((len(t), n))
            if len(t):
                _lib.check(_lib.lib().mpx_colloc_diff_matrix(_lib.dptr(x), n, _lib.dptr(t), len(t), int(order), _lib.dptr(D)))
        return _mat(D)

    def get_quadrature_weights(self, key, tau0=None, tau1=None):
        x = self._nodes(key)
        a = self.tau0 if tau0 is None else tau0
        b = self.tau1 if tau1 is None else tau1
        w = np.empty(len(x))
        _lib.check(_lib.lib().mpx_colloc_quad_weights(_lib.dptr(x), len(x), float(a), float(b), _lib.dptr(w)))
        return _mat(w.reshape(-1, 1))

    def get_interpolation_matrix(self, taus, degree):
        x = self._nodes(degree)
        t = np.ascontiguousarray(taus, dtype=float)
        C = np.empty((len(t), len(x)))
        if len(t):
            _lib.check(_lib.lib().mpx_colloc_interp_matrix(_lib.dptr(x), len(x), _lib.dptr(t), len(t), _lib.dptr(C)))
        return _mat(C)

    def get_diff_matrices(self, poly_orders=None, order=1):
        keys = self.unique_polys if poly_orders is None else set(poly_orders)
        return {d: self.diff_matrix_fn(self, d, order=order) for d in keys}

    def get_interpolation_Dmatrices_at(self, taus, keys=None, order=1):
        keys = self.poly_orders if keys is None else keys
        return {i: self.diff_matrix_fn(self, key, taus=taus[i], order=order) for i, key in enumerate(keys)}

    def get_quad_weight_matrices(self, keys=None, tau0=None, tau1=None):
        keys = self.unique_polys if keys is None else set(keys)
        a = self.tau0 if tau0 is None else tau0
        b = self.tau1 if tau1 is None else tau1
        return {k: self.quad_matrix_fn(self, k, tau0=a, tau1=b) for k in keys}

    def get_interpolation_matrices(self, taus, poly_orders=None):
        poly_orders = self.poly_orders if poly_orders is None else poly_orders
        return {i: self.get_interpolation_matrix(taus[i], d) for i, d in enumerate(poly_orders)}

    # composites: dense, as the reference returns them (mpopt.py:4015-4131).  The GPU path never
    # forms them; they exist for callers and parity tests.
    def get_composite_differentiation_matrix(self, poly_orders=None, order=1):
        D = self.get_diff_matrices(poly_orders, order=order)
        poly_orders = self.poly_orders if poly_orders is None else poly_orders
        n = sum(poly_orders) + 1
        out = np.zeros((n, n))
        start = 0
        for i, p in enumerate(poly_orders):
            if i == 0:
                out[0:p + 1, 0:p + 1] = D[p]
            else:
                out[start + 1:start + 1 + p, start:start + p + 1] = np.asarray(D[p])[1:, :]
            start += p
        return _mat(out)

    def get_composite_quadrature_weights(self, poly_orders=None, tau0=None, tau1=None):
        poly_orders = self.poly_orders if poly_orders is None else poly_orders
        W = self.get_quad_weight_matrices(poly_orders, tau0=tau0, tau1=tau1)
        parts = [np.asarray(W[poly_orders[0]]).ravel()[:1]] + [np.asarray(W[p]).ravel()[1:] for p in poly_orders]
        return _mat(np.concatenate(parts).reshape(1, -1))

    def _composite(self, blocks, taus, poly_orders):
        n_nodes = sum(poly_orders) + 1
        n_taus = [len(t) for t in taus]
        out = np.zeros((sum(n_taus), n_nodes))
        r = c = 0
        for i, p in enumerate(poly_orders):
            if n_taus[i]:
                out[r:r + n_taus[i], c:c + p + 1] = blocks[i]
            r += n_taus[i]
            c += p
        return out

    def get_composite_interpolation_matrix(self, taus, poly_orders=None):
        C = self.get_interpolation_matrices(taus, poly_orders)
        poly_orders = self.poly_orders if poly_orders is None else poly_orders
        return self._composite(C, taus, poly_orders)

    def get_composite_interpolation_Dmatrix_at(self, taus, poly_orders=None, order=1):
        D = self.get_interpolation_Dmatrices_at(taus, keys=poly_orders, order=order)
        poly_orders = self.poly_orders if poly_orders is None else poly_orders
        return self._composite(D, taus, poly_orders)


# ---------------------------------------------------------------------------------------------
class _NlpSymbol:
    """Placeholder for the reference's SX entries of the nlp dict (f, x, g, p)."""

    def __init__(self, name, size, oracle, shape=None):
        self.name, self.size, self.oracle = name, size, oracle
        self._shape = (size, 1) if shape is None else shape

    def size1(self):
        return self.size

    @property
    def shape(self):
        return self._shape

    def __repr__(self):
        return f"<{self.name}[{self.size}] of {self.oracle.__class__.__name__}>"


class mpopt:
    """Transcription of a multi-phase OCP to an NLP on a pseudo-spectral grid (mpopt.py:31-855)."""

    _GRID_TYPE = "fixed"
    _MAX_GRID_POINTS = 15
    _MUTE_ = False
    _DEVICE = 0  # HIP device ordinal of the libmpx context (one process per GPU: set to LOCAL_RANK)

    def __init__(self, problem, n_segments=1, poly_orders=[9], scheme="LGR", **kwargs):
        self.n_segments = n_segments
        self.device = kwargs.get("device", self._DEVICE)
        self.poly_orders = [poly_orders] * n_segments if isinstance(poly_orders, (int, np.integer)) else list(poly_orders)
        self._ocp = copy.deepcopy(problem)
        self.colloc_scheme = scheme
        self.reset_mpopt()

    def reset_mpopt(self):
        assert len(self.poly_orders) == self.n_segments
        self._Npoints = sum(self.poly_orders) + 1
        self._collocation_approximation_computed = False
        self._variables_created = False
        self._nlpsolver_initialized = False
        self.grid_type = [self._GRID_TYPE] * self._ocp.n_phases
        self.max_grid_points = [self._MAX_GRID_POINTS] * self._ocp.n_phases
        self.oracle = None
        self.__dict__.pop("_resid_plans", None)  # residual plans belong to the previous context

    def compute_numerical_approximation(self, scheme=None):
        scheme = self.colloc_scheme if scheme is None else scheme
        self.collocation = Collocation(self.poly_orders, scheme)
        self._taus = self.collocation.roots
        self.tau0, self.tau1 = self.collocation.tau0, self.collocation.tau1
        self._collocation_approximation_computed = True

    @property
    def _compD(self):
        return self.collocation.get_composite_differentiation_matrix()

    @property
    def _compW(self):
        return self.collocation.get_composite_quadrature_weights()

    def create_variables(self):
        """The reference creates CasADi symbols here (mpopt.py:105-152); the handles below only keep the
        attribute names alive.  ``_optimization_vars_per_phase`` reproduces the reference's formula
        (``n_phases * na``, mpopt.py:130-134) although the vector holds ``na`` parameters per phase."""
        o, N = self._ocp, self._Npoints
        self._optimization_vars_per_phase = N * (o.nx + o.nu) + o.n_phases * o.na + 2
        self.X = [_NlpSymbol(f"X{ph}", N * o.nx, None) for ph in range(o.n_phases)]
        self.U = [_NlpSymbol(f"U{ph}", N * o.nu, None) for ph in range(o.n_phases)]
        self.A = _NlpSymbol("A", o.na * o.n_phases, None)
        self.t0, self.tf = _NlpSymbol("t0", o.n_phases, None), _NlpSymbol("tf", o.n_phases, None)
        self.seg_widths = _NlpSymbol("h_seg", self.n_segments * o.n_phases, None)
        self._variables_created = True

    def validate(self):
        """Consistency checks of the reference's ``mpopt.validate`` (mpopt.py:983-987): grid definition."""
        assert len(self.poly_orders) == self.n_segments and self.n_segments > 0
        assert all(int(p) == p and p >= 1 for p in self.poly_orders)
        assert self._Npoints == sum(self.poly_orders) + 1
        self._ocp.validate()

    def discretize_phase(self, phase):
        """(G, Gmin, Gmax, J) of one phase: handles with ``.shape`` for G and J, bound arrays as in the
        reference (mpopt.py:415-462)."""
        if self.oracle is None:
            self.create_nlp()
        lo, hi = self._phase_row_bounds(phase)
        return (_NlpSymbol(f"G{phase}", len(lo), self.oracle), lo, hi, _NlpSymbol(f"J{phase}", 1, self.oracle, shape=(1, 1)))

    # ---- bounds -------------------------------------------------------------------------
    def _midu_rows(self, phase):
        o = self._ocp
        return bool(o.midu[phase]) and bool((np.asarray(o.lbu[phase]) > -np.inf).any() or (np.asarray(o.ubu[phase]) < np.inf).any())

    def get_nlp_variables(self, phase):
        """(Z, Zmin, Zmax) of one phase; layout [vec(X); vec(U); t0; tf; A], state-major."""
        o, N = self._ocp, self._Npoints
        sx, su, sa, st = np.asarray(o.scale_x, float), np.asarray(o.scale_u, float), np.asarray(o.scale_a, float), o.scale_t
        xmin = np.tile(np.asarray(o.lbx[phase], float) * sx, (N, 1))
        xmax = np.tile(np.asarray(o.ubx[phase], float) * sx, (N, 1))
        if phase == 0:  # initial state fixed through equal bounds
            xmin[0] = xmax[0] = np.asarray(o.x00[0], float) * sx
        zmin = np.concatenate([xmin.T.ravel(), np.repeat(np.asarray(o.lbu[phase], float) * su, N),
                               np.asarray(o.lbt0[phase], float).ravel() * st, np.asarray(o.lbtf[phase], float).ravel() * st,
                               np.asarray(o.lba[phase], float) * sa])
        zmax = np.concatenate([xmax.T.ravel(), np.repeat(np.asarray(o.ubu[phase], float) * su, N),
                               np.asarray(o.ubt0[phase], float).ravel() * st, np.asarray(o.ubtf[phase], float).ravel() * st,
                               np.asarray(o.uba[phase], float) * sa])
        n = len(zmin)
        return (_NlpSymbol(f"z{phase}", n, self.oracle), zmin, zmax)

    def _phase_row_bounds(self, phase):
        o, N, S = self._ocp, self._Npoints, self.n_segments
        prog = self.oracle.program.phases[phase]
        lo, hi = [], []

        def block(n, a, b):
            lo.append(np.full(n, a, dtype=float)), hi.append(np.full(n, b, dtype=float))

        block(o.nx * N, o.LB_DYNAMICS, o.UB_DYNAMICS)
        block(prog.nc * N, o.LB_PATH_CONSTRAINTS, o.UB_PATH_CONSTRAINTS)
        if o.diff_u[phase]:
            block(o.nu * N, o.lbdu[phase], o.ubdu[phase])
        if self._midu_rows(phase):
            n_mid = N - 1
            lo.append(np.repeat(np.asarray(o.lbu[phase], float) * np.asarray(o.scale_u, float), n_mid))
            hi.append(np.repeat(np.asarray(o.ubu[phase], float) * np.asarray(o.scale_u, float), n_mid))
        if o.du_continuity[phase] and S > 1:
            block(o.nu * (S - 1), 0, 0)
        block(prog.ntc, o.LB_TERMINAL_CONSTRAINTS, o.UB_TERMINAL_CONSTRAINTS)
        return np.concatenate(lo), np.concatenate(hi)

    # ---- the reference's per-block builders (mpopt.py:144-413): same names, argument lists and bound vectors; the
    # first element of every tuple is a handle with .shape where the reference returns CasADi expressions (the
    # rows themselves are evaluated by the GPU oracle, never formed symbolically)
    def init_segment_width(self):
        self.seg_widths = _NlpSymbol("h_seg", self.n_segments * self._ocp.n_phases, None)

    def _block(self, name, lo, hi):
        lo, hi = np.asarray(lo, float).ravel(), np.asarray(hi, float).ravel()
        return (_NlpSymbol(name, len(lo), self.oracle) if len(lo) else [], lo if len(lo) else [], hi if len(hi) else [])

    def _n_path(self, phase):
        o = self._ocp
        if not o.has_path_constraints(phase):
            return 0
        return len(np.atleast_1d(o.get_path_constraints(phase)(o.x00[phase], o.u00[phase], o.t00[phase], o.a0[phase])))

    def get_discretized_dynamics_constraints_and_cost_matrices(self, phase=0):
        """(f, c, q): handles for the N node values of h*Sx*dyn, the path constraints and h*L (mpopt.py:154-212)."""
        N, o = self._Npoints, self._ocp
        return (_NlpSymbol(f"f{phase}", N * o.nx, self.oracle, shape=(N, o.nx)),
                _NlpSymbol(f"c{phase}", N * self._n_path(phase), self.oracle, shape=(N, self._n_path(phase))),
                _NlpSymbol(f"q{phase}", N, self.oracle, shape=(N, 1)))

    def get_nlp_constraints_for_dynamics(self, f=[], phase=0):  # mpopt.py:214-237
        o, n = self._ocp, self._ocp.nx * self._Npoints
        return self._block(f"F{phase}", np.full(n, o.LB_DYNAMICS), np.full(n, o.UB_DYNAMICS))

    def get_nlp_constraints_for_path_contraints(self, c=[], phase=0):  # mpopt.py:239-262
        o, n = self._ocp, self._n_path(phase) * self._Npoints
        return self._block(f"C{phase}", np.full(n, o.LB_PATH_CONSTRAINTS), np.full(n, o.UB_PATH_CONSTRAINTS))

    def get_nlp_constraints_for_terminal_contraints(self, phase=0):  # mpopt.py:264-300: (TC, TCmin, TCmax, J)
        o = self._ocp
        n = len(np.atleast_1d(o.get_terminal_constraints(phase)(o.xf0[phase], o.tf0[phase], o.x00[phase], o.t00[phase], o.a0[phase]))) \
            if o.has_terminal_constraints(phase) else 0
        return self._block(f"TC{phase}", np.full(n, o.LB_TERMINAL_CONSTRAINTS), np.full(n, o.UB_TERMINAL_CONSTRAINTS)) + \
            (_NlpSymbol(f"mayer{phase}", 1, self.oracle, shape=(1, 1)),)

    def get_nlp_constraints_for_control_input_slope(self, phase=0):  # mpopt.py:302-328
        o = self._ocp
        n = o.nu * self._Npoints if o.diff_u[phase] else 0
        return self._block(f"DU{phase}", np.full(n, float(o.lbdu[phase])), np.full(n, float(o.ubdu[phase])))

    def get_nlp_constrains_for_control_input_at_mid_colloc_points(self, phase=0):  # mpopt.py:330-377
        o, n_mid = self._ocp, self._Npoints - 1
        if not self._midu_rows(phase):
            return ([], [], [])
        su = np.asarray(o.scale_u, float)
        return self._block(f"mU{phase}", np.repeat(np.asarray(o.lbu[phase], float) * su, n_mid), np.repeat(np.asarray(o.ubu[phase], float) * su, n_mid))

    def get_nlp_constrains_for_control_slope_continuity_across_segments(self, phase=0):  # mpopt.py:379-413
        o = self._ocp
        if self.n_segments == 1 or not o.du_continuity[phase]:
            return ([], [], [])
        n = o.nu * (self.n_segments - 1)
        return self._block(f"dU{phase}", np.zeros(n), np.zeros(n))

    def get_event_constraints(self):
        o = self._ocp
        if o.n_phases < 2:
            return ([], [], [])
        n = len(o.phase_links)
        sx = np.asarray(o.scale_x, float)
        emin = [np.concatenate([np.asarray(o.lbe[ph], float) * sx for ph in range(n)]), np.zeros(o.nu * n), np.zeros(n)]
        emax = [np.concatenate([np.asarray(o.ube[ph], float) * sx for ph in range(n)]), np.zeros(o.nu * n), np.zeros(n)]
        return ([_NlpSymbol(f"E{k}", len(emin[k]), self.oracle) for k in range(3)], emin, emax)

    def create_nlp(self):
        """Returns ``(nlp_prob, nlp_bounds)`` with the reference's keys.  ``f,x,g,p`` are handles on
        the GPU oracle object (``nlp_prob["oracle"]``) instead of CasADi expressions."""
        o = self._ocp
        self.compute_numerical_approximation()
        self.create_variables()
        self.oracle = NlpFunctions(o, self.n_segments, self.poly_orders, self.colloc_scheme, tau0=self.tau0,
                                   tau1=self.tau1, midu_rows=[self._midu_rows(ph) for ph in range(o.n_phases)],
                                   device=self.device)
        zmin, zmax, gmin, gmax = [], [], [], []
        for ph in range(o.n_phases):
            _, a, b = self.get_nlp_variables(ph)
            zmin.append(a), zmax.append(b)
            a, b = self._phase_row_bounds(ph)
            gmin.append(a), gmax.append(b)
        if o.n_phases > 1:
            _, emin, emax = self.get_event_constraints()
            gmin.extend(emin), gmax.extend(emax)
        self.Zmin, self.Zmax = np.concatenate(zmin), np.concatenate(zmax)
        self.Gmin, self.Gmax = np.concatenate(gmin), np.concatenate(gmax)
        assert len(self.Zmin) == self.oracle.n_z and len(self.Gmin) == self.oracle.n_g, "layout mismatch with libmpx"
        orc = self.oracle
        nlp_prob = {"f": _NlpSymbol("f", 1, orc), "x": _NlpSymbol("x", orc.n_z, orc), "g": _NlpSymbol("g", orc.n_g, orc),
                    "p": _NlpSymbol("p", orc.n_p, orc), "oracle": orc}
        self.Z, self.G, self.J = nlp_prob["x"], nlp_prob["g"], nlp_prob["f"]  # mpopt.py:595-627 (handles with .shape)
        nlp_bounds = {"lbg": self.Gmin, "ubg": self.Gmax, "lbx": self.Zmin, "ubx": self.Zmax}
        return (nlp_prob, nlp_bounds)

    # ---- initial guess ------------------------------------------------------------------
    def init_solution_per_phase(self, phase):
        o, N = self._ocp, self._Npoints
        sx, su, sa, st = np.asarray(o.scale_x, float), np.asarray(o.scale_u, float), np.asarray(o.scale_a, float), o.scale_t
        x00, xf0 = np.asarray(o.x00[phase], float) * sx, np.asarray(o.xf0[phase], float) * sx
        u00, uf0 = np.asarray(o.u00[phase], float) * su, np.asarray(o.uf0[phase], float) * su
        t00, tf0 = np.asarray(o.t00[phase], float) * st, np.asarray(o.tf0[phase], float) * st
        a0 = np.asarray(o.a0[phase], float) * sa
        ts = np.linspace(t00, tf0, N)  # linear in node index, (N,1)
        X = np.array([x00 + (xf0 - x00) / (tf0 - t00) * (t - t00) for t in ts])
        U = np.array([u00 + (uf0 - u00) / (tf0 - t00) * (t - t00) for t in ts])
        return np.concatenate([X.T.ravel(), _ref_control_order(U), t00, tf0, a0])

    def initialize_solution(self):
        return np.concatenate([self.init_solution_per_phase(ph) for ph in range(self._ocp.n_phases)])

    def get_segment_width_parameters(self, solution):
        return [1.0 / self.n_segments] * (self.n_segments * self._ocp.n_phases)

    # ---- solver -------------------------------------------------------------------------
    def create_solver(self, solver="ipopt", options={}):
        from .solver import NlpSolver

        nlp_problem, self.nlp_bounds = self.create_nlp()
        defaults = {"ipopt.max_iter": 2000, "ipopt.acceptable_tol": 1e-4, "ipopt.print_level": 0, "ipopt.sb": "yes",
                    "print_time": 0} if solver == "ipopt" else {}
        defaults.update(options)
        self.nlp_solver = NlpSolver("solver", solver, nlp_problem, defaults)
        self._nlpsolver_initialized = True

    def solve(self, initial_solution=None, reinitialize_nlp=False, solver="ipopt", nlp_solver_options={},
              mpopt_options={}, **kwargs):
        start = time.monotonic()
        if (not self._nlpsolver_initialized) or reinitialize_nlp:
            self.create_solver(solver=solver, options=nlp_solver_options)
        if "nlp_sw_params" in mpopt_options:
            self._nlp_sw_params = mpopt_options["nlp_sw_params"]
        else:
            self._nlp_sw_params = self.get_segment_width_parameters(initial_solution)
        inputs = self.get_solver_warm_start_input_parameters(initial_solution)
        inputs["p"] = self._nlp_sw_params
        t_ocp = time.monotonic()
        solution = self.nlp_solver(**inputs, **self.nlp_bounds)
        end = time.monotonic()
        if not self._MUTE_:
            print("\n *********** MPOPT Summary ********** \n")
            print(" Optimal cost (J): ", solution["f"], "\n")
            print(f" Solved in {round((end - start) * 1e3, 3)} ms")
            print(f" \t OCP transcription time  : {round((t_ocp - start) * 1e3, 3)} ms")
            print(f" \t NLP solution time       : {round((end - t_ocp) * 1e3, 3)} ms")
        return solution

    def get_solver_warm_start_input_parameters(self, solution=None):
        target = {"x": "x0", "x0": "x0", "lam_x": "lam_x0", "lam_x0": "lam_x0", "lam_g": "lam_g0", "lam_g0": "lam_g0"}
        inputs = {}
        if solution is not None:
            for key in solution:
                if key in target:
                    inputs[target[key]] = solution[key]
        if "x0" not in inputs:
            inputs["x0"] = self.initialize_solution()
        return inputs

    # ---- minimal result extraction (post-processing proper is out of scope) ----------------
    def get_trajectories(self, solution, phase=0):
        """Unscaled (x, u, t, t0, tf, a) of one phase from a solution vector."""
        o, N = self._ocp, self._Npoints
        n_zp = N * (o.nx + o.nu) + 2 + o.na
        z = np.asarray(solution["x"], float).ravel()[phase * n_zp:(phase + 1) * n_zp]
        X = z[:o.nx * N].reshape(o.nx, N).T / np.asarray(o.scale_x, float)
        U = z[o.nx * N:(o.nx + o.nu) * N].reshape(o.nu, N).T / np.asarray(o.scale_u, float)
        t0, tf = z[(o.nx + o.nu) * N] / o.scale_t, z[(o.nx + o.nu) * N + 1] / o.scale_t
        a = z[(o.nx + o.nu) * N + 2:] / np.asarray(o.scale_a, float)
        p = np.asarray(self._nlp_sw_params, float).reshape(o.n_phases, self.n_segments)[phase]
        t = np.empty(N)
        t_seg0, idx = t0, 0
        for s, deg in enumerate(self.poly_orders):
            h = (tf - t0) / (self.tau1 - self.tau0) * p[s]
            taus = self._taus[deg]
            for k in range(0 if s == 0 else 1, deg + 1):
                t[idx] = t_seg0 + h * (taus[k] - self.tau0)
                idx += 1
            t_seg0 += h * (self.tau1 - self.tau0)
        return X, U, t, t0, tf, a

    # ---- post-solve: off-node interpolation and dynamics residuals (mpopt.py:1152-1236, 1360-1573) --
    def get_residual_grid_taus(self, phase=0, grid_type=None):
        """Non-collocation target points per segment: "fixed" (equally spaced over the phase),
        "mid-points" (between consecutive nodes), "spectral" (interior nodes of degree _MAX_GRID_POINTS+2)."""
        grid_type = self.grid_type[phase] if grid_type is None else grid_type
        if grid_type == "fixed":
            n_nodes = max(sum(self.poly_orders) + 2, self._MAX_GRID_POINTS + 2)
            target = np.linspace(self.tau0, self.tau1, n_nodes)
            S = self.n_segments
            taus = self.compute_interpolation_taus_corresponding_to_original_grid(
                target, self._nlp_sw_params[S * phase:S * (phase + 1)], tau0=self.tau0, tau1=self.tau1)
            taus[0] = taus[0][:-1]
            return taus
        if grid_type == "mid-points":
            return [(self.collocation._taus_fn(d)[:-1] + self.collocation._taus_fn(d)[1:]) / 2.0 for d in self.poly_orders]
        if grid_type == "spectral":
            return [np.array(self.collocation._taus_fn(self._MAX_GRID_POINTS + 2)[1:-1]) for _ in self.poly_orders]
        return None

    @staticmethod
    def compute_interpolation_taus_corresponding_to_original_grid(nodes_req, seg_widths, tau0=0, tau1=1):
        cum = np.append(0, np.cumsum(seg_widths))
        assert abs(cum[-1] - 1) < 1e-6
        scaled = (np.asarray(nodes_req, float) - tau0) / (tau1 - tau0)
        out = []
        for i, w in enumerate(seg_widths):
            t = scaled[scaled > cum[i]]  # the start node is excluded, the end node included
            t = t[t <= cum[i + 1]]
            out.append(tau0 + (tau1 - tau0) * ((t - cum[i]) / w))
        return out

    @staticmethod
    def get_interpolated_time_grid(t_orig, taus, poly_orders, tau0, tau1):
        t_orig = np.asarray(t_orig, float).ravel()
        t_seg = [t_orig[0]] + [t_orig[sum(poly_orders[:i + 1])] for i in range(len(poly_orders))]
        return np.concatenate([t_seg[i] + (t_seg[i + 1] - t_seg[i]) * ((np.asarray(taus[i], float) - tau0) / (tau1 - tau0))
                               for i in range(len(t_seg) - 1)]).reshape(-1, 1)

    def _residual_plan(self, phase, target_nodes, deriv_order=1):
        if self.oracle is None:
            self.create_nlp()
        key = (phase, deriv_order, tuple(np.asarray(t, float).tobytes() for t in target_nodes))
        cache = self.__dict__.setdefault("_resid_plans", {})
        if key not in cache:
            cache.clear()
            cache[key] = self.oracle.residual_plan(phase, target_nodes, deriv_order)
        return cache[key]

    # ---- post-solve: second derivatives of the interpolating polynomials (mpopt.py:1238-1358) -----------
    def get_state_second_derivative_single_phase(self, solution, phase=0, nodes=None, grid_type=None, residual_type=None):
        """(ti_phase, ddx_phase, ddu_phase): per segment, arrays of shape (n_taus, nx, 1) / (n_taus, nu, 1) with
        D2_at . X and D2_at . U (scaled variables), ``None`` for segments without target points.  The products
        are evaluated by the GPU kernel mpx_resid_* with second-derivative rows."""
        target = self.get_residual_grid_taus(phase=phase, grid_type=self.grid_type[phase]) if nodes is None else nodes
        plan = self._residual_plan(phase, target, deriv_order=2)
        r = plan.eval(np.asarray(solution["x"], float).ravel(), np.asarray(self._nlp_sw_params, float), what=("ti", "dxi", "dui"))
        o, S = self._ocp, self.n_segments
        ddx = r["dxi"]
        ddu = r["dui"] if o.nu else np.zeros((plan.n_pts, 0))
        ti_phase, ddx_phase, ddu_phase = [None] * S, [None] * S, [None] * S
        for s in range(S):
            a, b = plan.seg_ptr[s], plan.seg_ptr[s + 1]
            if a == b:
                continue
            ddx_phase[s], ddu_phase[s] = ddx[a:b, :, None].copy(), ddu[a:b, :, None].copy()
            if residual_type == "relative":
                ddx_phase[s] = ddx_phase[s] / ddx_phase[s].max()
                ddu_phase[s] = ddu_phase[s] / ddu_phase[s].max()
            ti_phase[s] = r["ti"][a:b].copy()
        return ti_phase, ddx_phase, ddu_phase

    def get_state_second_derivative(self, solution, grid_type="spectral", nodes=None, plot=False, fig=None, axs=None):
        P = self._ocp.n_phases
        ti, DDx, DDu = [None] * P, [None] * P, [None] * P
        for phase in range(P):
            target = self.get_residual_grid_taus(phase, grid_type=grid_type) if nodes is None else nodes[phase]
            ti[phase], DDx[phase], DDu[phase] = self.get_state_second_derivative_single_phase(solution, phase, nodes=target)
        return ti, DDx, DDu

    def interpolate_single_phase(self, solution, phase=0, target_nodes=None, grid_type=None, options={}):
        """(Xi, Ui, ti, a, DXi, DUi, target_nodes, t0, tf) like the reference (mpopt.py:1489-1543); the
        products C.X, D_at.X are evaluated by the GPU kernel mpx_resid_*."""
        if target_nodes is None:
            target_nodes = self.get_residual_grid_taus(phase=phase, grid_type=grid_type)
        plan = self._residual_plan(phase, target_nodes)
        r = plan.eval(np.asarray(solution["x"], float).ravel(), np.asarray(self._nlp_sw_params, float))
        o, N = self._ocp, self._Npoints
        n_zp = N * (o.nx + o.nu) + 2 + o.na
        zp = np.asarray(solution["x"], float).ravel()[phase * n_zp:(phase + 1) * n_zp]
        a = zp[(o.nx + o.nu) * N + 2:]
        t0 = np.array([zp[(o.nx + o.nu) * N] / o.scale_t])
        tf = np.array([zp[(o.nx + o.nu) * N + 1] / o.scale_t])
        nu_empty = np.zeros((plan.n_pts, 0))
        return (_mat(r["xi"]), _mat(r.get("ui", nu_empty)), _mat(r["ti"].reshape(-1, 1)), _mat(a.reshape(-1, 1)),
                _mat(r["dxi"]), _mat(r.get("dui", nu_empty)), target_nodes, t0, tf)

    def get_dynamics_residuals_single_phase(self, solution, phase=0, target_nodes=None):
        """(ti_phase, residual_phase, dyn_phase): per-segment lists, ``None``/[] for empty segments
        (mpopt.py:1428-1487)."""
        if target_nodes is None:
            target_nodes = self.get_residual_grid_taus(phase=phase)
        plan = self._residual_plan(phase, target_nodes)
        r = plan.eval(np.asarray(solution["x"], float).ravel(), np.asarray(self._nlp_sw_params, float), what=("ti", "dyn", "resid"))
        ti = [t if t is not None else [] for t in plan.split(r["ti"])]
        return ti, plan.split(r["resid"]), plan.split(r["dyn"])

    def get_dynamics_residuals(self, solution, nodes=None, grid_type=None, residual_type=None, plot=False, fig=None, axs=None):
        ti, residuals = [None] * self._ocp.n_phases, [None] * self._ocp.n_phases
        for phase in range(self._ocp.n_phases):
            target = nodes[phase] if nodes is not None else self.get_residual_grid_taus(phase, grid_type=grid_type)
            ti[phase], residuals[phase], dyn = self.get_dynamics_residuals_single_phase(solution, phase, target_nodes=target)
            if residual_type == "relative":  # scaled by the largest |h*Sx*dyn| of the phase (mpopt.py:1403-1418)
                mx = np.zeros(self._ocp.nx)
                for seg in dyn:
                    if seg is not None:
                        mx = np.maximum(mx, np.abs(seg).max(axis=0))
                residuals[phase] = [None if r is None else r / mx for r in residuals[phase]]
        return ti, residuals

    # ---- post-solve: states re-integrated from the dynamics (mpopt.py:989-1150) -------------------
    def compute_states_from_solution_dynamics(self, solution, phase=0, nodes=None):
        """(xint_phase, u_phase, ti_phase, residual_phase) per segment: the states obtained by
        integrating h_s*Sx*dyn over [tau0, tau_i] with the interpolatory quadrature on the target points
        themselves, and the difference to the interpolated states.  Interpolation and dynamics come from
        the GPU kernel (xi, ui, ti, dyn); the remaining per-segment n_s x n_s quadrature product is host
        arithmetic."""
        target = self.get_residual_grid_taus(phase=phase, grid_type=self.grid_type[phase]) if nodes is None else nodes
        plan = self._residual_plan(phase, target)
        z = np.asarray(solution["x"], float).ravel()
        r = plan.eval(z, np.asarray(self._nlp_sw_params, float), what=("ti", "xi", "ui", "dyn"))
        o, N = self._ocp, self._Npoints
        X = z[phase * (N * (o.nx + o.nu) + 2 + o.na):][:o.nx * N].reshape(o.nx, N).T  # scaled states at the nodes
        starts = np.concatenate([[0], np.cumsum(self.poly_orders)])
        S = self.n_segments
        xint, res, uph, tph = [None] * S, [None] * S, [None] * S, [None] * S
        L = _lib.lib()
        for s in range(S):
            a, b = plan.seg_ptr[s], plan.seg_ptr[s + 1]
            if a == b:
                continue
            taus = np.ascontiguousarray(target[s], dtype=float)
            n = len(taus)
            Q = np.empty((n, n))  # Q[i, j] = int_{tau0}^{tau_i} l_j, Lagrange basis on the target points
            w = np.empty(n)
            for i in range(n):
                _lib.check(L.mpx_colloc_quad_weights(_lib.dptr(taus), n, float(self.tau0), float(taus[i]), _lib.dptr(w)))
                Q[i] = w
            xint[s] = X[starts[s]][None, :] + Q @ r["dyn"][a:b]
            res[s] = list(r["xi"][a:b] - xint[s])
            uph[s] = (r["ui"][a:b] if o.nu else np.zeros((n, 0))).reshape(n, 1, o.nu)
            tph[s] = r["ti"][a:b]
        return xint, uph, tph, res

    def get_states_residuals(self, solution, phases=None, nodes=None, residual_type=None, plot=False, fig=None, axs=None):
        P = self._ocp.n_phases
        x_int, u_int, residuals, ti = [None] * P, [None] * P, [None] * P, [None] * P
        for phase in (range(P) if phases is None else phases):
            target = self.get_residual_grid_taus(phase, grid_type=self.grid_type[phase]) if nodes is None else nodes[phase]
            x_int[phase], u_int[phase], ti[phase], residuals[phase] = self.compute_states_from_solution_dynamics(solution, phase, nodes=target)
            if residual_type == "relative":
                mx = np.zeros(self._ocp.nx)
                for seg in x_int[phase]:
                    if seg is not None:
                        mx = np.maximum(mx, np.abs(np.asarray(seg)).max(axis=0))
                residuals[phase] = [None if r_ is None else np.asarray(r_) / mx for r_ in residuals[phase]]
        return x_int, u_int, ti, residuals

    def init_trajectories(self, phase=0):
        """Callable ``(z, seg_widths) -> (x, u, t, t0, tf, a)`` of one phase: scaled x, u, a, unscaled t, t0, tf
        (the CasADi Function of mpopt.py:857-882)."""
        o = self._ocp
        sx, su, sa = np.asarray(o.scale_x, float), np.asarray(o.scale_u, float), np.asarray(o.scale_a, float)

        def trajectories(z, seg_widths):
            keep = getattr(self, "_nlp_sw_params", None)
            self._nlp_sw_params = np.asarray(seg_widths, float).ravel()
            try:
                X, U, t, t0, tf, a = self.get_trajectories({"x": z}, phase)
            finally:
                if keep is not None:
                    self._nlp_sw_params = keep
            return (_mat(X * sx), _mat(U * su), _mat(t.reshape(-1, 1)), t0, tf, _mat((a * sa).reshape(-1, 1)))

        return trajectories

    def process_results(self, solution, plot=False, scaling=False, residual_x=False, residual_dx=True):
        """``post_process`` object with the solution, the per-phase trajectory extractors and, like the reference
        (mpopt.py:884-981), the residuals it was asked for in ``options["residuals"]``.  ``plot`` is ignored."""
        o = self._ocp
        trajectories = [self.init_trajectories(phase) for phase in range(o.n_phases)]
        resid_value = {}
        if residual_x:
            x_int, u_int, ti, res_x = self.get_states_residuals(solution)
            resid_value["t_x"] = [ti, res_x]
        if residual_dx:
            tdx, res_dx = self.get_dynamics_residuals(solution)
            resid_value["t_dx"] = [tdx, res_dx]
        else:  # the reference drops the state residuals too in this case (mpopt.py:921-925)
            resid_value = None

        def interpolate(phase, taus):
            Xi, Ui, *_ = self.interpolate_single_phase(solution, phase=phase, target_nodes=taus)
            return Xi.full(), Ui.full()

        options = {"nx": o.nx, "nu": o.nu, "na": o.na, "nPh": o.n_phases, "ns": self.n_segments, "poly_orders": self.poly_orders,
                   "N": self._Npoints, "phases_to_plot": o.phases_to_plot, "scale_x": o.scale_x, "scale_u": o.scale_u,
                   "scale_a": o.scale_a, "scale_t": o.scale_t, "scaling": scaling, "colloc_scheme": self.colloc_scheme,
                   "tau0": self.tau0, "tau1": self.tau1, "interpolation_depth": 3, "seg_widths": self._nlp_sw_params,
                   "residuals": resid_value, "interpolate": interpolate}
        post = post_process(solution, trajectories, options)
        if plot:
            for phases in o.phases_to_plot:
                post.plot_phases(phases, residuals=residual_x or residual_dx)
        return post


class mpopt_h_adaptive(mpopt):
    """Iterative refinement of the segment widths with a fixed number of segments
    (mpopt.py:2273-2874).  The NLP structure never changes -- only the parameter vector ``p`` -- so one
    libmpx context (tables, patterns, code object) serves the whole loop; the dynamics residuals that
    drive the refinement come from the GPU kernel ``mpx_resid_*``.  The refinement rules themselves are
    O(n_segments) host arithmetic."""

    _SEG_WIDTH_MIN = 1e-5
    _SEG_WIDTH_MAX = 1
    _TOL_SEG_WIDTH_CHANGE = 0.05
    _TOL_RESIDUAL = 1e-2
    _DEFAULT_METHOD = "residual"
    _DEFAULT_SUB_METHOD = "equal_area"
    _THRESHOLD_SLOPE = 1e-1

    def __init__(self, problem, n_segments=1, poly_orders=[9], scheme="LGR", **kwargs):
        super().__init__(problem=problem, n_segments=n_segments, poly_orders=poly_orders, scheme=scheme, **kwargs)
        P = self._ocp.n_phases
        self.lbh, self.ubh = [self._SEG_WIDTH_MIN] * P, [self._SEG_WIDTH_MAX] * P
        self.tol_residual = [self._TOL_RESIDUAL] * P
        self.fig = self.axs = None
        self.plot_residual_evolution = False

    def _say(self, *a):
        if not self._MUTE_:
            print(*a)

    def solve(self, initial_solution=None, reinitialize_nlp=False, solver="ipopt", nlp_solver_options={}, mpopt_options={},
              max_iter=10, **kwargs):
        start = time.monotonic()
        if (not self._nlpsolver_initialized) or reinitialize_nlp:
            nlp_solver_options.setdefault("ipopt.print_level", 0)
            self.create_solver(solver=solver, options=nlp_solver_options)
        if mpopt_options == {}:
            mpopt_options = {"method": self._DEFAULT_METHOD, "sub_method": self._DEFAULT_SUB_METHOD}
        self.iter_count, self.iter_info = 0, dict()
        sw_old = []
        new_sw, max_error = self.get_segment_width_parameters(initial_solution, options=mpopt_options)
        solution = initial_solution
        if max_error is not None and max_error < min(self.tol_residual):
            self.iter_info[self.iter_count] = max_error
            self._say(f"Solved to acceptable tolerance {min(self.tol_residual)}", max_error)
        else:
            for it in range(max_iter):
                self._nlp_sw_params = new_sw
                if self.iter_count > 0:
                    self.iter_info[self.iter_count] = max_error
                    if self.iter_count > 4:  # stagnation of the max error over the last four iterations
                        mean_error = np.mean(list(self.iter_info.values())[-4:])
                        if abs(max_error - mean_error) < 0.05 * abs(max_error):
                            self._say("Stopping the iterations: Change in max error is < 5%")
                            self._nlp_sw_params = sw_old
                            break
                if it > 0:
                    cur, old = np.asarray(self._nlp_sw_params, float), np.asarray(sw_old, float)
                    if (np.abs(cur - old) / cur <= self._TOL_SEG_WIDTH_CHANGE).all():
                        self._say("Stopping the iterations: Change in width less than 5%", max_error)
                        self._nlp_sw_params = sw_old
                        break
                inputs = self.get_solver_warm_start_input_parameters(initial_solution)
                inputs["p"] = self._nlp_sw_params
                solution = self.nlp_solver(**inputs, **self.nlp_bounds)
                initial_solution = solution
                sw_old = copy.deepcopy(self._nlp_sw_params)
                new_sw, max_error = self.get_segment_width_parameters(initial_solution, options=mpopt_options)
                self.iter_count += 1
                if max_error is not None and max_error < min(self.tol_residual):
                    self.iter_info[self.iter_count] = max_error
                    self._say(f"Solved to acceptable tolerance {min(self.tol_residual)}", max_error)
                    break
                if it == max_iter - 1:
                    self.iter_info[self.iter_count] = max_error
                    self._say("Stopping the iterations: Iteration limit exceeded")
        self._say(f"H-Adaptive Iter., max_residual : {self.iter_count}, {max_error}")
        self._say(" Optimal cost (J): ", solution["f"], f"\n Solved in {round((time.monotonic() - start) * 1e3, 3)} ms\n")
        return solution

    def get_segment_width_parameters(self, solution, options={"method": "residual", "sub_method": "merge_split"}):
        S, P = self.n_segments, self._ocp.n_phases
        default = [1 / S] * (S * P)
        if S == 1 or solution is None:
            return default, None
        if not hasattr(self, "_nlp_sw_params"):
            self._nlp_sw_params = default
        method = options.get("method")
        if method == "control_slope":
            return self.compute_seg_width_based_on_input_slope(solution)
        if method == "residual":
            return self.compute_seg_width_based_on_residuals(solution, method=options.get("sub_method", "equal_area"))
        return default, None

    def _phase_max_residual(self, residuals_phase):
        return max([np.abs(np.asarray(r)).max() if r is not None else 0 for r in residuals_phase])

    def compute_seg_width_based_on_residuals(self, solution, method="merge_split"):
        S = self.n_segments
        ti, residuals = self.get_dynamics_residuals(solution)
        widths, max_error = [], 0
        for ph in range(self._ocp.n_phases):
            mx = self._phase_max_residual(residuals[ph])
            max_error = max(max_error, mx)
            old = self._nlp_sw_params[S * ph:S * (ph + 1)]
            if mx < self.tol_residual[ph]:
                widths.append(old)
                continue
            new = self.refine_segment_widths_based_on_residuals(residuals[ph], old, ERR_TOL=self.tol_residual[ph], method=method)
            if method == "equal_area":  # damped update
                new = 0.4 * np.array(new) + 0.6 * np.array(old)
            widths.append(new)
        return np.concatenate(widths), max_error

    def refine_segment_widths_based_on_residuals(self, residuals, segment_widths, ERR_TOL=1e-3, method="merge_split"):
        if method == "merge_split":
            mx = [np.abs(np.asarray(r)).max() if r is not None else 0 for r in residuals]
            return self.merge_split_segments_based_on_residuals(mx, segment_widths, ERR_TOL=ERR_TOL)
        if method == "equal_area":
            r1d = np.concatenate([np.linalg.norm(np.asarray(r), 2, axis=1) if r is not None else [0] for r in residuals])
            return self.get_roots_wrt_equal_area(r1d, self.n_segments)
        return segment_widths

    @staticmethod
    def get_roots_wrt_equal_area(residuals, n_segments):
        """Segment boundaries that split the area under the residual curve (trapezoids over an equally
        spaced abscissa) into n_segments equal parts."""
        r = np.asarray(residuals, float)
        cum = np.append(0, np.cumsum(0.5 * (r[:-1] + r[1:])))
        cum = cum / cum[-1]
        target = (np.arange(n_segments) + 1) / n_segments
        j = np.maximum(np.searchsorted(cum, target, side="left"), 1)  # first index with cum >= target (cum[0] = 0 < target)
        pos = (j - 1 + (target - cum[j - 1]) / (cum[j] - cum[j - 1])) / (len(r) - 1)
        return list(np.diff(np.append(0, pos)))

    @staticmethod
    def merge_split_segments_based_on_residuals(max_residuals, segment_widths, ERR_TOL=1e-3):
        """Consecutive segments below the tolerance are merged into one; the segments freed that way are
        spent on splitting every run above the tolerance evenly."""
        ok = np.asarray(max_residuals) < ERR_TOL
        ns = len(segment_widths)
        starts = np.flatnonzero(np.append(True, ok[1:] != ok[:-1]))  # runs of equal flag
        ends = np.append(starts[1:], ns)
        bad_runs = [k for k, s0 in enumerate(starts) if not ok[s0]]
        if len(starts) == ns or not bad_runs:
            return segment_widths
        w = np.asarray(segment_widths, float)
        run_w = [w[a:b].sum() for a, b in zip(starts, ends)]
        n_free = ns - len(starts)
        per_bad = [1 + n_free // len(bad_runs)] * len(bad_runs)
        per_bad[-1] += n_free % len(bad_runs)
        out, b = [], 0
        for k, s0 in enumerate(starts):
            if ok[s0]:
                out.append(run_w[k])
            else:
                out += [run_w[k] / per_bad[b]] * per_bad[b]
                b += 1
        return np.array(out)

    def _node_plan_nodes(self):
        """Target points = the collocation nodes themselves (node 0 once), so that the kernel's dui
        equals compD.U of the reference (mpopt.py:2770)."""
        return [self.collocation._taus_fn(d) if s == 0 else self.collocation._taus_fn(d)[1:] for s, d in enumerate(self.poly_orders)]

    def compute_seg_width_based_on_input_slope(self, solution):
        S = self.n_segments
        ti, residuals = self.get_dynamics_residuals(solution)
        widths, max_error = [], 0.0
        for ph in range(self._ocp.n_phases):
            mx = self._phase_max_residual(residuals[ph])
            max_error = max(max_error, mx)
            old = self._nlp_sw_params[S * ph:S * (ph + 1)]
            if mx < self.tol_residual[ph]:
                widths.append(old)
                continue
            plan = self._residual_plan(ph, self._node_plan_nodes())
            r = plan.eval(np.asarray(solution["x"], float).ravel(), np.asarray(self._nlp_sw_params, float), what=("ti", "dui"))
            _, _, _, t0, tf, _ = self.get_trajectories(solution, ph)
            times = self.compute_time_at_max_values(None, r["ti"], np.abs(r["dui"]), threshold=self._THRESHOLD_SLOPE)
            if len(times) == 0:
                widths.append(old)
                continue
            new = np.clip(self.compute_segment_widths_at_times(times, S, t0, tf), self.lbh[ph], self.ubh[ph])
            widths.append(new / new.sum())
        return np.concatenate(widths), max_error

    @staticmethod
    def compute_time_at_max_values(t_grid, t_orig, du_orig, threshold=0):
        """Interior node times whose control-slope 2-norm reaches the threshold, ordered by that norm
        (ascending, stable) -- the reference's ordering, kept as is."""
        t, nrm = np.asarray(t_orig, float).ravel()[1:-1], np.linalg.norm(np.asarray(du_orig, float), 2, axis=1)[1:-1]
        keep = nrm >= threshold
        t, nrm = t[keep], nrm[keep]
        return t[np.argsort(nrm, kind="stable")] if len(t) else np.array([])

    @staticmethod
    def compute_segment_widths_at_times(times, n_segments, t0, tf):
        times = np.asarray(times, float)
        n_avail = len(times)
        if n_avail > n_segments - 2:
            cut = np.sort(times[:n_segments])
            w = np.concatenate([[cut[0] - t0], np.diff(cut[:n_segments - 1]), [tf - cut[n_segments - 2]]])
        else:
            cut = np.sort(times)
            sw0, sw_end = cut[0] - t0, tf - cut[-1]
            n_req = n_segments - (n_avail - 1)
            n_start = 1 if n_req == 2 else 1 + int(sw0 / (sw0 + sw_end) * (n_req - 1))
            n_end = n_req - n_start
            w = np.concatenate([[sw0 / n_start] * n_start, np.diff(cut), [sw_end / n_end] * n_end])
        return np.asarray(w, float) / (tf - t0)


class mpopt_adaptive(mpopt):
    """Segment widths as decision variables, solved for together with the trajectories
    (mpopt.py:2877-3375).  Per phase Z = [X; U; t0; tf; A; W] and G = [F; C; DU; TC; SW] with
    SW = [sum(W) - 1; mid-point controls; mid-point states; W_s * mid-point dynamics residuals].
    The NLP oracles run on the GPU through an assembled libmpx context (mpopt_amd/adaptive.py);
    post-processing (trajectories, residuals) reuses the fixed-width machinery with the optimal widths.

    Examples :
        >>> ocp = mp.OCP(n_states=2, n_controls=1, n_phases=1)
        >>> ocp.dynamics[0] = lambda x, u, t: [x[1], u[0] - 1.5]
        >>> ocp.running_costs[0] = lambda x, u, t: u[0]
        >>> ocp.terminal_constraints[0] = lambda xf, tf, x0, t0: [xf[0], xf[1]]
        >>> ocp.x00[0] = [10, -2]; ocp.lbu[0] = 0; ocp.ubu[0] = 3; ocp.lbtf[0] = 3; ocp.ubtf[0] = 5
        >>> opt = mp.mpopt_adaptive(ocp, n_segments=3, poly_orders=[2]*3)
        >>> solution = opt.solve()
    """

    _SEG_WIDTH_MIN = 1e-4
    _SEG_WIDTH_MAX = 1.0
    _TOL_RESIDUAL = 1e-3

    def __init__(self, problem, n_segments=1, poly_orders=[9], scheme="LGR", **kwargs):
        super().__init__(problem, n_segments=n_segments, poly_orders=poly_orders, scheme=scheme, **kwargs)
        self.mid_residuals = True
        n_ph = self._ocp.n_phases
        self.lbh = [self._SEG_WIDTH_MIN] * n_ph
        self.ubh = [self._SEG_WIDTH_MAX] * n_ph
        self.tol_residual = [self._TOL_RESIDUAL] * n_ph
        self._post = None

    def _n_zp(self):
        o = self._ocp
        return self._Npoints * (o.nx + o.nu) + 2 + o.na + self.n_segments

    def get_nlp_variables(self, phase=0):
        """(Z, Zmin, Zmax): the base layout followed by the S segment widths (mpopt.py:2927-2979)."""
        sym, zmin, zmax = super().get_nlp_variables(phase)
        S = self.n_segments
        zmin = np.concatenate([zmin, np.full(S, float(self.lbh[phase]))])
        zmax = np.concatenate([zmax, np.full(S, float(self.ubh[phase]))])
        return (_NlpSymbol(f"z{phase}", len(zmin), self.oracle), zmin, zmax)

    def init_solution_per_phase(self, phase):
        return np.concatenate([super().init_solution_per_phase(phase), np.full(self.n_segments, 1.0 / self.n_segments)])

    def _bounded(self, phase):
        o = self._ocp
        fin = lambda lo, hi: bool((np.asarray(lo, float) > -np.inf).any() or (np.asarray(hi, float) < np.inf).any())
        return fin(o.lbu[phase], o.ubu[phase]), fin(o.lbx[phase], o.ubx[phase])

    def get_nlp_constrains_for_segment_widths(self, phase=0):
        """(SW, SWmin, SWmax) (mpopt.py:3034-3136); SW is a handle, the rows live in the GPU oracle."""
        o, n_mid = self._ocp, self._Npoints - 1
        lo, hi = [np.zeros(1)], [np.zeros(1)]
        u_b, x_b = self._bounded(phase)
        if u_b:
            lo.append(np.repeat(np.asarray(o.lbu[phase], float) * np.asarray(o.scale_u, float), n_mid))
            hi.append(np.repeat(np.asarray(o.ubu[phase], float) * np.asarray(o.scale_u, float), n_mid))
        if x_b:
            lo.append(np.repeat(np.asarray(o.lbx[phase], float) * np.asarray(o.scale_x, float), n_mid))
            hi.append(np.repeat(np.asarray(o.ubx[phase], float) * np.asarray(o.scale_x, float), n_mid))
        if self.mid_residuals:
            lo.append(np.full(o.nx * n_mid, -float(self.tol_residual[phase])))
            hi.append(np.full(o.nx * n_mid, float(self.tol_residual[phase])))
        lo, hi = np.concatenate(lo), np.concatenate(hi)
        return (_NlpSymbol(f"SW{phase}", len(lo), self.oracle), lo, hi)

    def _phase_row_bounds(self, phase):
        """[F; C; DU; TC; SW] (mpopt.py:3166-3172)."""
        o, N = self._ocp, self._Npoints
        lay = self._layout
        nc = (lay.rows[phase]["DU"] - lay.rows[phase]["C"]) // N
        ntc = lay.rows[phase]["sum"] - lay.rows[phase]["TC"]
        lo = [np.full(o.nx * N, float(o.LB_DYNAMICS)), np.full(nc * N, float(o.LB_PATH_CONSTRAINTS))]
        hi = [np.full(o.nx * N, float(o.UB_DYNAMICS)), np.full(nc * N, float(o.UB_PATH_CONSTRAINTS))]
        if o.diff_u[phase]:
            lo.append(np.full(o.nu * N, float(o.lbdu[phase]))), hi.append(np.full(o.nu * N, float(o.ubdu[phase])))
        lo.append(np.full(ntc, float(o.LB_TERMINAL_CONSTRAINTS))), hi.append(np.full(ntc, float(o.UB_TERMINAL_CONSTRAINTS)))
        _, a, b = self.get_nlp_constrains_for_segment_widths(phase)
        return np.concatenate(lo + [a]), np.concatenate(hi + [b])

    def create_nlp(self):
        from .adaptive import build_adaptive_oracle

        o = self._ocp
        self.compute_numerical_approximation()
        self.create_variables()
        self.oracle, self._layout = build_adaptive_oracle(o, self.n_segments, self.poly_orders, self.collocation,
                                                          mid_residuals=self.mid_residuals, device=self.device)
        zmin, zmax, gmin, gmax = [], [], [], []
        for ph in range(o.n_phases):
            _, a, b = self.get_nlp_variables(ph)
            zmin.append(a), zmax.append(b)
            a, b = self._phase_row_bounds(ph)
            gmin.append(a), gmax.append(b)
        if o.n_phases > 1:
            _, emin, emax = self.get_event_constraints()
            gmin.extend(emin), gmax.extend(emax)
        self.Zmin, self.Zmax = np.concatenate(zmin), np.concatenate(zmax)
        self.Gmin, self.Gmax = np.concatenate(gmin), np.concatenate(gmax)
        orc = self.oracle
        assert len(self.Zmin) == orc.n_z and len(self.Gmin) == orc.n_g, "layout mismatch with libmpx"
        nlp_prob = {"f": _NlpSymbol("f", 1, orc), "x": _NlpSymbol("x", orc.n_z, orc), "g": _NlpSymbol("g", orc.n_g, orc),
                    "p": _NlpSymbol("p", self.n_segments * o.n_phases, orc), "oracle": orc}
        self.Z, self.G, self.J = nlp_prob["x"], nlp_prob["g"], nlp_prob["f"]
        return (nlp_prob, {"lbg": self.Gmin, "ubg": self.Gmax, "lbx": self.Zmin, "ubx": self.Zmax})

    def discretize_phase(self, phase):
        if self.oracle is None:
            self.create_nlp()
        lo, hi = self._phase_row_bounds(phase)
        return (_NlpSymbol(f"G{phase}", len(lo), self.oracle), lo, hi, _NlpSymbol(f"J{phase}", 1, self.oracle, shape=(1, 1)))

    def create_solver(self, solver="ipopt", options={}):
        from .solver import NlpSolver

        nlp_problem, self.nlp_bounds = self.create_nlp()
        nlp_problem.pop("p", None)  # the widths are part of x (mpopt.py:3190-3192)
        defaults = {"ipopt.max_iter": 2000, "ipopt.acceptable_tol": 1e-4, "ipopt.print_level": 0, "ipopt.sb": "yes",
                    "print_time": 0} if solver == "ipopt" else {}
        defaults.update(options)
        self.nlp_solver = NlpSolver("solver", solver, nlp_problem, defaults)
        self._nlpsolver_initialized = True

    def segment_widths(self, solution):
        """(n_phases * S,) optimal width fractions, phase-major (what the reference prints, mpopt.py:3243-3245)."""
        z = np.asarray(solution["x"], float).ravel()
        n_zp, S = self._n_zp(), self.n_segments
        return np.concatenate([z[(ph + 1) * n_zp - S:(ph + 1) * n_zp] for ph in range(self._ocp.n_phases)])

    def solve(self, initial_solution=None, reinitialize_nlp=False, solver="ipopt", nlp_solver_options={}, mpopt_options={}, **kwargs):
        if (not self._nlpsolver_initialized) or reinitialize_nlp:
            self.create_solver(solver=solver, options=nlp_solver_options)
        if initial_solution is None and mpopt_options.get("warm_start_fixed_width", True):
            # The stand-ins for IPOPT (solver.py) are not reliable from the linear default guess on this problem
            # class (no proper restoration phase); start from the equal-width solution on the same grid instead (the
            # reference starts IPOPT from the linear guess, mpopt.py:3235-3239 -- {"warm_start_fixed_width": False}).
            fixed = mpopt(self._ocp, self.n_segments, self.poly_orders, self.colloc_scheme, device=self.device)
            zf = np.asarray(fixed.solve(solver=solver, nlp_solver_options=nlp_solver_options)["x"], float).ravel()
            n0, S = self._n_zp() - self.n_segments, self.n_segments
            initial_solution = {"x0": np.concatenate([np.concatenate([zf[ph * n0:(ph + 1) * n0], np.full(S, 1.0 / S)])
                                                      for ph in range(self._ocp.n_phases)])}
        inputs = self.get_solver_warm_start_input_parameters(initial_solution)
        solution = self.nlp_solver(**inputs, **self.nlp_bounds)
        self._nlp_sw_params = self.segment_widths(solution)
        if not self._MUTE_:
            print(f"Optimal segment width fractions: {self._nlp_sw_params}")
        return solution

    # ---- post-processing: fixed-width machinery with the optimal widths as parameters ----------------
    def _as_fixed_width(self, solution):
        """(base mpopt on the same grid, solution without the width variables)."""
        if self._post is None:
            self._post = mpopt(self._ocp, self.n_segments, self.poly_orders, self.colloc_scheme, device=self.device)
            self._post.compute_numerical_approximation()
        z = np.asarray(solution["x"], float).ravel()
        n_zp, S = self._n_zp(), self.n_segments
        self._post._nlp_sw_params = self.segment_widths(solution)
        self._post.grid_type, self._post.max_grid_points = self.grid_type, self.max_grid_points
        return self._post, {"x": np.concatenate([z[ph * n_zp:(ph + 1) * n_zp - S] for ph in range(self._ocp.n_phases)])}

    def get_trajectories(self, solution, phase=0):
        post, sol = self._as_fixed_width(solution)
        return post.get_trajectories(sol, phase)

    def interpolate_single_phase(self, solution, phase=0, target_nodes=None, grid_type=None, options={}):
        post, sol = self._as_fixed_width(solution)
        return post.interpolate_single_phase(sol, phase, target_nodes, grid_type, options)

    def get_dynamics_residuals_single_phase(self, solution, phase=0, target_nodes=None):
        post, sol = self._as_fixed_width(solution)
        return post.get_dynamics_residuals_single_phase(sol, phase, target_nodes)

    def get_dynamics_residuals(self, solution, nodes=None, grid_type=None, residual_type=None, plot=False, fig=None, axs=None):
        post, sol = self._as_fixed_width(solution)
        return post.get_dynamics_residuals(sol, nodes, grid_type, residual_type)

    def compute_states_from_solution_dynamics(self, solution, phase=0, nodes=None):
        post, sol = self._as_fixed_width(solution)
        return post.compute_states_from_solution_dynamics(sol, phase, nodes)

    def get_states_residuals(self, solution, phases=None, nodes=None, residual_type=None, plot=False, fig=None, axs=None):
        post, sol = self._as_fixed_width(solution)
        return post.get_states_residuals(sol, phases, nodes, residual_type)


class mpopt_ph_adaptive(mpopt):
    """Iterative refinement of polynomial degrees and segment widths (mpopt.py:4316-4596; the scheme of Patterson, Hager
    & Rao, doi:10.1016/j.jfranklin.2015.05.028).  Constructor, attributes and ``get_abs_max_residual`` follow the
    reference; ``solve_ph`` implements the loop the reference sketches (there it stops at an undefined name,
    mpopt.py:4443): solve, measure the relative state residuals per segment, raise the degree by 3 where they exceed
    ``max_residual``, re-solve, and split segments whose second derivative grew (non-smooth) instead of raising their
    degree further.  Every re-solve builds a new transcription (new degrees = new kernels, compiled once and cached)."""

    _SEG_WIDTH_MIN = 1e-5
    _SEG_WIDTH_MAX = 1
    _TOL_SEG_WIDTH_CHANGE = 0.05
    _TOL_RESIDUAL = 1e-2

    def __init__(self, problem, n_segments=1, poly_orders=[9], scheme="LGR", grid_type="spectral", max_residual=1e-4,
                 poly_order_min=3, poly_order_max=16, seg_min=1, seg_max=20, n_grid_points=20, non_smooth_threshold=1.05, **kwargs):
        super().__init__(problem=problem, n_segments=n_segments, poly_orders=poly_orders, scheme=scheme, **kwargs)
        self.poly_order_min = min(poly_order_min, min(self.poly_orders))
        self.poly_order_max = max(poly_order_max, max(self.poly_orders))
        self.min_segments, self.max_segments = min(seg_min, n_segments), max(seg_max, n_segments)
        self._MAX_GRID_POINTS, self._TOL_RESIDUAL, self._GRID_TYPE = n_grid_points, max_residual, grid_type
        self.max_residual, self.n_grid_points, self.non_smooth_threshold = max_residual, n_grid_points, non_smooth_threshold
        n_ph = self._ocp.n_phases
        self.lbh, self.ubh = [self._SEG_WIDTH_MIN] * n_ph, [self._SEG_WIDTH_MAX] * n_ph
        self.tol_residual = [self._TOL_RESIDUAL] * n_ph
        self.fig, self.axs, self.plot_residual_evolution = None, None, False
        self.reset_mpopt()

    @staticmethod
    def get_abs_max_residual(residual):
        """Per phase, per segment: [index of the largest |residual| per state, that largest value] (mpopt.py:4400-4420)."""
        out = [None] * len(residual)
        for i_phase, r_phase in enumerate(residual):
            out[i_phase] = [[np.abs(np.array(r_seg)).argmax(axis=0), np.abs(np.array(r_seg)).max(axis=0)] for r_seg in r_phase]
        return out

    def _regrid(self, poly_orders, widths):
        self.poly_orders = [int(p) for p in poly_orders]
        self.n_segments = len(self.poly_orders)
        self._nlp_sw_params = np.asarray(widths, float)
        self.reset_mpopt()

    def _segment_residuals(self, solution):
        _, _, _, res = self.get_states_residuals(solution, residual_type="relative")
        mx = self.get_abs_max_residual([[r for r in ph if r is not None] for ph in res])
        return [np.array([np.max(seg[1]) for seg in ph]) for ph in mx]

    def solve_ph(self, max_iter=1, solve_dict={}):
        if self._ocp.n_phases != 1:
            raise NotImplementedError("solve_ph: the refinement loop is defined for single-phase problems (one grid per phase)")
        clip = lambda orders: [min(max(self.poly_order_min, int(p)), self.poly_order_max) for p in orders]
        widths = np.full(self.n_segments, 1.0 / self.n_segments)
        solution = None
        for it in range(max_iter):
            opts = dict(solve_dict, mpopt_options=dict(solve_dict.get("mpopt_options", {}), nlp_sw_params=widths))
            solution = self.solve(reinitialize_nlp=True, **opts)
            seg_res = self._segment_residuals(solution)[0]
            bad = seg_res > self.max_residual
            if not bad.any():
                return solution
            taus = [self.collocation._taus_fn(clip([p + 3])[0]) for p in self.poly_orders]
            _, ddx, _ = self.get_state_second_derivative(solution, nodes=[taus])
            coarse_orders, coarse_dd = list(self.poly_orders), [np.abs(v).max() if v is not None else 0.0 for v in ddx[0]]
            # finer polynomial where the residual is too large, same segments
            self._regrid(clip([p + 3 * int(b) for p, b in zip(coarse_orders, bad)]), widths)
            opts = dict(solve_dict, mpopt_options=dict(solve_dict.get("mpopt_options", {}), nlp_sw_params=widths))
            solution = self.solve(reinitialize_nlp=True, **opts)
            seg_res = self._segment_residuals(solution)[0]
            bad = seg_res > self.max_residual
            if not bad.any():
                return solution
            _, ddx_new, _ = self.get_state_second_derivative(solution, nodes=[taus])
            orders, new_w = [], []
            for s, (p, w) in enumerate(zip(self.poly_orders, widths)):
                grew = ddx_new[0][s] is not None and coarse_dd[s] > 0 and np.abs(ddx_new[0][s]).max() / coarse_dd[s] > self.non_smooth_threshold
                if bad[s] and grew and len(orders) + (self.n_segments - s) < self.max_segments:
                    orders += [coarse_orders[s]] * 2  # non-smooth: split, keep the degree
                    new_w += [w / 2] * 2
                else:
                    orders.append(p)
                    new_w.append(w)
            if it + 1 == max_iter:
                break  # the returned solution belongs to the grid it was computed on: no regrid after the last solve
            widths = np.asarray(new_w)
            self._regrid(clip(orders), widths)
        return solution


def _ref_control_order(U):
    """The reference flattens the control guess node-major (``np.concatenate`` of an (N, nu)
    array, mpopt.py:678-689) although the decision vector is control-major -- identical for
    nu == 1 and for constant guesses.  Reproduced as is: parity beats tidiness."""
    return np.concatenate(U)


class post_process:
    """Solution data of the optimizer for further processing and plots (mpopt.py:1576-2270).

        >>> post = post_process(solution, trajectories, options)      # what mpopt.process_results builds

    ``trajectories[phase](z, seg_widths) -> (x, u, t, t0, tf, a)`` (scaled x, u, a; unscaled t), as returned by
    ``mpopt.init_trajectories``.  ``options["interpolate"]``, when present, is a callable
    ``(phase, taus) -> (Xi, Ui)`` evaluating the collocation polynomials on the GPU (kernel mpx_resid_*); without it
    the composite interpolation matrix is applied on the host like the reference does (mpopt.py:1804-1812)."""

    _INTERPOLATION_NODES_PER_SEG = 50

    def __init__(self, solution={}, trajectories=None, options={}):
        self.solution, self.trajectories, self.options = solution, trajectories, options
        self.phases = self.options["phases_to_plot"][0] if "phases_to_plot" in self.options else [0]
        self.nx, self.nu, self.na = self.options.get("nx", 1), self.options.get("nu", 1), self.options.get("na", 0)
        self.scaling = self.options.get("scaling", False)
        self.tau0 = self.options.get("tau0", CollocationRoots._TAU_MIN)
        self.tau1 = self.options.get("tau1", CollocationRoots._TAU_MAX)
        self.residuals = options.get("residuals")

    def get_trajectories(self, phase=0):
        """(x, u, t, a) of one phase; unscaled unless ``options["scaling"]`` (mpopt.py:1633-1661)."""
        x, u, t, t0, tf, a = self.trajectories[phase](self.solution["x"], self.options["seg_widths"])
        x, u, t, a = (np.asarray(v.full() if hasattr(v, "full") else v, float) for v in (x, u, t, a))
        if not self.scaling:
            # a (na x 1) divided by the 1-D scale vector broadcasts to na x na -- the reference does exactly this
            # (mpopt.py:1655-1659); reproduced so that callers indexing its result keep working
            return (x / self.options.get("scale_x", 1.0), u / self.options.get("scale_u", 1.0), t, a / self.options.get("scale_a", 1.0))
        return (x, u, t, a)

    def get_original_data(self, phases=[]):
        if not phases:
            phases = self.phases
        x, u, t, a = self.get_trajectories(phases[0])
        for phase in phases[1:]:
            xp, up, tp, ap = self.get_trajectories(phase)
            x, u, t, a = np.vstack((x, xp)), np.vstack((u, up)), np.vstack((t, tp)), np.vstack((a, ap))
        return (x, u, t, a)

    def get_interpolation_taus(self, n=75, taus_orig=None, method="uniform"):
        if method == "uniform" or taus_orig is None:
            return np.linspace(self.tau0, self.tau1, n)
        return self.get_non_uniform_interpolation_grid(taus_orig, n)

    @staticmethod
    def get_non_uniform_interpolation_grid(taus_orig, n=75):
        """Insert mid-points until there are ``n`` points, at most 6 times (mpopt.py:1712-1738)."""
        taus = np.asarray(taus_orig, float)
        count = 0
        while len(taus) < n:
            out = np.empty(2 * len(taus) - 1)
            out[0::2], out[1::2] = taus, (taus[:-1] + taus[1:]) / 2.0
            taus = out
            count += 1
            if count > 5:
                break
        return taus

    @staticmethod
    def get_interpolated_time_grid(t_orig, taus, poly_orders, tau0, tau1):
        return mpopt.get_interpolated_time_grid(t_orig, taus, poly_orders, tau0, tau1).ravel()

    def get_interpolated_data(self, phases, taus=[]):
        """(x, u, t, a) on a refined grid: by default ``_INTERPOLATION_NODES_PER_SEG`` equally spaced points per
        segment (mpopt.py:1773-1831)."""
        poly_orders = list(self.options["poly_orders"])
        if len(taus) == 0:
            taus = [self.get_interpolation_taus(n=self._INTERPOLATION_NODES_PER_SEG)[1:] for _ in poly_orders]
            taus[0] = np.append(self.tau0, taus[0])
        gpu = self.options.get("interpolate")
        compI = None
        xs, us, ts, as_ = [], [], [], []
        for phase in phases:
            x_orig, u_orig, t_orig, a = self.get_original_data([phase])
            if gpu is not None:
                scale_x = 1.0 if self.scaling else self.options.get("scale_x", 1.0)
                scale_u = 1.0 if self.scaling else self.options.get("scale_u", 1.0)
                Xi, Ui = gpu(phase, taus)
                x, u = Xi / scale_x, Ui / scale_u
            else:
                if compI is None:
                    compI = Collocation(poly_orders, self.options.get("colloc_scheme", "LGR")).get_composite_interpolation_matrix(taus, poly_orders)
                x, u = np.dot(compI, x_orig), np.dot(compI, u_orig)
            xs.append(x), us.append(u), as_.append(a)
            ts.append(self.get_interpolated_time_grid(t_orig, taus, poly_orders, self.tau0, self.tau1))
        return (np.vstack(xs), np.vstack(us), np.hstack(ts), np.hstack(as_) if len(phases) > 1 else as_[0])

    def get_data(self, phases=[], interpolate=False):
        if not phases:
            phases = self.phases
        return self.get_interpolated_data(phases) if interpolate else self.get_original_data(phases)

    # ---- plots (mpopt.py:1858-2270): thin matplotlib front-ends over the data methods above -----------------
    @staticmethod
    def _plt():
        import matplotlib

        if not matplotlib.get_backend():  # pragma: no cover
            matplotlib.use("Agg")
        import matplotlib.pyplot as plt

        return plt

    @staticmethod
    def plot_curve(ax, x, t, name=None, ylabel="", tics=["-"] * 15, legend_index=None):
        """Columns of ``x`` against ``t`` on one axis; legend ``name i`` (or ``name legend_index[i]``)."""
        x = np.asarray(x, float).reshape(len(np.asarray(t).ravel()), -1)
        for i in range(x.shape[1]):
            label = None if name is None else f"{name} {legend_index[i] if legend_index is not None else i}"
            ax.plot(np.asarray(t).ravel(), x[:, i], tics[i % len(tics)], label=label)
        if name is not None:
            ax.legend()
        ax.set(ylabel=ylabel)
        ax.grid(True)

    def plot_all(self, x, u, t, tics=None, fig=None, axs=None, legend=True, name=""):
        """States on the first axis, controls on the second."""
        plt = self._plt()
        tics = ["-"] * 15 if tics is None else tics
        if fig is None and axs is None:
            fig, axs = plt.subplots(2, 1, sharex=True)
        self.plot_curve(axs[0], x, t, name=(name + "state") if legend else None, ylabel="State variables", tics=tics)
        self.plot_curve(axs[1], u, t, name=(name + "control") if legend else None, ylabel="Control variables", tics=tics)
        axs[1].set(xlabel="Time, s")
        return fig, axs

    def plot_phases(self, phases=None, interpolate=True, residuals=True, fig=None, axs=None, tics=["-"] * 15, name=""):
        """States and controls of the given phases (refined grid by default, nodes as markers); a third axis with
        the dynamics residuals when they were computed by ``process_results``."""
        plt = self._plt()
        if phases is None:
            phases = self.options["phases_to_plot"][0] if "phases_to_plot" in self.options else self.phases
        phases = list(phases)
        with_res = bool(residuals) and self.residuals is not None and "t_dx" in self.residuals
        if fig is None and axs is None:
            fig, axs = plt.subplots(3 if with_res else 2, 1, sharex=True)
        x, u, t, _ = self.get_original_data(phases)
        if interpolate:
            xi, ui, ti, _ = self.get_interpolated_data(phases)
            self.plot_all(xi, ui, ti, tics=tics, fig=fig, axs=axs, name=name)
            self.plot_all(x, u, t, tics=["."] * 15, fig=fig, axs=axs, legend=False)
        else:
            self.plot_all(x, u, t, tics=tics, fig=fig, axs=axs, name=name)
        if with_res:
            tr, rr = self.residuals["t_dx"]
            self.plot_residuals(tr, rr, phases=phases, fig=fig, axs=axs[2])
        return fig, axs

    def plot_phase(self, phase=0, interpolate=True, fig=None, axs=None):
        return self.plot_phases([phase], interpolate, fig=fig, axs=axs)

    def plot_single_variable(self, var_data, t, dims, name=None, ylabel=None, axis=1, fig=None, axs=None, tics=["-"] * 15):
        """One subplot per entry of ``dims`` (an entry may be a list of columns sharing a subplot), stacked along ``axis``."""
        plt = self._plt()
        groups = [list(np.atleast_1d(d)) for d in dims]
        if fig is None and axs is None:
            fig, axs = plt.subplots(*( (len(groups), 1) if axis == 1 else (1, len(groups)) ), squeeze=False)
        flat = np.asarray(axs).ravel()
        var_data = np.asarray(var_data, float)
        for ax, cols in zip(flat, groups):
            self.plot_curve(ax, var_data[:, cols], t, name=name, ylabel=ylabel or "", tics=tics, legend_index=cols)
        flat[-1].set(xlabel="Time, s")
        return fig, axs

    def _plot_var(self, which, dims, phases, axis, interpolate, fig, axs, tics, name, ylabel):
        phases = list(self.phases if phases is None else phases)
        x, u, t, _ = self.get_data(phases, interpolate=interpolate)
        data = x if which == "x" else u
        dims = list(range(data.shape[1])) if dims is None else dims
        return self.plot_single_variable(data, t, dims, name=name, ylabel=ylabel, axis=axis, fig=fig, axs=axs,
                                         tics=["-"] * 15 if tics is None else tics)

    def plot_x(self, dims=None, phases=None, axis=1, interpolate=True, fig=None, axs=None, tics=["-"] * 15):
        return self._plot_var("x", dims, phases, axis, interpolate, fig, axs, tics, "state", "State variables")

    def plot_u(self, dims=None, phases=None, axis=1, interpolate=True, fig=None, axs=None, tics=None, name="control", ylabel="Control variables"):
        return self._plot_var("u", dims, phases, axis, interpolate, fig, axs, tics, name, ylabel)

    @staticmethod
    def sort_residual_data(time, residuals, phases=[0]):
        """(r, t): 2-norm over the states of every residual point, phases concatenated (mpopt.py:2210-2229)."""
        rs, ts = [], []
        for phase in phases:
            for seg_t, seg_r in zip(time[phase], residuals[phase]):
                if seg_r is None or len(np.atleast_1d(seg_t)) == 0:
                    continue
                rs.append(np.linalg.norm(np.asarray(seg_r, float).reshape(len(np.asarray(seg_t).ravel()), -1), 2, axis=1))
                ts.append(np.asarray(seg_t, float).ravel())
        r = np.concatenate(rs) if rs else np.zeros(0)
        t = np.concatenate(ts) if ts else np.zeros(0)
        return (r.reshape(-1, 1), t)

    @classmethod
    def plot_residuals(cls, time, residuals, phases=[0], name=None, fig=None, axs=None, tics=["."] * 15):
        plt = cls._plt()
        if fig is None and axs is None:
            fig, axs = plt.subplots(1, 1)
        r, t = cls.sort_residual_data(time, residuals, phases=phases)
        cls.plot_curve(axs, r, t, name, ylabel="residuals", tics=tics, legend_index=[""] * 15)
        cls.plot_curve(axs, r, t, ylabel="residuals", tics=["-"] * 15, legend_index=[""] * 15)
        axs.set(xlabel="Time, s")
        return fig, axs


def solve(ocp, n_segments=1, poly_orders=9, scheme="LGR", plot=True, solve_dict=dict(), residual_x=False, residual_dx=True):
    mpo = mpopt(ocp, n_segments=n_segments, poly_orders=poly_orders, scheme=scheme)
    solution = mpo.solve(**solve_dict)
    post = mpo.process_results(solution, plot=plot, residual_x=residual_x, residual_dx=residual_dx)
    return (mpo, post)


def get_segment_boundaries():
    """Placeholder of the reference's module surface (mpopt.py:4311-4313: a function with an empty body)."""
    return None


def __getattr__(name):
    """``mp.plt`` like the reference module (mpopt.py:25), imported on first use."""
    if name == "plt":
        return post_process._plt()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")

"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain numpy / sympy restatement of the reference's algorithm for the collocation hot path
(/root/reference/mpopt/mpopt.py).  It exists to CHECK the HIP kernels; it is never the thing
that is shipped or measured as the product:  only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  ``mpopt_amd`` never does.

Pinning:
  * tables (roots, D, W, interpolation, composites): pinned against ``tests/golden/tables.npz``,
    produced by importing the reference's CollocationRoots / Collocation ("numerical" back-end);
  * z0, bounds, f, g, jac_g, grad_f, hess_l: pinned against ``tests/golden/nlp_*.npz``, produced by
    running the reference's own ``mpopt.create_nlp()`` over a sympy-backed stand-in for CasADi
    (tests/golden/casadi_shim.py) because CasADi itself (casadi==3.6.0, requirements.txt:4) is not
    installable here.  CasADi's own floating-point evaluation order and AD are therefore NOT
    pinned ("parity unpinned" for that last step); differences are O(1e-16) relative.

Each function cites the reference lines it follows.  Derivatives take an independent route from
the product (sympy differentiation of per-node expressions + numpy assembly, versus the
product's own tracer + generated device code + C++ pattern builder).
"""
import numpy as np
import scipy.sparse as sp
import scipy.special


# ---------------------------------------------------------------------------------------------
# CollocationRoots (mpopt.py:4134-4276)
# ---------------------------------------------------------------------------------------------
def roots(scheme, deg, tau_min=-1.0, tau_max=1.0):
    def scale(r):  # mpopt.py:4203, 4224, 4250, 4274
        return tau_min + (tau_max - tau_min) / 2 * (np.asarray(r, dtype=float) + 1)

    if scheme == "LG":  # mpopt.py:4199-4203
        return scale(np.append(-1, np.polynomial.legendre.leggauss(deg - 1)[0]))
    if scheme in ("LGR", "LGL"):  # mpopt.py:4216-4229, 4242-4255
        if deg > 1:
            r = scipy.special.j_roots(deg - 1, 1.0, 0.0 if scheme == "LGR" else 1.0)[0]
            return scale(np.append(np.append(-1, r), 1.0))
        return np.array([tau_min, tau_max], dtype=float) if deg == 1 else np.array([0.0])
    if scheme == "CGL":  # mpopt.py:4270-4274
        return scale(np.array([np.cos(np.pi * j / deg) for j in range(deg + 1)])[::-1])
    # unknown scheme -> equally spaced (mpopt.py:4182-4188); argument is the node count
    return np.linspace(tau_min, tau_max, deg) if deg > 1 else np.array([tau_min, tau_max], dtype=float)


# ---------------------------------------------------------------------------------------------
# Collocation, D_MATRIX_METHOD == "numerical" (mpopt.py:3815-3905, 3987-4131)
# ---------------------------------------------------------------------------------------------
def lagrange_polys(nodes):  # mpopt.py:4006-4011
    n = len(nodes)
    polys = []
    for j in range(n):
        p = np.poly1d([1])
        for i in range(n):
            if i != j:
                p *= np.poly1d([1, -nodes[i]]) / (nodes[j] - nodes[i])
        polys.append(p)
    return polys


def diff_matrix(nodes, taus=None, order=1):  # mpopt.py:3815-3849
    at = nodes if taus is None else taus
    polys = lagrange_polys(nodes)
    D = np.zeros((len(at), len(nodes)))
    for j, p in enumerate(polys):
        d = np.polyder(p) if order == 1 else np.polyder(np.polyder(p))
        for i in range(len(at)):
            D[i, j] = d(at[i])
    return D


def quad_weights(nodes, a, b):  # mpopt.py:3851-3882
    w = np.zeros(len(nodes))
    for i, p in enumerate(lagrange_polys(nodes)):
        P = np.polyint(p)
        w[i] = P(b) - P(a)
    return w


def interp_matrix(nodes, taus):  # mpopt.py:3884-3905
    C = np.zeros((len(taus), len(nodes)))
    for j, p in enumerate(lagrange_polys(nodes)):
        for i in range(len(taus)):
            C[i, j] = p(taus[i])
    return C


# ---------------------------------------------------------------------------------------------
# The same definitions in multi-precision arithmetic.  The reference's "numerical" back-end above loses
# digits as the degree grows (np.poly1d coefficient products: ~1e-9 at p=20, ~4e-4 at p=30, see
# tests/test_tables.py and DESIGN.md), while its default "symbolic" back-end (CasADi AD of the
# product form, mpopt.py:3832-3840) does not.  Above degree 10 the oracle therefore evaluates the
# definitions l_j^(k)(tau_i), int l_j exactly (mpmath) instead of imitating the rounding failure.
# ---------------------------------------------------------------------------------------------
def _mp_basis_coeffs(nodes):
    import mpmath as mpm

    # (monomial coefficients of degree-n basis polynomials on [-1, 1] grow like 2^n and their evaluation cancels as much again:
    # 50 digits are exact to 1e-16 up to degree ~60 only -- at degree 100 the weights came out wrong in the 4th digit)
    mpm.mp.dps = max(50, 30 + len(nodes))
    xs = [mpm.mpf(float(v)) for v in nodes]
    out = []
    for j in range(len(xs)):
        c = [mpm.mpf(1)]  # highest power first
        for i in range(len(xs)):
            if i != j:
                den = xs[j] - xs[i]
                new = [mpm.mpf(0)] * (len(c) + 1)
                for k, ck in enumerate(c):
                    new[k] += ck / den
                    new[k + 1] -= ck * xs[i] / den
                c = new
        out.append(c)
    return out


def _mp_polyval(c, t):
    import mpmath as mpm

    v = mpm.mpf(0)
    for ck in c:
        v = v * t + ck
    return v


def exact_tables_monomial(nodes, taus, order, a=None, b=None):
    """order 0/1/2: matrix l_j^(order)(taus_i); order 'w': weights over [a, b] -- from the monomial coefficients of the basis
    polynomials (O(n^3) multi-precision operations with n + 30 digits: minutes at degree 255).  The form the oracle used up to round
    5; kept as the independent check of ``exact_tables`` (tests/test_oracle.py)."""
    import mpmath as mpm

    C = _mp_basis_coeffs(nodes)
    n = len(nodes)
    if order == "w":
        out = np.zeros(n)
        for j, c in enumerate(C):
            I = [ck / (n - k) for k, ck in enumerate(c)] + [mpm.mpf(0)]
            out[j] = float(_mp_polyval(I, mpm.mpf(float(b))) - _mp_polyval(I, mpm.mpf(float(a))))
        return out
    out = np.zeros((len(taus), n))
    for j, c in enumerate(C):
        d = c
        for _ in range(order):
            m = len(d) - 1
            d = [ck * (m - k) for k, ck in enumerate(d[:-1])] or [mpm.mpf(0)]
        for i, t in enumerate(taus):
            out[i, j] = float(_mp_polyval(d, mpm.mpf(float(t))))
    return out


def _mp_gauss_legendre(nq):
    """Gauss-Legendre nodes / weights on [-1, 1] in the current mpmath precision (Newton on the three-term recurrence from
    numpy's double-precision nodes)."""
    import mpmath as mpm

    xs, ws = [], []
    for x0 in np.polynomial.legendre.leggauss(nq)[0]:
        x = mpm.mpf(float(x0))
        for _ in range(6):
            p0, p1 = mpm.mpf(1), x
            for k in range(2, nq + 1):
                p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
            dp = nq * (x * p1 - p0) / (x * x - 1)
            x = x - p1 / dp
        p0, p1 = mpm.mpf(1), x
        for k in range(2, nq + 1):
            p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
        dp = nq * (x * p1 - p0) / (x * x - 1)
        xs.append(x)
        ws.append(2 / ((1 - x * x) * dp * dp))
    return xs, ws


def exact_tables(nodes, taus, order, a=None, b=None):
    """order 0/1/2: matrix l_j^(order)(taus_i); order 'w': weights int_a^b l_j -- the DEFINITIONS evaluated in 50-digit arithmetic
    through the barycentric form (well conditioned at every degree, O(n^2) operations per table): l_j(t) = (lam_j / (t - x_j)) /
    sum_m lam_m / (t - x_m); l_j'(x_i) = (lam_j / lam_i) / (x_i - x_j), l_i'(x_i) = - sum_{j != i} l_j'(x_i) (the derivatives of a
    partition of unity sum to zero); l_j^(k) at other points and l_j'' by exact re-interpolation (l_j' has degree < n); the weights
    by a Gauss-Legendre rule that is exact for degree n - 1.  Equal to the monomial form above wherever that one has digits left
    (tests/test_oracle.py: degrees 12 ... 100)."""
    import mpmath as mpm

    mpm.mp.dps = 50
    xs = [mpm.mpf(float(v)) for v in nodes]
    n = len(xs)
    lam = []
    for j in range(n):
        v = mpm.mpf(1)
        for m in range(n):
            if m != j:
                v *= xs[j] - xs[m]
        lam.append(1 / v)

    def basis_row(t):
        for j in range(n):
            if t == xs[j]:
                return [mpm.mpf(1 if m == j else 0) for m in range(n)]
        s = [lam[j] / (t - xs[j]) for j in range(n)]
        tot = mpm.fsum(s)
        return [v / tot for v in s]

    if order == "w":
        gx, gw = _mp_gauss_legendre(n // 2 + 1)
        a, b = mpm.mpf(float(a)), mpm.mpf(float(b))
        half, mid = (b - a) / 2, (b + a) / 2
        acc = [mpm.mpf(0)] * n
        for x, w in zip(gx, gw):
            row = basis_row(mid + half * x)
            acc = [s + w * v for s, v in zip(acc, row)]
        return np.array([float(v * half) for v in acc])
    if order == 0:
        return np.array([[float(v) for v in basis_row(mpm.mpf(float(t)))] for t in taus]).reshape(len(taus), n)
    D1 = [[mpm.mpf(0)] * n for _ in range(n)]
    for i in range(n):
        for j in range(n):
            if i != j:
                D1[i][j] = (lam[j] / lam[i]) / (xs[i] - xs[j])
        D1[i][i] = -mpm.fsum(D1[i])
    Dk = D1
    if order == 2:
        Dk = [[mpm.fsum(D1[i][m] * D1[m][j] for m in range(n)) for j in range(n)] for i in range(n)]
    out = np.zeros((len(taus), n))
    for i, t in enumerate(taus):
        t = mpm.mpf(float(t))
        hit = [k for k in range(n) if t == xs[k]]
        if hit:
            out[i] = [float(v) for v in Dk[hit[0]]]
        else:
            row = basis_row(t)
            out[i] = [float(mpm.fsum(row[m] * Dk[m][j] for m in range(n))) for j in range(n)]
    return out


class Grid:
    """Tables of one grid: what mpopt.compute_numerical_approximation caches (mpopt.py:95-103).
    ``method``: "numerical" = the reference's np.poly1d arithmetic (pinned by the goldens),
    "exact" = 50-digit evaluation, "auto" = numerical up to degree 10, exact above."""

    def __init__(self, poly_orders, scheme, tau0=-1.0, tau1=1.0, method="auto"):
        self.orders = [int(p) for p in poly_orders]
        self.scheme, self.tau0, self.tau1 = scheme, float(tau0), float(tau1)
        self.S = len(self.orders)
        self.N = sum(self.orders) + 1
        self.taus = {d: roots(scheme, d, tau0, tau1) for d in set(self.orders)}
        self.exact = {d: (method == "exact" or (method == "auto" and d > 10)) for d in self.taus}
        self.D = {d: self._diff(d, None, 1) for d in self.taus}
        self.w = {d: (exact_tables(self.taus[d], None, "w", tau0, tau1) if self.exact[d] else quad_weights(self.taus[d], tau0, tau1))
                  for d in self.taus}
        self.start = np.concatenate([[0], np.cumsum(self.orders)]).astype(int)

    def _diff(self, d, taus, order):
        if self.exact[d]:
            return exact_tables(self.taus[d], self.taus[d] if taus is None else taus, order)
        return diff_matrix(self.taus[d], taus, order)

    def _interp(self, d, taus):
        return exact_tables(self.taus[d], taus, 0) if self.exact[d] else interp_matrix(self.taus[d], taus)

    def comp_D(self):  # mpopt.py:4015-4039
        out = np.zeros((self.N, self.N))
        for i, p in enumerate(self.orders):
            if i == 0:
                out[0:p + 1, 0:p + 1] = self.D[p]
            else:
                st = self.start[i]
                out[st + 1:st + 1 + p, st:st + 1 + p] = self.D[p][1:, :]
        return out

    def comp_W(self):  # mpopt.py:4041-4064 (w_0 of later segments is dropped)
        return np.concatenate([self.w[self.orders[0]][:1]] + [self.w[p][1:] for p in self.orders])

    def comp_interp(self, taus_list, deriv=0):  # mpopt.py:4066-4131
        n_t = [len(t) for t in taus_list]
        out = np.zeros((sum(n_t), self.N))
        r = 0
        for i, p in enumerate(self.orders):
            if n_t[i]:
                blk = self._interp(p, taus_list[i]) if deriv == 0 else self._diff(p, taus_list[i], deriv)
                out[r:r + n_t[i], self.start[i]:self.start[i] + p + 1] = blk
            r += n_t[i]
        return out

    def comp_I_mid(self):  # mpopt.py:350-359
        return self.comp_interp([list((self.taus[p][:-1] + self.taus[p][1:]) / 2.0) for p in self.orders])

    def node_seg_point(self):
        """(segment, point) of every node following the loop at mpopt.py:189-195, 208."""
        seg, pt = np.zeros(self.N, int), np.zeros(self.N, int)
        s, k = 0, 0
        for i in range(self.N):
            if k > self.orders[s]:
                s, k = s + 1, 1
            seg[i], pt[i] = s, k
            k += 1
        return seg, pt


# ---------------------------------------------------------------------------------------------
# Transcription (mpopt.py:105-639)
# ---------------------------------------------------------------------------------------------
def _almost_everywhere(e):
    """Derivatives of the piecewise functions (Abs, sign, Max, Min) as AD takes them -- away from the kinks: sympy's delta terms
    (DiracDelta, unevaluated d sign / d Heaviside) are zero almost everywhere, and CasADi's AD never forms them (d|x| = sign x,
    d sign = 0, d fmax = the active argument's).  Nested lists of expressions in, the same shape out."""
    import sympy as sy

    if isinstance(e, (list, tuple)):
        return [_almost_everywhere(v) for v in e]
    e = sy.sympify(e)
    if not e.has(sy.DiracDelta, sy.Derivative, sy.Subs):
        return e
    e = e.replace(lambda x: isinstance(x, sy.Subs) and isinstance(x.expr, sy.Derivative) and x.expr.expr.func in (sy.sign, sy.Heaviside), lambda x: sy.S.Zero)
    e = e.replace(lambda x: isinstance(x, sy.Derivative) and x.expr.func in (sy.sign, sy.Heaviside), lambda x: sy.S.Zero)
    return e.replace(lambda x: isinstance(x, sy.DiracDelta), lambda x: sy.S.Zero)


class OracleNLP:
    """f, g and derivatives of the NLP that ``mpopt.create_nlp`` builds, evaluated on the CPU."""

    def __init__(self, ocp, n_segments, poly_orders, scheme="LGR", tau0=-1.0, tau1=1.0, table_method="auto"):
        import sympy

        self.sympy = sympy
        self.ocp = o = ocp
        orders = [poly_orders] * n_segments if isinstance(poly_orders, (int, np.integer)) else list(poly_orders)
        assert len(orders) == n_segments
        self.grid = G = Grid(orders, scheme, tau0, tau1, table_method)
        self.S, self.N = G.S, G.N
        self.nx, self.nu, self.na, self.n_ph = o.nx, o.nu, o.na, o.n_phases
        self.sx, self.su, self.sa = (np.asarray(v, float) for v in (o.scale_x, o.scale_u, o.scale_a))
        self.st = float(o.scale_t)
        self.compD, self.compW = G.comp_D(), G.comp_W()
        self.seg, self.pt = G.node_seg_point()
        self.n_zp = self.N * (self.nx + self.nu) + 2 + self.na + self._extra_vars_per_phase()  # mpopt.py:537-543
        self.n_z = self.n_zp * self.n_ph
        self.n_p = self.S * self.n_ph
        self.has_path = [o.has_path_constraints(ph) for ph in range(self.n_ph)]
        self.has_tc = [o.has_terminal_constraints(ph) for ph in range(self.n_ph)]
        self.midu = [bool(o.midu[ph]) and bool((np.asarray(o.lbu[ph]) > -np.inf).any() or (np.asarray(o.ubu[ph]) < np.inf).any())
                     for ph in range(self.n_ph)]
        self.I_mid = G.comp_I_mid()
        if self.S > 1:  # mpopt.py:398-403
            Dat = G.comp_interp([np.array([tau0, tau1])] * self.S, deriv=1)
            self.D_cont = Dat[1:-1][::2] - Dat[2:-1][::2]
        self._build_symbolic()
        self.n_g = len(self.g(np.zeros(self.n_z) + 0.5, np.full(self.n_p, 1.0 / self.S)))

    # -- helpers ---------------------------------------------------------------------------
    def _extra_vars_per_phase(self):
        return 0

    def split(self, z, ph):
        zp = np.asarray(z, float)[ph * self.n_zp:(ph + 1) * self.n_zp]
        N, nx, nu = self.N, self.nx, self.nu
        X = zp[:nx * N].reshape(nx, N).T
        U = zp[nx * N:(nx + nu) * N].reshape(nu, N).T
        return X, U, zp[(nx + nu) * N], zp[(nx + nu) * N + 1], zp[(nx + nu) * N + 2:(nx + nu) * N + 2 + self.na]

    def zidx(self, ph, kind, comp=0, node=0):
        base, N, nx, nu = ph * self.n_zp, self.N, self.nx, self.nu
        return {"X": base + comp * N + node, "U": base + (nx + comp) * N + node, "t0": base + (nx + nu) * N,
                "tf": base + (nx + nu) * N + 1, "A": base + (nx + nu) * N + 2 + comp}[kind]

    def node_times(self, t0, tf, w):
        """h_seg and t of every node, accumulated like mpopt.py:180-198."""
        G = self.grid
        h_node, t_node = np.zeros(self.N), np.zeros(self.N)
        t_seg0, s = t0, 0
        h = (tf - t0) / (G.tau1 - G.tau0) * w[0]
        for i in range(self.N):
            if self.seg[i] != s:
                s = self.seg[i]
                t_seg0 = t_seg0 + h * (G.tau1 - G.tau0)
                h = (tf - t0) / (G.tau1 - G.tau0) * w[s]
            h_node[i] = h
            t_node[i] = t_seg0 + h * (G.taus[G.orders[s]][self.pt[i]] - G.tau0)
        return h_node, t_node

    # -- values: a line-by-line restatement ------------------------------------------------------
    def phase_parts(self, z, p, ph):
        o, G = self.ocp, self.grid
        X, U, t0v, tfv, A = self.split(z, ph)
        t0, tf = t0v / self.st, tfv / self.st  # mpopt.py:175-176
        a = A / self.sa  # mpopt.py:177
        w = np.asarray(p, float)[ph * self.S:(ph + 1) * self.S]
        h, t = self.node_times(t0, tf, w)
        dyn, pc, rc = o.get_dynamics(ph), o.get_path_constraints(ph), o.get_running_costs(ph)
        f = np.zeros((self.N, self.nx))
        c = []
        q = np.zeros(self.N)
        for i in range(self.N):
            x, u = X[i] / self.sx, U[i] / self.su  # mpopt.py:196-197
            f[i] = h[i] * self.sx * np.array([float(v) for v in dyn(x, u, t[i], a)])  # mpopt.py:201
            if self.has_path[ph]:
                c.append([float(v) for v in pc(x, u, t[i], a)])  # mpopt.py:204
            q[i] = h[i] * float(rc(x, u, t[i], a))  # mpopt.py:206
        F = (self.compD @ X - f).T.ravel()  # mpopt.py:227-232, state-major
        parts = [F]
        if self.has_path[ph]:
            parts.append(np.array(c).T.ravel())  # mpopt.py:255
        if o.diff_u[ph]:
            parts.append((self.compD @ U).T.ravel())  # mpopt.py:315-321
        if self.midu[ph]:
            parts.append((self.I_mid @ U).T.ravel())  # mpopt.py:357-366
        if self.S > 1 and o.du_continuity[ph]:
            parts.append((self.D_cont @ U).T.ravel())  # mpopt.py:394-408
        x0, xf = X[0] / self.sx, X[-1] / self.sx  # mpopt.py:277-278
        if self.has_tc[ph]:
            parts.append(np.array([float(v) for v in o.get_terminal_constraints(ph)(xf, tf, x0, t0, a)]))  # 289
        J = float(o.get_terminal_costs(ph)(xf, tf, x0, t0, a)) + float(self.compW @ q)  # mpopt.py:297-298, 455
        return np.concatenate(parts), J

    def events(self, z):  # mpopt.py:464-521
        o = self.ocp
        if self.n_ph < 2:
            return np.zeros(0)
        ex, eu, et = [], [], []
        for (i, j) in o.phase_links:
            Xi, Ui, _, tfi, _ = self.split(z, i)
            Xj, Uj, t0j, _, _ = self.split(z, j)
            ex.append(Xj[0] - Xi[-1]), eu.append(Uj[0] - Ui[-1]), et.append([t0j - tfi])
        return np.concatenate([np.concatenate(ex), np.concatenate(eu), np.concatenate(et)])

    def g(self, z, p):  # mpopt.py:612-624
        return np.concatenate([self.phase_parts(z, p, ph)[0] for ph in range(self.n_ph)] + [self.events(z)])

    def f(self, z, p):
        return sum(self.phase_parts(z, p, ph)[1] for ph in range(self.n_ph))

    # -- post-solve: off-node interpolation and dynamics residuals -----------------------------------
    def residuals(self, z, p, phase, taus_per_segment):
        """Restatement of mpopt.interpolate_single_phase + get_dynamics_residuals_single_phase
        (mpopt.py:1428-1573).  Returns dict ti, xi, ui, dxi, dui, dyn, resid (points concatenated over
        segments in order)."""
        o, G = self.ocp, self.grid
        X, U, t0v, tfv, A = self.split(z, phase)
        t0, tf = t0v / self.st, tfv / self.st  # init_trajectories: t0, tf unscaled (mpopt.py:872)
        w = np.asarray(p, float)[phase * self.S:(phase + 1) * self.S]
        _, t_node = self.node_times(t0, tf, w)  # time_grid of the transcription (mpopt.py:209, 873)
        taus = [np.asarray(t, float) for t in taus_per_segment]
        CI = G.comp_interp(taus, 0)  # mpopt.py:1516-1521
        CD = G.comp_interp(taus, 1)
        Xi, Ui, DXi, DUi = CI @ X, CI @ U, CD @ X, CD @ U  # mpopt.py:1523-1526
        # get_interpolated_time_grid (mpopt.py:1545-1573)
        t_seg = [t_node[0]] + [t_node[sum(G.orders[:i + 1])] for i in range(self.S)]
        ti = np.concatenate([t_seg[i] + (t_seg[i + 1] - t_seg[i]) * ((taus[i] - G.tau0) / (G.tau1 - G.tau0)) for i in range(self.S)])
        dyn = o.get_dynamics(phase)
        F = np.zeros((len(ti), self.nx))
        idx = 0
        for s in range(self.S):  # mpopt.py:1451-1480
            h_seg = (tf - t0) / (G.tau1 - G.tau0) * w[s]
            for _ in range(len(taus[s])):
                f = dyn(Xi[idx] / self.sx, Ui[idx] / self.su, ti[idx], A / self.sa)
                F[idx] = h_seg * (np.array([float(v) for v in f]) * self.sx)
                idx += 1
        # states re-integrated from the dynamics (compute_states_from_solution_dynamics, mpopt.py:1030-1066):
        # quadrature weights over [tau0, tau_i] of the Lagrange basis on the target points of the segment
        xint = np.zeros_like(Xi)
        idx = 0
        for s in range(self.S):
            n = len(taus[s])
            if n == 0:
                continue
            xstart = X[G.start[s]]
            for i in range(n):
                q = exact_tables(taus[s], None, "w", G.tau0, taus[s][i]) if n > 11 else quad_weights(taus[s], G.tau0, taus[s][i])
                xint[idx + i] = xstart + q @ F[idx:idx + n]
            idx += n
        return dict(ti=ti, xi=Xi, ui=Ui, dxi=DXi, dui=DUi, dyn=F, resid=DXi - F, xint=xint, xres=Xi - xint)

    def residuals_of_segments(self, z, p, phase, taus_per_segment, segments):
        """The same quantities as ``residuals`` for a SAMPLE of segments, without forming the composite matrices (they are
        (n_points x N) dense: 1 GB at 4000 segments).  Per segment s: Xi = C_s X_s, DXi = D_s X_s with the segment's own
        interpolation / differentiation rows (mpopt.py:1516-1526 restricted to the block of segment s, 4088-4095, 4123-4130),
        ti from the segment's end times (mpopt.py:1545-1573), F = h_s Sx dyn (mpopt.py:1451-1480).
        Returns {s: dict(ti, xi, ui, dxi, dui, dyn, resid)}."""
        o, G = self.ocp, self.grid
        X, U, t0v, tfv, A = self.split(z, phase)
        t0, tf = t0v / self.st, tfv / self.st
        w = np.asarray(p, float)[phase * self.S:(phase + 1) * self.S]
        wc = np.concatenate([[0.0], np.cumsum(w)])
        dyn = o.get_dynamics(phase)
        out = {}
        for s in segments:
            d, st = G.orders[s], G.start[s]
            taus = np.asarray(taus_per_segment[s], float)
            C, D = G._interp(d, taus), G._diff(d, taus, 1)
            Xs, Us = X[st:st + d + 1], U[st:st + d + 1]
            Xi, Ui, DXi, DUi = C @ Xs, C @ Us, D @ Xs, D @ Us
            ts0, ts1 = t0 + (tf - t0) * wc[s], t0 + (tf - t0) * wc[s + 1]
            ti = ts0 + (ts1 - ts0) * ((taus - G.tau0) / (G.tau1 - G.tau0))
            h_seg = (tf - t0) / (G.tau1 - G.tau0) * w[s]
            F = np.array([h_seg * (np.array([float(v) for v in dyn(Xi[k] / self.sx, Ui[k] / self.su, ti[k], A / self.sa)]) * self.sx)
                          for k in range(len(taus))]).reshape(len(taus), self.nx)
            out[s] = dict(ti=ti, xi=Xi, ui=Ui, dxi=DXi, dui=DUi, dyn=F, resid=DXi - F)
        return out

    # -- bounds and initial guess ---------------------------------------------------------------
    def bounds(self):
        """(lbx, ubx, lbg, ubg) following mpopt.py:546-570, 234-235, 257-258, 291-292, 323-324,
        368-369, 410-411, 459-460, 491-519."""
        o, N = self.ocp, self.N
        zmin, zmax, gmin, gmax = [], [], [], []
        for ph in range(self.n_ph):
            xmin = [np.asarray(o.lbx[ph], float) * self.sx] * N
            xmax = [np.asarray(o.ubx[ph], float) * self.sx] * N
            if ph == 0:
                xmin[0] = xmax[0] = np.asarray(o.x00[0], float) * self.sx
            zmin.append(np.concatenate([np.concatenate(np.array(xmin).T), np.repeat(np.asarray(o.lbu[ph], float) * self.su, N),
                                        np.asarray(o.lbt0[ph], float).ravel() * self.st, np.asarray(o.lbtf[ph], float).ravel() * self.st,
                                        np.asarray(o.lba[ph], float) * self.sa]))
            zmax.append(np.concatenate([np.concatenate(np.array(xmax).T), np.repeat(np.asarray(o.ubu[ph], float) * self.su, N),
                                        np.asarray(o.ubt0[ph], float).ravel() * self.st, np.asarray(o.ubtf[ph], float).ravel() * self.st,
                                        np.asarray(o.uba[ph], float) * self.sa]))
            lo, hi = [np.zeros(self.nx * N)], [np.zeros(self.nx * N)]
            if self.has_path[ph]:
                nc = len(o.get_path_constraints(ph)(o.x00[ph], o.u00[ph], o.t00[ph], o.a0[ph]))
                lo.append(np.full(nc * N, -np.inf)), hi.append(np.zeros(nc * N))
            if o.diff_u[ph]:
                lo.append(np.full(self.nu * N, float(o.lbdu[ph]))), hi.append(np.full(self.nu * N, float(o.ubdu[ph])))
            if self.midu[ph]:
                lo.append(np.repeat(np.asarray(o.lbu[ph], float) * self.su, N - 1))
                hi.append(np.repeat(np.asarray(o.ubu[ph], float) * self.su, N - 1))
            if self.S > 1 and o.du_continuity[ph]:
                lo.append(np.zeros(self.nu * (self.S - 1))), hi.append(np.zeros(self.nu * (self.S - 1)))
            if self.has_tc[ph]:
                ntc = len(o.get_terminal_constraints(ph)(o.xf0[ph], o.tf0[ph], o.x00[ph], o.t00[ph], o.a0[ph]))
                lo.append(np.zeros(ntc)), hi.append(np.zeros(ntc))
            gmin.append(np.concatenate(lo)), gmax.append(np.concatenate(hi))
        if self.n_ph > 1:
            n = len(o.phase_links)
            gmin += [np.concatenate([np.asarray(o.lbe[k], float) * self.sx for k in range(n)]), np.zeros(self.nu * n), np.zeros(n)]
            gmax += [np.concatenate([np.asarray(o.ube[k], float) * self.sx for k in range(n)]), np.zeros(self.nu * n), np.zeros(n)]
        return np.concatenate(zmin), np.concatenate(zmax), np.concatenate(gmin), np.concatenate(gmax)

    def initial_guess(self):  # mpopt.py:641-708
        o, N, out = self.ocp, self.N, []
        for ph in range(self.n_ph):
            x00, xf0 = np.asarray(o.x00[ph], float) * self.sx, np.asarray(o.xf0[ph], float) * self.sx
            u00, uf0 = np.asarray(o.u00[ph], float) * self.su, np.asarray(o.uf0[ph], float) * self.su
            t00, tf0 = np.asarray(o.t00[ph], float) * self.st, np.asarray(o.tf0[ph], float) * self.st
            a0 = np.asarray(o.a0[ph], float) * self.sa
            ts = np.linspace(t00, tf0, N)
            zx = np.concatenate(np.array([x00 + (xf0 - x00) / (tf0 - t00) * (t - t00) for t in ts]).T)
            zu = np.concatenate(np.array([u00 + (uf0 - u00) / (tf0 - t00) * (t - t00) for t in ts]))
            out.append(np.concatenate([zx, zu, t00, tf0, a0]))
        return np.concatenate(out)

    # -- derivatives: sympy per node, numpy assembly ------------------------------------------------
    def _build_symbolic(self):
        """Per-phase symbolic node functions in the NLP variables.  Mirrors what CasADi's AD sees:
        f_i = h_s*Sx*dyn(X_i/Sx, U_i/Su, t_i, A/Sa) with h_s, t_i functions of (t0, tf, w)."""
        sy, o = self.sympy, self.ocp
        nx, nu, na = self.nx, self.nu, self.na
        self.sym = []
        for ph in range(self.n_ph):
            X = sy.symbols(f"X0:{nx}", real=True)
            U = sy.symbols(f"U0:{nu}", real=True) if nu else ()
            A = sy.symbols(f"A0:{na}", real=True) if na else ()
            t0v, tfv, kap, th = sy.symbols("t0v tfv kap th", real=True)
            x = [X[a] / sy.Float(self.sx[a]) for a in range(nx)]
            u = [U[b] / sy.Float(self.su[b]) for b in range(nu)]
            a_ = [A[c] / sy.Float(self.sa[c]) for c in range(na)]
            t0, tf = t0v / sy.Float(self.st), tfv / sy.Float(self.st)
            h = (tf - t0) * kap
            t = t0 + (tf - t0) * th
            dyn = [sy.sympify(v) for v in o.get_dynamics(ph)(x, u, t, a_)]
            fx = [h * sy.Float(self.sx[a]) * dyn[a] for a in range(nx)]
            c = [sy.sympify(v) for v in o.get_path_constraints(ph)(x, u, t, a_)] if self.has_path[ph] else []
            q = h * sy.sympify(o.get_running_costs(ph)(x, u, t, a_))
            v = list(X) + list(U) + [t0v, tfv] + list(A)
            args = v + [kap, th]
            XF = sy.symbols(f"XF0:{nx}", real=True)
            XI = sy.symbols(f"XI0:{nx}", real=True)
            xf = [XF[a] / sy.Float(self.sx[a]) for a in range(nx)]
            x0 = [XI[a] / sy.Float(self.sx[a]) for a in range(nx)]
            M = sy.sympify(o.get_terminal_costs(ph)(xf, tf, x0, t0, a_))
            TC = [sy.sympify(e) for e in o.get_terminal_constraints(ph)(xf, tf, x0, t0, a_)] if self.has_tc[ph] else []
            tvars = list(XF) + [tfv] + list(XI) + [t0v] + list(A)
            L = lambda exprs, ar: sy.lambdify(ar, _almost_everywhere(exprs), "numpy")
            d = dict(nc=len(c), ntc=len(TC), v=v, tvars=tvars,
                     vals=L(fx + c + [q], args),
                     jac=L([[sy.diff(e, s) for s in v] for e in fx + c + [q]], args),
                     jacw=L([[sy.diff(e, kap), sy.diff(e, th)] for e in fx + c + [q]], args),
                     hes=L([[[sy.diff(e, s1, s2) for s2 in v] for s1 in v] for e in fx + c + [q]], args),
                     tvals=L([M] + TC, tvars),
                     tjac=L([[sy.diff(e, s) for s in tvars] for e in [M] + TC], tvars),
                     thes=L([[[sy.diff(e, s1, s2) for s2 in tvars] for s1 in tvars] for e in [M] + TC], tvars))
            self.sym.append(d)

    def _node_args(self, z, p, ph):
        G = self.grid
        X, U, t0v, tfv, A = self.split(z, ph)
        w = np.asarray(p, float)[ph * self.S:(ph + 1) * self.S]
        wcum = np.concatenate([[0.0], np.cumsum(w)[:-1]])
        kap = w[self.seg] / (G.tau1 - G.tau0)
        tk = np.array([(G.taus[G.orders[s]][k] - G.tau0) / (G.tau1 - G.tau0) for s, k in zip(self.seg, self.pt)])
        th = wcum[self.seg] + w[self.seg] * tk
        ones = np.ones(self.N)
        return [X[:, a] for a in range(self.nx)] + [U[:, b] for b in range(self.nu)] + [t0v * ones, tfv * ones] + \
               [A[c] * ones for c in range(self.na)] + [kap, th]

    def _term_args(self, z, ph):
        X, U, t0v, tfv, A = self.split(z, ph)
        return list(X[-1]) + [tfv] + list(X[0]) + [t0v] + list(A)

    def _vcols(self, ph, i):
        return ([self.zidx(ph, "X", a, i) for a in range(self.nx)] + [self.zidx(ph, "U", b, i) for b in range(self.nu)]
                + [self.zidx(ph, "t0"), self.zidx(ph, "tf")] + [self.zidx(ph, "A", c) for c in range(self.na)])

    def _tcols(self, ph):
        N = self.N
        return ([self.zidx(ph, "X", a, N - 1) for a in range(self.nx)] + [self.zidx(ph, "tf")]
                + [self.zidx(ph, "X", a, 0) for a in range(self.nx)] + [self.zidx(ph, "t0")] + [self.zidx(ph, "A", c) for c in range(self.na)])

    def _bc(self, v):
        return np.broadcast_to(np.asarray(v, float), (self.N,))

    def row_offsets(self, ph_target=None):
        """Row offset of every block, per phase: dict with F, C, DU, mU, dU, TC, and 'events'."""
        o, N, r, out = self.ocp, self.N, 0, []
        for ph in range(self.n_ph):
            d = {"F": r}
            r += self.nx * N
            d["C"] = r
            r += self.sym[ph]["nc"] * N
            d["DU"] = r
            r += self.nu * N if o.diff_u[ph] else 0
            d["mU"] = r
            r += self.nu * (N - 1) if self.midu[ph] else 0
            d["dU"] = r
            r += self.nu * (self.S - 1) if (self.S > 1 and o.du_continuity[ph]) else 0
            d["TC"] = r
            r += self.sym[ph]["ntc"]
            out.append(d)
        return out, r

    def jac_g(self, z, p):
        """Sparse (n_g x n_z) Jacobian of g; explicit zeros are not stored."""
        o, N, nx, nu = self.ocp, self.N, self.nx, self.nu
        offs, ev_row = self.row_offsets()
        R, C, V = [], [], []

        def add(r, c, v):
            R.append(np.asarray(r).ravel()), C.append(np.asarray(c).ravel()), V.append(np.asarray(v, float).ravel())

        nodes = np.arange(N)
        for ph in range(self.n_ph):
            d, off = self.sym[ph], offs[ph]
            J = d["jac"](*self._node_args(z, p, ph))
            Dr, Dc = np.nonzero(self.compD)
            for a in range(nx):
                add(off["F"] + a * N + Dr, self.zidx(ph, "X", a, 0) + Dc, self.compD[Dr, Dc])
                for k, col in enumerate(zip(*[self._vcols(ph, i) for i in range(N)])):
                    add(off["F"] + a * N + nodes, np.array(col), -self._bc(J[a][k]))
            for j in range(d["nc"]):
                for k, col in enumerate(zip(*[self._vcols(ph, i) for i in range(N)])):
                    add(off["C"] + j * N + nodes, np.array(col), self._bc(J[nx + j][k]))
            if o.diff_u[ph]:
                for b in range(nu):
                    add(off["DU"] + b * N + Dr, self.zidx(ph, "U", b, 0) + Dc, self.compD[Dr, Dc])
            if self.midu[ph]:
                Ir, Ic = np.nonzero(self.I_mid)
                for b in range(nu):
                    add(off["mU"] + b * (N - 1) + Ir, self.zidx(ph, "U", b, 0) + Ic, self.I_mid[Ir, Ic])
            if self.S > 1 and o.du_continuity[ph]:
                Cr, Cc = np.nonzero(self.D_cont)
                for b in range(nu):
                    add(off["dU"] + b * (self.S - 1) + Cr, self.zidx(ph, "U", b, 0) + Cc, self.D_cont[Cr, Cc])
            if d["ntc"]:
                TJ = d["tjac"](*self._term_args(z, ph))
                for j in range(d["ntc"]):
                    add(off["TC"] + j + np.zeros(len(TJ[1 + j]), int), np.array(self._tcols(ph)), np.array(TJ[1 + j], float))
        if self.n_ph > 1:
            r = ev_row
            for kind, cnt in (("X", nx), ("U", nu)):
                for (i, j) in o.phase_links:
                    for a in range(cnt):
                        add([r, r], [self.zidx(j, kind, a, 0), self.zidx(i, kind, a, N - 1)], [1.0, -1.0])
                        r += 1
            for (i, j) in o.phase_links:
                add([r, r], [self.zidx(j, "t0"), self.zidx(i, "tf")], [1.0, -1.0])
                r += 1
        M = sp.coo_matrix((np.concatenate(V), (np.concatenate(R), np.concatenate(C))), shape=(self.n_g, self.n_z)).tocsr()
        M.eliminate_zeros()
        return M

    def grad_f(self, z, p):
        g = np.zeros(self.n_z)
        for ph in range(self.n_ph):
            d = self.sym[ph]
            J = d["jac"](*self._node_args(z, p, ph))
            dq = J[self.nx + d["nc"]]
            for k, col in enumerate(zip(*[self._vcols(ph, i) for i in range(self.N)])):
                np.add.at(g, np.array(col), self.compW * self._bc(dq[k]))
            TJ = d["tjac"](*self._term_args(z, ph))
            np.add.at(g, np.array(self._tcols(ph)), np.array(TJ[0], float))
        return g

    def grad_gamma(self, z, p, sigma, lam):
        """``nlp_grad`` (derived by ca.nlpsol at mpopt.py:757): (d gamma / d x, d gamma / d p) of gamma = sigma*f + lam^T g.
        x part: sigma * grad_f + jac_g^T lam through the full sparse Jacobian (the product never forms it on this path).
        p part: a width w_s enters node i through kap_i = w_s / (tau1 - tau0) (i in segment s: h_s, mpopt.py:184) and through
        th_i = sum_{r<s(i)} w_r + w_s(i) * tk_i (the running t_seg0 and the node's own offset, mpopt.py:192-198)."""
        lam = np.asarray(lam, float)
        ggx = sigma * self.grad_f(z, p) + self.jac_g(z, p).T @ lam
        G, N, nx = self.grid, self.N, self.nx
        offs, _ = self.row_offsets()
        ggp = np.zeros(self.n_p)
        tk = np.array([(G.taus[G.orders[s]][k] - G.tau0) / (G.tau1 - G.tau0) for s, k in zip(self.seg, self.pt)])
        for ph in range(self.n_ph):
            d, off = self.sym[ph], offs[ph]
            Jw = d["jacw"](*self._node_args(z, p, ph))
            wts = [-lam[off["F"] + a * N:off["F"] + (a + 1) * N] for a in range(nx)] + \
                  [lam[off["C"] + j * N:off["C"] + (j + 1) * N] for j in range(d["nc"])] + [sigma * self.compW]
            gk = sum(wt * self._bc(Jw[e][0]) for e, wt in enumerate(wts))   # d gamma_i / d kap_i
            gth = sum(wt * self._bc(Jw[e][1]) for e, wt in enumerate(wts))  # d gamma_i / d th_i
            for s in range(self.S):
                own = self.seg == s
                ggp[ph * self.S + s] = (gk[own] / (G.tau1 - G.tau0) + gth[own] * tk[own]).sum() + gth[self.seg > s].sum()
        return ggx, ggp

    def hess_l(self, z, p, sigma, lam):
        """Dense symmetric Hessian of sigma*f + lam^T g."""
        N, nx = self.N, self.nx
        offs, _ = self.row_offsets()
        H = np.zeros((self.n_z, self.n_z))
        lam = np.asarray(lam, float)
        for ph in range(self.n_ph):
            d, off = self.sym[ph], offs[ph]
            Hs = d["hes"](*self._node_args(z, p, ph))
            nv = len(d["v"])
            cols = np.array([self._vcols(ph, i) for i in range(N)])  # N x nv
            wts = [-lam[off["F"] + a * N:off["F"] + (a + 1) * N] for a in range(nx)] + \
                  [lam[off["C"] + j * N:off["C"] + (j + 1) * N] for j in range(d["nc"])] + [sigma * self.compW]
            for e, wt in enumerate(wts):
                for k1 in range(nv):
                    for k2 in range(nv):
                        np.add.at(H, (cols[:, k1], cols[:, k2]), wt * self._bc(Hs[e][k1][k2]))
            TH = d["thes"](*self._term_args(z, ph))
            tc = np.array(self._tcols(ph))
            twts = [sigma] + [lam[off["TC"] + j] for j in range(d["ntc"])]
            for e, wt in enumerate(twts):
                H[np.ix_(tc, tc)] += wt * np.array(TH[e], float)
        return H


# ---------------------------------------------------------------------------------------------
# mpopt_adaptive (mpopt.py:2877-3375): segment widths are decision variables, no NLP parameters
# ---------------------------------------------------------------------------------------------
class OracleAdaptiveNLP(OracleNLP):
    """f, g and derivatives of the NLP that ``mpopt_adaptive.create_nlp`` builds (``p`` dropped,
    mpopt.py:3190-3192).  Values are a line-by-line restatement that runs on floats *or* on sympy
    symbols; derivatives are sympy differentiation of the whole-NLP expressions (what CasADi's AD does
    to the reference's SX graph) -- an independent route from the product's per-point AD + sparse
    assembly.  Meant for the small grids the adaptive variant is used with."""

    SEG_WIDTH_MIN, SEG_WIDTH_MAX, TOL_RESIDUAL = 1e-4, 1.0, 1e-3  # mpopt.py:2896-2898

    def __init__(self, ocp, n_segments, poly_orders, scheme="LGR", tau0=-1.0, tau1=1.0, table_method="auto", mid_residuals=True):
        self._lam_cache = {}
        self.mid_residuals = bool(mid_residuals)  # mpopt.py:2920, 3087
        super().__init__(ocp, n_segments, poly_orders, scheme, tau0, tau1, table_method)
        self.n_p = 0

    def _extra_vars_per_phase(self):
        return self.S  # mpopt.py:2947

    def _build_symbolic(self):  # derivatives are taken on the whole NLP (see _symbolic)
        G = self.grid
        self.taus_mid = [(G.taus[d][:-1] + G.taus[d][1:]) / 2.0 for d in G.orders]  # mpopt.py:3043-3048
        self.I_mid_all = G.comp_interp(self.taus_mid, 0)  # mpopt.py:3057-3059
        self.D_mid_all = G.comp_interp(self.taus_mid, 1)  # mpopt.py:3060-3062
        self.u_bounded = [bool((np.asarray(self.ocp.lbu[ph]) > -np.inf).any() or (np.asarray(self.ocp.ubu[ph]) < np.inf).any())
                          for ph in range(self.n_ph)]
        self.x_bounded = [bool((np.asarray(self.ocp.lbx[ph]) > -np.inf).any() or (np.asarray(self.ocp.ubx[ph]) < np.inf).any())
                          for ph in range(self.n_ph)]

    # -- values (generic in the number type) ----------------------------------------------------------
    def _phase_generic(self, zp, ph):
        o, G, N, nx, nu, na, S = self.ocp, self.grid, self.N, self.nx, self.nu, self.na, self.S
        zp = np.asarray(zp, dtype=object)
        X = zp[:nx * N].reshape(nx, N).T
        U = zp[nx * N:(nx + nu) * N].reshape(nu, N).T
        k = (nx + nu) * N
        t0v, tfv, A, W = zp[k], zp[k + 1], zp[k + 2:k + 2 + na], zp[k + 2 + na:k + 2 + na + S]
        t0, tf = t0v / self.st, tfv / self.st  # mpopt.py:175-176
        a = [A[c] / self.sa[c] for c in range(na)]
        dyn, pc, rc = o.get_dynamics(ph), o.get_path_constraints(ph), o.get_running_costs(ph)
        dtau = G.tau1 - G.tau0
        # node loop, mpopt.py:180-206
        f, c, q, tgrid = [], [], [], []
        t_seg0, s, h = t0, 0, (tf - t0) / dtau * W[0]
        for i in range(N):
            if self.seg[i] != s:
                s = self.seg[i]
                t_seg0 = t_seg0 + h * dtau
                h = (tf - t0) / dtau * W[s]
            t = t_seg0 + h * (G.taus[G.orders[s]][self.pt[i]] - G.tau0)
            tgrid.append(t)
            x = [X[i, b] / self.sx[b] for b in range(nx)]
            u = [U[i, b] / self.su[b] for b in range(nu)]
            d = list(dyn(x, u, t, a))
            f.append([h * (self.sx[b] * d[b]) for b in range(nx)])
            if self.has_path[ph]:
                c.append(list(pc(x, u, t, a)))
            q.append(h * rc(x, u, t, a))
        f = np.array(f, dtype=object).reshape(N, nx)
        D = self.compD.astype(object)
        parts = [(D.dot(X) - f).T.ravel()]  # F, mpopt.py:227-232
        if self.has_path[ph]:
            parts.append(np.array(c, dtype=object).reshape(N, -1).T.ravel())
        if o.diff_u[ph]:
            parts.append(D.dot(U).T.ravel())  # mpopt.py:315-321
        x0 = [X[0, b] / self.sx[b] for b in range(nx)]
        xf = [X[N - 1, b] / self.sx[b] for b in range(nx)]
        if self.has_tc[ph]:
            parts.append(np.array(list(o.get_terminal_constraints(ph)(xf, tf, x0, t0, a)), dtype=object))
        J = o.get_terminal_costs(ph)(xf, tf, x0, t0, a)
        for i in range(N):
            if self.compW[i] != 0.0:
                J = J + self.compW[i] * q[i]
        # --- widths block, mpopt.py:3034-3136
        sw = [np.array([sum(W[1:], W[0]) - 1.0], dtype=object)]
        Im, Dm = self.I_mid_all.astype(object), self.D_mid_all.astype(object)
        xi, ui, dxi = Im.dot(X), Im.dot(U), Dm.dot(X)
        ti = [(tgrid[i] + tgrid[i + 1]) / 2.0 for i in range(N - 1)]
        if self.u_bounded[ph]:
            sw.append(ui.T.ravel())
        if self.x_bounded[ph]:
            sw.append(xi.T.ravel())
        idx = 0
        for s in range(S if self.mid_residuals else 0):
            h_seg = (tfv - t0v) / self.st / dtau * W[s]
            n_mid = len(self.taus_mid[s])
            if n_mid == 0:
                continue
            blk = []
            for m in range(n_mid):
                x = [xi[idx, b] / self.sx[b] for b in range(nx)]
                u = [ui[idx, b] / self.su[b] for b in range(nu)]
                d = list(dyn(x, u, ti[idx], a))
                blk.append([W[s] * (dxi[idx, b] - h_seg * (self.sx[b] * d[b])) for b in range(nx)])
                idx += 1
            sw.append(np.array(blk, dtype=object).reshape(n_mid, nx).T.ravel())
        parts.append(np.concatenate(sw))
        return np.concatenate(parts), J

    def _nlp_generic(self, z):
        z = np.asarray(z, dtype=object)
        gs, f = [], 0
        for ph in range(self.n_ph):
            gp, J = self._phase_generic(z[ph * self.n_zp:(ph + 1) * self.n_zp], ph)
            gs.append(gp)
            f = f + J
        if self.n_ph > 1:  # events, mpopt.py:464-521
            N, ex, eu, et = self.N, [], [], []
            for (i, j) in self.ocp.phase_links:
                for a in range(self.nx):
                    ex.append(z[self.zidx(j, "X", a, 0)] - z[self.zidx(i, "X", a, N - 1)])
                for b in range(self.nu):
                    eu.append(z[self.zidx(j, "U", b, 0)] - z[self.zidx(i, "U", b, N - 1)])
                et.append(z[self.zidx(j, "t0")] - z[self.zidx(i, "tf")])
            gs.append(np.array(ex + eu + et, dtype=object))
        return np.concatenate(gs), f

    def g(self, z, p=None):
        return np.array([float(v) for v in self._nlp_generic([float(v) for v in z])[0]])

    def f(self, z, p=None):
        return float(self._nlp_generic([float(v) for v in z])[1])

    # -- bounds and initial guess (mpopt.py:2927-3032, 3034-3136, 3138-3174) ----------------------------
    def bounds(self):
        o, N = self.ocp, self.N
        zmin, zmax, gmin, gmax = [], [], [], []
        base_nzp = N * (self.nx + self.nu) + 2 + self.na
        blx, bux, _, _ = OracleNLP.bounds(self._as_base_for_bounds())  # per-phase variable bounds of the base class
        for ph in range(self.n_ph):
            zmin += [blx[ph * base_nzp:(ph + 1) * base_nzp], np.full(self.S, self.SEG_WIDTH_MIN)]
            zmax += [bux[ph * base_nzp:(ph + 1) * base_nzp], np.full(self.S, self.SEG_WIDTH_MAX)]
            lo, hi = [np.zeros(self.nx * N)], [np.zeros(self.nx * N)]
            if self.has_path[ph]:
                nc = len(o.get_path_constraints(ph)(o.x00[ph], o.u00[ph], o.t00[ph], o.a0[ph]))
                lo.append(np.full(nc * N, -np.inf)), hi.append(np.zeros(nc * N))
            if o.diff_u[ph]:
                lo.append(np.full(self.nu * N, float(o.lbdu[ph]))), hi.append(np.full(self.nu * N, float(o.ubdu[ph])))
            if self.has_tc[ph]:
                ntc = len(o.get_terminal_constraints(ph)(o.xf0[ph], o.tf0[ph], o.x00[ph], o.t00[ph], o.a0[ph]))
                lo.append(np.zeros(ntc)), hi.append(np.zeros(ntc))
            lo.append(np.zeros(1)), hi.append(np.zeros(1))  # sum of widths = 1
            n_mid = N - 1
            if self.u_bounded[ph]:
                lo.append(np.repeat(np.asarray(o.lbu[ph], float) * self.su, n_mid))
                hi.append(np.repeat(np.asarray(o.ubu[ph], float) * self.su, n_mid))
            if self.x_bounded[ph]:
                lo.append(np.repeat(np.asarray(o.lbx[ph], float) * self.sx, n_mid))
                hi.append(np.repeat(np.asarray(o.ubx[ph], float) * self.sx, n_mid))
            if self.mid_residuals:
                lo.append(np.full(self.nx * n_mid, -self.TOL_RESIDUAL)), hi.append(np.full(self.nx * n_mid, self.TOL_RESIDUAL))
            gmin.append(np.concatenate(lo)), gmax.append(np.concatenate(hi))
        if self.n_ph > 1:
            n = len(o.phase_links)
            gmin += [np.concatenate([np.asarray(o.lbe[k], float) * self.sx for k in range(n)]), np.zeros(self.nu * n), np.zeros(n)]
            gmax += [np.concatenate([np.asarray(o.ube[k], float) * self.sx for k in range(n)]), np.zeros(self.nu * n), np.zeros(n)]
        return np.concatenate(zmin), np.concatenate(zmax), np.concatenate(gmin), np.concatenate(gmax)

    def _as_base_for_bounds(self):
        """A shallow stand-in exposing only what OracleNLP.bounds reads for the per-phase *variable* bounds."""
        class _B:
            pass

        b = _B()
        for k in ("ocp", "N", "n_ph", "nx", "nu", "sx", "su", "sa", "st", "has_path", "has_tc", "S"):
            setattr(b, k, getattr(self, k))
        b.midu = [False] * self.n_ph
        return b

    def initial_guess(self):  # mpopt.py:2981-3032
        base = OracleNLP.initial_guess(self)
        n0 = self.N * (self.nx + self.nu) + 2 + self.na
        return np.concatenate([np.concatenate([base[ph * n0:(ph + 1) * n0], np.full(self.S, 1.0 / self.S)]) for ph in range(self.n_ph)])

    # -- derivatives: sympy on the whole NLP ----------------------------------------------------------
    def _symbolic(self):
        if "syms" not in self._lam_cache:
            sy = self.sympy
            zs = list(sy.symbols(f"z0:{self.n_z}", real=True))
            g, f = self._nlp_generic(zs)
            self._lam_cache.update(syms=zs, g=[sy.sympify(e) for e in g], f=sy.sympify(f), pos={s: i for i, s in enumerate(zs)})
        return self._lam_cache

    def jac_g(self, z, p=None):
        sy, C = self.sympy, self._symbolic()
        if "jac" not in C:
            R, Cc, E = [], [], []
            for i, e in enumerate(C["g"]):
                for s in e.free_symbols:
                    d = sy.diff(e, s)
                    if d != 0:
                        R.append(i), Cc.append(C["pos"][s]), E.append(d)
            C["jac"] = (np.array(R), np.array(Cc), sy.lambdify(C["syms"], E, "math", cse=True))
        R, Cc, fn = C["jac"]
        M = sp.coo_matrix((np.array(fn(*[float(v) for v in z]), float), (R, Cc)), shape=(self.n_g, self.n_z)).tocsr()
        M.eliminate_zeros()
        return M

    def grad_f(self, z, p=None):
        sy, C = self.sympy, self._symbolic()
        if "grad" not in C:
            C["grad"] = sy.lambdify(C["syms"], [sy.diff(C["f"], s) for s in C["syms"]], "math", cse=True)
        return np.array(C["grad"](*[float(v) for v in z]), float)

    def grad_gamma(self, z, p, sigma, lam):
        """(d gamma / d x, empty): the adaptive NLP has no parameters (mpopt.py:3190-3192)."""
        return sigma * self.grad_f(z) + self.jac_g(z).T @ np.asarray(lam, float), np.zeros(0)

    def hess_l(self, z, p, sigma, lam):
        """Dense symmetric Hessian of sigma*f + lam^T g (lam, sigma enter as symbols: one lambdify)."""
        sy, C = self.sympy, self._symbolic()
        if "hess" not in C:
            ls = list(sy.symbols(f"lam0:{self.n_g}", real=True))
            sg = sy.Symbol("sigma_f", real=True)
            lag = sg * C["f"] + sum(l * e for l, e in zip(ls, C["g"]))
            R, Cc, E = [], [], []
            for s in C["syms"]:
                d1 = sy.diff(lag, s)
                if d1 == 0:
                    continue
                for s2 in C["syms"][C["pos"][s]:]:
                    if s2 is not s and s2 not in d1.free_symbols:
                        continue
                    d2 = sy.diff(d1, s2)
                    if d2 != 0:
                        R.append(C["pos"][s]), Cc.append(C["pos"][s2]), E.append(d2)
            C["hess"] = (np.array(R, int), np.array(Cc, int), sy.lambdify(C["syms"] + ls + [sg], E, "math", cse=True))
        R, Cc, fn = C["hess"]
        H = np.zeros((self.n_z, self.n_z))
        if len(R):
            H[R, Cc] = np.array(fn(*[float(v) for v in z], *[float(v) for v in lam], float(sigma)), float)
        return H + np.triu(H, 1).T

    # -- derivatives: exact operator-overloading AD of the value code above (oracle/sparse_ad.py) -----------------------------------
    # The same quantities as jac_g / grad_f / hess_l, for grids where sympy on the whole NLP is not practical (the bench size
    # 20 x 5, 100 x 3, ...): _nlp_generic runs on sparse hyper-dual numbers.  Equal to the sympy route to rounding on the five golden
    # cases (tests/test_oracle.py).
    def _ad_pass(self, z, order):
        from .sparse_ad import SD

        old, SD.ORDER = SD.ORDER, order
        try:
            return self._nlp_generic([SD.var(float(v), i) for i, v in enumerate(z)])
        finally:
            SD.ORDER = old

    def ad_first(self, z):
        """-> (f, g, grad_f dense, jac_g CSR) from ONE first-order pass."""
        from .sparse_ad import gradient, value

        g, f = self._ad_pass(z, 1)
        grad = np.zeros(self.n_z)
        for k, v in gradient(f).items():
            grad[k] = v
        R, Cc, V = [], [], []
        for r, e in enumerate(g):
            for k, v in gradient(e).items():
                R.append(r), Cc.append(k), V.append(v)
        J = sp.coo_matrix((np.array(V, float), (np.array(R, int), np.array(Cc, int))), shape=(self.n_g, self.n_z)).tocsr()
        return value(f), np.array([value(e) for e in g]), grad, J

    def ad_hess_l(self, z, sigma, lam):
        """Upper triangle (CSR) of the Hessian of sigma f + lam^T g from one second-order pass."""
        from .sparse_ad import SD, hessian

        g, f = self._ad_pass(z, 2)
        acc = {}
        for wt, e in [(float(sigma), f)] + [(float(l), e) for l, e in zip(lam, g)]:
            if wt == 0.0 or not isinstance(e, SD):
                continue
            for k, v in hessian(e).items():
                acc[k] = acc.get(k, 0.0) + wt * v
        keys = list(acc)
        return sp.coo_matrix((np.array([acc[k] for k in keys], float), (np.array([k[0] for k in keys], int), np.array([k[1] for k in keys], int))),
                             shape=(self.n_z, self.n_z)).tocsr()

"""Transcription with the segment widths as decision variables (the reference's ``mpopt_adaptive``,
mpopt.py:2877-3375) expressed as point functions + sparse maps for ``mpopt_amd.assembly``.

Per phase (mpopt.py:2927-2979, 3138-3174):

    Z = [vec(X); vec(U); t0; tf; A; W]                       (state-major, W = the S segment widths)
    G = [F; C; DU; TC; SW],   SW = [sum(W) - 1; U at mid-points (if u is bounded); X at mid-points (if x is
                                    bounded); per segment  W_s * (D_mid X - h_s Sx dyn(I_mid X, I_mid U, t_mid, a))]

and the event rows of multi-phase problems after the last phase (mpopt.py:464-521).  The NLP has no
parameters (mpopt.py:3190-3192).  Three point functions per phase carry everything non-linear:

    node     loc = [X_i, U_i, t0, tf, A, w_s, sum_{r<s} w_r]        out = [h Sx dyn, path, h L]
    mid      loc = [I_mid X, I_mid U, D_mid X, t0, tf, A, w_s, sum_{r<s} w_r]
                                                                   out = [w_s (D_mid X - h Sx dyn)]
    end      loc = [X_N, tf, X_0, t0, A]                           out = [Mayer cost, terminal constraints]

Because the widths enter the node times through the running sum, time-dependent callables couple a row
to every earlier width; the assembly expands that fan-out on the host, the kernels stay per-point.
"""
import numpy as np
import scipy.sparse as sp

from .assembly import AssembledNlpFunctions, PointFunction, PointSet
from .codegen import _as_list
from .expr import symvec


class AdaptiveLayout:
    """Index arithmetic of the decision vector and the constraint rows."""

    def __init__(self, ocp, n_segments, poly_orders, nc, ntc, u_bounded, x_bounded, mid_residuals):
        o = ocp
        self.nx, self.nu, self.na, self.n_ph = o.nx, o.nu, o.na, o.n_phases
        self.S, self.orders = int(n_segments), [int(d) for d in poly_orders]
        self.N = sum(self.orders) + 1
        self.start = np.concatenate([[0], np.cumsum(self.orders)]).astype(int)  # first node of every segment
        self.n_zp = self.N * (self.nx + self.nu) + 2 + self.na + self.S
        self.n_z = self.n_zp * self.n_ph
        N, r = self.N, 0
        self.rows = []
        for ph in range(self.n_ph):
            d = {"F": r}
            r += self.nx * N
            d["C"] = r
            r += nc[ph] * N
            d["DU"] = r
            r += self.nu * N if o.diff_u[ph] else 0
            d["TC"] = r
            r += ntc[ph]
            d["sum"] = r
            r += 1
            d["mU"] = r
            r += self.nu * (N - 1) if u_bounded[ph] else 0
            d["mX"] = r
            r += self.nx * (N - 1) if x_bounded[ph] else 0
            d["res"] = r
            r += self.nx * (N - 1) if mid_residuals else 0
            self.rows.append(d)
        self.row_events = r
        if self.n_ph > 1:
            r += len(o.phase_links) * (self.nx + self.nu + 1)
        self.n_g = r

    def X(self, ph, a, i):
        return ph * self.n_zp + a * self.N + i

    def U(self, ph, b, i):
        return ph * self.n_zp + (self.nx + b) * self.N + i

    def t0(self, ph):
        return ph * self.n_zp + (self.nx + self.nu) * self.N

    def tf(self, ph):
        return self.t0(ph) + 1

    def A(self, ph, c):
        return self.t0(ph) + 2 + c

    def W(self, ph, s):
        return self.t0(ph) + 2 + self.na + s


def _bounded(lo, hi):
    return bool((np.asarray(lo, float) > -np.inf).any() or (np.asarray(hi, float) < np.inf).any())


def _trace_sizes(ocp):
    """Number of path / terminal constraint rows per phase (one throw-away call of the user functions)."""
    nc, ntc = [], []
    for ph in range(ocp.n_phases):
        nc.append(len(_as_list(ocp.get_path_constraints(ph)(ocp.x00[ph], ocp.u00[ph], ocp.t00[ph], ocp.a0[ph]))) if ocp.has_path_constraints(ph) else 0)
        ntc.append(len(_as_list(ocp.get_terminal_constraints(ph)(ocp.xf0[ph], ocp.tf0[ph], ocp.x00[ph], ocp.t00[ph], ocp.a0[ph])))
                   if ocp.has_terminal_constraints(ph) else 0)
    return nc, ntc


def build_adaptive_oracle(ocp, n_segments, poly_orders, collocation, mid_residuals=True, device=0, with_device=None, verbose=False):
    """-> (AssembledNlpFunctions, AdaptiveLayout).  ``collocation``: the product's ``Collocation`` (native tables)."""
    o = ocp
    nx, nu, na = o.nx, o.nu, o.na
    sx = [float(v) for v in np.asarray(o.scale_x, float).reshape(-1)]
    su = [float(v) for v in np.asarray(o.scale_u, float).reshape(-1)]
    sa = [float(v) for v in np.asarray(o.scale_a, float).reshape(-1)]
    st = float(o.scale_t)
    tau0, tau1 = float(collocation.tau0), float(collocation.tau1)
    inv_dtau = 1.0 / (tau1 - tau0)
    orders = [int(d) for d in poly_orders]
    nc, ntc = _trace_sizes(o)
    u_b = [_bounded(o.lbu[ph], o.ubu[ph]) for ph in range(o.n_phases)]
    x_b = [_bounded(o.lbx[ph], o.ubx[ph]) for ph in range(o.n_phases)]
    lay = AdaptiveLayout(o, n_segments, orders, nc, ntc, u_b, x_b, mid_residuals)
    N, S, n_z, n_g = lay.N, lay.S, lay.n_z, lay.n_g
    # grid tables (mpopt.py:4015-4131; mid-points mpopt.py:3043-3062)
    roots = {d: np.asarray(collocation._taus_fn(d), float) for d in set(orders)}
    compD = sp.coo_matrix(np.asarray(collocation.get_composite_differentiation_matrix(orders)))
    compW = np.asarray(collocation.get_composite_quadrature_weights(orders)).ravel()
    taus_mid = [(roots[d][:-1] + roots[d][1:]) / 2.0 for d in orders]
    I_mid = sp.csr_matrix(collocation.get_composite_interpolation_matrix(taus_mid, orders))
    D_mid = sp.csr_matrix(collocation.get_composite_interpolation_Dmatrix_at(taus_mid, orders))
    seg = np.concatenate([[0]] + [np.full(d, s) for s, d in enumerate(orders)]).astype(int)  # owner segment of a node
    pt = np.concatenate([[0]] + [np.arange(1, d + 1) for d in orders]).astype(int)           # its point inside the segment
    tk = np.array([(roots[orders[s]][k] - tau0) * inv_dtau for s, k in zip(seg, pt)])
    mseg = np.concatenate([np.full(d, s) for s, d in enumerate(orders)]).astype(int)          # segment of a mid-point
    tmid = np.concatenate([(taus_mid[s] - tau0) * inv_dtau for s in range(S)])
    n_mid = N - 1

    Gz_r, Gz_c, Gz_v = [], [], []
    g0 = np.zeros(n_g)

    def lin(rows, cols, vals):
        Gz_r.append(np.asarray(rows, np.int64).ravel()), Gz_c.append(np.asarray(cols, np.int64).ravel()), Gz_v.append(np.asarray(vals, float).ravel())

    sets = []
    for ph in range(o.n_phases):
        R = lay.rows[ph]
        dyn_f, path_f, cost_f = o.get_dynamics(ph), o.get_path_constraints(ph), o.get_running_costs(ph)

        def scaled(loc_x, loc_u, loc_a):
            return (symvec([loc_x[a] * (1.0 / sx[a]) for a in range(nx)]), symvec([loc_u[b] * (1.0 / su[b]) for b in range(nu)]),
                    symvec([loc_a[c] * (1.0 / sa[c]) for c in range(na)]))

        # ---- node function (mpopt.py:175-206) ---------------------------------------------------
        n_loc = nx + nu + 2 + na + 2
        iT0, iTF, iA, iWS, iWC = nx + nu, nx + nu + 1, nx + nu + 2, nx + nu + 2 + na, nx + nu + 3 + na

        def node_build(loc, cst, ph=ph):
            x, u, a_ = scaled(loc[:nx], loc[nx:nx + nu], loc[iA:iA + na])
            t0, tf = loc[iT0] / st, loc[iTF] / st
            h = (tf - t0) * (loc[iWS] * cst[1])
            t = t0 + (tf - t0) * (loc[iWC] + loc[iWS] * cst[0])
            dyn = _as_list(dyn_f(x, u, t, a_))
            if len(dyn) != nx:
                raise ValueError(f"phase {ph}: dynamics returned {len(dyn)} values for {nx} states")
            out = [h * (sx[a] * dyn[a]) for a in range(nx)]
            if nc[ph]:
                out += _as_list(path_f(x, u, t, a_))
            return out + [h * cost_f(x, u, t, a_)]

        fn = PointFunction(n_loc, 2, node_build)
        lr, lc, lv = [], [], []

        def sel(p, v, col, coef=1.0):
            lr.append(p * n_loc + v), lc.append(col), lv.append(coef)

        for i in range(N):
            for a in range(nx):
                sel(i, a, lay.X(ph, a, i))
            for b in range(nu):
                sel(i, nx + b, lay.U(ph, b, i))
            sel(i, iT0, lay.t0(ph)), sel(i, iTF, lay.tf(ph))
            for c in range(na):
                sel(i, iA + c, lay.A(ph, c))
            sel(i, iWS, lay.W(ph, seg[i]))
            for r in range(seg[i]):  # running sum of the earlier widths (mpopt.py:186-195)
                sel(i, iWC, lay.W(ph, r))
        L = sp.coo_matrix((lv, (lr, lc)), shape=(N * n_loc, n_z))
        gr, gc_, gv = [], [], []
        for i in range(N):
            for a in range(nx):  # F = D X - f  (mpopt.py:227-232)
                gr.append(R["F"] + a * N + i), gc_.append(i * fn.n_out + a), gv.append(-1.0)
            for j in range(nc[ph]):
                gr.append(R["C"] + j * N + i), gc_.append(i * fn.n_out + nx + j), gv.append(1.0)
        G = sp.coo_matrix((gv, (gr, gc_)), shape=(n_g, N * fn.n_out))
        fw = np.zeros((N, fn.n_out))
        fw[:, -1] = compW  # J += compW . q  (mpopt.py:3163)
        sets.append(PointSet(fn, N, L, np.stack([tk, np.full(N, inv_dtau)], axis=1), G, fw))
        for a in range(nx):
            lin(R["F"] + a * N + compD.row, lay.X(ph, a, 0) + compD.col, compD.data)
        if o.diff_u[ph]:  # mpopt.py:315-321
            for b in range(nu):
                lin(R["DU"] + b * N + compD.row, lay.U(ph, b, 0) + compD.col, compD.data)

        # ---- end-point function (mpopt.py:277-298) ------------------------------------------------
        n_tl = 2 * nx + 2 + na

        def term_build(loc, cst, ph=ph):
            xf = symvec([loc[a] * (1.0 / sx[a]) for a in range(nx)])
            a_ = symvec([loc[2 * nx + 2 + c] * (1.0 / sa[c]) for c in range(na)])
            x0 = symvec([loc[nx + 1 + a] * (1.0 / sx[a]) for a in range(nx)])
            tf, t0 = loc[nx] / st, loc[2 * nx + 1] / st
            out = [o.get_terminal_costs(ph)(xf, tf, x0, t0, a_)]
            if ntc[ph]:
                out += _as_list(o.get_terminal_constraints(ph)(xf, tf, x0, t0, a_))
            return out

        tfn = PointFunction(n_tl, 0, term_build)
        cols = ([lay.X(ph, a, N - 1) for a in range(nx)] + [lay.tf(ph)] + [lay.X(ph, a, 0) for a in range(nx)] + [lay.t0(ph)]
                + [lay.A(ph, c) for c in range(na)])
        L = sp.coo_matrix((np.ones(n_tl), (np.arange(n_tl), cols)), shape=(n_tl, n_z))
        G = sp.coo_matrix((np.ones(ntc[ph]), (R["TC"] + np.arange(ntc[ph]), 1 + np.arange(ntc[ph]))), shape=(n_g, tfn.n_out))
        fw = np.zeros((1, tfn.n_out))
        fw[0, 0] = 1.0
        sets.append(PointSet(tfn, 1, L, np.zeros((1, 0)), G, fw))

        # ---- widths block (mpopt.py:3034-3136) --------------------------------------------------------
        lin(np.full(S, R["sum"]), [lay.W(ph, s) for s in range(S)], np.ones(S))
        g0[R["sum"]] = -1.0
        Im = I_mid.tocoo()
        if u_b[ph]:
            for b in range(nu):
                lin(R["mU"] + b * n_mid + Im.row, lay.U(ph, b, 0) + Im.col, Im.data)
        if x_b[ph]:
            for a in range(nx):
                lin(R["mX"] + a * n_mid + Im.row, lay.X(ph, a, 0) + Im.col, Im.data)
        if mid_residuals:
            m_loc = 2 * nx + nu + 2 + na + 2
            jD, jT0, jTF, jA = nx + nu, 2 * nx + nu, 2 * nx + nu + 1, 2 * nx + nu + 2
            jWS, jWC = jA + na, jA + na + 1

            def mid_build(loc, cst, ph=ph):
                x, u, a_ = scaled(loc[:nx], loc[nx:nx + nu], loc[jA:jA + na])
                t0, tf = loc[jT0] / st, loc[jTF] / st
                h = (tf - t0) * (loc[jWS] * cst[1])
                t = t0 + (tf - t0) * (loc[jWC] + loc[jWS] * cst[0])
                dyn = _as_list(dyn_f(x, u, t, a_))
                return [loc[jWS] * (loc[jD + a] - h * (sx[a] * dyn[a])) for a in range(nx)]

            mfn = PointFunction(m_loc, 2, mid_build)
            lr, lc, lv = [], [], []
            for m in range(n_mid):
                for mat, base in ((I_mid, 0), (D_mid, jD)):
                    lo, hi = mat.indptr[m], mat.indptr[m + 1]
                    for a in range(nx):
                        lr += [m * m_loc + base + a] * (hi - lo)
                        lc += list(lay.X(ph, a, 0) + mat.indices[lo:hi])
                        lv += list(mat.data[lo:hi])
                lo, hi = I_mid.indptr[m], I_mid.indptr[m + 1]
                for b in range(nu):
                    lr += [m * m_loc + nx + b] * (hi - lo)
                    lc += list(lay.U(ph, b, 0) + I_mid.indices[lo:hi])
                    lv += list(I_mid.data[lo:hi])
                lr += [m * m_loc + jT0, m * m_loc + jTF]
                lc += [lay.t0(ph), lay.tf(ph)]
                lv += [1.0, 1.0]
                for c in range(na):
                    lr.append(m * m_loc + jA + c), lc.append(lay.A(ph, c)), lv.append(1.0)
                lr.append(m * m_loc + jWS), lc.append(lay.W(ph, mseg[m])), lv.append(1.0)
                for r in range(mseg[m]):
                    lr.append(m * m_loc + jWC), lc.append(lay.W(ph, r)), lv.append(1.0)
            L = sp.coo_matrix((lv, (lr, lc)), shape=(n_mid * m_loc, n_z))
            # rows: per segment, state-major over the segment's mid-points (mpopt.py:3120-3124)
            gr, gc_ = [], []
            m0 = np.concatenate([[0], np.cumsum(orders)])
            for m in range(n_mid):
                s = mseg[m]
                for a in range(nx):
                    gr.append(R["res"] + nx * m0[s] + a * orders[s] + (m - m0[s])), gc_.append(m * nx + a)
            G = sp.coo_matrix((np.ones(len(gr)), (gr, gc_)), shape=(n_g, n_mid * nx))
            sets.append(PointSet(mfn, n_mid, L, np.stack([tmid, np.full(n_mid, inv_dtau)], axis=1), G, np.zeros((n_mid, nx))))

    if o.n_phases > 1:  # events (mpopt.py:464-521): x, u, t continuity across linked phases
        r = lay.row_events
        for kind, cnt in (("X", nx), ("U", nu)):
            for (i, j) in o.phase_links:
                for a in range(cnt):
                    f = lay.X if kind == "X" else lay.U
                    lin([r, r], [f(j, a, 0), f(i, a, N - 1)], [1.0, -1.0])
                    r += 1
        for (i, j) in o.phase_links:
            lin([r, r], [lay.t0(j), lay.tf(i)], [1.0, -1.0])
            r += 1
    Gz = sp.coo_matrix((np.concatenate(Gz_v), (np.concatenate(Gz_r), np.concatenate(Gz_c))), shape=(n_g, n_z))
    orc = AssembledNlpFunctions(n_z, n_g, sets, Gz, g0, device=device, with_device=with_device, verbose=verbose)
    return orc, lay

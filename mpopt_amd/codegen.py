"""Per-problem code generation: OCP callables -> ``__device__`` node/terminal functions.

For every phase the user's ``dynamics / path_constraints / running_costs`` are traced exactly
as the reference evaluates them at a collocation node (mpopt.py:175-206):

    x = X/scale_x, u = U/scale_u, a = A/scale_a, t0 = t0_var/scale_t, tf = tf_var/scale_t,
    h = (tf - t0)/(tau1 - tau0) * w_s,  t = t_seg0 + h (tau_k - tau0),
    f_i = h * scale_x * dyn(x,u,t,a),  c_i = path(x,u,t,a),  q_i = h * L(x,u,t,a)

but in terms of the *NLP variables* (scaled X, U, t0_var, tf_var, A) plus three per-node
constants (kap = w_s/(tau1-tau0), th = normalised node time, W = composite quadrature weight),
so that symbolic differentiation yields directly the entries of jac_g / grad_f / hess_l that the
reference obtains from CasADi's AD inside ``nlpsol`` (mpopt.py:757).  Terminal cost and terminal
constraints (mpopt.py:277-298) are traced the same way in (xf, tf, x0, t0, a).

Outputs: HIP source text (structs ``mpxgen::Phase<PH>``) and the structural description
(which derivative entries exist) that the C library turns into the COO patterns.
"""
import hashlib

import numpy as np

from .expr import Tracer, symvec

# column kinds of a node row / terminal variable kinds (must match mpx.h)
COL_X, COL_U, COL_T0, COL_TF, COL_A = 0, 1, 2, 3, 4
ROW_F, ROW_C = 0, 1
TV_XF, TV_TF, TV_X0, TV_T0, TV_A = 0, 1, 2, 3, 4


def _as_list(v):
    if v is None:
        return None
    if isinstance(v, (list, tuple)):
        return list(v)
    if isinstance(v, np.ndarray):
        return list(v.reshape(-1))
    return [v]


class PhaseProgram:
    """Traced node + terminal program of one phase."""

    def __init__(self, ocp, phase):
        self.phase = phase
        nx, nu, na = ocp.nx, ocp.nu, ocp.na
        self.nx, self.nu, self.na = nx, nu, na
        tr = self.tr = Tracer()
        sx = [float(v) for v in np.asarray(ocp.scale_x, dtype=float).reshape(-1)]
        su = [float(v) for v in np.asarray(ocp.scale_u, dtype=float).reshape(-1)]
        sa = [float(v) for v in np.asarray(ocp.scale_a, dtype=float).reshape(-1)]
        st = float(ocp.scale_t)
        # ---- node program ------------------------------------------------------------
        Xs = [tr.var(f"Xs[{a}]") for a in range(nx)]
        Us = [tr.var(f"Us[{b}]") for b in range(nu)]
        As = [tr.var(f"As[{c}]") for c in range(na)]
        t0v, tfv = tr.var("t0v"), tr.var("tfv")
        kap, th, W = tr.var("kap"), tr.var("th"), tr.var("W")
        x = symvec([Xs[a] * (1.0 / sx[a]) for a in range(nx)])
        u = symvec([Us[b] * (1.0 / su[b]) for b in range(nu)])
        a_ = symvec([As[c] * (1.0 / sa[c]) for c in range(na)])
        t0, tf = t0v / st, tfv / st
        dt = tf - t0
        h = dt * kap
        t = t0 + dt * th
        dyn = _as_list(ocp.get_dynamics(phase)(x, u, t, a_))
        if len(dyn) != nx:
            raise ValueError(f"phase {phase}: dynamics returned {len(dyn)} values for {nx} states")
        self.t_node = t
        self.fx = [h * (sx[a] * tr.wrap(dyn[a])) for a in range(nx)]
        if ocp.has_path_constraints(phase):
            self.c = [tr.wrap(v) for v in _as_list(ocp.get_path_constraints(phase)(x, u, t, a_))]
        else:
            self.c = []
        self.nc = len(self.c)
        L = tr.wrap(ocp.get_running_costs(phase)(x, u, t, a_))
        self.qW = W * (h * L)
        # does anything of a node depend on the node TIME (t = t0 + (tf - t0) th)?  If no phase does, the prefix sums of the segment
        # widths (th's only input besides the width itself) are never used, and libmpx skips the kernel that forms them
        self.time_dependent = any(e.depends_on({"th"}) for e in self.fx + self.c + [self.qW])
        self.node_vars = ([(COL_X, a, Xs[a]) for a in range(nx)] + [(COL_U, b, Us[b]) for b in range(nu)]
                          + [(COL_T0, 0, t0v), (COL_TF, 0, tfv)] + [(COL_A, c, As[c]) for c in range(na)])
        nv = self.node_vars
        # first derivatives
        self.dd = []            # d fx[a] / d Xs[a]  (merged with the D-block diagonal)
        self.jv = []            # (row_kind,row_comp,col_kind,col_comp, expr) structural entries
        rows = [(ROW_F, a, self.fx[a]) for a in range(nx)] + [(ROW_C, j, self.c[j]) for j in range(self.nc)]
        dmemo = {id(v[2]): {} for v in nv}
        for rk, rc, e in rows:
            for ck, cc, v in nv:
                d = tr.diff(e, v, dmemo[id(v)])
                if rk == ROW_F and ck == COL_X and cc == rc:
                    self.dd.append(d)
                    continue
                if not d.is_zero:
                    # defect row = D.X - f  (mpopt.py:232): the entry is -df/dv
                    self.jv.append((rk, rc, ck, cc, tr.neg(d) if rk == ROW_F else d))
        self.gn = [tr.diff(self.qW, v, dmemo[id(v)]) for ck, cc, v in nv if ck in (COL_X, COL_U)]
        self.gr = [self.qW] + [tr.diff(self.qW, v, dmemo[id(v)]) for ck, cc, v in nv if ck not in (COL_X, COL_U)]
        # second derivatives of the node Lagrangian
        sig = tr.var("sig")
        lF = [tr.var(f"lF[{a}]") for a in range(nx)]
        lC = [tr.var(f"lC[{j}]") for j in range(self.nc)]
        lag = sig * self.qW
        for a in range(nx):
            lag = lag - lF[a] * self.fx[a]
        for j in range(self.nc):
            lag = lag + lC[j] * self.c[j]
        self.hn, self.hc = [], []
        for i, (k1, c1, v1) in enumerate(nv):
            g1 = tr.diff(lag, v1, dmemo[id(v1)])
            if g1.is_zero:
                continue
            for (k2, c2, v2) in nv[i:]:
                d2 = tr.diff(g1, v2, dmemo[id(v2)])
                if d2.is_zero:
                    continue
                if k1 in (COL_X, COL_U):
                    self.hn.append((k1, c1, k2, c2, d2))
                else:
                    self.hc.append((k1, c1, k2, c2, d2))
        # first derivatives of the node Lagrangian: the nonlinear part of nlp_grad's grad_gamma_x (the D / interpolation blocks are
        # linear rows, contracted in the kernel) and, through kap = w_s / (tau1 - tau0) and th = sum_{r<s} w_r + w_s (tau_k - tau0) /
        # (tau1 - tau0) (mpopt.py:184-198), everything grad_gamma_p needs
        self.gl_x = [tr.diff(lag, v, dmemo[id(v)]) for ck, cc, v in nv if ck in (COL_X, COL_U)]
        self.gl_r = [tr.diff(lag, v, dmemo[id(v)]) for ck, cc, v in nv if ck not in (COL_X, COL_U)]
        self.gl_k, self.gl_th = tr.diff(lag, kap), tr.diff(lag, th)
        # ---- terminal program --------------------------------------------------------
        XF = [tr.var(f"XF[{a}]") for a in range(nx)]
        X0 = [tr.var(f"X0[{a}]") for a in range(nx)]
        xf = symvec([XF[a] * (1.0 / sx[a]) for a in range(nx)])
        x0 = symvec([X0[a] * (1.0 / sx[a]) for a in range(nx)])
        self.mayer = tr.wrap(ocp.get_terminal_costs(phase)(xf, tf, x0, t0, a_))
        if ocp.has_terminal_constraints(phase):
            self.tc = [tr.wrap(v) for v in _as_list(ocp.get_terminal_constraints(phase)(xf, tf, x0, t0, a_))]
        else:
            self.tc = []
        self.ntc = len(self.tc)
        self.term_vars = ([(TV_XF, a, XF[a]) for a in range(nx)] + [(TV_TF, 0, tfv)]
                          + [(TV_X0, a, X0[a]) for a in range(nx)] + [(TV_T0, 0, t0v)]
                          + [(TV_A, c, As[c]) for c in range(na)])
        tv = self.term_vars
        self.mg = [(k, c, tr.diff(self.mayer, v)) for k, c, v in tv]
        self.mg = [m for m in self.mg if not m[2].is_zero]
        self.tj = []
        for j, e in enumerate(self.tc):
            for k, c, v in tv:
                d = tr.diff(e, v)
                if not d.is_zero:
                    self.tj.append((j, k, c, d))
        lT = [tr.var(f"lT[{j}]") for j in range(self.ntc)]
        lagT = sig * self.mayer
        for j in range(self.ntc):
            lagT = lagT + lT[j] * self.tc[j]
        self.th = []
        for i, (k1, c1, v1) in enumerate(tv):
            g1 = tr.diff(lagT, v1)
            if g1.is_zero:
                continue
            for (k2, c2, v2) in tv[i:]:
                d2 = tr.diff(g1, v2)
                if not d2.is_zero:
                    self.th.append((k1, c1, k2, c2, d2))
        self.tg = [tr.diff(lagT, v) for k, c, v in tv]  # dense over (XF, tf, X0, t0, A): terminal part of grad_gamma_x

    # -- emission ----------------------------------------------------------------------
    def _names(self):
        n = {f"Xs[{a}]": f"Xs[{a}]" for a in range(self.nx)}
        n.update({f"Us[{b}]": f"Us[{b}]" for b in range(self.nu)})
        n.update({f"As[{c}]": f"As[{c}]" for c in range(self.na)})
        n.update({f"XF[{a}]": f"XF[{a}]" for a in range(self.nx)})
        n.update({f"X0[{a}]": f"X0[{a}]" for a in range(self.nx)})
        n.update({f"lF[{a}]": f"lF[{a}]" for a in range(self.nx)})
        n.update({f"lC[{j}]": f"lC[{j}]" for j in range(self.nc)})
        n.update({f"lT[{j}]": f"lT[{j}]" for j in range(self.ntc)})
        for s in ("t0v", "tfv", "kap", "th", "W", "sig"):
            n[s] = s
        return n

    def source(self, flags):
        tr, names, ph = self.tr, self._names(), self.phase
        NRED = 3 + self.na
        out = [f"template <> struct Phase<{ph}> {{"]
        out.append(f"  static constexpr int NX = {self.nx}, NU = {self.nu}, NA = {self.na}, NC = {self.nc}, NTC = {self.ntc};")
        out.append(f"  static constexpr int NJV = {len(self.jv)}, NHN = {len(self.hn)}, NHC = {len(self.hc)};")
        out.append(f"  static constexpr int NMG = {len(self.mg)}, NTJ = {len(self.tj)}, NTH = {len(self.th)}, NRED = {NRED};")
        out.append(f"  static constexpr bool DIFF_U = {'true' if flags['diff_u'] else 'false'}, MIDU = {'true' if flags['midu'] else 'false'};")
        # D-block of state a holds a variable entry (its diagonal, merged with -d fx_a/d X_a) or is constant
        ddnz = ", ".join("true" if not e.is_zero else "false" for e in self.dd)
        out.append(f"  static constexpr bool DD_VARIABLE[{max(self.nx, 1)}] = {{{ddnz}}};")
        node_sig = ("const double* __restrict__ Xs, const double* __restrict__ Us, double t0v, double tfv, "
                    "const double* __restrict__ As, double kap, double th, double W")
        term_sig = ("const double* __restrict__ XF, double tfv, const double* __restrict__ X0, double t0v, "
                    "const double* __restrict__ As")
        # unscaled time of a point with normalised position th in the phase (mpopt.py:198, 1564-1571)
        out.append("  __device__ static __forceinline__ double node_time(double t0v, double tfv, double th) {")
        out.append("    double t;")
        out += tr.emit([("t", self.t_node)], names, "    ")
        out.append("    return t;")
        out.append("  }")
        # fg
        out.append(f"  __device__ static __forceinline__ void fg({node_sig}, double* fx, double* c, double& qW) {{")
        asg = [(f"fx[{a}]", e) for a, e in enumerate(self.fx)] + [(f"c[{j}]", e) for j, e in enumerate(self.c)] + [("qW", self.qW)]
        out += tr.emit(asg, names, "    ")
        out.append("  }")
        # fgj
        out.append(f"  __device__ static __forceinline__ void fgj({node_sig}, double* fx, double* c, double* dd, double* jv, double* gn, double* gr) {{")
        asg = ([(f"fx[{a}]", e) for a, e in enumerate(self.fx)] + [(f"c[{j}]", e) for j, e in enumerate(self.c)]
               + [(f"dd[{a}]", e) for a, e in enumerate(self.dd)] + [(f"jv[{k}]", s[4]) for k, s in enumerate(self.jv)]
               + [(f"gn[{k}]", e) for k, e in enumerate(self.gn)] + [(f"gr[{k}]", e) for k, e in enumerate(self.gr)])
        out += tr.emit(asg, names, "    ")
        out.append("  }")
        # hess
        out.append(f"  __device__ static __forceinline__ void hess({node_sig}, double sig, const double* __restrict__ lF, const double* __restrict__ lC, double* hn, double* hc) {{")
        asg = [(f"hn[{k}]", s[4]) for k, s in enumerate(self.hn)] + [(f"hc[{k}]", s[4]) for k, s in enumerate(self.hc)]
        out += tr.emit(asg, names, "    ")
        out.append("  }")
        # gradl: d/d(X, U), d/d(t0, tf, A), d/dkap, d/dth of  sig * qW - lF . fx + lC . c  (nlp_grad, mpx_node_gradl_*)
        out.append(f"  __device__ static __forceinline__ void gradl({node_sig}, double sig, const double* __restrict__ lF, const double* __restrict__ lC, double* gx, double* gr, double& gk, double& gth) {{")
        asg = ([(f"gx[{k}]", e) for k, e in enumerate(self.gl_x)] + [(f"gr[{k}]", e) for k, e in enumerate(self.gl_r)]
               + [("gk", self.gl_k), ("gth", self.gl_th)])
        out += tr.emit(asg, names, "    ")
        out.append("  }")
        # terminal
        out.append(f"  __device__ static __forceinline__ void term_fg({term_sig}, double& M, double* tc) {{")
        out += tr.emit([("M", self.mayer)] + [(f"tc[{j}]", e) for j, e in enumerate(self.tc)], names, "    ")
        out.append("  }")
        out.append(f"  __device__ static __forceinline__ void term_fgj({term_sig}, double& M, double* tc, double* mg, double* tj) {{")
        asg = ([("M", self.mayer)] + [(f"tc[{j}]", e) for j, e in enumerate(self.tc)]
               + [(f"mg[{k}]", s[2]) for k, s in enumerate(self.mg)] + [(f"tj[{k}]", s[3]) for k, s in enumerate(self.tj)])
        out += tr.emit(asg, names, "    ")
        out.append("  }")
        out.append(f"  __device__ static __forceinline__ void term_hess({term_sig}, double sig, const double* __restrict__ lT, double* th) {{")
        out += tr.emit([(f"th[{k}]", s[4]) for k, s in enumerate(self.th)], names, "    ")
        out.append("  }")
        out.append(f"  __device__ static __forceinline__ void term_gradl({term_sig}, double sig, const double* __restrict__ lT, double* tg) {{")
        out += tr.emit([(f"tg[{k}]", e) for k, e in enumerate(self.tg)], names, "    ")
        out.append("  }")
        out.append("};")
        return "\n".join(out)

    def structure(self, flags):
        """Packed int32 description of one phase (format documented in include/mpx.h)."""
        s = [self.nc, self.ntc, int(flags["diff_u"]), int(flags["midu"]), int(flags["du_continuity"])]
        s.append(len(self.jv))
        for rk, rc, ck, cc, _ in self.jv:
            s += [rk, rc, ck, cc]
        s.append(len(self.hn))
        for k1, c1, k2, c2, _ in self.hn:
            s += [k1, c1, k2, c2]
        s.append(len(self.hc))
        for k1, c1, k2, c2, _ in self.hc:
            s += [k1, c1, k2, c2]
        s.append(len(self.mg))
        for k, c, _ in self.mg:
            s += [k, c]
        s.append(len(self.tj))
        for j, k, c, _ in self.tj:
            s += [j, k, c]
        s.append(len(self.th))
        for k1, c1, k2, c2, _ in self.th:
            s += [k1, c1, k2, c2]
        return s


class ProblemProgram:
    """All phases of an OCP: generated source + structure + instantiation list."""

    def __init__(self, ocp, degrees, midu_rows):
        """``degrees``: distinct polynomial degrees on the grid; ``midu_rows[ph]``: whether the
        mid-point control rows are emitted for that phase (mpopt.py:346, 363-365)."""
        self.ocp = ocp
        self.degrees = sorted(set(int(d) for d in degrees))
        self.phases = [PhaseProgram(ocp, ph) for ph in range(ocp.n_phases)]
        self.flags = [dict(diff_u=bool(ocp.diff_u[ph]), midu=bool(midu_rows[ph]),
                           du_continuity=bool(ocp.du_continuity[ph])) for ph in range(ocp.n_phases)]

    def structure(self):
        s = []
        for ph, prog in enumerate(self.phases):
            s += prog.structure(self.flags[ph])
        return np.asarray(s, dtype=np.int32)

    def source(self):
        nph = len(self.phases)
        # (fp contract(off), here and in mpx_kernels.h: a product and a sum are fused only where the source says fma().  The node
        # functions are inlined into several kernels -- node_body, light_body, resid_body, gradl_body --, and with the compiler's
        # default every one of them made its own fusing choices: the same g came out with different last bits from different passes)
        parts = ["// generated by mpopt_amd.codegen -- do not edit", "#include <hip/hip_runtime.h>", "#pragma clang fp contract(off)",
                 f"#define MPX_NPH {nph}", "namespace mpxgen {", "template <int PH> struct Phase;"]
        parts += [p.source(self.flags[k]) for k, p in enumerate(self.phases)]
        parts.append("}  // namespace mpxgen")
        parts.append('#include "mpx_kernels.h"')
        for ph in range(nph):
            for d in self.degrees:
                parts.append(f"MPX_INSTANTIATE_NODE({ph}, {d})")
        if len(self.degrees) > 1:  # mixed-degree grid: the hess_l node pass runs over node-ordered tiles (mpx_kernels.h)
            parts += [f"MPX_INSTANTIATE_HESS_BY_NODE({ph})" for ph in range(nph)]
        for ph in range(nph):
            for d in self.degrees:
                parts.append(f"MPX_INSTANTIATE_GRADL({ph}, {d})")
                if 12 < d <= 31:  # light passes of the high-degree buckets on the matrix cores (mpx_kernels.h: light_body)
                    low = [q for q in self.degrees if q != d]  # (the one low degree as a compile-time constant, light_body)
                    parts.append(f"MPX_INSTANTIATE_LIGHT_PF({ph}, {d}, {low[0] if len(low) == 1 and low[0] <= 12 else 0})")
                if d <= 12 and len(self.degrees) == 1 and self.light_low_chunks(d) >= 2:  # ... and of single-degree grids of low degree (light_low_body)
                    parts.append(f"MPX_INSTANTIATE_LIGHT_LOW({ph}, {d})")
                if d >= 32 and len(self.degrees) == 1:  # ... and of single-degree grids of HIGH degree: evaluation points as a matrix dimension (light_high_body)
                    parts.append(f"MPX_INSTANTIATE_LIGHT_HIGH({ph}, {d})")
        if nph > 1 and len(self.degrees) == 1:  # all phases of a single-degree grid in ONE launch (mpx_kernels.h: node_all, light_low_all)
            d = self.degrees[0]
            parts.append(f"MPX_INSTANTIATE_NODE_ALL({d})")
            if d <= 12 and self.light_low_chunks(d) >= 2:
                parts.append(f"MPX_INSTANTIATE_LIGHT_LOW_ALL({d})")
        parts.append("MPX_INSTANTIATE_BOUNDARY()")
        # (read by libmpx at load: 0 = no node function of any phase uses the node time, the widths' prefix sums are not needed)
        parts.append('extern "C" __device__ __attribute__((used)) const int mpx_time_dependent = '
                     f"{1 if any(p.time_dependent for p in self.phases) else 0};")
        parts += self._resident_source()
        return "\n".join(parts) + "\n"

    def light_low_chunks(self, d):
        """64-node chunks per span of the low-degree light kernels, exactly as light_low_body (mpx_kernels.h: CAP0 / CHL) and the
        host's plan (mpx_layout.cpp: lplan, `ok = chl >= 2`) compute them: the X / U rows of MPX_LIGHT_WAVES = 4 wavefronts share 52 KB
        of LDS.  Problems with many states + controls (nx + nu > ~11 at degree 3) get fewer than two chunks: the host would not use
        the kernels, and from ~24 inputs on their span rows do not fit the LDS of a workgroup at all (static_assert in the kernel) --
        so they are not instantiated (the node kernels serve the light passes of such problems, as before round 4)."""
        nin = self.phases[0].nx + self.phases[0].nu
        cap0 = 53248 // (8 * 4 * max(nin, 1))
        return max(1, (cap0 - 2 * d - 8) // 64)

    def _resident_source(self):
        """The resident kernel of single evaluations (mpx_kernels.h: resident_loop): one workgroup per tile stays on the device and
        is fed through mapped host memory.  Generated only for single-degree grids of low degree whose node kernels (first-order
        and hess_l variant of every phase, inlined into ONE kernel) fit the LDS of a workgroup."""
        if len(self.degrees) != 1 or self.degrees[0] > 12:
            return []
        d = self.degrees[0]
        slots = (256 // d) * (d + 1)
        lds = sum(2 * (2 * (p.nx + p.nu) * slots * 8 + 2 * 4 * max(3 + p.na, len(p.hc), 1) * 8) for p in self.phases)
        if lds > 56 * 1024:
            return []
        out = ['extern "C" __global__ __launch_bounds__(MPX_TILE) void mpx_resident(const MpxResidentArgs R) {',
               "  mpxk::resident_loop(R,", "    [&](const MpxNodeArgs& A, int mode, int bx) {"]
        for ph in range(len(self.phases)):
            out.append(f"      {'if' if ph == 0 else 'else if'} (A.phase == {ph}) {{ if (mode == MPX_MODE_HESS) mpxk::node_body<{ph}, {d}, MPX_MODE_HESS>(A, bx); "
                       f"else mpxk::node_body<{ph}, {d}, MPX_MODE_FGJ>(A, bx); }}")
        out += ["    },", "    [&](const MpxBoundArgs& G, int mode) { if (mode == MPX_MODE_HESS) mpxk::boundary_body<MPX_MODE_HESS>(G, 0); else mpxk::boundary_body<MPX_MODE_FGJ>(G, 0); });",
                "}"]
        return out

    def key(self, extra=""):
        return hashlib.sha256((self.source() + extra).encode()).hexdigest()[:24]

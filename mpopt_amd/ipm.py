"""A compact primal-dual interior-point driver for the NLPs of this package.

OUT OF SCOPE of the hot path (IPOPT + MUMPS inside CasADi are what the reference uses for
the outer iteration, mpopt.py:757, 804).  It exists so that ``mp.solve`` runs end to end with the GPU oracles where
neither CasADi nor IPOPT is installed: every f / g / grad_f / jac_g / hess_l value it consumes comes from the HIP
kernels through the C ABI; the linear algebra of the Newton steps is SciPy's sparse LU on the host.

Algorithm: the textbook line-search barrier method in IPOPT's formulation (Waechter & Biegler 2006, sections 2-3),
reduced to what these collocation problems need:

    min f(x)   s.t.  g_E(x) = b_E,   g_I(x) - s = 0,   l <= (x, s) <= u
    barrier phi_mu = f - mu * sum log(distance to finite bounds), monotone mu update (Fiacco-McCormick),
    primal-dual Newton steps with diagonal Sigma = z_l/(y-l) + z_u/(u-y), fraction-to-the-boundary rule,
    filter line search with a second-order correction and a Gauss-Newton feasibility step as crude restoration,
    inertia handled by increasing a diagonal regularisation until the step has non-negative curvature.

Fixed variables (l == u, e.g. the initial state) are removed from the step, like IPOPT's
``fixed_variable_treatment = make_parameter``.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class InteriorPoint:
    def __init__(self, orc, p, lbx, ubx, lbg, ubg, tol=1e-8, max_iter=2000, acceptable_tol=1e-6, print_level=0):
        self.o, self.p = orc, p
        self.tol, self.max_iter, self.acc_tol, self.verbose = float(tol), int(max_iter), float(acceptable_tol), int(print_level)
        n, m = orc.n_z, orc.n_g
        lbx, ubx = np.asarray(lbx, float).ravel(), np.asarray(ubx, float).ravel()
        lbg, ubg = np.asarray(lbg, float).ravel(), np.asarray(ubg, float).ravel()
        self.n, self.m = n, m
        self.eq = np.nonzero(lbg == ubg)[0]
        self.iq = np.nonzero(lbg != ubg)[0]
        self.b_eq = lbg[self.eq]
        self.nI = len(self.iq)
        self.l = np.concatenate([lbx, lbg[self.iq]])
        self.u = np.concatenate([ubx, ubg[self.iq]])
        self.fixed = self.l == self.u
        self.free = np.nonzero(~self.fixed)[0]
        self.has_l = np.isfinite(self.l) & ~self.fixed
        self.has_u = np.isfinite(self.u) & ~self.fixed
        jr, jc = orc.jac_pattern()
        hr, hc = orc.hess_pattern()
        self.jr, self.jc, self.hr, self.hc = jr.astype(np.int64), jc.astype(np.int64), hr.astype(np.int64), hc.astype(np.int64)
        # row of a g row in c = [equalities; inequalities]
        self.crow = np.empty(m, np.int64)
        self.crow[self.eq] = np.arange(len(self.eq))
        self.crow[self.iq] = len(self.eq) + np.arange(self.nI)
        self.n_eval = {"f": 0, "g": 0, "grad_f": 0, "jac_g": 0, "hess_l": 0}
        self.fs, self.gs = 1.0, np.ones(m)

    # -- oracle access -------------------------------------------------------------------------------
    def _ev(self, what, x, **kw):
        """Oracle call in the SCALED problem (IPOPT's gradient-based scaling: objective factor fs, constraint-row factors gs)."""
        for w in what:
            self.n_eval[w] += 1
        if "lam_g" in kw:
            kw = dict(kw, lam_g=np.asarray(kw["lam_g"]) * self.gs, sigma=np.asarray(kw["sigma"]) * self.fs)
        r = self.o.eval(what, x, self.p, pinned=False, **kw)
        out = {}
        for k, v in r.items():
            if k == "f":
                out[k] = v * self.fs
            elif k == "grad_f":
                out[k] = v * self.fs
            elif k == "g":
                out[k] = v * self.gs
            elif k == "jac_g":
                out[k] = v * self.gs[self.jr]
            else:
                out[k] = v
        return out

    def _setup_scaling(self, x):
        """fs = min(1, 100 / max|grad f|), gs_i = min(1, 100 / max_j |J_ij|) at the starting point (nlp_scaling_max_gradient)."""
        self.fs, self.gs = 1.0, np.ones(self.m)
        r = self.o.eval(["grad_f", "jac_g"], x, self.p)
        gmax = np.abs(r["grad_f"]).max() if self.n else 0.0
        fs = min(1.0, 100.0 / gmax) if gmax > 0 else 1.0
        rowmax = np.zeros(self.m)
        np.maximum.at(rowmax, self.jr, np.abs(r["jac_g"]))
        gs = np.where(rowmax > 100.0, 100.0 / np.maximum(rowmax, 1e-300), 1.0)
        self.fs, self.gs = max(fs, 1e-8), np.maximum(gs, 1e-8)
        # rescale the row bounds that were split in __init__
        self.b_eq = self.b_eq * self.gs[self.eq]
        self.l[self.n:] = self.l[self.n:] * self.gs[self.iq]
        self.u[self.n:] = self.u[self.n:] * self.gs[self.iq]

    def _c(self, g, s):
        return np.concatenate([g[self.eq] - self.b_eq, g[self.iq] - s])

    def _A(self, jv):
        """Jacobian of c with respect to y = (x, s)."""
        ny = self.n + self.nI
        A = sp.coo_matrix((jv, (self.crow[self.jr], self.jc)), shape=(self.m, ny)).tocsr()
        if self.nI:
            A = A + sp.coo_matrix((-np.ones(self.nI), (len(self.eq) + np.arange(self.nI), self.n + np.arange(self.nI))), shape=(self.m, ny))
        return A.tocsr()

    def _W(self, hv):
        ny = self.n + self.nI
        up = sp.coo_matrix((hv, (self.hr, self.hc)), shape=(ny, ny)).tocsr()
        return up + sp.triu(up, 1).T

    def _lam_g(self, lam):
        out = np.empty(self.m)
        out[self.eq] = lam[:len(self.eq)]
        out[self.iq] = lam[len(self.eq):]
        return out

    # -- barrier pieces ------------------------------------------------------------------------------
    def _phi(self, f, y, mu):
        return f - mu * (np.log(y[self.has_l] - self.l[self.has_l]).sum() + np.log(self.u[self.has_u] - y[self.has_u]).sum())

    def _max_step(self, v, dv, lo_mask, hi_mask, tau):
        a = 1.0
        d = dv[lo_mask]
        neg = d < 0
        if neg.any():
            a = min(a, float((-tau * (v[lo_mask] - self.l[lo_mask])[neg] / d[neg]).min()))
        d = dv[hi_mask]
        pos = d > 0
        if pos.any():
            a = min(a, float((tau * (self.u[hi_mask] - v[hi_mask])[pos] / d[pos]).min()))
        return a

    def solve(self, x0, lam0=None):
        n, nI, m, free = self.n, self.nI, self.m, self.free
        l, u, has_l, has_u = self.l, self.u, self.has_l, self.has_u
        x = np.array(x0, float).ravel()
        # interior starting point (IPOPT: bound_push / bound_frac = 1e-2)
        def push(v, lo, hi):
            v = v.copy()
            pl = np.minimum(1e-2 * np.maximum(1.0, np.abs(lo)), 1e-2 * (hi - lo))
            pu = np.minimum(1e-2 * np.maximum(1.0, np.abs(hi)), 1e-2 * (hi - lo))
            both = np.isfinite(lo) & np.isfinite(hi)
            v[both] = np.clip(v[both], (lo + pl)[both], (hi - pu)[both])
            ol = np.isfinite(lo) & ~np.isfinite(hi)
            v[ol] = np.maximum(v[ol], lo[ol] + 1e-2 * np.maximum(1.0, np.abs(lo[ol])))
            ou = ~np.isfinite(lo) & np.isfinite(hi)
            v[ou] = np.minimum(v[ou], hi[ou] - 1e-2 * np.maximum(1.0, np.abs(hi[ou])))
            return v

        fx = self.fixed[:n]
        x[fx] = l[:n][fx]
        x[~fx] = push(x[~fx], l[:n][~fx], u[:n][~fx])
        self._setup_scaling(x)
        l, u = self.l, self.u
        r = self._ev(["f", "g", "grad_f", "jac_g"], x)
        s = r["g"][self.iq].copy()
        fs = self.fixed[n:]
        s[fs] = l[n:][fs]
        s[~fs] = push(s[~fs], l[n:][~fs], u[n:][~fs])
        y = np.concatenate([x, s])
        zl, zu = np.where(has_l, 1.0, 0.0), np.where(has_u, 1.0, 0.0)
        lam = np.zeros(m)
        if lam0 is not None:  # multipliers of the unscaled problem -> scaled
            l0 = np.asarray(lam0, float) * self.fs / self.gs
            lam = np.concatenate([l0[self.eq], l0[self.iq]])
        mu, dw_last = 0.1, 0.0
        filt, filt_mu = [], None
        th_init = float(np.abs(self._c(r["g"], y[n:])).sum())
        status, it = "Maximum_Iterations_Exceeded", 0

        def kkt_error(mu_, gradL, c):
            sd = max(100.0, (np.abs(lam).sum() + zl.sum() + zu.sum()) / max(1, m + has_l.sum() + has_u.sum())) / 100.0
            comp = 0.0
            if has_l.any():
                comp = max(comp, np.abs((y - l)[has_l] * zl[has_l] - mu_).max())
            if has_u.any():
                comp = max(comp, np.abs((u - y)[has_u] * zu[has_u] - mu_).max())
            return max(np.abs(gradL[free]).max() / sd if len(free) else 0.0, np.abs(c).max() if m else 0.0, comp / sd)

        for it in range(self.max_iter):
            f, g, gf, jv = float(r["f"]), r["g"], r["grad_f"], r["jac_g"]
            A = self._A(jv)
            c = self._c(g, y[n:])
            grad = np.concatenate([gf, np.zeros(nI)])
            gradL = grad + A.T @ lam - zl + zu
            e0 = kkt_error(0.0, gradL, c)
            if self.verbose:
                print(f"{it:4d} f={f: .8e} inf_pr={np.abs(c).max() if m else 0:.2e} err={e0:.2e} mu={mu:.1e}")
            if e0 <= self.tol:
                status = "Solve_Succeeded"
                break
            while mu > self.tol / 10 and kkt_error(mu, gradL, c) <= 10.0 * mu:
                mu = max(self.tol / 10, min(0.2 * mu, mu ** 1.5))
            tau = max(0.99, 1.0 - mu)
            hv = self._ev(["hess_l"], y[:n], lam_g=self._lam_g(lam), sigma=1.0)["hess_l"]
            W = self._W(hv)
            tiny = np.finfo(float).tiny ** 0.5
            dl = np.where(has_l, np.maximum(y - l, tiny), 1.0)
            du = np.where(has_u, np.maximum(u - y, tiny), 1.0)
            sigma = np.where(has_l, zl / dl, 0.0) + np.where(has_u, zu / du, 0.0)
            rhs_y = -(grad + A.T @ lam - np.where(has_l, mu / dl, 0.0) + np.where(has_u, mu / du, 0.0))
            Af = A[:, free]
            Wf = (W + sp.diags(sigma))[free][:, free]
            c1 = np.abs(c).sum()
            dphi_base = (grad - np.where(has_l, mu / dl, 0.0) + np.where(has_u, mu / du, 0.0))[free]
            dw, dc = 0.0, 1e-8 * mu ** 0.25
            dy = dlam = None
            for trial in range(60):
                K = sp.bmat([[Wf + dw * sp.identity(len(free)), Af.T], [Af, -dc * sp.identity(m)]], format="csc")
                try:
                    lu = spla.splu(K)
                    sol = lu.solve(np.concatenate([rhs_y[free], -c]))
                    ok = np.isfinite(sol).all()
                except RuntimeError:
                    ok = False
                if ok:
                    dyf, dlam = sol[:len(free)], sol[len(free):]
                    curv = float(dyf @ (Wf @ dyf) + dw * (dyf @ dyf))
                    # SuperLU reports no inertia; its practical consequences are tested instead: positive curvature
                    # along the step and, at (nearly) feasible points, descent of the barrier objective
                    good = curv > 1e-12 * float(dyf @ dyf) and (c1 > 1e-6 * max(1.0, th_init) or float(dphi_base @ dyf) < 0)
                    if good or dw >= 1e12:
                        dy = np.zeros(n + nI)
                        dy[free] = dyf
                        break
                if not ok:  # singular: dependent constraint rows (the constraint regularisation was too small)
                    dc = min(1e-2, dc * 100.0)
                dw = max(1e-4, dw_last / 3.0) if dw == 0.0 else dw * (8.0 if dw_last > 0 else 100.0)
                if dw > 1e16:
                    break
            if dy is None:
                if self.verbose:
                    print(f"   step computation failed: dw={dw:.1e} dc={dc:.1e} finite W={np.isfinite(hv).all()} A={np.isfinite(jv).all()} "
                          f"rhs={np.isfinite(rhs_y).all()} sigma_max={sigma.max():.2e} ok={ok} curv={curv if ok else None} "
                          f"dphi={float(dphi_base @ dyf) if ok else None} c1={c1:.2e}")
                status = "Error_In_Step_Computation"
                break
            dw_last = dw
            dzl = np.where(has_l, mu / dl - zl - zl / dl * dy, 0.0)
            dzu = np.where(has_u, mu / du - zu + zu / du * dy, 0.0)
            a_max = self._max_step(y, dy, has_l, has_u, tau)
            a_du = 1.0
            for z_, dz_ in ((zl, dzl), (zu, dzu)):
                neg = dz_ < 0
                if neg.any():
                    a_du = min(a_du, float((-tau * z_[neg] / dz_[neg]).min()))
            # filter line search (Waechter & Biegler 2006, section 2.3) on (theta, phi) = (||c||_1, barrier objective),
            # with one second-order correction when the first trial increases the infeasibility
            dphi = float(dphi_base @ dy[free])
            phi0, th0 = self._phi(f, y, mu), c1
            if mu != filt_mu:  # a new barrier problem: the filter starts over
                filt, filt_mu = [], mu
            th_max = 1e4 * max(1.0, th_init)
            th_min = 1e-4 * max(1.0, th_init)

            def acceptable(th_t, phi_t, a_):
                if not (np.isfinite(th_t) and np.isfinite(phi_t)) or th_t > th_max:
                    return False, False
                for (tf_, pf_) in filt:
                    if th_t >= tf_ and phi_t >= pf_:
                        return False, False
                ftype = th0 <= th_min and dphi < 0 and a_ * min(-dphi, 1e100) ** 2.3 > 1.0 * th0 ** 1.1
                if ftype:
                    return phi_t <= phi0 + 1e-8 * a_ * dphi + 10 * np.finfo(float).eps * abs(phi0), True
                return (th_t <= (1 - 1e-5) * th0 or phi_t <= phi0 - 1e-5 * th0), False

            a, accepted, ftype, soc_done = a_max, False, False, False
            spec = None  # backtracking candidates evaluated ahead, as one batched oracle call
            for ls in range(30):
                yt = y + a * dy
                if ls >= 1 and (ls - 1) % 8 == 0:
                    # the first trial failed: the next eight step lengths cost one launch instead of eight (a batch of
                    # evaluation points is the GPU oracle's native mode; IPOPT's sequential line search cannot use it)
                    cand = np.stack([(y + a * 0.5 ** k * dy)[:n] for k in range(8)])
                    rb = self._ev(["f", "g"], cand)
                    spec = [{"f": rb["f"][k], "g": rb["g"][k]} for k in range(8)]
                rt = self._ev(["f", "g"], yt[:n]) if ls == 0 else spec[(ls - 1) % 8]
                ct = self._c(rt["g"], yt[n:])
                tht = float(np.abs(ct).sum())
                accepted, ftype = acceptable(tht, self._phi(float(rt["f"]), yt, mu) if np.isfinite(rt["f"]) else np.inf, a)
                if accepted:
                    break
                if ls == 0 and not soc_done and np.isfinite(tht) and tht >= th0:
                    soc_done = True  # second-order correction with the factorisation of this iteration
                    csoc = a * c + ct
                    sol2 = lu.solve(np.concatenate([rhs_y[free], -csoc]))
                    if np.isfinite(sol2).all():
                        dy2 = np.zeros(n + nI)
                        dy2[free] = sol2[:len(free)]
                        a2 = self._max_step(y, dy2, has_l, has_u, tau)
                        y2 = y + a2 * dy2
                        r2 = self._ev(["f", "g"], y2[:n])
                        c2 = self._c(r2["g"], y2[n:])
                        ok2, ft2 = acceptable(float(np.abs(c2).sum()), self._phi(float(r2["f"]), y2, mu) if np.isfinite(r2["f"]) else np.inf, a2)
                        if ok2:
                            yt, a, accepted, ftype, dlam = y2, a2, True, ft2, sol2[len(free):]
                            break
                a *= 0.5
            if not accepted:
                # no acceptable step: a crude restoration -- a damped Gauss-Newton step on the infeasibility
                JJ = (Af @ Af.T + 1e-8 * sp.identity(m)).tocsc()
                try:
                    dyr = np.zeros(n + nI)
                    dyr[free] = -Af.T @ spla.splu(JJ).solve(c)
                except RuntimeError:
                    dyr = None
                done = False
                if dyr is not None and th0 > self.tol:
                    ar = self._max_step(y, dyr, has_l, has_u, tau)
                    for _ in range(30):
                        yt = y + ar * dyr
                        rt = self._ev(["f", "g"], yt[:n])
                        tht = float(np.abs(self._c(rt["g"], yt[n:])).sum())
                        if np.isfinite(tht) and tht <= (1 - 1e-4 * ar) * th0:
                            done = True
                            break
                        ar *= 0.5
                if done:
                    filt.append(((1 - 1e-5) * th0, phi0 - 1e-5 * th0))
                    y = yt
                    r = self._ev(["f", "g", "grad_f", "jac_g"], y[:n])
                    continue
                if mu > self.tol / 10:  # stuck at this barrier parameter: tighten it and continue
                    mu = max(self.tol / 10, 0.2 * mu)
                    r = self._ev(["f", "g", "grad_f", "jac_g"], y[:n])
                    continue
                status = "Search_Direction_Becomes_Too_Small"
                break
            if not ftype:
                filt.append(((1 - 1e-5) * th0, phi0 - 1e-5 * th0))
            nu, dmerit = 0.0, dphi
            if self.verbose > 1:
                print(f"      alpha={a:.2e} alpha_max={a_max:.2e} alpha_du={a_du:.2e} dw={dw:.1e} |dy|={np.abs(dy).max():.2e} nu={nu:.1e} dmerit={dmerit:.2e} ls={ls}")
            y = yt
            # an iterate that rounds onto a bound would make the barrier terms infinite: keep a floor on the slacks
            # (IPOPT moves the bound instead, slack_move = eps^(3/4))
            floor = 1e-13
            y[has_l] = np.maximum(y[has_l], l[has_l] + floor * np.maximum(1.0, np.abs(l[has_l])))
            y[has_u] = np.minimum(y[has_u], u[has_u] - floor * np.maximum(1.0, np.abs(u[has_u])))
            lam = lam + a * dlam
            zl = zl + a_du * dzl
            zu = zu + a_du * dzu
            # keep the bound multipliers within a factor of the central path (IPOPT eq. (16))
            if has_l.any():
                zl[has_l] = np.clip(zl[has_l], mu / (1e10 * (y - l)[has_l]), 1e10 * mu / (y - l)[has_l])
            if has_u.any():
                zu[has_u] = np.clip(zu[has_u], mu / (1e10 * (u - y)[has_u]), 1e10 * mu / (u - y)[has_u])
            r = self._ev(["f", "g", "grad_f", "jac_g"], y[:n])
        else:
            it = self.max_iter
        if status != "Solve_Succeeded":
            f, g, gf, jv = float(r["f"]), r["g"], r["grad_f"], r["jac_g"]
            A = self._A(jv)
            gradL = np.concatenate([gf, np.zeros(nI)]) + A.T @ lam - zl + zu
            if kkt_error(0.0, gradL, self._c(g, y[n:])) <= self.acc_tol:
                status = "Solved_To_Acceptable_Level"
        lam_g = self._lam_g(lam) * self.gs / self.fs  # back to the unscaled problem
        return {"x": y[:n].copy(), "f": float(r["f"]) / self.fs, "g": r["g"] / self.gs, "lam_g": lam_g, "lam_x": (zu - zl)[:n] / self.fs, "status": status,
                "iter_count": it, "success": status in ("Solve_Succeeded", "Solved_To_Acceptable_Level")}

"""``NlpSolver``: the callable that ``mpopt.solve`` invokes where the reference calls the object
returned by ``ca.nlpsol`` (mpopt.py:757, 804).

The outer NLP iteration (IPOPT + MUMPS inside CasADi in the reference) is OUT OF SCOPE of this
build (only the oracle functions are the hot path).  So that ``mp.solve``
still runs end to end -- and so that the published optimum of the reference can anchor the whole
pipeline -- the GPU oracles are driven here by stand-ins: a compact primal-dual barrier method in
IPOPT's formulation (mpopt_amd/ipm.py; default) with SciPy's ``trust-constr`` as the fallback when it
does not converge (``options["standin"]``: "auto" | "ipm" | "trust-constr").  They are stand-in
drivers, not a product claim: every f/g/grad/jac/hess value they consume comes from the HIP kernels
through the C ABI.
"""
import time

import numpy as np
import scipy.sparse as sp
from scipy.optimize import Bounds, NonlinearConstraint, minimize


class NlpSolver:
    def __init__(self, name, solver, nlp_problem, options=None):
        self.name, self.solver = name, solver
        self.oracle = nlp_problem["oracle"]
        self.options = dict(options or {})
        self.stats = {}

    def __call__(self, x0=None, p=None, lbx=None, ubx=None, lbg=None, ubg=None, lam_x0=None, lam_g0=None):
        orc = self.oracle
        p = np.zeros(0) if p is None else np.asarray(p, dtype=float).ravel()
        return self._with_lam_p(self._solve(x0, p, lbx, ubx, lbg, ubg, lam_g0), p)

    def _with_lam_p(self, sol, p):
        """What CasADi's Nlpsol does after the last iterate (option ``calc_lam_p``, default true): ONE evaluation of ``nlp_grad`` at
        the solution with ``lam_f = 1`` and the final multipliers; ``lam_p = -grad_gamma_p`` (the reference's tests assert the key,
        tests/test_examples.py:44-45; "nlp_grad ... n_eval 1" in its recorded solves, moon_lander.ipynb:206).  On the GPU
        (mpx_eval_grad_gamma)."""
        if p.size and self.options.get("calc_lam_p", True):
            q = self.oracle.eval_grad_gamma(np.asarray(sol["x"], float).ravel(), p, np.asarray(sol["lam_g"], float).ravel(), 1.0,
                                            what=("grad_gamma_p",))
            sol["lam_p"] = -q["grad_gamma_p"]
            if isinstance(self.stats.get("n_eval"), dict):
                self.stats["n_eval"]["nlp_grad"] = 1
        return sol

    def _solve(self, x0, p, lbx, ubx, lbg, ubg, lam_g0):
        standin = self.options.get("standin", "auto")
        if standin == "trust-constr":
            return self._trust_constr(x0, p, lbx, ubx, lbg, ubg)
        sol = self._interior_point(x0, p, lbx, ubx, lbg, ubg, lam_g0)
        if standin == "ipm" or self.stats["success"]:
            return sol
        # "auto": the barrier method did not converge (it has no proper restoration phase) -- let SciPy's trust-region
        # interior point continue from where it stopped and keep the better of the two end points
        first, viol = dict(self.stats), self._violation(sol, lbx, ubx, lbg, ubg)
        start = sol["x"] if np.isfinite(sol["x"]).all() and viol < 1e-2 else x0
        sol2 = self._trust_constr(start, p, lbx, ubx, lbg, ubg)
        viol2 = self._violation(sol2, lbx, ubx, lbg, ubg)
        self.stats["first_attempt"] = first
        if self.stats["success"] or viol2 < viol or (viol2 <= 10 * max(viol, 1e-8) and sol2["f"] < sol["f"]):
            return sol2
        self.stats = dict(first, second_attempt=self.stats)
        return sol

    @staticmethod
    def _violation(sol, lbx, ubx, lbg, ubg):
        x, g = np.asarray(sol["x"], float).ravel(), np.asarray(sol["g"], float).ravel()
        v = [np.maximum(np.asarray(lbx, float) - x, 0).max(initial=0.0), np.maximum(x - np.asarray(ubx, float), 0).max(initial=0.0)]
        if g.size:
            v += [np.maximum(np.asarray(lbg, float) - g, 0).max(initial=0.0), np.maximum(g - np.asarray(ubg, float), 0).max(initial=0.0)]
        return float(max(v))

    def _interior_point(self, x0, p, lbx, ubx, lbg, ubg, lam_g0):
        """Barrier method of mpopt_amd/ipm.py (the default stand-in: same problem formulation and options as IPOPT)."""
        from .ipm import InteriorPoint

        t0 = time.perf_counter()
        ip = InteriorPoint(self.oracle, p, lbx, ubx, lbg, ubg, tol=float(self.options.get("ipopt.tol", 1e-8)),
                           max_iter=int(self.options.get("ipopt.max_iter", 2000)),
                           acceptable_tol=float(self.options.get("ipopt.acceptable_tol", 1e-4)),
                           print_level=int(self.options.get("ipopt.print_level", 0)))
        res = ip.solve(np.asarray(x0, dtype=float).ravel(), lam_g0)
        self.stats = {"iter_count": res["iter_count"], "success": res["success"], "return_status": res["status"], "n_eval": ip.n_eval,
                      "t_wall_s": time.perf_counter() - t0}
        return {"x": res["x"], "f": res["f"], "g": res["g"], "lam_x": res["lam_x"], "lam_g": res["lam_g"], "lam_p": np.zeros_like(p)}

    def _trust_constr(self, x0, p, lbx, ubx, lbg, ubg):
        """SciPy's interior-point trust-region method (options={"standin": "trust-constr"})."""
        orc = self.oracle
        n, m = orc.n_z, orc.n_g
        x0 = np.clip(np.asarray(x0, dtype=float).ravel(), lbx, ubx)
        jr, jc = orc.jac_pattern()
        hr, hc = orc.hess_pattern()
        cnt = {"f": 0, "g": 0, "grad_f": 0, "jac_g": 0, "hess_l": 0}
        t_eval = [0.0]

        def timed(what, *a, **k):
            t = time.perf_counter()
            r = orc.eval(what, *a, **k)
            t_eval[0] += time.perf_counter() - t
            for w in what:
                cnt[w] += 1
            return r

        def fun(x):
            return float(timed(["f"], x, p)["f"])

        def grad(x):
            return timed(["grad_f"], x, p)["grad_f"]

        def con(x):
            return timed(["g"], x, p)["g"]

        def jac(x):
            return sp.csr_matrix((timed(["jac_g"], x, p, pinned=True)["jac_g"], (jr, jc)), shape=(m, n))

        def hess_con(x, v):
            h = timed(["hess_l"], x, p, lam_g=v, sigma=0.0, pinned=True)["hess_l"]
            return _sym(h, hr, hc, n)

        def hess_obj(x):
            h = timed(["hess_l"], x, p, lam_g=np.zeros(m), sigma=1.0, pinned=True)["hess_l"]
            return _sym(h, hr, hc, n)

        max_iter = int(self.options.get("ipopt.max_iter", 2000))
        tol = float(self.options.get("ipopt.tol", 1e-8))
        res = minimize(fun, x0, jac=grad, hess=hess_obj, method="trust-constr",
                       bounds=Bounds(lbx, ubx, keep_feasible=False),
                       constraints=[NonlinearConstraint(con, lbg, ubg, jac=jac, hess=hess_con)] if m else [],
                       options={"maxiter": max_iter, "gtol": tol, "xtol": 1e-14, "barrier_tol": min(tol, 1e-10), "verbose": 0,
                                "sparse_jacobian": True})
        self.stats = {"iter_count": res.nit, "success": bool(res.success), "return_status": res.message,
                      "n_eval": cnt, "t_oracle_s": t_eval[0]}
        lam_g = np.asarray(res.v[0]) if m and len(res.v) else np.zeros(m)
        lam_x = np.asarray(res.v[-1]) if len(res.v) > (1 if m else 0) else np.zeros(n)
        return {"x": res.x, "f": float(res.fun), "g": con(res.x) if m else np.zeros(0), "lam_x": lam_x, "lam_g": lam_g,
                "lam_p": np.zeros_like(p)}


def _sym(vals, r, c, n):
    up = sp.coo_matrix((vals, (r, c)), shape=(n, n)).tocsr()
    return up + sp.triu(up, 1).T

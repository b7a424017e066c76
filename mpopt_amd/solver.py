"""``NlpSolver``: the callable that ``mpopt.solve`` invokes where the reference calls the object
returned by ``ca.nlpsol`` (mpopt.py:757, 804).

The outer NLP iteration (IPOPT + MUMPS inside CasADi in the reference) is OUT OF SCOPE of this
build (only the oracle functions are the hot path).  So that ``mp.solve``
still runs end to end -- and so that the published optimum of the reference can anchor the whole
pipeline -- the GPU oracles are driven here by SciPy's interior-point ``trust-constr`` method.
It is a stand-in driver, not a product claim: every f/g/grad/jac/hess value it consumes comes from
the HIP kernels through the C ABI.
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

"""The grids the reference publishes numbers for (CasADi's per-call timing table of the documentation
notebooks -- moon lander 10 x 6 LGR and 2 x 30 CGL / LGL, hypersensitive 5 x 50 LGR / CGL / LGL, Van der Pol 1 x 25 LGR, two-phase
Schwartz 1 x 20 per phase LGR; /root/reference/docs/source/notebooks/moon_lander.ipynb:171-210, 280-319, 373-412,
hypersensitive.ipynb:165-204, 267-306, 360-399, vanderpol.ipynb:177-216, twophaseschwartz.ipynb:195-234) as GPU parity rows: all five
outputs + nlp_grad of every grid against the numpy / sympy oracle and, per entry, against the C oracle (the CPU column of the
per-call table in profiles/r6_report.md is timed with that code); 10 x 6 also against the golden of the reference's own NLP; the
problem sizes against IPOPT's recorded size report."""
import numpy as np
import pytest
import scipy.sparse as sp

import mpopt_amd as M
from mpopt_amd import mp
import problems
from helpers import assert_by_class, grad_classes, hess_classes, jac_classes, load_golden, rel_err
from oracle.mpopt_oracle import OracleNLP
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu
TOL = 1e-10
# "Number of variables / equality constraints / inequality constraints" of IPOPT's banner in the same notebook cells: variables with
# lb == ub are not counted by IPOPT (fixed_variable_treatment = make_parameter)
IPOPT_SIZES = {"moon_lander_10x6_LGR": (182, 124, 60), "hyper_sensitive_5x50_LGR": (501, 252, None), "van_der_pol_1x25_LGR": (76, 52, 25),
               "schwartz_1x20_LGR": (125, 88, 41)}


@pytest.mark.parametrize("name", list(problems.PUBLISHED_GRIDS))
def test_published_grid_parity(name):
    builder, S, P, scheme, cnames, st, midu, _, _ = problems.PUBLISHED_GRIDS[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, P, scheme)
    nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    if name in IPOPT_SIZES:
        nv, neq, nineq = IPOPT_SIZES[name]
        lbx, ubx, lbg, ubg = (np.asarray(bounds[k], float) for k in ("lbx", "ubx", "lbg", "ubg"))
        assert int((lbx < ubx).sum()) == nv and int((lbg == ubg).sum()) == neq
        if nineq is not None:
            assert int((lbg < ubg).sum()) == nineq
    O = OracleNLP(ocp, S, P, scheme)
    assert (o.n_z, o.n_g) == (O.n_z, O.n_g) and np.array_equal(O.initial_guess(), mpo.initialize_solution())
    rng = np.random.default_rng(7)
    z0 = mpo.initialize_solution()
    z = z0 + 0.05 * np.abs(z0) * rng.uniform(-1, 1, o.n_z) + 0.05 * rng.uniform(-1, 1, o.n_z)
    w = rng.uniform(0.3, 1.7, (ocp.n_phases, S))
    p = (w / w.sum(axis=1, keepdims=True)).ravel()
    lam, sig = rng.standard_normal(o.n_g), float(rng.uniform(0.2, 2.0))
    r = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], z, p, lam_g=lam, sigma=sig)
    q = o.eval_grad_gamma(z, p, lam, sig)
    jr, jc = o.jac_pattern()
    hr, hc = o.hess_pattern()
    # numpy / sympy oracle (tables in 50-digit arithmetic above degree 10)
    assert rel_err(r["f"], O.f(z, p)) < TOL and rel_err(r["g"], O.g(z, p)) < TOL and rel_err(r["grad_f"], O.grad_f(z, p)) < TOL
    Jo = sp.csr_matrix(O.jac_g(z, p))
    d = sp.coo_matrix((r["jac_g"], (jr, jc)), shape=(o.n_g, o.n_z)).tocsr() - Jo
    assert (abs(d).max() if d.nnz else 0.0) < TOL * max(1.0, abs(Jo).max())
    Ho = sp.csr_matrix(np.triu(O.hess_l(z, p, sig, lam)))
    d = sp.coo_matrix((r["hess_l"], (hr, hc)), shape=(o.n_z, o.n_z)).tocsr() - Ho
    assert (abs(d).max() if d.nnz else 0.0) < TOL * max(1.0, abs(Ho).max())
    gx, gp = O.grad_gamma(z, p, sig, lam)
    assert rel_err(q["grad_gamma_x"], gx) < TOL and rel_err(q["grad_gamma_p"], gp) < TOL
    # the C oracle (binary64: what the CPU column of the report times), per entry and per entry class
    C = COracle(cnames, S, P, scheme, scale_x=ocp.scale_x, scale_u=ocp.scale_u, scale_a=ocp.scale_a if ocp.na else None, scale_t=st, midu=midu)
    c, c2 = C.eval(z, p), C.eval(z * 1.01, p)
    Jal, Jal2 = (np.asarray(sp.coo_matrix((v["jac_val"], (v["jac_row"], v["jac_col"])), shape=(o.n_g, o.n_z)).tocsr()[jr, jc]).ravel() for v in (c, c2))
    assert rel_err(r["f"], c["f"]) < TOL and rel_err(r["g"], c["g"]) < TOL
    assert_by_class(r["jac_g"], Jal, jac_classes(o, jr, jc, Jal, Jal2), TOL, f"{name} (published grid) jac_g")
    assert_by_class(r["grad_f"], c["grad_f"], grad_classes(o), TOL, f"{name} (published grid) grad_f")
    assert_by_class(r["hess_l"], np.asarray(C.hess_matrix(z, p, sig, lam)[hr, hc]).ravel(), hess_classes(o, hr, hc), TOL, f"{name} (published grid) hess_l")


def test_moon_lander_10x6_against_the_reference_golden():
    """tests/golden/nlp_moon_lander_10x6_LGR.npz: the reference's own create_nlp() (over the sympy stand-in for CasADi) at the grid of
    its published timing table -- bounds, initial guess, f, g, grad_f, jac_g, hess_l, nlp_grad."""
    G = load_golden("moon_lander_10x6_LGR")
    ocp = problems.moon_lander(mp, M.math)
    mpo = mp.mpopt(ocp, 10, 6, "LGR")
    nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    for k in ("lbx", "ubx", "lbg", "ubg"):
        assert np.array_equal(np.asarray(bounds[k], float), G[k])
    assert np.array_equal(mpo.initialize_solution(), G["z0"])
    r = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], G["z"], G["p"], lam_g=G["lam"], sigma=float(G["sigma"]))
    assert rel_err(r["f"], G["f"]) < TOL and rel_err(r["g"], G["g"]) < TOL and rel_err(r["grad_f"], G["grad_f"]) < TOL
    J = sp.coo_matrix((r["jac_g"], o.jac_pattern()), shape=(o.n_g, o.n_z)).tocsr()
    Jg = sp.coo_matrix((G["jac_val"], (G["jac_row"], G["jac_col"])), shape=(o.n_g, o.n_z)).tocsr()
    assert abs(J - Jg).max() < TOL * max(1.0, abs(Jg).max())
    assert set(zip(G["jac_row"].tolist(), G["jac_col"].tolist())) <= set(zip(*[a.tolist() for a in o.jac_pattern()]))
    H = sp.coo_matrix((r["hess_l"], o.hess_pattern()), shape=(o.n_z, o.n_z)).tocsr()
    Hg = sp.coo_matrix((G["hess_val"], (G["hess_row"], G["hess_col"])), shape=(o.n_z, o.n_z)).tocsr()
    assert abs(H - Hg).max() < TOL * max(1.0, abs(Hg).max())
    q = o.eval_grad_gamma(G["z"], G["p"], G["lam"], float(G["sigma"]))
    assert rel_err(q["grad_gamma_x"], G["grad_gamma_x"]) < TOL and rel_err(q["grad_gamma_p"], G["grad_gamma_p"]) < TOL

"""End-to-end anchors through mp.solve (GPU oracles + SciPy stand-in for IPOPT, see mpopt_amd/solver.py):
the reference's analytic-solution test and its published optima."""
import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp
import problems

pytestmark = pytest.mark.gpu


def test_analytic_solution():
    """Reference tests/test_mpopt.py:1124-1133: 1 segment, degree 5, |x - (-2t^2+6t+1)| < 1e-6, |u - 2(t-1)| < 1e-6."""
    mp.mpopt._MUTE_ = True
    ocp = problems.analytic_solution(mp, M.math)
    mpo = mp.mpopt(ocp, 1, 5)
    sol = mpo.solve()
    post = mpo.process_results(sol, plot=False)
    x, u, t, _ = post.get_data()
    assert mpo.oracle.has_device
    assert (abs(x - (-2 * t * t + 6 * t + 1)) < 1e-6).all()
    assert (abs(u - 2 * (t - 1)) < 1e-6).all()
    assert set(sol) >= {"f", "g", "lam_g", "lam_p", "lam_x", "x"}  # reference tests/test_examples.py:41-48


def test_moon_lander_published_optimum():
    """Published by the reference: J* = 8.24677 for 20x3 LGR (docs/source/notebooks/getting_started.ipynb:428)
    and 8.2477255075783038 for 10x6 LGR (moon_lander.ipynb:185, IPOPT with acceptable_tol 1e-4)."""
    mp.mpopt._MUTE_ = True
    ocp = problems.moon_lander(mp, M.math)
    mpo, post = mp.solve(ocp, n_segments=20, poly_orders=3, scheme="LGR", plot=False)
    sol = post.solution
    assert abs(sol["f"] - 8.24677) < 1e-5
    g = sol["g"]
    b = mpo.nlp_bounds
    assert (g >= b["lbg"] - 1e-7).all() and (g <= b["ubg"] + 1e-7).all()
    mpo2, post2 = mp.solve(ocp, n_segments=10, poly_orders=6, scheme="LGR", plot=False)
    assert abs(post2.solution["f"] - 8.2477255075783038) < 1e-4
    # warm start from the previous solution converges immediately to the same point
    sol2 = mpo2.solve(initial_solution=post2.solution)
    assert abs(sol2["f"] - post2.solution["f"]) < 1e-7

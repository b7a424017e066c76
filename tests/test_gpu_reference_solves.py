"""The solve-level tests of the reference's own suite (/root/reference/tests/test_mpopt.py) run against
``mpopt_amd.mp`` on the GPU: same fixtures (grids, options), same calls, same assertions, restated compactly.
The outer NLP iteration is the SciPy stand-in for IPOPT (mpopt_amd/solver.py); every oracle value comes from the
HIP kernels.  Line numbers refer to the reference's test file."""
import matplotlib

matplotlib.use("Agg")
import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp
import problems

pytestmark = pytest.mark.gpu


def reference_test_ocp():  # fixture test_ocp, 87-110
    ocp = mp.OCP(n_states=2, n_controls=2, n_phases=2)
    ocp.dynamics = [lambda x, u, t: [u[0], u[0]]] * 2
    ocp.path_constraints = [lambda x, u, t: [x[0] + 1, u[0]]] * 2
    ocp.running_costs = [lambda x, u, t: u[0]] * 2
    ocp.terminal_constraints = [lambda xf, tf, x0, t0: [-xf[0]]] * 2
    ocp.terminal_costs = [lambda xf, tf, x0, t0: tf] * 2
    for phase in range(2):
        ocp.lbu[phase], ocp.ubu[phase] = -1.0, 1.0
        ocp.lbtf[phase], ocp.ubtf[phase] = 1.0, 1.0
    ocp.validate()
    return ocp


def check_solution_and_post(mpo, sol, plots=False):
    """The common tail of the reference's *_solve tests (e.g. 417-428)."""
    for key in ["x", "f"]:
        assert key in sol
    post = mpo.process_results(sol, plot=False)
    if plots:  # 592-594
        for fn in (post.plot_phases, post.plot_x, post.plot_u):
            fig, axs = fn()
            assert fig is not None
        matplotlib.pyplot.close("all")
    x, u, t, _ = post.get_data()
    xi, ui, ti, _ = post.get_data(interpolate=True)
    assert x.shape[0] == u.shape[0] == t.shape[0]
    assert xi.shape[0] == ui.shape[0] == ti.shape[0]
    return post


def test_mpopt_solve():  # 410-413: the 2-phase fixture on the default grid (1 segment of degree 9)
    mp.mpopt._MUTE_ = True
    mpo = mp.mpopt(reference_test_ocp())
    mpo.validate()
    mpo.solve()
    for key in ["lbx", "lbg", "ubx", "ubg"]:
        assert key in mpo.nlp_bounds


def test_moon_lander_mpopt_solve():  # 241-247, 416-428: mp.solve first, then a fresh optimizer with the optional rows
    mp.mpopt._MUTE_ = True
    ocp = problems.moon_lander(mp, M.math)
    mpo, post = mp.solve(ocp, n_segments=20, poly_orders=3, scheme="LGR", plot=False)
    mpo = mp.mpopt(ocp, 20, 3)
    mpo.validate()
    mpo._ocp.diff_u[0] = 1
    mpo._ocp.midu[0] = 0
    mpo._ocp.du_continuity[0] = 1
    sol = mpo.solve()
    check_solution_and_post(mpo, sol)
    assert abs(float(sol["f"]) - 8.2468) < 5e-3  # docs/source/notebooks/getting_started.ipynb:428 (without the slope rows)


@pytest.mark.parametrize("scheme", ["LGR", "LGL", "CGL"])
def test_van_der_pol_mpopt_solve(scheme):  # 564-600: one segment of degree 15, three schemes, plots for all
    mp.mpopt._MUTE_ = True
    mpo = mp.mpopt(problems.van_der_pol(mp, M.math), 1, 15, scheme)
    mpo.validate()
    sol = mpo.solve()
    check_solution_and_post(mpo, sol, plots=True)
    assert mpo.nlp_solver.stats["success"] and abs(float(sol["f"]) - 2.8737) < 2e-3  # the known optimum of this problem


def test_two_phase_schwartz_mpopt_solve_and_residuals():  # 554-562, 729-743
    mp.mpopt._MUTE_ = True
    mpo = mp.mpopt(problems.two_phase_schwartz(mp, M.math), 1, 15, "LGL")
    mpo.validate()
    sol = mpo.solve()
    check_solution_and_post(mpo, sol)
    taus = [mpo.collocation._taus_fn(deg)[1:-1] for deg in mpo.poly_orders]
    for phase, bound in ((0, 1e-1), (1, 1.0)):
        time, residual, _ = mpo.get_dynamics_residuals_single_phase(sol, phase, taus)
        assert max(abs(np.array(err)).max() for err in residual) < bound


def test_mpopt_interpolate_single_phase():  # 660-726
    mp.mpopt._MUTE_ = True
    mpo = mp.mpopt(problems.van_der_pol(mp, M.math), 1, 15, "LGR")
    mpo.validate()
    sol = mpo.solve()
    for nodes in (None, np.array([[mpo.tau0, mpo.tau1] for _ in range(mpo.n_segments)])):
        Xi, Ui, ti, a, DXi, DUi, target_nodes, t0, tf = mpo.interpolate_single_phase(sol, phase=0, target_nodes=nodes)
        assert Xi.size() == DXi.size() and Ui.size() == DUi.size()
        assert a.size() == (mpo._ocp.na, 1)
        assert ti.size() == (sum(len(node) for node in target_nodes), 1)


def test_hyper_sensitive_mpopt_solve():  # 284-289, 486-496: 15 segments of degree 15
    mp.mpopt._MUTE_ = True
    mpo = mp.mpopt(problems.hyper_sensitive(mp, M.math), 15, 15)
    mpo.validate()
    sol = mpo.solve()
    check_solution_and_post(mpo, sol)
    assert mpo.nlp_solver.stats["success"]
    g = mpo.oracle.eval(["g"], sol["x"], mpo._nlp_sw_params)["g"]
    assert (g >= mpo.Gmin - 1e-7).all() and (g <= mpo.Gmax + 1e-7).all()


@pytest.mark.parametrize("grid_type", [None, "mid-points", "spectral"])
def test_moon_lander_h_adaptive_solve(grid_type):  # 249-255, 431-470
    mp.mpopt._MUTE_ = True
    mpo = mp.mpopt_h_adaptive(problems.moon_lander(mp, M.math), 10, 4)
    mpo.validate()
    if grid_type is not None:
        mpo.grid_type[0] = grid_type
    sol = mpo.solve(max_iter=3) if grid_type is None else mpo.solve(max_iter=2, mpopt_options={"method": "residual", "sub_method": "equal_area"})
    check_solution_and_post(mpo, sol)


def test_ph_adaptive_refinement_loop():
    """mpopt_ph_adaptive.solve_ph (the loop the reference sketches at mpopt.py:4422-4596): degrees go up where the
    relative state residual exceeds the tolerance, the grid stays consistent and the cost approaches the optimum."""
    mp.mpopt._MUTE_ = True
    opt = mp.mpopt_ph_adaptive(problems.moon_lander(mp, M.math), n_segments=3, poly_orders=[2] * 3, max_residual=1e-3)
    sol = opt.solve_ph(max_iter=3)
    assert len(opt.poly_orders) == opt.n_segments == len(opt._nlp_sw_params) and abs(sum(opt._nlp_sw_params) - 1) < 1e-12
    assert max(opt.poly_orders) > 2 and opt.n_segments <= opt.max_segments
    assert abs(float(sol["f"]) - 8.2462) < 0.05
    check_solution_and_post(opt, sol)

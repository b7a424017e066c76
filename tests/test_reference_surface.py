"""The reference's own tests (tests/test_mpopt.py) for the classes on the path, run against
``mpopt_amd.mp``: same calls, same assertions.  Line numbers refer to /root/reference/tests/test_mpopt.py."""
import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp
import problems

FIXTURES = {
    "test_ocp": (problems.generic_two_phase, 2, [2, 3]),          # 231-233
    "moon_lander": (problems.moon_lander, 3, 4),
    "van_der_pol": (problems.van_der_pol, 2, [3, 4]),
    "hyper_sensitive": (problems.hyper_sensitive, 4, 3),
    "two_phase_schwartz": (problems.two_phase_schwartz, 3, 3),
}


@pytest.fixture(params=list(FIXTURES))
def test_mpo(request):
    builder, S, po = FIXTURES[request.param]
    mp.mpopt._MUTE_ = True
    mpo = mp.mpopt(builder(mp, M.math), S, po)
    mpo.validate()
    return mpo


def test_mpopt_collocation_basis(test_mpo):  # 333-346
    test_mpo.compute_numerical_approximation()
    poly_orders, Npoints = test_mpo.poly_orders, test_mpo._Npoints
    assert len(test_mpo._taus) == len(set(poly_orders))
    assert test_mpo._compW.shape == (1, Npoints)
    for p in poly_orders:
        assert len(test_mpo._taus[p]) == p + 1
    assert test_mpo._compD.shape == (Npoints, Npoints)
    assert test_mpo._collocation_approximation_computed


def test_mpopt_casadi_variables(test_mpo):  # 349-356
    test_mpo.create_variables()
    for name in ("X", "U", "t0", "tf", "seg_widths"):
        assert getattr(test_mpo, name) is not None


def test_mpopt_ocp_discretization(test_mpo):  # 359-364
    for phase in range(test_mpo._ocp.n_phases):
        G, Gmin, Gmax, J = test_mpo.discretize_phase(phase)
        assert G.shape[0] == Gmin.shape[0] == Gmax.shape[0]
        assert J.shape == (1, 1)


def test_mpopt_event_constraints(test_mpo):  # 367-373
    E, Emin, Emax = test_mpo.get_event_constraints()
    if test_mpo._ocp.n_phases == 1:
        assert E == Emin == Emax == []
    assert len(E) == len(Emin) == len(Emax)


def test_mpopt_nlp_vars_init(test_mpo):  # 376-385
    test_mpo.create_variables()
    for phase in range(test_mpo._ocp.n_phases):
        Z, Zmin, Zmax = test_mpo.get_nlp_variables(phase)
        assert Z.shape[0] == Zmin.shape[0] == Zmax.shape[0] == test_mpo._optimization_vars_per_phase


def test_mpopt_nlp_init(test_mpo):  # 388-400
    nlp_prob, nlp_bounds = test_mpo.create_nlp()
    assert nlp_prob["x"].shape[0] == nlp_bounds["lbx"].shape[0] == nlp_bounds["ubx"].shape[0]
    assert nlp_prob["g"].shape[0] == nlp_bounds["lbg"].shape[0] == nlp_bounds["ubg"].shape[0]
    assert nlp_prob["f"].shape[0] == 1


def test_mpopt_init_solution(test_mpo):  # 403-407
    test_mpo.create_variables()
    Z0 = test_mpo.initialize_solution()
    assert Z0.shape[0] == test_mpo._optimization_vars_per_phase * test_mpo._ocp.n_phases


def test_mpopt_get_residual_grid_taus(test_mpo):  # 637-660 (without the solve: widths are the defaults)
    test_mpo.compute_numerical_approximation()
    test_mpo._nlp_sw_params = test_mpo.get_segment_width_parameters(None)
    for gt in ("fixed", "mid-points", "spectral"):
        taus = test_mpo.get_residual_grid_taus(grid_type=gt)
        taus_1D = np.concatenate(taus)
        # (1e-12 slack: with widths 1/3 the reference's own formula yields 1.0000000000000004, golden-verified)
        assert taus_1D.min() >= test_mpo.tau0 - 1e-12 and taus_1D.max() <= test_mpo.tau1 + 1e-12
        assert len(taus) == test_mpo.n_segments
    assert test_mpo.get_residual_grid_taus(grid_type="do-not-know-any") is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["moon_lander", "van_der_pol", "two_phase_schwartz"])
def test_solve_and_postprocess(name):
    """416-428, 554-602, 678-727: solve, result keys, data shapes, interpolate_single_phase sizes."""
    builder, S, po = FIXTURES[name]
    mp.mpopt._MUTE_ = True
    mpo = mp.mpopt(builder(mp, M.math), S, po)
    if name == "moon_lander":  # 417-419
        mpo._ocp.diff_u[0], mpo._ocp.midu[0], mpo._ocp.du_continuity[0] = 1, 0, 1
    sol = mpo.solve()
    for key in ["x", "f"]:
        assert key in sol
    for key in ["lbx", "lbg", "ubx", "ubg"]:  # 410-413
        assert key in mpo.nlp_bounds
    post = mpo.process_results(sol, plot=False)
    x, u, t, _ = post.get_data()
    xi, ui, ti, _ = post.get_data(interpolate=True)
    assert x.shape[0] == u.shape[0] == t.shape[0]
    assert xi.shape[0] == ui.shape[0] == ti.shape[0]
    Xi, Ui, ti, a, DXi, DUi, target_nodes, t0, tf = mpo.interpolate_single_phase(sol, phase=0)
    assert Xi.size() == DXi.size() and Ui.size() == DUi.size()
    assert a.size() == (mpo._ocp.na, 1)
    assert ti.size() == (sum(len(n) for n in target_nodes), 1)
    ends = np.array([[mpo.tau0, mpo.tau1] for _ in range(mpo.n_segments)])
    Xi, Ui, ti, a, DXi, DUi, target_nodes, t0, tf = mpo.interpolate_single_phase(sol, phase=0, target_nodes=ends)
    assert Xi.size() == DXi.size() and ti.size() == (2 * mpo.n_segments, 1)
    # 730-744: dynamics residuals at a solution are small on this coarse grid
    ti_r, residuals = mpo.get_dynamics_residuals(sol, grid_type="spectral")
    mx = max(np.abs(r).max() for r in residuals[0] if r is not None)
    assert np.isfinite(mx) and mx < 4


@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "schwartz_4x3_LGL", "kitchen_sink_mixed_CGL"])
def test_post_process_original_data_matches_reference(name):
    """post_process.get_data() (mpopt.py:1633-1690, 1833-1858) at the golden sample point against what the
    reference's own process_results(...).get_data() returned (tests/golden/make_golden.py post)."""
    import os
    from helpers import GOLDEN, load_golden

    builder, S, po, scheme = problems.GOLDEN_CASES[name]
    G, P = load_golden(name), np.load(os.path.join(GOLDEN, f"post_{name}.npz"))
    mp.mpopt._MUTE_ = True
    mpo = mp.mpopt(builder(mp, M.math), S, po, scheme)
    mpo.compute_numerical_approximation()
    mpo._nlp_sw_params = G["p"]
    post = mp.post_process({"x": G["z"]}, [mpo.init_trajectories(ph) for ph in range(mpo._ocp.n_phases)],
                           {"phases_to_plot": mpo._ocp.phases_to_plot, "seg_widths": G["p"], "scale_x": mpo._ocp.scale_x,
                            "scale_u": mpo._ocp.scale_u, "scale_a": mpo._ocp.scale_a, "poly_orders": mpo.poly_orders,
                            "colloc_scheme": scheme, "tau0": mpo.tau0, "tau1": mpo.tau1})
    x, u, t, a = post.get_data()
    assert x.shape == P["orig/x"].shape and a.shape == P["orig/a"].shape
    for got, key in ((x, "orig/x"), (u, "orig/u"), (t, "orig/t"), (a, "orig/a")):
        assert got.size == 0 or np.abs(got - P[key]).max() <= 1e-12 * max(1.0, np.abs(P[key]).max()), key
    # host interpolation path (no GPU callable in the options): composite interpolation matrix, like the reference
    xi, ui, ti, ai = post.get_data(interpolate=True)
    assert xi.shape == P["interp/x"].shape and ti.shape == P["interp/t"].shape and ai.shape == P["interp/a"].shape
    for got, key in ((xi, "interp/x"), (ui, "interp/u"), (ti, "interp/t")):
        assert np.abs(got - P[key]).max() <= 1e-10 * max(1.0, np.abs(P[key]).max()), key
    assert np.array_equal(mp.post_process.get_non_uniform_interpolation_grid(np.array([-1.0, -0.2, 0.5, 1.0]), 20), P["grid/non_uniform"])
    import matplotlib
    matplotlib.use("Agg")
    fig, axs = post.plot_phases()
    assert len(axs) == 2 and len(axs[0].lines) >= x.shape[1]
    fig, axs = post.plot_x()
    fig, axs = post.plot_u()
    fig, axs = post.plot_phases(interpolate=False)
    fig, ax = mp.post_process.plot_residuals([[np.array([0.0, 1.0]), None]], [[np.array([[1.0, 2.0], [0.5, 0.1]]), None]])
    assert len(ax.lines) == 2
    matplotlib.pyplot.close("all")


def test_mpopt_block_builders_agree_with_phase_bounds(test_mpo):
    """The reference's per-block builders (mpopt.py:214-413): concatenated in the reference's row order
    [F; C; DU; mU; dU; TC] (mpopt.py:458) they reproduce the bounds of discretize_phase."""
    test_mpo.create_nlp()
    test_mpo.init_segment_width()
    for phase in range(test_mpo._ocp.n_phases):
        f, c, q = test_mpo.get_discretized_dynamics_constraints_and_cost_matrices(phase)
        assert f.shape == (test_mpo._Npoints, test_mpo._ocp.nx) and q.shape == (test_mpo._Npoints, 1)
        F = test_mpo.get_nlp_constraints_for_dynamics(f, phase)
        C = test_mpo.get_nlp_constraints_for_path_contraints(c, phase)
        TC = test_mpo.get_nlp_constraints_for_terminal_contraints(phase)
        assert len(TC) == 4 and TC[3].shape == (1, 1)
        blocks = [F, C, test_mpo.get_nlp_constraints_for_control_input_slope(phase),
                  test_mpo.get_nlp_constrains_for_control_input_at_mid_colloc_points(phase),
                  test_mpo.get_nlp_constrains_for_control_slope_continuity_across_segments(phase), TC[:3]]
        lo = np.concatenate([np.asarray(b[1], float) for b in blocks])
        hi = np.concatenate([np.asarray(b[2], float) for b in blocks])
        G, Gmin, Gmax, J = test_mpo.discretize_phase(phase)
        assert np.array_equal(lo, Gmin) and np.array_equal(hi, Gmax)
        for b in blocks:
            assert (b[0] == [] and len(b[1]) == 0) or b[0].shape[0] == len(b[1])


def test_module_surface_used_by_the_reference_examples():
    """Names the reference's examples/ scripts touch on ``mp`` and on the optimizer objects."""
    import matplotlib
    matplotlib.use("Agg")
    assert mp.plt.__name__ == "matplotlib.pyplot"
    for name in ("OCP", "mpopt", "mpopt_h_adaptive", "mpopt_adaptive", "mpopt_ph_adaptive", "post_process", "solve", "Collocation",
                 "CollocationRoots", "get_segment_boundaries"):
        assert hasattr(mp, name), name
    assert mp.get_segment_boundaries() is None  # (mpopt.py:4311-4313: an empty function)
    mpo = mp.mpopt(problems.moon_lander(mp, M.math), 3, 3)
    nlp, bounds = mpo.create_nlp()
    assert mpo.Z.shape[0] == len(bounds["lbx"]) and mpo.G.shape[0] == len(bounds["lbg"])
    h = mp.mpopt_h_adaptive(problems.moon_lander(mp, M.math), 3, 3)
    for name in ("tol_residual", "plot_residual_evolution", "_THRESHOLD_SLOPE", "_SEG_WIDTH_MIN"):
        assert hasattr(h, name), name
    a = mp.mpopt_adaptive(problems.moon_lander(mp, M.math), 3, 3)
    for name in ("tol_residual", "mid_residuals", "lbh", "ubh"):
        assert hasattr(a, name), name


def test_ph_adaptive_surface_and_max_residual():
    """mpopt_ph_adaptive (mpopt.py:4316-4420): constructor attributes and get_abs_max_residual."""
    mpo = mp.mpopt_ph_adaptive(problems.moon_lander(mp, M.math), n_segments=3, poly_orders=[2] * 3)
    assert (mpo.poly_order_min, mpo.poly_order_max, mpo.min_segments, mpo.max_segments) == (2, 16, 1, 20)
    assert mpo.max_residual == 1e-4 and mpo.grid_type == ["spectral"] and mpo.max_grid_points == [20]
    assert mpo.lbh == [1e-5] and mpo.ubh == [1] and mpo.tol_residual == [1e-4]
    res = [[np.array([[1.0, -3.0], [-2.0, 0.5]]), np.array([[0.1, 0.2]])]]
    out = mp.mpopt_ph_adaptive.get_abs_max_residual(res)
    assert np.array_equal(out[0][0][0], [1, 0]) and np.array_equal(out[0][0][1], [2.0, 3.0]) and np.array_equal(out[0][1][1], [0.1, 0.2])

"""GPU parity against the golden vectors produced by the reference's own transcription code
(tests/golden/make_golden.py).  Tolerance: 1e-10 relative for FP64 values (BASELINE.json
north_star); index data (rows/cols) must match exactly."""
import numpy as np
import pytest

import problems
from helpers import (align_coo, assert_by_class, assert_coo_close, build_case, grad_classes, hess_classes, jac_classes, load_golden,
                     rel_err)

TOL = 1e-10
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(problems.GOLDEN_CASES))
def test_golden_point(name):
    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=True)
    assert o.has_device
    r = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], G["z"], G["p"], lam_g=G["lam"], sigma=float(G["sigma"]))
    assert rel_err(r["f"], G["f"]) < TOL
    assert rel_err(r["g"], G["g"]) < TOL
    assert rel_err(r["grad_f"], G["grad_f"]) < TOL
    jr, jc = o.jac_pattern()
    assert_coo_close(jr, jc, r["jac_g"], G["jac_row"], G["jac_col"], G["jac_val"], TOL, "jac_g")
    hr, hc = o.hess_pattern()
    assert (hr <= hc).all()
    assert_coo_close(hr, hc, r["hess_l"], G["hess_row"], G["hess_col"], G["hess_val"], TOL, "hess_l")
    # ... and PER ENTRY, one floor per entry class (helpers.py): which Jacobian entries are grid constants is read off the numpy
    # oracle (pinned to these goldens at 1e-12, tests/test_oracle.py) evaluated at a second point
    from oracle.mpopt_oracle import OracleNLP

    O = OracleNLP(ocp, *problems.GOLDEN_CASES[name][1:])
    ja = align_coo(jr, jc, G["jac_row"], G["jac_col"], G["jac_val"], "jac_g")
    J2 = O.jac_g(G["z0"], G["p_equal"]).toarray()
    assert_by_class(r["jac_g"], ja, jac_classes(o, jr, jc, ja, J2[jr, jc]), TOL, f"{name} jac_g")
    assert_by_class(r["hess_l"], align_coo(hr, hc, G["hess_row"], G["hess_col"], G["hess_val"], "hess_l"), hess_classes(o, hr, hc), TOL, f"{name} hess_l")
    assert_by_class(r["grad_f"], G["grad_f"], grad_classes(o), TOL, f"{name} grad_f")


@pytest.mark.parametrize("name", list(problems.GOLDEN_CASES))
def test_golden_initial_guess_equal_widths(name):
    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=True)
    z0 = mpo.initialize_solution()
    assert np.array_equal(z0, G["z0"])
    p = np.asarray(mpo.get_segment_width_parameters(None))
    r = o.eval(["f", "g"], z0, p)
    assert rel_err(r["f"], G["f_z0_equal"]) < TOL
    assert rel_err(r["g"], G["g_z0_equal"]) < TOL


@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "kitchen_sink_mixed_CGL"])
def test_batch_matches_single(name):
    """A batch of B points equals B single evaluations bit for bit (fixed-order reductions)."""
    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=True)
    rng = np.random.default_rng(5)
    B = 7
    Z = G["z"][None, :] + 0.01 * rng.standard_normal((B, o.n_z))
    lam = rng.standard_normal((B, o.n_g))
    sig = rng.uniform(0.5, 1.5, B)
    what = ["f", "g", "grad_f", "jac_g", "hess_l"]
    rb = o.eval(what, Z, G["p"], lam_g=lam, sigma=sig)
    for b in range(B):
        r1 = o.eval(what, Z[b], G["p"], lam_g=lam[b], sigma=sig[b])
        for k in what:
            assert np.array_equal(rb[k][b], r1[k]), (k, b)
    # per-point widths
    P = np.stack([np.roll(G["p"].reshape(ocp.n_phases, -1), b, axis=1).ravel() for b in range(B)])
    rp = o.eval(["f", "g", "jac_g"], Z, P)
    for b in range(B):
        r1 = o.eval(["f", "g", "jac_g"], Z[b], P[b])
        for k in ("f", "g", "jac_g"):
            assert np.array_equal(rp[k][b], r1[k]), (k, b)


@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "kitchen_sink_mixed_CGL", "adaptive_kitchen_sink_mixed_LGR"])
def test_casadi_external_entry_points(name):
    """nlp_f / nlp_g / nlp_grad_f / nlp_jac_g / nlp_hess_l with CasADi's generated-code calling
    convention (mpx_casadi.cpp), values in compressed-column order, against the reference goldens."""
    import ctypes
    from mpopt_amd import _lib

    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=True)
    o.make_current()
    L = _lib.lib()
    LL = ctypes.POINTER(ctypes.c_longlong)

    def call(fn, ins, outs):
        arg = (ctypes.c_void_p * len(ins))(*[a.ctypes.data if a is not None else None for a in ins])
        res = (ctypes.c_void_p * len(outs))(*[a.ctypes.data if a is not None else None for a in outs])
        assert getattr(L, fn)(arg, res, None, None, 0) == 0

    def ccs_dense(fn, idx, vals):
        f = getattr(L, fn + "_sparsity_out")
        f.restype, f.argtypes = LL, [ctypes.c_longlong]
        sp = f(idx)
        nrow, ncol = sp[0], sp[1]
        M_ = np.zeros((nrow, ncol))
        k = 0
        for j in range(ncol):
            for q in range(sp[2 + j], sp[2 + j + 1]):
                M_[sp[2 + ncol + 1 + q], j] = vals[k]
                k += 1
        return M_

    z, p, lam, sig = G["z"].copy(), G["p"].copy(), G["lam"].copy(), np.array([float(G["sigma"])])
    f, g, gr = np.zeros(1), np.zeros(o.n_g), np.zeros(o.n_z)
    jv, hv = np.zeros(o.nnz_jac), np.zeros(o.nnz_hess)
    call("nlp_f", [z, p], [f])
    assert rel_err(f[0], G["f"]) < TOL
    call("nlp_g", [z, p], [g])
    assert rel_err(g, G["g"]) < TOL
    f[:] = 0
    call("nlp_grad_f", [z, p], [f, gr])
    assert rel_err(f[0], G["f"]) < TOL and rel_err(gr, G["grad_f"]) < TOL
    g[:] = 0
    call("nlp_jac_g", [z, p], [g, jv])
    Jr = np.zeros((o.n_g, o.n_z))
    Jr[G["jac_row"], G["jac_col"]] = G["jac_val"]
    assert rel_err(g, G["g"]) < TOL and rel_err(ccs_dense("nlp_jac_g", 1, jv), Jr) < TOL
    call("nlp_hess_l", [z, p, sig, lam], [hv])
    Hr = np.zeros((o.n_z, o.n_z))
    Hr[G["hess_row"], G["hess_col"]] = G["hess_val"]
    assert rel_err(ccs_dense("nlp_hess_l", 0, hv), Hr) < TOL
    # opt-in: page-lock the caller's arrays on first sight (CasADi's work vectors persist over a solve): same values
    assert L.mpx_current_pin_buffers(1) == 0
    jv2, hv2, g2 = np.zeros(o.nnz_jac), np.zeros(o.nnz_hess), np.zeros(o.n_g)
    for _ in range(2):
        call("nlp_jac_g", [z, p], [g2, jv2])
        call("nlp_hess_l", [z, p, sig, lam], [hv2])
    assert np.array_equal(jv2, jv) and np.array_equal(hv2, hv) and np.array_equal(g2, g)
    assert L.mpx_current_pin_buffers(0) == 0
    # NULL argument = zeros, NULL result = not requested (CasADi convention)
    call("nlp_jac_g", [z, p], [None, jv])
    call("nlp_hess_l", [z, p, None, None], [hv])
    assert np.abs(hv).max() == 0.0
    o.close()


def test_width_cache_on_host_path():
    """mpx_eval skips the width upload + prefix kernel while p is unchanged (IPOPT's call pattern); a
    changed p, a changed batch size, interleaved residual/device calls must all invalidate correctly."""
    name = "kitchen_sink_mixed_CGL"
    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=True)
    ocp2, mpo2, fresh = build_case(name, with_device=True)
    z, p1 = G["z"], G["p"]
    p2 = np.asarray(mpo.get_segment_width_parameters(None))
    what = ["f", "g", "jac_g"]

    def same(a, b):
        return all(np.array_equal(a[k], b[k]) for k in what)

    seq = [p1, p1, p2, p2, p1, p1]
    for k, p in enumerate(seq):
        r = o.eval(what, z, p)
        ref = fresh.eval(what, z * 1.0, p.copy())  # `fresh` alternates p every other call as well
        assert same(r, ref), k
        if k == 2:  # a residual evaluation in between recomputes the prefix sums with other widths
            plan = o.residual_plan(0, mpo.__class__.get_residual_grid_taus.__get__(mpo)(0, "mid-points") if hasattr(mpo, "collocation") else None)
            plan.eval(z, p1)
    rb = o.eval(what, np.stack([z, z]), p1)  # batch size change
    assert np.array_equal(rb["g"][0], o.eval(what, z, p1)["g"])
    assert rel_err(o.eval(what, z, p1)["g"], G["g"]) < TOL


@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "schwartz_4x3_LGL", "kitchen_sink_mixed_CGL"])
def test_process_results_interpolated_data_on_gpu(name):
    """mpopt.process_results(...).get_data(interpolate=True): the refined-grid states/controls come from the GPU
    interpolation kernel (mpx_resid_*) and equal what the reference's post_process returned (mpopt.py:1773-1831)."""
    import os
    from helpers import GOLDEN

    G, P = load_golden(name), np.load(os.path.join(GOLDEN, f"post_{name}.npz"))
    ocp, mpo, o = build_case(name, with_device=True)
    mpo.create_nlp()
    mpo._nlp_sw_params = G["p"]
    post = mpo.process_results({"x": G["z"]}, plot=False, residual_x=True, residual_dx=True)
    assert set(post.residuals) == {"t_x", "t_dx"} and len(post.residuals["t_dx"][1]) == ocp.n_phases
    x, u, t, a = post.get_data()
    xi, ui, ti, ai = post.get_data(interpolate=True)
    assert xi.shape == P["interp/x"].shape and ui.shape == P["interp/u"].shape and ti.shape == P["interp/t"].shape
    assert rel_err(x, P["orig/x"]) < 1e-12 and rel_err(t, P["orig/t"]) < 1e-12
    assert rel_err(xi, P["interp/x"]) < TOL and rel_err(ui, P["interp/u"]) < TOL and rel_err(ti, P["interp/t"]) < TOL
    assert mpo.process_results({"x": G["z"]}, residual_dx=False).residuals is None
    o.close()


@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "hyper_sensitive_5x3_LGR", "kitchen_sink_mixed_CGL", "dae_vdp_mixed_CGL"])
def test_state_second_derivative_matches_reference(name):
    """mpopt.get_state_second_derivative[_single_phase] (mpopt.py:1238-1358): D2_at.X, D2_at.U on the spectral grid
    and on a ragged custom grid (empty segments), GPU kernel with second-derivative rows vs the reference's output."""
    import os
    from helpers import GOLDEN

    G, D = load_golden(name), np.load(os.path.join(GOLDEN, f"ddx_{name}.npz"))
    ocp, mpo, o = build_case(name, with_device=True)
    mpo.create_nlp()
    mpo._nlp_sw_params = G["p"]
    sol = {"x": G["z"]}
    for ph in range(ocp.n_phases):
        for gt in ("spectral", "custom"):
            key = f"ph{ph}/{gt}"
            ptr = D[key + "/seg_ptr"]
            nodes = [D[key + "/taus"][ptr[s]:ptr[s + 1]] for s in range(len(ptr) - 1)]
            ti, ddx, ddu = mpo.get_state_second_derivative_single_phase(sol, ph, nodes=nodes)
            assert [v is None for v in ddx] == [len(t) == 0 for t in nodes]
            cat = lambda L: np.concatenate([v for v in L if v is not None])
            assert cat(ddx).shape == D[key + "/ddx"].shape and cat(ddu).shape == D[key + "/ddu"].shape
            assert rel_err(cat(ddx), D[key + "/ddx"]) < 1e-9 and rel_err(cat(ddu), D[key + "/ddu"]) < 1e-9
            assert rel_err(cat(ti), D[key + "/ti"]) < TOL
    ti, DDx, DDu = mpo.get_state_second_derivative(sol)  # default spectral grid, all phases
    assert len(DDx) == ocp.n_phases and rel_err(np.concatenate([v for v in DDx[0] if v is not None]), D["ph0/spectral/ddx"]) < 1e-9
    o.close()

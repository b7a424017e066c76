#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in this container.

Usage (build container only; /root/reference does not exist on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

What runs from the reference (imported from /root/reference, never copied):
  * CollocationRoots / Collocation with ``D_MATRIX_METHOD = "numerical"`` (mpopt.py:3706-4276):
    roots, D (order 1 and 2, at nodes and at arbitrary taus), quadrature weights (incl.
    sub-intervals), interpolation matrices and the four composite builders;
  * ``OCP`` + ``mpopt.create_nlp()`` + ``initialize_solution()`` (mpopt.py:574-708) on the
    problems of tests/problems.py.  CasADi is absent from the image, so the module
    ``tests/golden/casadi_shim.py`` (sympy-backed) stands in for it: the reference's
    transcription code builds f, g, x, p as sympy expressions, which this script evaluates and
    differentiates (Jacobian of g, gradient of f, Hessian of sigma*f + lam^T g) with sympy.

The output files hold DATA only (inputs and expected outputs).
"""
import os
import sys
import time

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np
import sympy as sp

import casadi_shim

sys.modules["casadi"] = casadi_shim
sys.path.insert(0, "/root/reference")
import mpopt.mpopt as ref  # noqa: E402  (the reference, unmodified)

import problems  # noqa: E402

ref.mpopt._MUTE_ = True
SCHEMES = ["LGR", "LGL", "CGL"]
DEGREES = [1, 2, 3, 4, 5, 6, 8, 10, 15, 20, 30]


def full(m):
    return np.array(m.full() if hasattr(m, "full") else m, dtype=float)


def make_tables():
    out = {}
    ref.Collocation.D_MATRIX_METHOD = "numerical"
    for tmin, tmax in [(-1, 1), (0, 1)]:
        ref.CollocationRoots._TAU_MIN, ref.CollocationRoots._TAU_MAX = tmin, tmax
        tag = f"t{tmin}_{tmax}".replace("-", "m")
        for scheme in SCHEMES + ["LG"]:
            for deg in DEGREES:
                key = f"{tag}/{scheme}/{deg}"
                if scheme == "LG":
                    # p nodes instead of p+1: roots only; deg=1 raises in the reference
                    if deg >= 2:
                        out[key + "/roots"] = np.asarray(ref.CollocationRoots("LG")._taus_fn(deg), dtype=float)
                    continue
                col = ref.Collocation([deg], scheme)
                roots = np.asarray(col.roots[deg], dtype=float)
                out[key + "/roots"] = roots
                out[key + "/D1"] = full(col.get_diff_matrix(deg))
                out[key + "/D2"] = full(col.get_diff_matrix(deg, order=2))
                out[key + "/w"] = full(col.get_quadrature_weights(deg)).ravel()
                if deg <= 10:
                    mids = (roots[:-1] + roots[1:]) / 2.0
                    out[key + "/mids"] = mids
                    out[key + "/C_mid"] = full(col.get_interpolation_matrix(mids, deg))
                    out[key + "/D1_mid"] = full(col.get_diff_matrix(deg, taus=mids))
                    out[key + "/D2_mid"] = full(col.get_diff_matrix(deg, taus=mids, order=2))
                    ends = np.array([col.tau0, col.tau1], dtype=float)
                    out[key + "/D1_ends"] = full(col.get_diff_matrix(deg, taus=ends))
                    a, b = roots[0], roots[min(2, deg)]
                    out[key + "/w_sub_ab"] = np.array([a, b])
                    out[key + "/w_sub"] = full(col.get_quadrature_weights(deg, tau0=a, tau1=b)).ravel()
        # composite builders
        for scheme in SCHEMES:
            for name, orders in [("20x3", [3] * 20), ("4x5", [5] * 4), ("3_10_3", [3, 10, 3]), ("2_4_3", [2, 4, 3])]:
                col = ref.Collocation(orders, scheme)
                key = f"{tag}/{scheme}/comp_{name}"
                out[key + "/orders"] = np.array(orders)
                out[key + "/compD"] = full(col.get_composite_differentiation_matrix())
                out[key + "/compW"] = full(col.get_composite_quadrature_weights()).ravel()
                taus_mid = [list((col._taus_fn(d)[:-1] + col._taus_fn(d)[1:]) / 2.0) for d in orders]
                out[key + "/compI_mid"] = full(col.get_composite_interpolation_matrix(taus_mid, orders))
                taus_end = [np.array([col.tau0, col.tau1]) for _ in orders]
                out[key + "/compDat_ends"] = full(col.get_composite_interpolation_Dmatrix_at(taus_end, orders, order=1))
    # the symbolic back-end through the shim (sympy AD, exact quadrature) for the 1e-5 cross-check
    ref.CollocationRoots._TAU_MIN, ref.CollocationRoots._TAU_MAX = -1, 1
    ref.Collocation.D_MATRIX_METHOD = "symbolic"
    for scheme in SCHEMES:
        for deg in [3, 5]:
            col = ref.Collocation([deg], scheme)
            out[f"symbolic/{scheme}/{deg}/compD"] = full(col.get_composite_differentiation_matrix())
            out[f"symbolic/{scheme}/{deg}/compW"] = full(col.get_composite_quadrature_weights()).ravel()
    ref.Collocation.D_MATRIX_METHOD = "numerical"
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **out)
    print(f"tables.npz: {len(out)} arrays")


def flat_syms(m):
    return list(m.a.reshape(-1, order="F"))


def make_case(name, builder, n_segments, poly_orders, scheme, adaptive=False):
    """``adaptive``: the reference's ``mpopt_adaptive`` (mpopt.py:2877-3375): the segment widths are part of
    ``x`` and the NLP has no parameters (create_solver pops ``p``, mpopt.py:3190-3192)."""
    t0 = time.time()
    ref.CollocationRoots._TAU_MIN, ref.CollocationRoots._TAU_MAX = -1, 1
    ref.Collocation.D_MATRIX_METHOD = "numerical"
    ocp = builder(ref, casadi_shim)
    mpo = (ref.mpopt_adaptive if adaptive else ref.mpopt)(ocp, n_segments, poly_orders, scheme)
    nlp, bounds = mpo.create_nlp()
    zs, ps = flat_syms(nlp["x"]), ([] if adaptive else flat_syms(nlp["p"]))
    g_exprs = [sp.sympify(e) for e in flat_syms(nlp["g"])]
    f_expr = sp.sympify(nlp["f"].scalar())
    n_z, n_p, n_g = len(zs), len(ps), len(g_exprs)
    z0 = np.asarray(mpo.initialize_solution(), dtype=float)
    lbx, ubx = np.asarray(bounds["lbx"], float), np.asarray(bounds["ubx"], float)
    lbg, ubg = np.asarray(bounds["lbg"], float), np.asarray(bounds["ubg"], float)
    assert len(z0) == n_z == len(lbx) and n_g == len(lbg)

    z, lam, sigma, rng = problems.sample_point(name, n_z, n_p, n_g, z0, lbx, ubx)
    p = np.zeros(0) if adaptive else problems.sample_widths(rng, n_segments, ocp.n_phases)
    p_equal = np.zeros(0) if adaptive else np.asarray(mpo.get_segment_width_parameters(None), dtype=float)
    zpos = {s: i for i, s in enumerate(zs)}
    args = zs + ps

    def ev(exprs, zz, pp):
        fn = sp.lambdify(args, exprs, "math", cse=True)
        return np.array(fn(*list(zz), *list(pp)), dtype=float)

    out = dict(z0=z0, lbx=lbx, ubx=ubx, lbg=lbg, ubg=ubg, z=z, p=p, p_equal=p_equal, lam=lam,
               sigma=np.array(sigma), n_segments=np.array(n_segments),
               poly_orders=np.array(mpo.poly_orders), n_phases=np.array(ocp.n_phases))
    # f, g at the sample point with non-uniform widths and at Z0 with equal widths
    out["g"] = ev(g_exprs, z, p)
    out["f"] = ev([f_expr], z, p)[0]
    out["g_z0_equal"] = ev(g_exprs, z0, p_equal)
    out["f_z0_equal"] = ev([f_expr], z0, p_equal)[0]
    # Jacobian of g (structural triplets)
    jr, jc, je = [], [], []
    for i, e in enumerate(g_exprs):
        for s in sorted(e.free_symbols & set(zs), key=lambda q: zpos[q]):
            d = sp.diff(e, s)
            if d != 0:
                jr.append(i), jc.append(zpos[s]), je.append(d)
    out["jac_row"], out["jac_col"] = np.array(jr), np.array(jc)
    out["jac_val"] = ev(je, z, p)
    # gradient of f
    gsyms = sorted(f_expr.free_symbols & set(zs), key=lambda q: zpos[q])
    gexp = [sp.diff(f_expr, s) for s in gsyms]
    grad = np.zeros(n_z)
    grad[[zpos[s] for s in gsyms]] = ev(gexp, z, p)
    out["grad_f"] = grad
    # Hessian of sigma*f + lam^T g, upper triangle (CasADi's nlp_hess_l convention)
    lag = sigma * f_expr + sum(float(l) * e for l, e in zip(lam, g_exprs))
    hr, hc, he = [], [], []
    lsyms = sorted(lag.free_symbols & set(zs), key=lambda q: zpos[q])
    for s in lsyms:
        d1 = sp.diff(lag, s)
        for s2 in sorted(d1.free_symbols & set(zs), key=lambda q: zpos[q]):
            if zpos[s2] < zpos[s]:
                continue
            d2 = sp.diff(d1, s2)
            if d2 != 0:
                hr.append(zpos[s]), hc.append(zpos[s2]), he.append(d2)
        # constant second derivative w.r.t. itself (d1 linear in s) is caught above because
        # s stays in d1.free_symbols only if non-linear; handle the purely quadratic case:
        if s not in d1.free_symbols:
            d2 = sp.diff(d1, s)
            if d2 != 0:
                hr.append(zpos[s]), hc.append(zpos[s]), he.append(d2)
    out["hess_row"], out["hess_col"] = np.array(hr, dtype=np.int64), np.array(hc, dtype=np.int64)
    out["hess_val"] = ev(he, z, p) if he else np.zeros(0)
    # nlp_grad, the sixth oracle ca.nlpsol derives (mpopt.py:757; "nlp_grad | ... n_eval 1" in every recorded solve,
    # docs/source/notebooks/moon_lander.ipynb:206): gradient of gamma = lam_f * f + lam_g^T g w.r.t. x and w.r.t. the
    # parameters p (the segment widths); -grad_gamma_p at the solution is the lam_p the solver returns (tests/test_examples.py:44-45)
    ggx = np.zeros(n_z)
    ggx[[zpos[s] for s in lsyms]] = ev([sp.diff(lag, s) for s in lsyms], z, p)
    out["grad_gamma_x"] = ggx
    ggp = np.zeros(n_p)
    ppos = {s: i for i, s in enumerate(ps)}
    psyms = sorted(lag.free_symbols & set(ps), key=lambda q: ppos[q])
    if psyms:
        ggp[[ppos[s] for s in psyms]] = ev([sp.diff(lag, s) for s in psyms], z, p)
    out["grad_gamma_p"] = ggp
    np.savez_compressed(os.path.join(HERE, f"nlp_{name}.npz"), **out)
    print(f"nlp_{name}.npz: n_z={n_z} n_g={n_g} nnz_j={len(jr)} nnz_h={len(hr)}  ({time.time()-t0:.1f}s)")


RESIDUAL_CASES = ["moon_lander_20x3_LGR", "hyper_sensitive_5x3_LGR", "kitchen_sink_mixed_CGL", "dae_vdp_mixed_CGL", "schwartz_4x3_LGL"]


def make_residuals(name, builder, n_segments, poly_orders, scheme):
    """Post-solve row: the reference's interpolate_single_phase and
    get_dynamics_residuals_single_phase (mpopt.py:1428-1543) at the golden sample point, for the
    three built-in residual grids and one custom grid."""
    t0 = time.time()
    ref.CollocationRoots._TAU_MIN, ref.CollocationRoots._TAU_MAX = -1, 1
    ref.Collocation.D_MATRIX_METHOD = "numerical"
    G = np.load(os.path.join(HERE, f"nlp_{name}.npz"))
    ocp = builder(ref, casadi_shim)
    mpo = ref.mpopt(ocp, n_segments, poly_orders, scheme)
    mpo.create_nlp()
    mpo._nlp_sw_params = list(G["p"])
    sol = {"x": G["z"]}
    rng = np.random.default_rng(7)
    out = {}
    for ph in range(ocp.n_phases):
        grids = {}
        for gt in ("fixed", "mid-points", "spectral"):
            try:
                grids[gt] = mpo.get_residual_grid_taus(ph, grid_type=gt)
            except ValueError:
                # mpopt.py:1188 wraps a ragged list in np.array(): an object array under the pinned
                # numpy 1.22, an error under numpy >= 1.24 (mixed degrees).  Same content as a list:
                assert gt == "mid-points"
                grids[gt] = [(mpo.collocation._taus_fn(d)[:-1] + mpo.collocation._taus_fn(d)[1:]) / 2.0 for d in mpo.poly_orders]
        custom = []
        for s, d in enumerate(mpo.poly_orders):  # ragged: some segments empty, node values included
            k = [0, 3, 1, 5][s % 4]
            pts = np.sort(rng.uniform(-1, 1, k))
            if s % 3 == 0 and k:
                pts[-1] = 1.0
            custom.append(pts)
        grids["custom"] = custom
        for gt, nodes in grids.items():
            nodes = [np.asarray(t, dtype=float) for t in nodes]
            Xi, Ui, ti, a, DXi, DUi, tg, t0_, tf_ = mpo.interpolate_single_phase(sol, phase=ph, target_nodes=nodes)
            tis, res, dyn = mpo.get_dynamics_residuals_single_phase(sol, ph, target_nodes=nodes)
            key = f"ph{ph}/{gt}"
            out[key + "/seg_ptr"] = np.concatenate([[0], np.cumsum([len(t) for t in nodes])])
            out[key + "/taus"] = np.concatenate(nodes) if sum(len(t) for t in nodes) else np.zeros(0)
            out[key + "/xi"], out[key + "/ui"] = full(Xi), full(Ui)
            out[key + "/dxi"], out[key + "/dui"] = full(DXi), full(DUi)
            out[key + "/ti"] = full(ti).ravel()
            out[key + "/t0tf"] = np.array([t0_[0], tf_[0]])
            out[key + "/resid"] = np.concatenate([np.asarray(r, float) for r in res if r is not None]) if any(r is not None for r in res) else np.zeros((0, ocp.nx))
            out[key + "/dyn"] = np.concatenate([np.asarray(r, float) for r in dyn if r is not None]) if any(r is not None for r in dyn) else np.zeros((0, ocp.nx))
            out[key + "/ti_seg"] = np.concatenate([np.asarray(t, float).ravel() for t in tis]) if len(tis) else np.zeros(0)
            # state-integral residuals (mpopt.py:989-1076)
            xint, uph, tph, rph = mpo.compute_states_from_solution_dynamics(sol, ph, nodes=nodes)
            cat = lambda L, w: np.concatenate([np.asarray(v, float).reshape(-1, w) for v in L if v is not None]) if any(v is not None for v in L) else np.zeros((0, w))
            out[key + "/xint"] = cat(xint, ocp.nx)
            out[key + "/xres"] = cat(rph, ocp.nx)
    np.savez_compressed(os.path.join(HERE, f"resid_{name}.npz"), **out)
    print(f"resid_{name}.npz: {len(out)} arrays ({time.time()-t0:.1f}s)")


POST_CASES = ["moon_lander_20x3_LGR", "schwartz_4x3_LGL", "kitchen_sink_mixed_CGL"]


def make_post(name, builder, n_segments, poly_orders, scheme):
    """post_process data methods (mpopt.py:1633-1858): the reference's
    process_results(...).get_data() and get_data(interpolate=True) at the golden sample point."""
    ref.CollocationRoots._TAU_MIN, ref.CollocationRoots._TAU_MAX = -1, 1
    ref.Collocation.D_MATRIX_METHOD = "numerical"
    G = np.load(os.path.join(HERE, f"nlp_{name}.npz"))
    mpo = ref.mpopt(builder(ref, casadi_shim), n_segments, poly_orders, scheme)
    mpo.create_nlp()
    mpo._nlp_sw_params = list(G["p"])
    post = mpo.process_results({"x": G["z"]}, plot=False, residual_dx=False)
    out = {}
    for tag, interp in (("orig", False), ("interp", True)):
        x, u, t, a = post.get_data(interpolate=interp)
        out[tag + "/x"], out[tag + "/u"], out[tag + "/t"] = np.asarray(x, float), np.asarray(u, float), np.asarray(t, float)
        out[tag + "/a"] = np.asarray(a, float)
    out["grid/non_uniform"] = ref.post_process.get_non_uniform_interpolation_grid(np.array([-1.0, -0.2, 0.5, 1.0]), 20)
    np.savez_compressed(os.path.join(HERE, f"post_{name}.npz"), **out)
    print(f"post_{name}.npz: " + ", ".join(f"{k}{v.shape}" for k, v in out.items()))


def make_second_derivative(name, builder, n_segments, poly_orders, scheme):
    """mpopt.get_state_second_derivative (mpopt.py:1238-1358) at the golden sample point, spectral grid + ragged custom grid."""
    ref.CollocationRoots._TAU_MIN, ref.CollocationRoots._TAU_MAX = -1, 1
    ref.Collocation.D_MATRIX_METHOD = "numerical"
    G = np.load(os.path.join(HERE, f"nlp_{name}.npz"))
    mpo = ref.mpopt(builder(ref, casadi_shim), n_segments, poly_orders, scheme)
    mpo.create_nlp()
    mpo._nlp_sw_params = list(G["p"])
    sol = {"x": G["z"]}
    rng = np.random.default_rng(13)
    out = {}
    for ph in range(mpo._ocp.n_phases):
        grids = {"spectral": mpo.get_residual_grid_taus(ph, grid_type="spectral"),
                 "custom": [np.sort(rng.uniform(-1, 1, [0, 3, 1, 4][s % 4])) for s in range(n_segments)]}
        for gt, nodes in grids.items():
            nodes = [np.asarray(t, float) for t in nodes]
            ti, ddx, ddu = mpo.get_state_second_derivative_single_phase(sol, ph, nodes=nodes)
            key = f"ph{ph}/{gt}"
            out[key + "/seg_ptr"] = np.concatenate([[0], np.cumsum([len(t) for t in nodes])])
            out[key + "/taus"] = np.concatenate(nodes)
            out[key + "/ddx"] = np.concatenate([np.asarray(v, float) for v in ddx if v is not None])
            out[key + "/ddu"] = np.concatenate([np.asarray(v, float) for v in ddu if v is not None])
            out[key + "/ti"] = np.concatenate([np.asarray(v, float).ravel() for v in ti if v is not None])
    np.savez_compressed(os.path.join(HERE, f"ddx_{name}.npz"), **out)
    print(f"ddx_{name}.npz: {len(out)} arrays")


def make_hadaptive():
    """h-adaptive refinement: the reference's static helpers on seeded inputs and its
    width-update rules (mpopt.py:2524-2874) at the golden sample points."""
    out = {}
    H = ref.mpopt_h_adaptive
    rng = np.random.default_rng(11)
    for k in range(6):
        n_seg = int(rng.integers(2, 9))
        res = np.abs(rng.standard_normal(int(rng.integers(n_seg + 3, 60)))) + 1e-3
        out[f"equal_area/{k}/residuals"], out[f"equal_area/{k}/n"] = res, np.array(n_seg)
        out[f"equal_area/{k}/widths"] = np.asarray(H.get_roots_wrt_equal_area(res, n_seg), float)
        mx = np.abs(rng.standard_normal(n_seg)) * 10.0 ** rng.integers(-4, 1, n_seg)
        w = rng.uniform(0.2, 1.0, n_seg)
        w /= w.sum()
        tol = float(10.0 ** rng.integers(-3, 0))
        out[f"merge_split/{k}/max_res"], out[f"merge_split/{k}/w"], out[f"merge_split/{k}/tol"] = mx, w, np.array(tol)
        out[f"merge_split/{k}/widths"] = np.asarray(H.merge_split_segments_based_on_residuals(list(mx), list(w), ERR_TOL=tol), float)
        t_orig = np.sort(rng.uniform(0, 5, 25))
        du = np.abs(rng.standard_normal((25, int(rng.integers(1, 3)))))
        thr = float(rng.uniform(0.2, 1.2))
        times = H.compute_time_at_max_values(None, t_orig, du, threshold=thr)
        out[f"max_values/{k}/t"], out[f"max_values/{k}/du"], out[f"max_values/{k}/thr"] = t_orig, du, np.array(thr)
        out[f"max_values/{k}/times"] = np.asarray(times, float)
        for n_seg2 in (3, 6, 40):
            tt = np.array(times, float).copy()
            if len(tt) == 0:
                continue
            out[f"widths_at_times/{k}/{n_seg2}"] = np.asarray(H.compute_segment_widths_at_times(tt, n_seg2, 0.0, 5.0), float)
    for name in ["hyper_sensitive_5x3_LGR", "moon_lander_20x3_LGR", "kitchen_sink_mixed_CGL", "schwartz_4x3_LGL"]:
        builder, S, po, scheme = problems.GOLDEN_CASES[name]
        G = np.load(os.path.join(HERE, f"nlp_{name}.npz"))
        for method, sub in (("residual", "equal_area"), ("residual", "merge_split"), ("control_slope", None)):
            ref.CollocationRoots._TAU_MIN, ref.CollocationRoots._TAU_MAX = -1, 1
            ref.Collocation.D_MATRIX_METHOD = "numerical"
            mpo = ref.mpopt_h_adaptive(builder(ref, casadi_shim), S, po, scheme)
            mpo.create_nlp()
            mpo._nlp_sw_params = list(G["p"])
            mpo.tol_residual = [1e-3] * mpo._ocp.n_phases
            opts = {"method": method}
            if sub:
                opts["sub_method"] = sub
            w, err = mpo.get_segment_width_parameters({"x": G["z"]}, options=opts)
            out[f"update/{name}/{method}/{sub}/widths"] = np.asarray(w, float)
            out[f"update/{name}/{method}/{sub}/max_error"] = np.array(float(err))
    np.savez_compressed(os.path.join(HERE, "hadaptive.npz"), **out)
    print(f"hadaptive.npz: {len(out)} arrays")


def main():
    only = sys.argv[1:]
    if not only or "tables" in only:
        make_tables()
    for name, (builder, s, po, scheme) in problems.GOLDEN_CASES.items():
        if only and name not in only:
            continue
        if only in (["residuals"], ["hadaptive"], ["adaptive"], ["post"], ["ddx"]):
            continue
        make_case(name, builder, s, po, scheme)
    for name, (builder, s, po, scheme) in problems.ADAPTIVE_CASES.items():
        if only in (["post"], ["ddx"]) or (only and name not in only and "adaptive" not in only):
            continue
        make_case(name, builder, s, po, scheme, adaptive=True)
    if not only or "hadaptive" in only:
        make_hadaptive()
    if only == ["ddx"] or not only:
        for name in RESIDUAL_CASES[:4]:
            make_second_derivative(name, *problems.GOLDEN_CASES[name])
        if only:
            return
    if only == ["post"]:
        for name in POST_CASES:
            make_post(name, *problems.GOLDEN_CASES[name])
        return
    if not only:
        for name in POST_CASES:
            make_post(name, *problems.GOLDEN_CASES[name])
    for name in RESIDUAL_CASES:
        if only and ("resid_" + name) not in only and "residuals" not in only:
            continue
        make_residuals(name, *problems.GOLDEN_CASES[name])


if __name__ == "__main__":
    main()

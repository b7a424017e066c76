"""Host logic without a GPU: C-ABI symbols, structure (sizes, COO patterns, z0, bounds) against the
reference goldens, loud failure without device code, the tracer, and the OCP container."""
import ctypes
import os
import re

import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp, _lib
from mpopt_amd.expr import Tracer
import problems
from helpers import build_case, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_of_the_header():
    hdr = open(os.path.join(ROOT, "include", "mpx.h")).read()
    declared = set(re.findall(r"\b(mpx_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"mpx_problem", "mpx_sizes", "mpx_ctx"}
    L = ctypes.CDLL(_lib.build_library())
    for name in sorted(declared):
        assert hasattr(L, name), f"libmpx.so does not export {name}"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


@pytest.mark.parametrize("name", list(problems.GOLDEN_CASES))
def test_structure_matches_reference(name):
    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=False)
    assert (o.n_z, o.n_g, o.n_p) == (len(G["z0"]), len(G["lbg"]), len(G["p"]))
    nlp, b = mpo.create_nlp()
    for k in ("lbx", "ubx", "lbg", "ubg"):
        assert np.array_equal(b[k], G[k]), k
    assert set(nlp) >= {"f", "x", "g", "p"}
    assert np.array_equal(mpo.initialize_solution(), G["z0"])
    assert np.array_equal(np.asarray(mpo.get_segment_width_parameters(None)), G["p_equal"])
    jr, jc = o.jac_pattern()
    mine = set(zip(jr.tolist(), jc.tolist()))
    ref = set(zip(G["jac_row"].tolist(), G["jac_col"].tolist()))
    assert len(mine) == o.nnz_jac, "duplicate Jacobian entries"
    assert ref <= mine
    # anything extra must sit in a differentiation/interpolation block (an exactly-zero table entry
    # that CasADi drops): same row has other entries in the same column block
    assert len(mine - ref) <= 0.02 * len(ref) + 8
    hr, hc = o.hess_pattern()
    hm = set(zip(hr.tolist(), hc.tolist()))
    assert len(hm) == o.nnz_hess and (hr <= hc).all()
    assert hm == set(zip(G["hess_row"].tolist(), G["hess_col"].tolist()))
    perm, colind = o.ccs_perm("jac")
    assert sorted(perm.tolist()) == list(range(o.nnz_jac)) and colind[-1] == o.nnz_jac
    cols, rows = jc[perm], jr[perm]
    assert (np.diff(cols) >= 0).all() and all((np.diff(rows[colind[j]:colind[j + 1]]) > 0).all() for j in range(0, o.n_z, 7))
    assert o.bytes_fgj == 8 * (2 * o.n_z + o.n_p + o.n_g + o.nnz_jac + 1)


def test_sizes_of_baseline_configs():
    """SURVEY.md section 8 size table (n_z, n_g) for BASELINE.json configs 1, 2, 4, 5."""
    for builder, S, po, scheme, n_z, n_g in [
        (problems.moon_lander, 20, 3, "LGR", 185, 184),
        (problems.moon_lander, 1000, 5, "LGR", 15005, 15004),
        (problems.two_phase_schwartz, 500, 3, "LGL", 2 * 4505, 6003 + 3002 + 4),
        (problems.hyper_sensitive, 4000, 3, "LGR", 24004, 12002),
    ]:
        o = M.NlpFunctions(builder(mp, M.math), S, [po] * S, scheme, with_device=False)
        assert (o.n_z, o.n_g) == (n_z, n_g)


def test_no_cpu_fallback():
    ocp, mpo, o = build_case("moon_lander_20x3_LGR", with_device=False)
    assert not o.has_device
    with pytest.raises(M.MpxError, match="no CPU fallback"):
        o.eval(["f"], mpo.initialize_solution(), np.full(20, 1 / 20))
    # the round-2 entry points refuse just as loudly: sharded exchange, device-side width update; structure queries still work
    o.shard_setup(2, 1)
    n, cuts = o.shard_info(_lib.MPX_JAC)
    assert n > 0 and cuts[0] == 0 and cuts[-1] == o.n_tiles
    tab = o.shard_table(_lib.MPX_HESS)
    assert tab.shape[1] == 6 and set(tab[:, 0]) <= {0, 1} and set(tab[:, 1]) <= {0, 2}
    buf = np.zeros(8)
    with pytest.raises(M.MpxError, match="no CPU fallback"):
        o.shard_pack(_lib.MPX_JAC, 1, buf, buf)
    with pytest.raises(M.MpxError, match="mpx_set_tile_range on a context in segment-sharded mode"):
        o.set_tile_range(0, 1)
    o.shard_setup(1, 0)
    o.set_tile_range(0, o.n_tiles)
    with pytest.raises(M.MpxError, match="no CPU fallback"):
        o.equal_area_widths_device(0, 1, 60, buf, buf, buf)
    o.geometry_reset()


def test_bad_inputs_are_reported_not_fatal():
    ocp = problems.moon_lander(mp, M.math)
    with pytest.raises(ValueError):
        M.NlpFunctions(ocp, 3, [3, 3], "LGR", with_device=False)
    with pytest.raises(M.MpxError):
        M.NlpFunctions(ocp, 2, [3, 0], "LGR", with_device=False)
    L = _lib.lib()
    assert L.mpx_create(None, None) < 0 and b"null" in L.mpx_last_error(None)


def test_every_degree_up_to_255_creates_and_the_lds_limit_applies_to_lds_tables_only(monkeypatch):
    """Round 6: degrees above MPX_TABLES_STREAM_ABOVE (68) stream their tables, so no degree <= 255 is refused for LDS (up to round 5:
    every degree >= 94, although the reference documents 1 x 100, getting_started.ipynb:743).  With the threshold forced above a
    degree whose tables cannot fit, the old message still names the cause."""
    ocp = problems.moon_lander(mp, M.math)
    for S, po in ((1, [60]), (1, [100]), (4, [128] * 4), (3, [3, 100, 3]), (1, [255])):
        o = M.NlpFunctions(ocp, S, po, "LGR", with_device=False)
        assert o.n_nodes == sum(po) + 1
        o.close()
    with pytest.raises(M.MpxError, match="outside 1..255"):
        M.NlpFunctions(ocp, 1, [256], "LGR", with_device=False)
    monkeypatch.setenv("MPX_TABLES_STREAM_ABOVE", "255")
    with pytest.raises(M.MpxError, match="LDS"):
        M.NlpFunctions(ocp, 1, [120], "LGR", with_device=False)


def test_tracer_derivatives_against_sympy():
    import sympy as sp

    tr = Tracer()
    x, y, z = tr.var("x"), tr.var("y"), tr.var("z")
    m = M.math
    e = (x * y - m.sin(z * x)) / (1.0 + y * y) + m.exp(-0.3 * x) * m.sqrt(2.0 + z * z) + x ** 3 - (y ** 2.5) + m.tanh(x * z) + m.log(3.0 + y)
    X, Y, Z = sp.symbols("x y z")
    se = (X * Y - sp.sin(Z * X)) / (1.0 + Y * Y) + sp.exp(-0.3 * X) * sp.sqrt(2.0 + Z * Z) + X ** 3 - (Y ** 2.5) + sp.tanh(X * Z) + sp.log(3.0 + Y)
    env = {"x": 0.7, "y": 1.3, "z": -0.4}
    sub = {X: 0.7, Y: 1.3, Z: -0.4}
    vs, svs = [x, y, z], [X, Y, Z]
    assert tr.evaluate([e], env)[0] == pytest.approx(float(se.subs(sub)), rel=1e-14)
    for v, sv in zip(vs, svs):
        d = tr.diff(e, v)
        assert tr.evaluate([d], env)[0] == pytest.approx(float(sp.diff(se, sv).subs(sub)), rel=1e-12)
        for w, sw in zip(vs, svs):
            d2 = tr.diff(d, w)
            assert tr.evaluate([d2], env)[0] == pytest.approx(float(sp.diff(se, sv, sw).subs(sub)), rel=1e-11, abs=1e-13)
    # structural zeros are exact
    assert tr.diff(x * y, z).is_zero and tr.diff(tr.diff(x * y + z, x), x).is_zero
    with pytest.raises(TypeError):
        bool(x > 0) if hasattr(x, "__gt__") else bool(x)


@pytest.mark.parametrize("nx,nu,nph,na", [(1, 1, 1, 0), (2, 1, 1, 0), (3, 2, 2, 0), (2, 2, 3, 2), (7, 3, 4, 1)])
def test_ocp_defaults(nx, nu, nph, na):
    """Mirror of the reference's OCP shape/default checks (tests/test_mpopt.py:28-85)."""
    ocp = mp.OCP(n_states=nx, n_controls=nu, n_phases=nph, n_params=na)
    ocp.validate()
    assert (ocp.nx, ocp.nu, ocp.n_phases, ocp.na) == (nx, nu, nph, na)
    assert ocp.x00.shape == (nph, nx) and ocp.u00.shape == (nph, nu) and ocp.a0.shape == (nph, na)
    assert ocp.lbt0[0] == 0 and ocp.ubt0[0] == 0 and (ocp.tf0 == 1).all()
    assert np.isinf(ocp.lbx).all() and np.isinf(ocp.ubu).all()
    assert ocp.phase_links == [(i, i + 1) for i in range(nph - 1)]
    assert (ocp.midu == 1).all() and (ocp.diff_u == 0).all() and (ocp.du_continuity == 0).all()
    assert not ocp.has_path_constraints(0) and not ocp.has_terminal_constraints(0)
    assert len(ocp.get_dynamics(0)(ocp.x00[0], ocp.u00[0], 0.0, ocp.a0[0])) == nx


def test_generated_source_is_deterministic_and_cached():
    ocp = problems.kitchen_sink(mp, M.math)
    a = M.NlpFunctions(ocp, 3, [2, 4, 3], "CGL", with_device=False)
    b = M.NlpFunctions(problems.kitchen_sink(mp, M.math), 3, [2, 4, 3], "CGL", with_device=False)
    assert a.source == b.source and (a.structure == b.structure).all()
    co1, path1 = _lib.compile_kernels(a.source)
    co2, path2 = _lib.compile_kernels(b.source)
    assert path1 == path2 and co1 == co2 and (co1[:4] == b"\x7fELF" or co1.startswith(b"__CLANG_OFFLOAD_BUNDLE__"))


@pytest.mark.parametrize("nx,nu,expect_low", [(20, 6, False), (14, 6, False), (8, 3, True), (2, 1, True)])
def test_wide_problems_compile_and_get_low_degree_light_kernels_only_when_they_fit(nx, nu, expect_low):
    """Round-4 advisor finding: MPX_INSTANTIATE_LIGHT_LOW was emitted for every single-degree grid of degree <= 12, and
    light_low_body's span rows (4 wavefronts x (nx + nu) rows) do not fit the LDS of a workgroup from ~24 inputs on -- the code
    object of OCP(n_states=20, n_controls=6) no longer compiled, create_nlp on a GPU raised.  The generator now instantiates them
    only where the host's plan would use them (>= 2 chunks per span, the same arithmetic as mpx_layout.cpp / mpx_kernels.h), and every
    width compiles (hipcc cross-compiles gfx950 without a GPU)."""
    ocp = mp.OCP(n_states=nx, n_controls=nu)
    ocp.dynamics[0] = lambda x, u, t: [u[i % nu] - 0.1 * x[i] * x[(i + 1) % nx] for i in range(nx)]
    ocp.running_costs[0] = lambda x, u, t: sum(x[i] * x[i] for i in range(nx)) + sum(u[i] * u[i] for i in range(nu))
    ocp.validate()
    o = M.NlpFunctions(ocp, 3, [3] * 3, "LGR", with_device=False)
    assert ("MPX_INSTANTIATE_LIGHT_LOW(0, 3)" in o.source) == expect_low
    assert (o.light_plan()[1] > 0) == expect_low  # the host's plan and the generator agree
    co, path = _lib.compile_kernels(o.source)
    assert co[:4] == b"\x7fELF" or co.startswith(b"__CLANG_OFFLOAD_BUNDLE__")
    o.close()


@pytest.mark.parametrize("name,expected", [("moon_lander", 0), ("hyper_sensitive", 0), ("van_der_pol", 0), ("two_phase_schwartz", 0),
                                           ("time_dependent", 1), ("kitchen_sink", 1)])
def test_generated_source_says_whether_a_node_function_uses_time(name, expected):
    """``mpx_time_dependent`` of the code object: libmpx forms the prefix sums of the segment widths (the node times' only other input)
    only for problems whose dynamics / path constraints / running costs use t (mpopt.py:192-198 is where t comes from)."""
    o = M.NlpFunctions(getattr(problems, name)(mp, M.math), 4, [3] * 4, "LGR", with_device=False)
    m = re.search(r"mpx_time_dependent = (\d);", o.source)
    assert m and int(m.group(1)) == expected


CASADI_FUNCS = {"nlp_f": (2, 1), "nlp_g": (2, 1), "nlp_grad_f": (2, 2), "nlp_jac_g": (2, 2), "nlp_hess_l": (4, 1)}


def _sparsity(ptr):
    nrow, ncol = ptr[0], ptr[1]
    colind = [ptr[2 + j] for j in range(ncol + 1)]
    rows = [ptr[2 + ncol + 1 + k] for k in range(colind[-1])]
    return nrow, ncol, colind, rows


def test_casadi_external_surface_metadata():
    """The five oracles with CasADi's generated-code convention: counts, names, CCS
    sparsities consistent with the COO patterns; evaluation without device code fails with rc != 0."""
    ocp, mpo, o = build_case("kitchen_sink_mixed_CGL", with_device=False)
    L = ctypes.CDLL(_lib.build_library())
    o.make_current()
    LL = ctypes.POINTER(ctypes.c_longlong)
    for name, (nin, nout) in CASADI_FUNCS.items():
        for suffix in ("", "_n_in", "_n_out", "_name_in", "_name_out", "_sparsity_in", "_sparsity_out", "_work", "_incref", "_decref", "_alloc_mem", "_init_mem", "_free_mem",
                       "_checkout", "_release", "_default_in"):
            assert hasattr(L, name + suffix), name + suffix
        getattr(L, name + "_n_in").restype = ctypes.c_longlong
        getattr(L, name + "_n_out").restype = ctypes.c_longlong
        assert getattr(L, name + "_n_in")() == nin and getattr(L, name + "_n_out")() == nout
        getattr(L, name + "_name_in").restype = ctypes.c_char_p
        getattr(L, name + "_name_in").argtypes = [ctypes.c_longlong]
        assert getattr(L, name + "_name_in")(0) == b"x" and getattr(L, name + "_name_in")(1) == b"p"
        getattr(L, name + "_sparsity_in").restype = LL
        getattr(L, name + "_sparsity_in").argtypes = [ctypes.c_longlong]
        assert _sparsity(getattr(L, name + "_sparsity_in")(0))[:2] == (o.n_z, 1)
        assert _sparsity(getattr(L, name + "_sparsity_in")(1))[:2] == (o.n_p, 1)
    L.nlp_hess_l_name_in.restype = ctypes.c_char_p
    assert L.nlp_hess_l_name_in(ctypes.c_longlong(2)) == b"lam_f" and L.nlp_hess_l_name_in(ctypes.c_longlong(3)) == b"lam_g"
    for fn, idx, pat, shape in (("nlp_jac_g", 1, o.jac_pattern(), (o.n_g, o.n_z)), ("nlp_hess_l", 0, o.hess_pattern(), (o.n_z, o.n_z))):
        f = getattr(L, fn + "_sparsity_out")
        f.restype, f.argtypes = LL, [ctypes.c_longlong]
        nrow, ncol, colind, rows = _sparsity(f(idx))
        assert (nrow, ncol) == shape and colind[-1] == len(pat[0])
        got = set()
        for j in range(ncol):
            seg = rows[colind[j]:colind[j + 1]]
            assert seg == sorted(seg) and len(set(seg)) == len(seg)
            got |= {(r, j) for r in seg}
        assert got == set(zip(pat[0].tolist(), pat[1].tolist()))
        if fn == "nlp_hess_l":
            assert all(r <= c for r, c in got)  # upper triangle, like triu:hess:gamma:x:x
    # no device code: the CasADi entry points report failure (non-zero), they do not compute on the CPU
    x, p = np.zeros(o.n_z), np.full(o.n_p, 0.5)
    arg = (ctypes.c_void_p * 2)(x.ctypes.data, p.ctypes.data)
    fval = np.zeros(1)
    res = (ctypes.c_void_p * 1)(fval.ctypes.data)
    assert L.nlp_f(arg, res, None, None, 0) != 0
    o.close()
    assert L.nlp_f(arg, res, None, None, 0) != 0  # no current context any more


def test_assembled_context_structure_without_gpu():
    """mpx_create_assembled (mpopt_adaptive): sizes and patterns of a structure-only context equal the
    reference's structural sparsity; evaluation without a code object fails loudly."""
    import mpopt_amd as M
    from mpopt_amd import mp
    from mpopt_amd.adaptive import build_adaptive_oracle
    from mpopt_amd.mpopt import Collocation
    from helpers import load_golden

    for name in ("adaptive_moon_lander_3x2_LGR", "adaptive_generic_two_phase_LGR"):
        builder, S, po, scheme = problems.ADAPTIVE_CASES[name]
        G = load_golden(name)
        orc, lay = build_adaptive_oracle(builder(mp, M.math), S, po, Collocation(po, scheme), with_device=False)
        assert (orc.n_z, orc.n_g, orc.n_p) == (len(G["z"]), len(G["g"]), 0)
        jr, jc = orc.jac_pattern()
        assert set(zip(jr.tolist(), jc.tolist())) == set(zip(G["jac_row"].tolist(), G["jac_col"].tolist()))
        hr, hc = orc.hess_pattern()
        assert set(zip(hr.tolist(), hc.tolist())) == set(zip(G["hess_row"].tolist(), G["hess_col"].tolist()))
        perm, colind = orc.ccs_perm("jac")
        assert colind[-1] == orc.nnz_jac and np.array_equal(np.sort(perm), np.arange(orc.nnz_jac))
        assert (np.diff(jc[perm]) >= 0).all()  # perm sorts the source-major stored order into compressed-column order
        with pytest.raises(M.MpxError):
            orc.eval(["f"], G["z"], None)
        # the term offsets of every point set that the generated source holds as compile-time constants (mpxgen::SetT, fused kernels)
        # are the running sums of the ELL term counts the host receives
        assert f"#define MPX_FUSE_SETS {len(orc.sets)}" in orc.source
        for k, (s_, e) in enumerate(zip(orc.sets, orc._ell)):
            fid, lt, mt = orc._set_consts[k]
            assert fid == orc.functions.index(s_.fn) and lt[0] == 0 and mt[0] == 0
            assert np.array_equal(np.diff(lt), np.asarray(e[0][0])) and np.array_equal(np.diff(mt), np.asarray(e[1][0]))
            assert f"template <> struct SetT<{k}>" in orc.source and "{" + ", ".join(str(int(x)) for x in lt) + "}" in orc.source
        orc.close()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: importing every module of the product must not pull it in, and no product
    source file may mention the package in an import."""
    import subprocess
    import sys

    code = ("import sys; import mpopt_amd, mpopt_amd.adaptive, mpopt_amd.assembly, mpopt_amd.solver, mpopt_amd.ipm, "
            "mpopt_amd.distributed, mpopt_amd.codegen, mpopt_amd.nlp; "
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]; assert not bad, bad")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mpopt_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), fn


def test_numpy_functions_on_traced_symbols():
    """The reference's examples apply numpy functions to CasADi symbols (np.sqrt / np.cos / np.sin / np.dot, 190 uses
    under examples/): the tracer accepts the same statements and produces the SAME program as the explicit spelling."""
    from mpopt_amd.codegen import ProblemProgram

    a = ProblemProgram(problems.ascent_numpy_style(mp, M.math, use_numpy=True), [4], [True])
    b = ProblemProgram(problems.ascent_numpy_style(mp, M.math, use_numpy=False), [4], [True])
    assert np.array_equal(a.structure(), b.structure())
    tr = Tracer()
    x, y = tr.var("x"), tr.var("y")
    env = {"x": 0.3, "y": 0.7}
    import math
    for e, want in ((np.arctan2(y, x), math.atan2(0.7, 0.3)), (np.maximum(x * x, y), 0.7), (np.minimum(x, y) * 2, 0.6),
                    (np.dot(np.array([x, y]), [2.0, 3.0]), 2.7), ((np.array([[1.0, 2.0], [3.0, 4.0]]) @ np.array([x, y]))[1], 3.7),
                    (np.sin(np.array([x, y]))[1], math.sin(0.7)), ((x - np.array([1.0, 2.0]))[1], -1.7), (M.math.norm_2([x, y]), math.sqrt(0.58)),
                    (M.math.vertcat(x, [y, 1.0])[1], 0.7), (np.power(x, 3), 0.027), (np.abs(-x), 0.3)):
        assert abs(tr.evaluate([tr.wrap(e)], env)[0] - want) < 1e-14
    d = tr.diff(np.arctan2(y, x), x)
    assert abs(tr.evaluate([d], env)[0] + 0.7 / 0.58) < 1e-14


def test_mixed_degree_row_span_plan(monkeypatch):
    """Mixed-degree grids: the tiles of the bucket with the most nodes cut the phase's nodes into contiguous spans and store
    the g / grad_f rows of their span themselves (mpx_get_tile_spans); single-degree grids and grids outside the limits of the
    scheme (a tile would have to fetch more than 256 foreign nodes, or need more than 150 KB of LDS per workgroup) keep none."""
    def spans(builder, S, po, scheme):
        ocp = builder(mp, M.math)
        mpo = mp.mpopt(ocp, S, po, scheme)
        mpo.compute_numerical_approximation()
        o = M.NlpFunctions(ocp, S, mpo.poly_orders, scheme, tau0=mpo.tau0, tau1=mpo.tau1, with_device=False)
        lo, ln, nf = o.tile_spans()
        return o, lo, ln, nf, int(np.sum(mpo.poly_orders)) + 1

    # config 3's pattern: degree-30 tiles absorb the degree-3 segments (and node 0) between them
    o, lo, ln, nf, N = spans(problems.van_der_pol, 48, [30 if s % 3 == 1 else 3 for s in range(48)], "CGL")
    act = ln > 0
    assert act.sum() == 2 and lo[act][0] == 0 and (lo[act][1:] == (lo[act] + ln[act])[:-1]).all() and (lo[act] + ln[act])[-1] == N
    assert nf[act].sum() == N - 16 * 30 and (nf[~act] == 0).all()
    # two phases, segment 0 of the most populated degree: node 0 is staged by its mini tile and fetched like a foreign node
    o, lo, ln, nf, N = spans(problems.kitchen_sink, 40, [5, 2, 3, 4] * 10, "LGR")
    act = ln > 0
    assert act.sum() == 2 * 1 and (ln[act] == N).all() and (nf[act] == N - 50).all()
    # single degree: nothing to absorb
    o, lo, ln, nf, N = spans(problems.moon_lander, 30, 4, "LGR")
    assert not ln.any() and not nf.any()
    # halves of different degree: one tile would fetch > 256 foreign nodes -> the unpack pass stays
    o, lo, ln, nf, N = spans(problems.van_der_pol, 400, [3] * 200 + [6] * 200, "LGL")
    assert not ln.any()
    # 7 states + 3 controls: the node kernel's own state/control tile + the span rows exceed the 64 KB a launch gets by default -- since
    # round 4 the dynamic LDS limit of the kernels is raised (up to 150 KB per workgroup) instead of falling back to the unpack pass
    o, lo, ln, nf, N = spans(problems.staged_ascent, 30, [3, 4, 3] * 10, "LGR")
    assert ln.any() and o.notes() == []
    monkeypatch.setenv("MPX_NO_ABSORB", "1")
    o, lo, ln, nf, N = spans(problems.van_der_pol, 48, [30 if s % 3 == 1 else 3 for s in range(48)], "CGL")
    assert not ln.any()


@pytest.mark.parametrize("builder,S,P,scheme,expect", [
    (problems.moon_lander, 20, 5, "LGR", True), (problems.hyper_sensitive, 12, 4, "LGL", True), (problems.generic_two_phase, 6, 4, "LGR", True),
    (problems.kitchen_sink, 6, 4, "LGR", False), (problems.moon_lander, 3, 2, "LGR", False), (problems.time_dependent, 10, 3, "LGR", False)])
def test_lane_plan_of_assembled_hessians(builder, S, P, scheme, expect, monkeypatch):
    """mpopt_amd/assembly_lanes.py: the point tasks of an mpopt_adaptive transcription fall into groups (a collocation segment each)
    unless every row couples everything (time-dependent dynamics: global rows, dropped for hess_l) or the problem is tiny.
    Structure of a plan: every entry of hess_l in exactly one group, group by group in the pattern's order; every raw value a row
    reads belongs to a task of ITS group (own or halo); the group's columns of z / lam_g cover what its tasks read for those
    entries; the generated source has one LaneGrpHES per group SHAPE and the pattern is still the same SET of (row, col) pairs as without
    the plan.  MPX_LANES_FGJ=1 (opt-in: the first-order pass the same way, with its global rows f, d f / d t0, d f / d tf through the
    scratch array): the same checks on the rows of g / grad_f / jac_g, every row in one group or global."""
    from mpopt_amd import assembly_lanes

    assert assembly_lanes._pieces([(10, 5, 0), (40, 138, 5)]) == [(10, 2, 0), (14, 0, 4), (40, 6, 5), (104, 6, 69), (168, 3, 133), (176, 1, 141)]  # <= 64 columns a piece
    monkeypatch.setenv("MPX_LANES_FGJ", "1")
    mpo = mp.mpopt_adaptive(builder(mp, M.math), S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    pl = o.lanes_plan
    assert (pl is not None) == expect
    monkeypatch.setenv("MPX_NO_LANES_CODE", "1")
    o2 = mp.mpopt_adaptive(builder(mp, M.math), S, P, scheme).create_nlp()[0]["oracle"]
    monkeypatch.delenv("MPX_NO_LANES_CODE")
    assert o2.lanes_plan is None and o2.lanes_plan_fgj is None and o2.lanes_source is None and "mpx_asml" not in o2.source
    key = lambda r, c: sorted(zip(np.asarray(r).tolist(), np.asarray(c).tolist()))
    assert key(*o.hess_pattern()) == key(*o2.hess_pattern()) and key(*o.jac_pattern()) == key(*o2.jac_pattern())
    assert "mpx_asml" not in o.source and "LaneGrp" not in o.source  # (the lane kernels are a translation unit of their own, attached at the first batch)
    if pl is None and o.lanes_plan_fgj is None:
        assert o.lanes_source is None
        return
    for plan in (pl, o.lanes_plan_fgj):
        if plan is None:
            continue
        Pq = assembly_lanes.Pass(o, plan.kind)
        ptr, src = Pq.ptr, Pq.src
        KIND = plan.kind.upper()
        seen = set(plan.global_rows)
        assert len(seen) == len(plan.global_rows) and (plan.kind == "fgj" or not seen)
        arr_rows = {a: [] for a in range(len(Pq.arrays))}
        for gi, g in enumerate(plan.groups):
            tasks = set(g["tasks"])
            for r in g["rows"]:
                assert r not in seen
                seen.add(r)
                arr_rows[Pq.array_of(r)[0]].append(r)
                for x in src[ptr[r]:ptr[r + 1]]:
                    if x >= 0:
                        kp = Pq.tasks[Pq.task_of[x]]
                        assert kp in tasks
                        lv, mv = g["use"][kp]
                        dl, dm = Pq.deps(o.sets[kp[0]].fn)[int(Pq.val_of[x])]
                        assert dl <= lv and dm <= mv
                    elif x <= -2:
                        assert -2 - int(x) in set(g["zcols"])
            zc = set(g["zcols"])
            for (k, p), (lv, _) in g["use"].items():
                s_ = o.sets[k]
                for v in lv:
                    lo, hi = s_.L.indptr[p * s_.fn.n_loc + v], s_.L.indptr[p * s_.fn.n_loc + v + 1]
                    assert {int(c) for c, d in zip(s_.L.indices[lo:hi], s_.L.data[lo:hi]) if d != 0} <= zc
            for kp, lst in g["scratch"].items():  # scratch slots are written by the group that owns the task, once
                assert kp in g["own"]
        assert seen == set(range(Pq.n_rows))  # every row: one group, or global
        # the reordered array (hess_l / jac_g): group by group, the global rows last
        last = len(Pq.arrays) - 1
        rows_last = arr_rows[last] + [r for r in plan.global_rows if Pq.array_of(r)[0] == last]
        assert rows_last == list(range(Pq.arrays[last][1], Pq.arrays[last][1] + Pq.arrays[last][2]))
        written = [sd for g in plan.groups for lst in g["scratch"].values() for _, sd in lst]
        assert sorted(written) == list(range(len(plan.sid)))
        text = o.lanes_source_text()
        assert f"#define MPX_LANE_{KIND}_GROUPS {len(plan.groups)}" in text and f"MPX_INSTANTIATE_LANES({plan.kind}, {KIND}," in text
        # one struct per group SHAPE (round 6), every group an entry of the shape / parameter tables
        assert 1 <= plan.n_shapes <= len(plan.groups) and f"#define MPX_LANE_{KIND}_SHAPES {plan.n_shapes}" in text
        for sh in range(plan.n_shapes):
            assert f"struct LaneGrp{KIND}<{sh}>" in text
        assert f"struct LaneGrp{KIND}<{plan.n_shapes}>" not in text


def test_lane_kernels_are_one_body_per_group_shape_not_per_segment():
    """Round 6: equal-degree interior segments share ONE generated body -- what differs between them (where their
    columns of z / lam_g and their rows of hess_l start) is a row of the code object's parameter table.  Moon lander 100 x 3 was 67
    bodies, 32 527 lines, 48.8 s of hipcc and a 459 KB code object; now a handful of shapes whatever the number of segments, and
    the source stops growing with the grid."""
    sizes = {}
    for S in (20, 100):
        o = mp.mpopt_adaptive(problems.moon_lander(mp, M.math), S, 3, "LGR").create_nlp()[0]["oracle"]
        assert o.lanes_plan is not None
        text = o.lanes_source_text()
        sizes[S] = (len(o.lanes_plan.groups), o.lanes_plan.n_shapes, len(text))
        o.close()
    assert sizes[100][0] >= 4 * sizes[20][0] and sizes[100][1] <= 6 and sizes[20][1] <= 6
    assert sizes[100][2] < 1.5 * sizes[20][2] and sizes[100][2] < 300 * 1024


def test_lane_kernel_failure_falls_back_once(monkeypatch):
    """whatever goes wrong when the lane kernels are generated / compiled / attached must not reach the caller of eval --
    the fused kernels serve the call, the failure is recorded once and not retried."""
    import warnings

    o = mp.mpopt_adaptive(problems.moon_lander(mp, M.math), 20, 3, "LGR").create_nlp()[0]["oracle"]
    o.code_object = b"x"  # (pretend there is a device: attach_lane_kernels returns early without one)
    monkeypatch.setattr(type(o), "LANES_MAX_SOURCE_BYTES", 1000)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        o.attach_lane_kernels()
        o.attach_lane_kernels()
    assert len(w) == 1 and "exceeds" in o.lanes_error and o.lanes_source is None and not o._wants_lanes(16, 4096)
    o.code_object = None
    o.close()


@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "dae_vdp_mixed_CGL", "schwartz_4x3_LGL", "kitchen_sink_mixed_CGL"])
def test_jac_variable_mask_marks_only_constants_as_constant(name):
    """mpx_pattern_jac_variable (round 6): every entry the library calls a constant of the grid has the same value at two unrelated
    points (z, p) of the reference's golden NLP / the numpy oracle; the constants are the bulk of the Jacobian; terminal rows and the
    variable node entries are marked variable."""
    import ctypes
    from oracle.mpopt_oracle import OracleNLP

    builder, S, po, scheme = problems.GOLDEN_CASES[name]
    ocp = builder(mp, M.math)
    o = M.NlpFunctions(ocp, S, [po] * S if isinstance(po, int) else list(po), scheme, with_device=False)
    var = np.zeros(o.nnz_jac, np.uint8)
    assert _lib.lib().mpx_pattern_jac_variable(o._ctx, var.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))) == 0
    O = OracleNLP(ocp, S, po, scheme)
    rng = np.random.default_rng(2)
    jr, jc = o.jac_pattern()
    vals = []
    for _ in range(2):
        z = O.initial_guess() + rng.uniform(-0.3, 0.3, o.n_z)
        w = rng.uniform(0.3, 1.7, (ocp.n_phases, S))
        J = O.jac_g(z, (w / w.sum(axis=1, keepdims=True)).ravel()).tocsr()
        vals.append(np.asarray(J[jr, jc]).ravel())
    const = var == 0
    assert np.array_equal(vals[0][const], vals[1][const]) and const.sum() > 0.4 * o.nnz_jac
    assert (vals[0][~const] != vals[1][~const]).mean() > 0.5  # (the variable set is a superset: structurally zero diagonals included)
    o.close()


def test_math_namespace_numpy_spellings_on_the_instance():
    """`M.math.arctan(x)` ... : the numpy spellings are static on the namespace INSTANCE as well (found by the random OCPs: the aliases
    had been assigned from the class attribute, i.e. as plain functions, and an instance bound them as methods)."""
    import sympy as sp

    m, tr = M.math, Tracer()
    x = tr.var("x")
    X = sp.Symbol("x")
    for name, spf, v in (("arcsin", sp.asin, 0.3), ("arccos", sp.acos, 0.3), ("arctan", sp.atan, 0.3), ("asin", sp.asin, 0.3), ("atan", sp.atan, 0.3)):
        f = getattr(m, name)
        assert f(v) == pytest.approx(float(spf(v)), rel=1e-15)
        assert tr.evaluate([f(x)], {"x": v})[0] == pytest.approx(float(spf(v)), rel=1e-15)
        assert tr.evaluate([tr.diff(f(x), x)], {"x": v})[0] == pytest.approx(float(sp.diff(spf(X), X).subs(X, v)), rel=1e-14)
        assert f(X) == spf(X)
    assert m.arctan2(1.0, 2.0) == m.atan2(1.0, 2.0) == pytest.approx(np.arctan2(1.0, 2.0))
    assert tr.evaluate([m.arctan2(x, 2.0)], {"x": 1.0})[0] == pytest.approx(np.arctan2(1.0, 2.0))


@pytest.mark.parametrize("seed,wide", problems.RANDOM_OCPS, ids=[("wide" if w else "smooth") + str(s) for s, w in problems.RANDOM_OCPS])
def test_random_ocps_host_side(seed, wide):
    """tests/problems.py random_ocp_case without a device: sizes, bounds, initial guess equal to the oracle's, and the structural
    patterns (the tracer's structural derivatives) hold every non-zero of the oracle's dense jac_g / hess_l (sympy's derivatives)."""
    from oracle.mpopt_oracle import OracleNLP

    builder, S, po, scheme = problems.random_ocp_case(seed, wide)
    ocp = builder(mp, M.math)
    o = M.NlpFunctions(ocp, S, po, scheme, with_device=False)
    O = OracleNLP(ocp, S, po, scheme)
    assert (o.n_z, o.n_g) == (O.n_z, O.n_g)
    rng = np.random.default_rng(seed)
    z = O.initial_guess() + 0.1 * rng.uniform(-1, 1, O.n_z)
    w = rng.uniform(0.3, 1.7, (ocp.n_phases, S))
    p = (w / w.sum(axis=1, keepdims=True)).ravel()
    jr, jc = o.jac_pattern()
    hr, hc = o.hess_pattern()
    assert len(set(zip(jr.tolist(), jc.tolist()))) == len(jr) and len(set(zip(hr.tolist(), hc.tolist()))) == len(hr) and np.all(hr <= hc)
    mask = np.zeros((o.n_g, o.n_z), bool)
    mask[jr, jc] = True
    Jd = O.jac_g(z, p)
    assert not np.any((np.asarray(Jd.todense() if hasattr(Jd, 'todense') else Jd) != 0) & ~mask)
    mask = np.zeros((o.n_z, o.n_z), bool)
    mask[hr, hc] = True
    assert not np.any((np.triu(O.hess_l(z, p, 0.7, rng.standard_normal(O.n_g))) != 0) & ~mask)
    o.close()


def test_evaluation_path_switches_are_read_once_per_process():
    """include/mpx.h mpx_env_dynamic / mpx_env_knob: without MPX_ENV_DYNAMIC the knobs of the evaluation path are a snapshot
    taken at first use -- a later setenv is not seen until mpx_env_dynamic(1), and mpx_env_dynamic(0) takes a new snapshot."""
    import subprocess
    import sys

    code = r"""
import os, sys
sys.path.insert(0, %r)
from mpopt_amd import _lib
L = _lib.lib()
assert L.mpx_env_knob(b"MPX_NO_LIGHT") is None and L.mpx_env_knob(b"MPX_BPB") == b"3" and L.mpx_env_knob(b"PATH") is None
os.environ["MPX_NO_LIGHT"] = "1"; os.environ["MPX_BPB"] = "5"
assert L.mpx_env_knob(b"MPX_NO_LIGHT") is None and L.mpx_env_knob(b"MPX_BPB") == b"3"      # the snapshot
assert L.mpx_env_dynamic(1) == 0
assert L.mpx_env_knob(b"MPX_NO_LIGHT") == b"1" and L.mpx_env_knob(b"MPX_BPB") == b"5"      # per call
del os.environ["MPX_BPB"]
assert L.mpx_env_knob(b"MPX_BPB") is None
assert L.mpx_env_dynamic(0) == 0                                                             # a new snapshot
del os.environ["MPX_NO_LIGHT"]
assert L.mpx_env_knob(b"MPX_NO_LIGHT") == b"1" and L.mpx_env_knob(b"MPX_BPB") is None
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("MPX_ENV_DYNAMIC", "MPX_NO_LIGHT")}
    env["MPX_BPB"] = "3"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
    # ... and with MPX_ENV_DYNAMIC in the environment (this test process: tests/conftest.py) every call reads the environment
    L = _lib.lib()
    os.environ["MPX_LANES_ORDER"] = "probe"
    try:
        assert L.mpx_env_knob(b"MPX_LANES_ORDER") == b"probe"
    finally:
        del os.environ["MPX_LANES_ORDER"]
    assert L.mpx_env_knob(b"MPX_LANES_ORDER") is None

"""Grids of polynomial degree 69 ... 255 (round 6): above MPX_TABLES_STREAM_ABOVE (mpx_device.h, default 68) the node kernels do
not keep the differentiation / mid-point tables in LDS -- every lane streams its rows from the transposed tables in global memory
(mpx_kernels.h: node_body, TAB_GLB).  Up to round 5 mpx_create refused every degree >= 94 (two tables of (P + 1)^2 doubles in the LDS
of one workgroup), although the reference documents and times `mp.solve(ocp, n_segments=1, poly_orders=100, scheme="LGR")`
(docs/source/notebooks/getting_started.ipynb:721-743; mpopt.py:3815-3849, 4015-4039 have no degree limit).

Parity: every output against the numpy / sympy oracle (tables pinned to 50-digit arithmetic above degree 10), per call through the
C ABI; bit-identity of the streamed mode with the LDS mode on degrees both can run; the published optimum of the 1 x 100 grid."""
import contextlib
import os

import numpy as np
import pytest
import scipy.sparse as sp

import mpopt_amd as M
from mpopt_amd import mp
import problems
from helpers import rel_err
from oracle.mpopt_oracle import OracleNLP

pytestmark = pytest.mark.gpu
TOL = 1e-10


@contextlib.contextmanager
def stream_above(n):
    """Contexts created inside are built for (and their kernels compiled with) MPX_TABLES_STREAM_ABOVE = n."""
    old = os.environ.get("MPX_TABLES_STREAM_ABOVE")
    os.environ["MPX_TABLES_STREAM_ABOVE"] = str(n)
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("MPX_TABLES_STREAM_ABOVE", None)
        else:
            os.environ["MPX_TABLES_STREAM_ABOVE"] = old


def random_point(o, mpo, seed, S, n_ph):
    rng = np.random.default_rng(seed)
    z0 = mpo.initialize_solution()
    z = z0 + 0.05 * np.abs(z0) * rng.uniform(-1, 1, o.n_z) + 0.05 * rng.uniform(-1, 1, o.n_z)
    w = rng.uniform(0.3, 1.7, (n_ph, S))
    return z, (w / w.sum(axis=1, keepdims=True)).ravel(), rng.standard_normal(o.n_g), float(rng.uniform(0.2, 2.0))


HIGH = problems.HIGH_DEGREE_CASES


@pytest.mark.parametrize("name", list(HIGH))
def test_high_degree_against_numpy_oracle(name):
    builder, S, po, scheme = HIGH[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    O = OracleNLP(ocp, S, po, scheme)
    assert (o.n_z, o.n_g) == (O.n_z, O.n_g)
    assert np.array_equal(O.initial_guess(), mpo.initialize_solution())
    z, p, lam, sig = random_point(o, mpo, 17, S, ocp.n_phases)
    r = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], z, p, lam_g=lam, sigma=sig)
    assert rel_err(r["f"], O.f(z, p)) < TOL
    assert rel_err(r["g"], O.g(z, p)) < TOL
    assert rel_err(r["grad_f"], O.grad_f(z, p)) < TOL
    jr, jc = o.jac_pattern()
    Jo = sp.csr_matrix(O.jac_g(z, p))
    d = sp.coo_matrix((r["jac_g"], (jr, jc)), shape=(o.n_g, o.n_z)).tocsr() - Jo
    assert (abs(d).max() if d.nnz else 0.0) < TOL * max(1.0, abs(Jo).max())
    hr, hc = o.hess_pattern()
    Ho = sp.csr_matrix(np.triu(O.hess_l(z, p, sig, lam)))
    d = sp.coo_matrix((r["hess_l"], (hr, hc)), shape=(o.n_z, o.n_z)).tocsr() - Ho
    assert (abs(d).max() if d.nnz else 0.0) < TOL * max(1.0, abs(Ho).max())
    # a light pass (single-degree grids: mpx_lighthigh_* on the matrix cores, another fixed order of the sums in f; mixed grids: the
    # node kernels in f / g mode), the separate calls and a batch give the same bits of g
    lg = o.eval(["f", "g"], z, p)
    assert np.array_equal(lg["g"], r["g"]) and abs(lg["f"] - r["f"]) <= 1e-13 * max(1.0, abs(r["f"]))
    rb = o.eval(["f", "g", "grad_f", "jac_g"], np.stack([z, z + 1e-3, z]), p)
    assert np.array_equal(rb["jac_g"][0], r["jac_g"]) and np.array_equal(rb["jac_g"][2], r["jac_g"]) and np.array_equal(rb["g"][2], r["g"])
    # nlp_grad: grad_gamma_x / grad_gamma_p (the D / C_mid transposes straight from global memory at these degrees)
    q = o.eval_grad_gamma(z, p, lam, sig)
    gx, gp = O.grad_gamma(z, p, sig, lam)
    assert rel_err(q["grad_gamma_x"], gx) < TOL and rel_err(q["grad_gamma_p"], gp) < TOL


BOTH = {
    "moon_lander_3x30_LGR": (problems.moon_lander, 3, 30, "LGR"),
    "vdp_mixed_3_30_3_CGL": (problems.van_der_pol, 12, [30 if s % 3 == 1 else 3 for s in range(12)], "CGL"),
    "kitchen_sink_16_21": (problems.kitchen_sink, 2, [16, 21], "LGL"),
    "dae_vdp_2x64_LGL": (problems.dae_vdp, 2, 64, "LGL"),
    "moon_lander_5x92_LGR": (problems.moon_lander, 5, 92, "LGR"),  # the largest degree of the LDS mode at nx + nu = 3 (137 KB of tables)
    "hyper_sensitive_20x13_CGL": (problems.hyper_sensitive, 20, 13, "CGL"),  # smallest degree outside the register tables
}


@pytest.mark.parametrize("name", list(BOTH))
def test_streamed_tables_equal_lds_tables_bit_for_bit(name):
    """The same grid through the LDS-table kernels (threshold 255) and through the streamed-table kernels (threshold 12: forced on
    for every degree above the register tables): the contractions are the same sequential fma chains and the Jacobian rows copies of
    the same table entries, so every output has the same bits -- all masks, a batch, per-point widths, the variable-only Jacobian."""
    builder, S, po, scheme = BOTH[name]
    ocp = builder(mp, M.math)
    res = []
    for thr in (255, 12):
        with stream_above(thr):
            mpo = mp.mpopt(ocp, S, po, scheme)
            o = mpo.create_nlp()[0]["oracle"]
            z, p, lam, sig = random_point(o, mpo, 5, S, ocp.n_phases)
            Z = np.stack([z, mpo.initialize_solution(), z * 1.01])
            Pw = np.stack([p, np.full_like(p, 1.0 / S), p[::-1].copy()])
            out = [o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], Z, Pw, lam_g=np.stack([lam] * 3), sigma=np.full(3, sig))]
            out.append(o.eval(["f", "g"], z, p))
            out.append(o.eval(["f", "grad_f"], z, p))
            out.append(o.eval(["g", "jac_g"], Z, p))
            out.append(o.eval_grad_gamma(z, p, lam, sig))
            res.append(out)
            o.close()
    for a, b in zip(*res):
        assert a.keys() == b.keys()
        for key in a:
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), (name, key)


def test_reference_documented_grid_solves_to_the_published_optimum():
    """docs/source/notebooks/getting_started.ipynb:721-743: `mp.solve(ocp, n_segments=1, poly_orders=100, scheme="LGR")` on the moon
    lander -> "Optimal cost (J): 8.24747", terminal time 4.164977 (outer iteration: the stand-in of mpopt_amd/solver.py; every
    function value, Jacobian and Hessian from the streamed-table kernels)."""
    mp.mpopt._MUTE_ = True
    ocp = problems.moon_lander(mp, M.math)
    mpo, post = mp.solve(ocp, n_segments=1, poly_orders=100, scheme="LGR", plot=False)
    sol = mpo.solve()
    assert abs(float(sol["f"]) - 8.24747) < 2e-4
    x, u, t, _ = post.get_data()
    assert abs(float(np.ravel(t)[-1]) - 4.164977) < 2e-3 and x.shape[0] == 101


def test_off_node_residuals_at_high_degree():
    """mpx_resid_<ph>_<deg> above degree 32 reads its interpolation / derivative rows inside the contraction instead of holding
    2 (P + 1) of them in registers: the same fma chains -- against the numpy oracle's per-segment residuals."""
    ocp = problems.van_der_pol(mp, M.math)
    S, po = 2, [100, 40]
    mpo = mp.mpopt(ocp, S, po, "LGR")
    o = mpo.create_nlp()[0]["oracle"]
    z, p, _, _ = random_point(o, mpo, 3, S, 1)
    O = OracleNLP(ocp, S, po, "LGR")
    rng = np.random.default_rng(4)
    taus = [np.sort(rng.uniform(-1, 1, 37)), np.sort(rng.uniform(-1, 1, 11))]
    plan = o.residual_plan(0, taus)
    r = plan.eval(z, p)
    ref = O.residuals(z, p, 0, taus)
    for key in ("ti", "xi", "ui", "dxi", "dui", "dyn", "resid"):
        assert rel_err(r[key], np.asarray(ref[key]).reshape(r[key].shape)) < TOL, key
    plan.close()


@pytest.mark.parametrize("name", ["moon_lander_1x100_LGR", "kitchen_sink_95_71"])
def test_jac_variable_only_at_streamed_degrees(name):
    """MPX_JAC_VARIABLE_ONLY through the streamed walk: after one full evaluation, rewriting only the (z, p)-dependent entries at new
    points reproduces the full evaluation bit for bit, and the constant entries are really skipped."""
    import torch
    from mpopt_amd._lib import MPX_JAC_VARIABLE_ONLY

    builder, S, po, scheme = HIGH[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    B = 3
    z0 = mpo.initialize_solution()
    Z1 = torch.tensor(z0[None, :] + 0.05 * rng.standard_normal((B, o.n_z)), device=dev)
    Z2 = torch.tensor(z0[None, :] + 0.05 * rng.standard_normal((B, o.n_z)), device=dev)
    p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
    mk = lambda *s: torch.full(s, float("nan"), dtype=torch.float64, device=dev)
    f, g, gr, jv = mk(B), mk(B, o.n_g), mk(B, o.n_z), mk(B, o.nnz_jac)
    o.eval_device(15, B, Z1, p, 0, None, None, f, g, gr, jv, None)
    o.eval_device(15 | MPX_JAC_VARIABLE_ONLY, B, Z2, p, 0, None, None, f, g, gr, jv, None)
    f2, g2, gr2, jv2 = mk(B), mk(B, o.n_g), mk(B, o.n_z), mk(B, o.nnz_jac)
    o.eval_device(15, B, Z2, p, 0, None, None, f2, g2, gr2, jv2, None)
    o.sync()
    for a, b in ((f, f2), (g, g2), (gr, gr2), (jv, jv2)):
        assert torch.equal(a, b)
    jn = mk(B, o.nnz_jac)
    o.eval_device(8 | MPX_JAC_VARIABLE_ONLY, B, Z2, p, 0, None, None, None, None, None, jn, None)
    o.sync()
    assert torch.isnan(jn).any() and not torch.isnan(jn).all()


LIGHT_HIGH = {
    "moon_lander_6x40_LGR": (problems.moon_lander, 6, 40, "LGR"),        # node kernels of this degree keep their tables in LDS
    "moon_lander_3x100_LGR": (problems.moon_lander, 3, 100, "LGR"),      # ... stream them
    "kitchen_sink_3x36_LGL": (problems.kitchen_sink, 3, 36, "LGL"),      # two phases, control-slope rows (D.U), path rows, parameters, time dependence
    "dae_vdp_2x69_CGL": (problems.dae_vdp, 2, 69, "CGL"),                # path row + parameter
    "hyper_sensitive_1x255_LGR": (problems.hyper_sensitive, 1, 255, "LGR"),  # 16 column tiles, 64 K steps; no mid-point rows
    "schwartz_2x33_LGR": (problems.two_phase_schwartz, 2, 33, "LGR"),    # P + 1 = 34: two nodes in the third column tile
}


@pytest.mark.parametrize("name", list(LIGHT_HIGH))
def test_high_degree_light_passes_on_the_matrix_cores(name, monkeypatch):
    """Round 6: nlp_f / nlp_g / nlp_grad_f WITHOUT the Jacobian values on single-degree grids of degree >= 32 (mpx_lighthigh_*,
    light_high_body): the contraction of a segment as a matrix product with 16 evaluation points as one dimension, v_mfma_f64_16x16x4_f64,
    the transposed tables as operands straight from L2.  Against the node kernels (MPX_NO_LIGHT=1): g and the node entries of grad_f bit
    for bit (the same sequential fused chains), f and the (t0, tf, a) sums -- another fixed order -- to rounding; for batches that are
    no multiple of 16, per-point widths; and against the numpy oracle.  Defects: mpopt.py:227-232; objective: mpopt.py:455."""
    from helpers import border_columns

    builder, S, P, scheme = LIGHT_HIGH[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    assert o.light_plan()[:2] == (P, S), "no high-degree light plan: the masks below would run the node kernels"
    z, p, lam, sig = random_point(o, mpo, 23, S, ocp.n_phases)
    rng = np.random.default_rng(7)
    node = np.ones(o.n_z, bool)
    node[border_columns(o)] = False
    masks = (["f"], ["g"], ["f", "grad_f"], ["f", "g", "grad_f"])
    for B in (1, 5, 16, 37):
        Z = z[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z))) + 0.01 * rng.uniform(-1, 1, (B, o.n_z))
        Z[0] = z
        w = rng.uniform(0.3, 1.7, (B, ocp.n_phases, S))
        Pw = (w / w.sum(axis=2, keepdims=True)).reshape(B, -1)
        for pp in (p, Pw):
            light = [o.eval(m, Z if B > 1 else Z[0], pp if B > 1 else (pp if pp.ndim == 1 else pp[0])) for m in masks]
            monkeypatch.setenv("MPX_NO_LIGHT", "1")
            heavy = [o.eval(m, Z if B > 1 else Z[0], pp if B > 1 else (pp if pp.ndim == 1 else pp[0])) for m in masks]
            monkeypatch.delenv("MPX_NO_LIGHT")
            for m, a, h in zip(masks, light, heavy):
                a, h = ({k: np.atleast_2d(v) if k != "f" else np.atleast_1d(v) for k, v in d.items()} for d in (a, h))
                if "g" in m:
                    assert np.array_equal(a["g"], h["g"]), (name, B, m)
                if "grad_f" in m:
                    assert np.array_equal(a["grad_f"][:, node], h["grad_f"][:, node]), (name, B, m)
                    sc = max(1.0, np.abs(h["grad_f"][:, ~node]).max())
                    assert np.abs(a["grad_f"][:, ~node] - h["grad_f"][:, ~node]).max() <= 1e-12 * sc, (name, B, m)
                if "f" in m:
                    assert np.abs(a["f"] - h["f"]).max() <= 1e-13 * max(1.0, np.abs(h["f"]).max()), (name, B, m)
                    assert np.array_equal(a["f"], np.atleast_1d(light[0]["f"])), (name, B, m)  # every light pass sums f in the same order
    O = OracleNLP(ocp, S, P, scheme)
    r = o.eval(["f", "g", "grad_f"], z, p)
    assert rel_err(r["f"], O.f(z, p)) < TOL and rel_err(r["g"], O.g(z, p)) < TOL and rel_err(r["grad_f"], O.grad_f(z, p)) < TOL

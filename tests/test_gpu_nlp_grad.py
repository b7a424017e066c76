"""nlp_grad on the GPU (through the C ABI): grad_gamma_x = sigma * grad_f + jac_g^T lam_g and grad_gamma_p = d gamma / d p, the sixth
oracle ca.nlpsol derives from mpopt's NLP (mpopt.py:757; "nlp_grad ... n_eval 1" in the reference's recorded solves,
docs/source/notebooks/moon_lander.ipynb:206), whose -grad_gamma_p is the lam_p the solver returns (tests/test_examples.py:44-45).

Checked against (a) the goldens the reference's own NLP produced (tests/golden/make_golden.py), (b) the numpy / sympy oracle at
reduced sizes, (c) the C oracle's hand-derived derivatives at BASELINE.json's full sizes, (d) the library's own generic route
(J^T lam from the stored Jacobian values), (e) central differences of gamma.  Tolerance 1e-10, PER ENTRY (helpers.assert_entries)."""
import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp
import problems
from helpers import assert_entries, build_case, load_golden
from oracle.mpopt_oracle import OracleNLP
from oracle.c_oracle import COracle
from test_gpu_parity import FULL, REDUCED, random_point

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.mark.parametrize("name", list(problems.GOLDEN_CASES) + list(problems.ADAPTIVE_CASES))
def test_golden_point(name):
    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=True)
    r = o.eval_grad_gamma(G["z"], G["p"], G["lam"], float(G["sigma"]))
    assert_entries(r["grad_gamma_x"], G["grad_gamma_x"], TOL, what=f"{name} grad_gamma_x")
    assert r["grad_gamma_p"].shape == G["grad_gamma_p"].shape
    assert_entries(r["grad_gamma_p"], G["grad_gamma_p"], TOL, what=f"{name} grad_gamma_p")
    # sigma = 0: J^T lam alone; lam = 0: sigma * grad_f
    r0 = o.eval_grad_gamma(G["z"], G["p"], 0 * G["lam"], 1.0)
    assert_entries(r0["grad_gamma_x"], G["grad_f"], TOL, what=f"{name} grad_gamma_x(lam = 0) = grad_f")
    # only one of the two outputs (what CasADi asks for by default is grad_gamma_p alone)
    rp = o.eval_grad_gamma(G["z"], G["p"], G["lam"], float(G["sigma"]), what=("grad_gamma_p",))
    assert "grad_gamma_x" not in rp and np.array_equal(rp["grad_gamma_p"], r["grad_gamma_p"])
    rx = o.eval_grad_gamma(G["z"], G["p"], G["lam"], float(G["sigma"]), what=("grad_gamma_x",))
    assert np.array_equal(rx["grad_gamma_x"], r["grad_gamma_x"])


@pytest.mark.parametrize("name", list(REDUCED))
def test_reduced_size_against_numpy_oracle(name):
    builder, S, po, scheme = REDUCED[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    O = OracleNLP(ocp, S, po, scheme)
    z, p, lam, sig = random_point(o, mpo, bounds, 11, S, ocp.n_phases)
    r = o.eval_grad_gamma(z, p, lam, sig)
    gx, gp = O.grad_gamma(z, p, sig, lam)
    assert_entries(r["grad_gamma_x"], gx, TOL, what=f"{name} grad_gamma_x")
    assert_entries(r["grad_gamma_p"], gp, TOL, what=f"{name} grad_gamma_p")


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_against_c_oracle_and_properties(name, monkeypatch):
    (builder, S, po, scheme), cnames, st, midu = FULL[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    C = COracle(cnames, S, po, scheme, scale_x=ocp.scale_x, scale_u=ocp.scale_u, scale_a=ocp.scale_a if ocp.na else None, scale_t=st, midu=midu)
    z, p, lam, sig = random_point(o, mpo, bounds, 23, S, ocp.n_phases)
    Z = np.stack([z, mpo.initialize_solution(), z * 1.01])
    LAM = np.stack([lam, lam[::-1].copy(), 0.5 * lam])
    SIG = np.array([sig, 1.0, 0.0])
    r = o.eval_grad_gamma(Z, p, LAM, SIG)
    for b in range(3):
        gx, gp = C.grad_gamma(Z[b], p, SIG[b], LAM[b])
        assert_entries(r["grad_gamma_x"][b], gx, TOL, what=f"{name}[{b}] grad_gamma_x")
        assert_entries(r["grad_gamma_p"][b], gp, TOL, what=f"{name}[{b}] grad_gamma_p")
    # the generic route of the library (fgj pass into scratch, J^T lam by columns) agrees with the fused pass
    monkeypatch.setenv("MPX_GRADL_GENERIC", "1")
    rg = o.eval_grad_gamma(Z, p, LAM, SIG)
    monkeypatch.delenv("MPX_GRADL_GENERIC")
    assert_entries(rg["grad_gamma_x"], r["grad_gamma_x"], 1e-11, what=f"{name} generic vs fused")
    assert np.array_equal(rg["grad_gamma_p"], r["grad_gamma_p"])
    # size-independent property: directional derivatives of gamma(z, p) = sigma f + lam^T g by central differences of the GPU's f, g
    rng = np.random.default_rng(5)
    v, q = rng.standard_normal(o.n_z), rng.standard_normal(o.n_p) * p
    eps = 1e-6
    gam = lambda zz, pp: sig * o.eval(["f"], zz, pp)["f"] + lam @ o.eval(["g"], zz, pp)["g"]
    dz = (gam(z + eps * v, p) - gam(z - eps * v, p)) / (2 * eps)
    assert abs(r["grad_gamma_x"][0] @ v - dz) < 2e-6 * max(1.0, abs(dz))
    dp = (gam(z, p + eps * q) - gam(z, p - eps * q)) / (2 * eps)
    assert abs(r["grad_gamma_p"][0] @ q - dp) < 2e-6 * max(1.0, abs(dp), np.abs(r["grad_gamma_p"][0] * q).sum())


@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "kitchen_sink_mixed_CGL", "schwartz_4x3_LGL"])
def test_batch_matches_single_and_device_pointers(name):
    """A batch equals its single evaluations bit for bit (fixed-order sums), shared and per-point widths; the device-pointer entry
    gives the bits of the host-pointer one."""
    import torch

    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=True)
    rng = np.random.default_rng(5)
    B = 7
    Z = G["z"][None, :] + 0.01 * rng.standard_normal((B, o.n_z))
    lam = rng.standard_normal((B, o.n_g))
    sig = rng.uniform(0.5, 1.5, B)
    rb = o.eval_grad_gamma(Z, G["p"], lam, sig)
    P = np.stack([np.roll(G["p"].reshape(ocp.n_phases, -1), b, axis=1).ravel() for b in range(B)])
    rp = o.eval_grad_gamma(Z, P, lam, sig)
    for b in range(B):
        r1 = o.eval_grad_gamma(Z[b], G["p"], lam[b], sig[b])
        r2 = o.eval_grad_gamma(Z[b], P[b], lam[b], sig[b])
        for k in ("grad_gamma_x", "grad_gamma_p"):
            assert np.array_equal(rb[k][b], r1[k]), (k, b)
            assert np.array_equal(rp[k][b], r2[k]), (k, b)
    dev = torch.device("cuda:0")
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    gx, gp = torch.empty((B, o.n_z), dtype=torch.float64, device=dev), torch.empty((B, o.n_p), dtype=torch.float64, device=dev)
    zt, pt, lt, st = t(Z), t(P), t(lam), t(sig)
    torch.cuda.synchronize()
    o.eval_grad_gamma_device(B, zt, pt, lt, st, gx, gp, p_per_point=1)
    o.sync()
    assert np.array_equal(gx.cpu().numpy(), rp["grad_gamma_x"]) and np.array_equal(gp.cpu().numpy(), rp["grad_gamma_p"])


@pytest.mark.parametrize("name", ["moon_lander_20x3_LGR", "kitchen_sink_mixed_CGL", "schwartz_4x3_LGL", "dae_vdp_mixed_CGL"])
def test_evaluation_points_per_workgroup_do_not_change_the_bits(name, monkeypatch):
    """Round 6: the node pass of nlp_grad takes several evaluation points per workgroup for large batches (MpxGradlArgs::bpb: 4 from 8192
    workgroups on) and adds a segment's column-0 sums where the previous segment's last node is owned (no `halo` round trip) -- a ragged batch
    through 1, 3, 4 and 5 points per workgroup equals its single evaluations bit for bit, shared and per-point widths."""
    G = load_golden(name)
    ocp, mpo, o = build_case(name, with_device=True)
    rng = np.random.default_rng(8)
    B = 11
    Z = G["z"][None, :] + 0.01 * rng.standard_normal((B, o.n_z))
    lam = rng.standard_normal((B, o.n_g))
    sig = rng.uniform(0.5, 1.5, B)
    P = np.stack([np.roll(G["p"].reshape(ocp.n_phases, -1), b, axis=1).ravel() for b in range(B)])
    singles = [o.eval_grad_gamma(Z[b], P[b], lam[b], sig[b]) for b in range(B)]
    for bpb in (1, 3, 4, 5):
        monkeypatch.setenv("MPX_GRADL_BPB", str(bpb))
        rp = o.eval_grad_gamma(Z, P, lam, sig)
        rs = o.eval_grad_gamma(Z, G["p"], lam, sig, what=("grad_gamma_p",))  # (grad_gamma_x = NULL: only the width sums)
        r0 = o.eval_grad_gamma(Z[0], G["p"], lam[0], sig[0])
        for b in range(B):
            for k in ("grad_gamma_x", "grad_gamma_p"):
                assert np.array_equal(rp[k][b], singles[b][k]), (bpb, k, b)
        assert np.array_equal(rs["grad_gamma_p"][0], r0["grad_gamma_p"])
    monkeypatch.delenv("MPX_GRADL_BPB")
    # ... and the generic route (fgj pass into scratch, J^T lam by columns) agrees
    monkeypatch.setenv("MPX_GRADL_GENERIC", "1")
    rg = o.eval_grad_gamma(Z, P, lam, sig)
    monkeypatch.delenv("MPX_GRADL_GENERIC")
    assert_entries(rg["grad_gamma_x"], rp["grad_gamma_x"], 1e-11, what=f"{name} generic vs fused (points per workgroup)")
    assert np.array_equal(rg["grad_gamma_p"], rp["grad_gamma_p"])
    o.close()


def test_solver_returns_lam_p():
    """mp.solve's result carries the real lam_p = -grad_gamma_p at the solution (the reference's tests assert the key,
    tests/test_examples.py:44-45; CasADi computes it with one nlp_grad call after the last iterate)."""
    ocp = problems.moon_lander(mp, M.math)
    mpo = mp.mpopt(ocp, 20, 3, "LGR")
    sol = mpo.solve()
    for k in ("f", "g", "lam_g", "lam_p", "lam_x", "x"):
        assert k in sol
    lam_p = np.asarray(sol["lam_p"], float).ravel()
    assert lam_p.shape == (20,) and np.abs(lam_p).max() > 1e-6
    o = mpo.nlp_solver.oracle
    assert mpo.nlp_solver.stats["n_eval"].get("nlp_grad") == 1
    p = np.asarray(mpo.get_segment_width_parameters(None), float)
    q = o.eval_grad_gamma(np.asarray(sol["x"], float).ravel(), p, np.asarray(sol["lam_g"], float).ravel(), 1.0)
    assert np.allclose(lam_p, -q["grad_gamma_p"], rtol=1e-12, atol=1e-14)
    assert abs(float(sol["f"]) - 8.24677) < 1e-4  # the published optimum (docs/source/notebooks/getting_started.ipynb:428)

"""SURVEY 8(d) config-5 protocol at full size: hypersensitive OCP, 4000 segments of degree 3 (LGR); segment widths drawn once as
Dirichlet(1), then five outer iterations of the h-adaptive loop with everything resident on the device and ONE libmpx context:
    dynamics residuals at the mid-points (mpx_resid_eval_device)  ->  nlp_hess_l (mpx_eval_device)  ->
    equal-area width update, damped (mpx_equal_area_widths_device; mpopt.py:2636-2659, 2587-2590).
Every iteration is checked: residuals against the numpy oracle on a sample of segments, hess_l against the C oracle (all
entries), the new widths against the reference's rule (numpy restatement pinned by tests/golden/hadaptive.npz) at 1e-10."""
import os

import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp
from mpopt_amd._lib import MPX_HESS, MPX_WIDTHS_UNCHANGED
import problems
from helpers import rel_err
from oracle.mpopt_oracle import OracleNLP
from oracle.c_oracle import COracle

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_config5_loop_device_resident():
    import scipy.sparse as sp
    import torch

    builder, S, po, scheme = problems.BENCH_CASES[3]
    assert (S, po) == (4000, 3)
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    dev = torch.device("cuda:0")
    o.set_stream(torch.cuda.current_stream().cuda_stream)
    O = OracleNLP(ocp, S, po, scheme)
    C = COracle(["hyper_sensitive"], S, po, scheme, scale_t=1e-3, midu=[0])
    rng = np.random.default_rng(20260928)
    B, n_iter = 3, 5
    Zh = mpo.initialize_solution()[None, :] + 0.05 * rng.standard_normal((B, o.n_z))
    ph = rng.dirichlet(np.ones(S), B)  # widths ~ Dirichlet(1), one draw per evaluation point
    lamh, sigh = rng.standard_normal((B, o.n_g)), rng.uniform(0.5, 1.5, B)
    mids = [(mpo.collocation._taus_fn(d)[:-1] + mpo.collocation._taus_fn(d)[1:]) / 2 for d in mpo.poly_orders]
    plan = o.residual_plan(0, mids)
    n_pts = plan.n_pts
    assert n_pts == 3 * S
    t = lambda a: torch.tensor(a, device=dev)
    Z, p, lam, sig = t(Zh), t(ph), t(lamh), t(sigh)
    p_new = torch.empty_like(p)
    R = torch.empty(B, n_pts, ocp.nx, dtype=torch.float64, device=dev)
    H = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
    hr, hc = o.hess_pattern()
    A = mp.mpopt_h_adaptive
    sample = sorted(set(rng.integers(0, S, 12).tolist()) | {0, 1, S - 1})
    for it in range(n_iter):
        plan.eval_device(B, Z, p, p_per_point=1, resid=R)
        o.eval_device(MPX_HESS | (MPX_WIDTHS_UNCHANGED if it % 2 else 0), B, Z, p, 1, lam, sig, None, None, None, None, H)
        o.equal_area_widths_device(0, B, n_pts, R, p, p_new, damping=0.4, p_in_per_point=1)
        o.sync()
        Rh, Hh, pn, pc = R.cpu().numpy(), H.cpu().numpy(), p_new.cpu().numpy(), p.cpu().numpy()
        for b in range(B):
            ro = O.residuals_of_segments(Zh[b], pc[b], 0, mids, sample)
            for s in sample:
                assert rel_err(Rh[b, 3 * s:3 * s + 3, :], ro[s]["resid"]) < TOL, (it, b, s)
            Hg = sp.coo_matrix((Hh[b], (hr, hc)), shape=(o.n_z, o.n_z)).tocsr()
            Hc = C.hess_matrix(Zh[b], pc[b], sigh[b], lamh[b])
            d = Hg - Hc
            assert (abs(d).max() if d.nnz else 0.0) < TOL * max(1.0, abs(Hc).max()), (it, b)
            r1d = np.linalg.norm(Rh[b], 2, axis=1)
            want = 0.4 * np.asarray(A.get_roots_wrt_equal_area(r1d, S)) + 0.6 * pc[b]
            assert np.abs(pn[b] - want).max() < TOL * want.max(), (it, b, np.abs(pn[b] - want).max())
            assert abs(pn[b].sum() - 1) < 1e-9 and pn[b].min() > 0
        p, p_new = p_new, p
    # the widths moved: the loop is not a fixed point of the first draw
    assert np.abs(p.cpu().numpy() - ph).max() > 1e-6


@pytest.mark.parametrize("case", ["hyper_sensitive_4000x3", "kitchen_sink_40x4", "dae_vdp_30x7"])
def test_mid_point_residuals_fused_into_the_hess_pass(case):
    """MPX_MID_RESID: the node kernels of the hess_l pass also write the dynamics residuals at the mid-points of every segment.  They
    equal the residual plan over the same target points (same fma chains: compared at 1e-13, and bit for bit where the compiler
    keeps the generated node function identical in both kernels), hess_l itself is unchanged bit for bit, and the numpy oracle
    confirms a sample of segments -- single phase at config-5 size, and a two-phase problem with parameters and several states.
    (Mixed-degree grids run hess_l over node-ordered tiles and refuse the flag: a residual plan serves them.)"""
    import torch
    from mpopt_amd._lib import MPX_MID_RESID

    builder, S, po, scheme = {"hyper_sensitive_4000x3": (problems.hyper_sensitive, 4000, 3, "LGR"),
                              "kitchen_sink_40x4": (problems.kitchen_sink, 40, 4, "LGR"),
                              "dae_vdp_30x7": (problems.dae_vdp, 30, 7, "LGL")}[case]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    B = 5
    Zh = mpo.initialize_solution()[None, :] + 0.05 * rng.standard_normal((B, o.n_z))
    w = rng.uniform(0.5, 1.5, (B, ocp.n_phases, S))
    ph = (w / w.sum(axis=2, keepdims=True)).reshape(B, -1)
    t = lambda a: torch.tensor(a, device=dev)
    Z, p, lam, sig = t(Zh), t(ph), t(rng.standard_normal((B, o.n_g))), t(rng.uniform(0.5, 1.5, B))
    N = o.n_nodes
    mids = [(mpo.collocation._taus_fn(d)[:-1] + mpo.collocation._taus_fn(d)[1:]) / 2 for d in mpo.poly_orders]
    H0 = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
    H1 = torch.full_like(H0, float("nan"))
    o.eval_device(MPX_HESS, B, Z, p, 1, lam, sig, None, None, None, None, H0)
    Rf = torch.full((B, ocp.n_phases * (N - 1), ocp.nx), float("nan"), dtype=torch.float64, device=dev)
    o.set_mid_resid_output(Rf)
    o.eval_device(MPX_HESS | MPX_MID_RESID, B, Z, p, 1, lam, sig, None, None, None, None, H1)
    o.sync()
    assert torch.equal(H0, H1)
    O = OracleNLP(ocp, S, po, scheme)
    for phase in range(ocp.n_phases):
        plan = o.residual_plan(phase, mids)
        assert plan.n_pts == N - 1
        Rp = torch.empty(B, N - 1, ocp.nx, dtype=torch.float64, device=dev)
        plan.eval_device(B, Z, p, p_per_point=1, resid=Rp)
        o.sync()
        got = Rf[:, phase * (N - 1):(phase + 1) * (N - 1), :]
        assert not torch.isnan(got).any()
        assert float((got - Rp).abs().max()) <= 1e-13 * max(1.0, float(Rp.abs().max())), (case, phase, float((got - Rp).abs().max()))
        start = np.concatenate([[0], np.cumsum(mpo.poly_orders)])
        for s in sorted(set(rng.integers(0, S, 6).tolist()) | {0, S - 1}):
            ro = O.residuals_of_segments(Zh[1], ph[1], phase, mids, [s])
            assert rel_err(got[1, start[s]:start[s + 1], :].cpu().numpy(), ro[s]["resid"]) < TOL, (case, phase, s)
        plan.close()
    if ocp.n_phases == 1:
        # the equal-area update leaves the prefix sums of ITS output on the device: evaluating at those widths with
        # MPX_WIDTHS_UNCHANGED (no prefix launch) gives the bits of the evaluation that recomputes them
        p2 = torch.empty_like(p)
        o.equal_area_widths_device(0, B, N - 1, Rf, p, p2, damping=0.4, p_in_per_point=1)
        Ha, Ra = torch.empty_like(H0), torch.empty_like(Rf)
        o.set_mid_resid_output(Ra)
        o.eval_device(MPX_HESS | MPX_MID_RESID | MPX_WIDTHS_UNCHANGED, B, Z, p2, 1, lam, sig, None, None, None, None, Ha)
        Hb, Rb = torch.empty_like(H0), torch.empty_like(Rf)
        o.set_mid_resid_output(Rb)
        o.eval_device(MPX_HESS | MPX_MID_RESID, B, Z, p2, 1, lam, sig, None, None, None, None, Hb)
        o.sync()
        assert torch.equal(Ha, Hb) and torch.equal(Ra, Rb) and not torch.equal(Ha, H0)
        p2h = p2.cpu().numpy()
        assert np.abs(p2h.sum(axis=1) - 1).max() < 1e-9 and p2h.min() > 0
        # ... and the flag is never trusted blindly: the GENERIC equal-area kernel (larger grids; MPX_EA_GENERIC=1) leaves no prefix
        # sums behind, neither does an update into another array -- the library then launches the prefix kernel after all instead of
        # evaluating with the sums of older widths (round-3 advisor finding)
        os.environ["MPX_EA_GENERIC"] = "1"
        try:
            p3 = torch.empty_like(p)
            o.equal_area_widths_device(0, B, N - 1, Rb, p2, p3, damping=0.4, p_in_per_point=1)
        finally:
            del os.environ["MPX_EA_GENERIC"]
        Hc, Hd = torch.empty_like(H0), torch.empty_like(H0)
        o.set_mid_resid_output(None)
        o.eval_device(MPX_HESS | MPX_WIDTHS_UNCHANGED, B, Z, p3, 1, lam, sig, None, None, None, None, Hc)
        o.eval_device(MPX_HESS, B, Z, p3, 1, lam, sig, None, None, None, None, Hd)
        o.sync()
        assert torch.equal(Hc, Hd) and not torch.equal(Hc, Hb)
        p4 = p3.clone()  # same widths at another address: the sums on the device describe p3, not p4
        o.equal_area_widths_device(0, B, N - 1, Rb, p2, p3, damping=0.4, p_in_per_point=1)  # fast kernel: prefix sums of p3
        o.eval_device(MPX_HESS | MPX_WIDTHS_UNCHANGED, B, Z, p4, 1, lam, sig, None, None, None, None, Hc)
        o.sync()
        assert torch.equal(Hc, Hd)
    o.set_mid_resid_output(None)
    with pytest.raises(M.MpxError):
        o.eval_device(MPX_HESS | MPX_MID_RESID, B, Z, p, 1, lam, sig, None, None, None, None, H1)
    o.close()
    if case == "dae_vdp_30x7":  # a mixed-degree grid refuses the flag
        mpo2 = mp.mpopt(ocp, 6, [3, 6, 3] * 2, scheme)
        o2 = mpo2.create_nlp()[0]["oracle"]
        R2 = torch.empty(B, o2.n_nodes - 1, ocp.nx, dtype=torch.float64, device=dev)
        o2.set_mid_resid_output(R2)
        z2 = torch.tensor(mpo2.initialize_solution()[None, :].repeat(B, 0), device=dev)
        p2 = torch.full((6,), 1 / 6, dtype=torch.float64, device=dev)
        h2 = torch.empty(B, o2.nnz_hess, dtype=torch.float64, device=dev)
        l2 = torch.zeros(B, o2.n_g, dtype=torch.float64, device=dev)
        with pytest.raises(M.MpxError, match="mixed-degree"):
            o2.eval_device(MPX_HESS | MPX_MID_RESID, B, z2, p2, 0, l2, sig, None, None, None, None, h2)
        o2.close()


def test_equal_area_device_rule_on_reference_vectors():
    """The device kernel on the reference's own equal-area vectors (tests/golden/hadaptive.npz), no damping."""
    import os
    import torch
    from helpers import GOLDEN

    Hh = np.load(os.path.join(GOLDEN, "hadaptive.npz"))
    dev = torch.device("cuda:0")
    for k in range(6):
        r, n = Hh[f"equal_area/{k}/residuals"], int(Hh[f"equal_area/{k}/n"])
        ocp = problems.hyper_sensitive(mp, M.math)  # nx = 1: residual samples are the norms themselves
        mpo = mp.mpopt(ocp, n, 3, "LGR")
        o = mpo.create_nlp()[0]["oracle"]
        R = torch.tensor(np.ascontiguousarray(r, float).reshape(1, -1, 1), device=dev)
        p_in = torch.full((n,), 1.0 / n, dtype=torch.float64, device=dev)
        p_out = torch.empty(1, n, dtype=torch.float64, device=dev)
        o.equal_area_widths_device(0, 1, len(r), R, p_in, p_out, damping=1.0)
        o.sync()
        assert np.allclose(p_out.cpu().numpy()[0], Hh[f"equal_area/{k}/widths"], rtol=1e-11, atol=1e-14), k


@pytest.mark.parametrize("S", [3, 40, 257, 4096])
def test_equal_area_kernels_over_sample_counts_and_residual_shapes(S):
    """mpx_equal_area_widths_device against the reference's rule (numpy restatement pinned by tests/golden/hadaptive.npz) over the
    shapes that steer the kernels: sample counts around the row lengths of the fast kernel (one to twelve trapezoids per lane, odd
    and even -- padded or not --, the last lanes empty or partly filled), one past its limit (generic kernel), more evaluation
    points than compute units (persistent workgroups, prefetch of the next point), and residual curves that are flat, zero over
    long stretches, a single spike (thousands of boundaries inside one trapezoid) or a ramp.  Damped, widths per point."""
    import torch

    dev = torch.device("cuda:0")
    mpo = mp.mpopt(problems.hyper_sensitive(mp, M.math), S, 3, "LGR")
    o = mpo.create_nlp()[0]["oracle"]
    A = mp.mpopt_h_adaptive
    rng = np.random.default_rng(S)

    def curves(B, n):
        r = rng.uniform(0.1, 1.0, (B, n))
        r[1 % B] = 1.0                                      # flat
        if n >= 6:
            r[2 % B, n // 3: 2 * n // 3 + 1] = 0.0          # no area over a third of the samples
        if B > 3:
            r[3] = 1e-6
            r[3, n // 2] = 1e3                              # a spike
        if B > 4:
            r[4] = np.linspace(0.0, 1.0, n)                 # ramp from zero
        if B > 5 and n >= 8:
            r[5, : n - n // 4] = 0.0                        # everything in the last quarter
        return r

    for n, B in ((2, 6), (3, 6), (13, 6), (65, 6), (1024, 6), (1025, 6), (1026, 6), (2049, 6), (2050, 6), (3074, 6), (5000, 6), (11264, 6),
                 (12288, 6), (12289, 6), (700, 600)):
        r = curves(B, n)
        p_in = rng.dirichlet(np.ones(S), B)
        R = torch.tensor(r.reshape(B, n, 1), device=dev)
        pi = torch.tensor(p_in, device=dev)
        po = torch.full((B, S), float("nan"), dtype=torch.float64, device=dev)
        o.equal_area_widths_device(0, B, n, R, pi, po, damping=0.4, p_in_per_point=1)
        o.sync()
        got = po.cpu().numpy()
        for b in range(B):
            want = 0.4 * np.asarray(A.get_roots_wrt_equal_area(r[b], S)) + 0.6 * p_in[b]
            assert np.abs(got[b] - want).max() < 1e-11, (n, b, np.abs(got[b] - want).max())
            assert abs(got[b].sum() - 1) < 1e-9 and got[b].min() > 0
    o.close()


@pytest.mark.parametrize("builder,S,nph", [(problems.van_der_pol, 37, 1), (problems.two_phase_schwartz, 300, 2), (problems.kitchen_sink, 9, 2)])
def test_equal_area_vector_residuals_and_phases(builder, S, nph):
    """The same rule on the 2-norms of VECTOR residual samples (nx = 2, 3) and per phase of a multi-phase context: the update of one
    phase leaves the other phase's widths alone, both phases' prefix sums are on the device afterwards (MPX_WIDTHS_UNCHANGED gives
    the bits of a recomputation), sample counts on both sides of the fast kernel's limit."""
    import torch

    dev = torch.device("cuda:0")
    ocp = builder(mp, M.math)
    assert ocp.n_phases == nph
    mpo = mp.mpopt(ocp, S, 3, "LGR")
    o = mpo.create_nlp()[0]["oracle"]
    A = mp.mpopt_h_adaptive
    rng = np.random.default_rng(5)
    B = 7
    nx = ocp.nx
    for n in (3 * S, 12289):
        p_in = np.concatenate([rng.dirichlet(np.ones(S), B) for _ in range(nph)], axis=1)
        pi = torch.tensor(p_in, device=dev)
        po = pi.clone()
        want = p_in.copy()
        for ph in range(nph):
            r = rng.uniform(0.0, 1.0, (B, n, nx)) * rng.uniform(0.1, 3.0, (B, n, 1))
            r[0, n // 4: n // 2] = 0.0
            R = torch.tensor(r, device=dev)
            o.equal_area_widths_device(ph, B, n, R, pi, po, damping=0.4, p_in_per_point=1)
            for b in range(B):
                want[b, ph * S:(ph + 1) * S] = 0.4 * np.asarray(A.get_roots_wrt_equal_area(np.linalg.norm(r[b], 2, axis=1), S)) + 0.6 * p_in[b, ph * S:(ph + 1) * S]
            o.sync()
            got = po.cpu().numpy()
            assert np.abs(got - want).max() < 1e-11, (n, ph, np.abs(got - want).max())  # (phases not yet updated: still p_in)
        if n <= 12288:  # both phases updated by the fast kernel: their prefix sums are on the device
            Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.01 * rng.standard_normal((B, o.n_z)), device=dev)
            lam, sig = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev), torch.ones(B, dtype=torch.float64, device=dev)
            H1, H2 = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
            o.eval_device(MPX_HESS | MPX_WIDTHS_UNCHANGED, B, Z, po, 1, lam, sig, None, None, None, None, H1)
            o.eval_device(MPX_HESS, B, Z, po, 1, lam, sig, None, None, None, None, H2)
            o.sync()
            assert torch.equal(H1, H2), n
    o.close()

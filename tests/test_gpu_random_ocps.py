"""Random OCPs (round 6): the user callables themselves are drawn at random (tests/problems.py: random_ocp_case -- phases, states,
controls, parameters, optional row blocks, scalings, expression trees over (x, u, t, a) / (xf, tf, x0, t0, a), grid, scheme), so the
tracer, the symbolic differentiation and the code generator (mpopt_amd/expr.py, codegen.py: what stands where CasADi's SX graph and AD
stand behind `ca.nlpsol`, reference mpopt.py:757) are checked on expressions nobody wrote by hand.  Every oracle function through
the generated kernels against the numpy / sympy oracle (oracle/mpopt_oracle.py: sympy differentiates the same callables), 1e-10 per
entry; indices exact: the structural patterns cover every non-zero of the oracle's dense derivatives."""
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


@pytest.mark.parametrize("seed,wide", problems.RANDOM_OCPS, ids=[("wide" if w else "smooth") + str(s) for s, w in problems.RANDOM_OCPS])
def test_random_ocp_against_numpy_oracle(seed, wide):
    builder, S, po, scheme = problems.random_ocp_case(seed, wide)
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    O = OracleNLP(ocp, S, po, scheme)
    assert (o.n_z, o.n_g) == (O.n_z, O.n_g)
    z0 = mpo.initialize_solution()
    assert np.array_equal(O.initial_guess(), z0)
    lbx, ubx, lbg, ubg = O.bounds()
    assert all(np.array_equal(np.asarray(bounds[k], float).ravel(), v) for k, v in (("lbx", lbx), ("ubx", ubx), ("lbg", lbg), ("ubg", ubg)))
    rng = np.random.default_rng(seed)
    jr, jc = o.jac_pattern()
    hr, hc = o.hess_pattern()
    for trial in range(3):
        z = z0 + 0.05 * np.abs(z0) * rng.uniform(-1, 1, o.n_z) + 0.1 * rng.uniform(-1, 1, o.n_z)
        w = rng.uniform(0.3, 1.7, (ocp.n_phases, S))
        p = (w / w.sum(axis=1, keepdims=True)).ravel()
        lam, sig = rng.standard_normal(o.n_g), float(rng.uniform(0.2, 2.0))
        r = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], z, p, lam_g=lam, sigma=sig)
        assert rel_err(r["f"], O.f(z, p)) < TOL
        assert rel_err(r["g"], O.g(z, p)) < TOL
        assert rel_err(r["grad_f"], O.grad_f(z, p)) < TOL
        Jd = O.jac_g(z, p)
        Jo = sp.csr_matrix(Jd)
        d = sp.coo_matrix((r["jac_g"], (jr, jc)), shape=(o.n_g, o.n_z)).tocsr() - Jo
        assert (abs(d).max() if d.nnz else 0.0) < TOL * max(1.0, abs(Jo).max())
        Hd = np.triu(O.hess_l(z, p, sig, lam))
        Ho = sp.csr_matrix(Hd)
        d = sp.coo_matrix((r["hess_l"], (hr, hc)), shape=(o.n_z, o.n_z)).tocsr() - Ho
        assert (abs(d).max() if d.nnz else 0.0) < TOL * max(1.0, abs(Ho).max())
        # the structural patterns hold every non-zero of the dense derivatives
        mask = np.zeros((o.n_g, o.n_z), bool)
        mask[jr, jc] = True
        assert not np.any((np.asarray(Jd.todense() if hasattr(Jd, 'todense') else Jd) != 0) & ~mask)
        mask = np.zeros((o.n_z, o.n_z), bool)
        mask[hr, hc] = True
        assert not np.any((Hd != 0) & ~mask)
        # light passes and the separate calls: the same bits of g; nlp_grad
        lg = o.eval(["f", "g"], z, p)
        assert np.array_equal(lg["g"], r["g"]) and abs(lg["f"] - r["f"]) <= 1e-13 * max(1.0, abs(r["f"]))
        q = o.eval_grad_gamma(z, p, lam, sig)
        gx, gp = O.grad_gamma(z, p, sig, lam)
        assert rel_err(q["grad_gamma_x"], gx) < TOL and rel_err(q["grad_gamma_p"], gp) < TOL
    # a batch: every member equal to its single evaluation
    Z = np.stack([z, z0, z * 0.99])
    rb = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], Z, p, lam_g=np.stack([lam] * 3), sigma=np.full(3, sig))
    for key in ("g", "grad_f", "jac_g", "hess_l"):
        assert np.array_equal(rb[key][0], r[key]), key
    o.close()


@pytest.mark.parametrize("seed,wide", problems.RANDOM_OCPS, ids=[("wide" if w else "smooth") + str(s) for s, w in problems.RANDOM_OCPS])
def test_random_ocp_with_widths_as_variables_against_the_exact_ad_oracle(seed, wide):
    """The same random OCPs through mpopt_adaptive (segment widths as decision variables, reference mpopt.py:2927-2979, 3034-3136):
    the assembled contexts (mpopt_amd/assembly.py: per-point derivatives + chain rule through the widths) against the oracle's restated
    value code differentiated exactly by sparse hyper-dual arithmetic (oracle/sparse_ad.py) -- single evaluations and a batch."""
    from oracle.mpopt_oracle import OracleAdaptiveNLP

    builder, S, po, scheme = problems.random_ocp_case(seed, wide)
    ocp = builder(mp, M.math)
    mpo = mp.mpopt_adaptive(ocp, S, po, scheme)
    nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    O = OracleAdaptiveNLP(ocp, S, po, scheme)
    z0 = O.initial_guess()
    assert (o.n_z, o.n_g) == (O.n_z, O.n_g) and np.array_equal(mpo.initialize_solution(), z0)
    lbx, ubx, lbg, ubg = O.bounds()
    assert np.array_equal(bounds["lbx"], lbx) and np.array_equal(bounds["ubx"], ubx) and np.array_equal(bounds["lbg"], lbg) and np.array_equal(bounds["ubg"], ubg)
    rng = np.random.default_rng(100 + seed)
    jr, jc = o.jac_pattern()
    hr, hc = o.hess_pattern()
    jset, hset = set(zip(jr.tolist(), jc.tolist())), set(zip(hr.tolist(), hc.tolist()))
    for B in (1, 33):
        Z = z0[None, :] * (1 + 0.05 * rng.uniform(-1, 1, (B, o.n_z))) + 0.05 * rng.uniform(-1, 1, (B, o.n_z))
        lam, sig = rng.standard_normal((B, o.n_g)), rng.uniform(0.3, 1.7, B)
        r = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], Z if B > 1 else Z[0], None, lam_g=lam if B > 1 else lam[0], sigma=sig if B > 1 else sig[0])
        if B == 1:
            r = {k: np.asarray(v)[None] for k, v in r.items()}
        for b in sorted({0, B - 1}):
            f, g, grad, J = O.ad_first(Z[b])
            H = O.ad_hess_l(Z[b], sig[b], lam[b])
            assert rel_err(r["f"][b], f) < TOL and rel_err(r["g"][b], g) < TOL and rel_err(r["grad_f"][b], grad) < TOL
            Jc, Hc = J.tocoo(), H.tocoo()
            assert set(zip(Jc.row[Jc.data != 0].tolist(), Jc.col[Jc.data != 0].tolist())) <= jset
            assert set(zip(Hc.row[Hc.data != 0].tolist(), Hc.col[Hc.data != 0].tolist())) <= hset
            assert np.abs(r["jac_g"][b] - np.asarray(J[jr, jc]).ravel()).max() < TOL * max(1.0, abs(J).max())
            assert np.abs(r["hess_l"][b] - np.asarray(H[hr, hc]).ravel()).max() < TOL * max(1.0, abs(H).max())
    o.close()

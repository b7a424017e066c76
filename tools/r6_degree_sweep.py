"""Every degree 32 ... 255 once (round 6): a two-segment moon lander / Van der Pol grid per degree, scheme cycling, a ragged batch of 19 -- the
light passes through the matrix-core kernels (mpx_lighthigh_*) against the same calls through the node kernels (MPX_NO_LIGHT=1: g and the node
entries of grad_f bit for bit, f and the (t0, tf) sums to rounding), and f, g, grad_f, jac_g, hess_l, nlp_grad's two outputs and the off-node residuals of three plans against the numpy oracle (tables in 50-digit
arithmetic) at 1e-10.  The suite's high-degree cases pick 12 degrees; this runs all 224 (tile tails of every residue of P + 1 mod 16 and mod 4).  Degrees 1 ... 31: a
single-degree grid of 40 segments and a mixed one in the pattern of BASELINE configs[2] per degree (mpx_lightlow_* / mpx_light_*).
    python tools/r6_degree_sweep.py compile LO HI      (no GPU: fills the in-tree kernel cache)
    python tools/r6_degree_sweep.py run LO HI          (GPU)"""
import os
import sys

os.environ.setdefault("MPX_ENV_DYNAMIC", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scipy.sparse as sp

import mpopt_amd as M
from mpopt_amd import mp, _lib
import problems


def cases(P):
    """Grids of degree P: (builder, S, orders, scheme, light plan expected)."""
    builder = [problems.moon_lander, problems.van_der_pol, problems.dae_vdp][P % 3]
    scheme = ["LGR", "LGL", "CGL"][(P // 3) % 3]
    if P >= 32:
        return [(builder, 2, [P, P], scheme, True)]
    # degrees 1 ... 31: a single-degree grid (register tables / mpx_lightlow_* up to 12, LDS tables / mpx_light_* on the matrix cores above) and
    # a mixed one in the pattern of BASELINE configs[2] (the degree between low-degree neighbours; for P <= 12 next to a degree-13 bucket)
    single = (builder, 40, [P] * 40, scheme, P >= 13 or P >= 2)
    mixed = (builder, 9, [3, P, 3, 3, P, 2, 3, P, 3] if P >= 13 else [P, 13, P, P, 13, P, 13, P, P], scheme, True)
    return [single, mixed]


def main():
    what, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    worst = 0.0
    for P, (builder, S, po, scheme, want_light) in [(P, c) for P in range(lo, hi) for c in cases(P)]:
        ocp = builder(mp, M.math)
        if what == "compile":
            o = M.NlpFunctions(ocp, S, po, scheme, with_device=False)
            _lib.compile_kernels(o.source)
            o.close()
            print(P, "compiled", flush=True)
            continue
        from oracle.mpopt_oracle import OracleNLP

        mpo = mp.mpopt(ocp, S, po, scheme)
        o = mpo.create_nlp()[0]["oracle"]
        has_light = o.light_plan()[1] > 0
        assert has_light or not want_light or P < 13, (P, S, "no light plan")
        rng = np.random.default_rng(P)
        z0 = mpo.initialize_solution()
        B = 19
        Z = z0[None, :] + 0.05 * np.abs(z0)[None, :] * rng.uniform(-1, 1, (B, o.n_z)) + 0.05 * rng.uniform(-1, 1, (B, o.n_z))
        w = rng.uniform(0.3, 1.7, (ocp.n_phases, S))
        p = (w / w.sum(axis=1, keepdims=True)).ravel()
        res = {}
        for no_light in (False, True):
            if no_light:
                os.environ["MPX_NO_LIGHT"] = "1"
            try:
                res[no_light] = [o.eval(["f", "g"], Z, p), o.eval(["f", "grad_f"], Z, p), o.eval(["f", "g", "grad_f"], Z[3], p), o.eval(["g"], Z[:16], p)]
            finally:
                os.environ.pop("MPX_NO_LIGHT", None)
        nn = o.n_z - 2 - ocp.na  # node entries of grad_f
        for a, b in zip(res[False], res[True]):
            if "g" in a:
                assert np.array_equal(a["g"], b["g"]), (P, "g")
            if "grad_f" in a:
                assert np.array_equal(np.asarray(a["grad_f"])[..., :nn], np.asarray(b["grad_f"])[..., :nn]), (P, "grad_f nodes")
                assert np.allclose(a["grad_f"], b["grad_f"], rtol=1e-12, atol=1e-13), (P, "grad_f sums")
            if "f" in a:
                assert np.allclose(a["f"], b["f"], rtol=1e-13, atol=1e-13), (P, "f")
        O = OracleNLP(ocp, S, po, scheme)
        lam, sig = rng.standard_normal((2, o.n_g)), rng.uniform(0.3, 1.7, 2)
        full = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], Z[:2], p, lam_g=lam, sigma=sig)
        jr, jc = o.jac_pattern()
        hr, hc = o.hess_pattern()
        for b in range(2):
            fo, go, qo = O.f(Z[b], p), O.g(Z[b], p), O.grad_f(Z[b], p)
            e = max(abs(full["f"][b] - fo) / max(1, abs(fo)), np.abs(full["g"][b] - go).max() / max(1, np.abs(go).max()),
                    np.abs(full["grad_f"][b] - qo).max() / max(1, np.abs(qo).max()),
                    abs(res[False][0]["f"][b] - fo) / max(1, abs(fo)), np.abs(res[False][0]["g"][b] - go).max() / max(1, np.abs(go).max()))
            Jo = sp.csr_matrix(O.jac_g(Z[b], p))
            d = sp.coo_matrix((full["jac_g"][b], (jr, jc)), shape=(o.n_g, o.n_z)).tocsr() - Jo
            e = max(e, (abs(d).max() if d.nnz else 0.0) / max(1.0, abs(Jo).max()))
            Ho = sp.csr_matrix(np.triu(O.hess_l(Z[b], p, sig[b], lam[b])))
            d = sp.coo_matrix((full["hess_l"][b], (hr, hc)), shape=(o.n_z, o.n_z)).tocsr() - Ho
            e = max(e, (abs(d).max() if d.nnz else 0.0) / max(1.0, abs(Ho).max()))
            q = o.eval_grad_gamma(Z[b], p, lam[b], sig[b])
            gx, gp = O.grad_gamma(Z[b], p, sig[b], lam[b])
            e = max(e, np.abs(q["grad_gamma_x"] - gx).max() / max(1, np.abs(gx).max()), np.abs(q["grad_gamma_p"] - gp).max() / max(1, np.abs(gp).max()))
            assert e < 1e-10, (P, e)
            worst = max(worst, e)
        # off-node residuals (mpx_resid_*: staged in LDS from degree 8 on where the span fits): a dense plan, a one-point-per-segment plan and a
        # plan with empty segments, single evaluation and the batch's point 18, against the numpy oracle
        for taus in ([np.sort(rng.uniform(-1, 1, 11 + sgi % 4)) for sgi in range(S)], [rng.uniform(-1, 1, 1) for _ in range(S)],
                     [np.sort(rng.uniform(-1, 1, 7)) if sgi % 2 == 0 else np.zeros(0) for sgi in range(S)]):
            plan = o.residual_plan(0, taus)
            rb = plan.eval(Z, p)
            for bb in (0, 18):
                r1 = plan.eval(Z[bb], p)
                ref = O.residuals(Z[bb], p, 0, taus)
                for key in ("xi", "ui", "dxi", "dyn", "resid"):
                    assert np.array_equal(r1[key], rb[key][bb]), (P, key)
                    want = np.asarray(ref[key], dtype=float).reshape(plan.n_pts, -1)
                    er = np.abs(np.asarray(r1[key]).reshape(plan.n_pts, -1) - want).max() / max(1.0, np.abs(want).max())
                    assert er < 1e-10, (P, key, er)
                    e = max(e, er)
            plan.close()
        worst = max(worst, e)
        o.close()
        print(P, S, scheme, builder.__name__, "light" if has_light else "no light plan", "ok", f"{e:.1e}", flush=True)
    if what == "run":
        print(f"degrees {lo}..{hi - 1}: light passes bit-identical to the node kernels; worst relative error against the numpy oracle {worst:.2e}")


if __name__ == "__main__":
    main()

"""Random grids with degrees above 31 (round 6 soak; the suite's random grids draw degrees up to 30): 1-3 distinct degrees from 2 ... 255 in runs
of random length, at most ~1200 nodes per phase, six problems (two phases, parameters, time dependence, D.U rows among them), three schemes.
Per seed: f, g, grad_f, jac_g, hess_l and nlp_grad against the numpy oracle at 1e-10; the light passes against the node kernels (g and the
node entries of grad_f bit for bit); a batch against its members.
    python tools/r6_high_soak.py compile LO HI | run LO HI"""
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

DEGS = [2, 3, 5, 12, 13, 20, 31, 32, 33, 47, 64, 68, 69, 70, 93, 94, 100, 127, 128, 129, 200, 255]


def case(seed):
    rng = np.random.default_rng(7000 + seed)
    degs = rng.choice(DEGS, size=int(rng.integers(1, 4)), replace=False)
    if degs.max() < 32:
        degs[0] = int(rng.choice([d for d in DEGS if d >= 32]))
    po = []
    while True:
        d = int(rng.choice(degs))
        run = int(rng.integers(1, 4))
        if sum(po) + d * run > 1200 or len(po) + run > 14:
            break
        po += [d] * run
    if not po:
        po = [int(degs.max())]
    builder = [problems.moon_lander, problems.van_der_pol, problems.dae_vdp, problems.kitchen_sink, problems.time_dependent, problems.hyper_sensitive][seed % 6]
    return builder, len(po), po, ["LGR", "LGL", "CGL"][(seed // 6) % 3]


def main():
    what, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    if what == "compile":
        from concurrent.futures import ThreadPoolExecutor

        srcs = []
        for seed in range(lo, hi):
            b, S, po, sc = case(seed)
            o = M.NlpFunctions(b(mp, M.math), S, po, sc, with_device=False)
            srcs.append(o.source)
            o.close()
        with ThreadPoolExecutor(8) as pool:
            list(pool.map(_lib.compile_kernels, list(dict.fromkeys(srcs))))
        print("compiled", len(set(srcs)))
        return
    from oracle.mpopt_oracle import OracleNLP

    worst = 0.0
    for seed in range(lo, hi):
        builder, S, po, scheme = case(seed)
        ocp = builder(mp, M.math)
        mpo = mp.mpopt(ocp, S, po, scheme)
        o = mpo.create_nlp()[0]["oracle"]
        O = OracleNLP(ocp, S, po, scheme)
        rng = np.random.default_rng(seed)
        z0 = mpo.initialize_solution()
        z = z0 + 0.05 * np.abs(z0) * rng.uniform(-1, 1, o.n_z) + 0.05 * rng.uniform(-1, 1, o.n_z)
        w = rng.uniform(0.3, 1.7, (ocp.n_phases, S))
        p = (w / w.sum(axis=1, keepdims=True)).ravel()
        lam, sig = rng.standard_normal(o.n_g), float(rng.uniform(0.3, 1.7))
        r = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], z, p, lam_g=lam, sigma=sig)
        fo, go, qo = O.f(z, p), O.g(z, p), O.grad_f(z, p)
        e = max(abs(r["f"] - fo) / max(1, abs(fo)), np.abs(r["g"] - go).max() / max(1, np.abs(go).max()), np.abs(r["grad_f"] - qo).max() / max(1, np.abs(qo).max()))
        jr, jc = o.jac_pattern()
        Jo = sp.csr_matrix(O.jac_g(z, p))
        d = sp.coo_matrix((r["jac_g"], (jr, jc)), shape=(o.n_g, o.n_z)).tocsr() - Jo
        e = max(e, (abs(d).max() if d.nnz else 0.0) / max(1.0, abs(Jo).max()))
        hr, hc = o.hess_pattern()
        Ho = sp.csr_matrix(np.triu(O.hess_l(z, p, sig, lam)))
        d = sp.coo_matrix((r["hess_l"], (hr, hc)), shape=(o.n_z, o.n_z)).tocsr() - Ho
        e = max(e, (abs(d).max() if d.nnz else 0.0) / max(1.0, abs(Ho).max()))
        q = o.eval_grad_gamma(z, p, lam, sig)
        gx, gp = O.grad_gamma(z, p, sig, lam)
        e = max(e, np.abs(q["grad_gamma_x"] - gx).max() / max(1, np.abs(gx).max()), np.abs(q["grad_gamma_p"] - gp).max() / max(1, np.abs(gp).max()))
        assert e < 1e-10, (seed, po, e)
        # light passes against the node kernels; a batch against its members
        Z = np.stack([z, z0, z * 1.01, z])
        res = {}
        for no_light in (False, True):
            if no_light:
                os.environ["MPX_NO_LIGHT"] = "1"
            try:
                res[no_light] = [o.eval(["f", "g"], Z, p), o.eval(["f", "grad_f"], Z, p), o.eval(["f", "g", "grad_f"], z, p)]
            finally:
                os.environ.pop("MPX_NO_LIGHT", None)
        nzp = o.n_z // ocp.n_phases
        node = np.zeros(o.n_z, bool)
        for ph in range(ocp.n_phases):
            node[ph * nzp: ph * nzp + (ocp.nx + ocp.nu) * o.n_nodes] = True
        for a, b in zip(res[False], res[True]):
            if "g" in a:
                assert np.array_equal(a["g"], b["g"]), (seed, po, "g")
            if "grad_f" in a:
                assert np.array_equal(np.asarray(a["grad_f"])[..., node], np.asarray(b["grad_f"])[..., node]), (seed, po, "grad_f nodes")
                assert np.allclose(a["grad_f"], b["grad_f"], rtol=1e-12, atol=1e-13)
            assert np.allclose(a["f"], b["f"], rtol=1e-13, atol=1e-13)
        assert np.array_equal(res[False][0]["g"][0], r["g"]) and np.array_equal(res[False][0]["g"][3], r["g"])
        rb = o.eval(["g", "jac_g", "hess_l"], Z, p, lam_g=np.stack([lam] * 4), sigma=np.full(4, sig))
        assert np.array_equal(rb["jac_g"][0], r["jac_g"]) and np.array_equal(rb["jac_g"][3], r["jac_g"]) and np.array_equal(rb["hess_l"][3], r["hess_l"])
        worst = max(worst, e)
        print(seed, builder.__name__, scheme, po, "light plan" if o.light_plan()[1] > 0 else "no light plan", f"{e:.1e}", flush=True)
        o.close()
    print(f"seeds {lo}..{hi - 1}: ok, worst relative error against the numpy oracle {worst:.2e}")


if __name__ == "__main__":
    main()

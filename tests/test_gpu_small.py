"""Single evaluations of small problems in ONE launch (round 6; mpx_small_<mode>, mpx_kernels.h: small_body): one workgroup runs the node
tiles in order, the boundary pass, the compressed-column permutation and raises the completion flag -- where a single evaluation was
2-4 dependent launches.  The same node_body tile by tile, so every output has the bits of the separate launches (MPX_NO_SMALL=1), through
host pointers (zero-copy and staged), device pointers, compressed-column order, the nlp_* entry points, two phases, degrees with the
tables in registers and in LDS.  What a maintainer running the reference's own examples feels: BASELINE.md section 1a / 1c grids."""
import ctypes

import numpy as np
import pytest

import mpopt_amd as M
from mpopt_amd import mp, _lib
import problems

pytestmark = pytest.mark.gpu

CASES = {
    "moon_lander_20x3_LGR": (problems.moon_lander, 20, 3, "LGR"),          # BASELINE configs[0]
    "moon_lander_10x6_LGR": (problems.moon_lander, 10, 6, "LGR"),          # moon_lander.ipynb:171-210
    "moon_lander_2x30_CGL": (problems.moon_lander, 2, 30, "CGL"),          # tables in LDS
    "hyper_sensitive_5x50_LGL": (problems.hyper_sensitive, 5, 50, "LGL"),  # 251 nodes, 3 tiles, 42 KB of tables
    "schwartz_1x20_LGR": (problems.two_phase_schwartz, 1, 20, "LGR"),      # two phases in one workgroup
    "kitchen_sink_3x4": (problems.kitchen_sink, 3, 4, "LGR"),              # time-dependent (prefix pass in front), parameters, all row blocks
    "dae_vdp_30x5": (problems.dae_vdp, 30, 5, "CGL"),                      # 151 nodes: node-0 tile + 1
}


@pytest.mark.parametrize("name", list(CASES))
def test_single_launch_equals_separate_launches_bitwise(name, monkeypatch):
    import torch

    builder, S, P, scheme = CASES[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    assert o.n_tiles <= 8
    rng = np.random.default_rng(3)
    z = mpo.initialize_solution() * (1 + 0.05 * rng.uniform(-1, 1, o.n_z)) + 0.02 * rng.uniform(-1, 1, o.n_z)
    w = rng.uniform(0.5, 1.5, (ocp.n_phases, S))
    p = (w / w.sum(axis=1, keepdims=True)).ravel()
    lam, sig = rng.standard_normal(o.n_g), 0.8
    masks = (["f"], ["g"], ["f", "grad_f"], ["g", "jac_g"], ["f", "g", "grad_f", "jac_g"], ["hess_l"], ["f", "g", "grad_f", "jac_g", "hess_l"])

    def run_all():
        out = []
        for m in masks:
            for kw in (dict(), dict(pinned=True), dict(ccs_order=True), dict(pinned=True, ccs_order=True)):
                r = o.eval(m, z, p, lam_g=lam, sigma=sig, **kw)
                out.append({k: np.array(v, copy=True) for k, v in r.items()})
        # device pointers
        dev = torch.device("cuda:0")
        Z, Pd = torch.tensor(z[None, :], device=dev), torch.tensor(p, device=dev)
        L, Sg = torch.tensor(lam[None, :], device=dev), torch.tensor([sig], device=dev)
        f, g, gr = (torch.zeros(s, dtype=torch.float64, device=dev) for s in ((1,), (1, o.n_g), (1, o.n_z)))
        jv, hv = torch.zeros((1, o.nnz_jac), dtype=torch.float64, device=dev), torch.zeros((1, o.nnz_hess), dtype=torch.float64, device=dev)
        o.eval_device(15, 1, Z, Pd, 0, None, None, f, g, gr, jv, None)
        o.eval_device(16, 1, Z, Pd, 0, L, Sg, None, None, None, None, hv)
        o.sync()
        out.append({k: v.cpu().numpy().copy() for k, v in dict(f=f, g=g, grad_f=gr, jac_g=jv, hess_l=hv).items()})
        # the CasADi-convention entry points in a solver's order (same-iterate cache, zero-copy, completion flag)
        Lb = _lib.lib()
        o.make_current()
        Lb.mpx_current_pin_buffers(1)
        zz, pp, ll, sg = z.copy(), p.copy(), lam.copy(), np.array([sig])
        f1, g1, q1, j1, h1 = np.zeros(1), np.zeros(o.n_g), np.zeros(o.n_z), np.zeros(max(o.nnz_jac, 1)), np.zeros(max(o.nnz_hess, 1))
        vp = lambda arrs: (ctypes.c_void_p * len(arrs))(*[a.ctypes.data if a is not None else None for a in arrs])
        for nme, a, r in (("nlp_f", [zz, pp], [f1]), ("nlp_g", [zz, pp], [g1]), ("nlp_grad_f", [zz, pp], [f1, q1]), ("nlp_jac_g", [zz, pp], [None, j1]),
                          ("nlp_hess_l", [zz, pp, sg, ll], [h1])):
            assert getattr(Lb, nme)(vp(a), vp(r), None, None, 0) == 0
        Lb.mpx_current_pin_buffers(0)
        Lb.mpx_set_current(None)
        out.append(dict(f=f1.copy(), g=g1.copy(), grad_f=q1.copy(), jac_g=j1.copy(), hess_l=h1.copy()))
        return out

    small = run_all()
    monkeypatch.setenv("MPX_NO_SMALL", "1")
    plain = run_all()
    for a, b in zip(small, plain):
        assert a.keys() == b.keys()
        for k in a:
            assert np.array_equal(a[k], b[k]), (name, k)
    # ... and the values themselves: finite, the compressed-column order really is the permutation of the native one
    perm, _ = o.ccs_perm("jac")
    assert np.array_equal(small[4 * 4 + 2]["jac_g"], small[4 * 4]["jac_g"][perm]) and np.isfinite(small[-1]["hess_l"]).all()

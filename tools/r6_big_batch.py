"""Batches whose output arrays pass 2^31 ELEMENTS (round 6): configs[1] (moon lander 1000 x 5: nnz_jac = 120 020) at B = 20 000 and 40 000 -- jac_g values of
19 / 38 GB, element offsets up to 4.8e9 -- and configs[2] (mixed grid, packed g) at B = 9 000; points on both sides of the 2^31 boundary against the same
points in a batch of 3 (bit for bit).    python tools/r6_big_batch.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mpopt_amd as M
from mpopt_amd import mp
from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_HESS, MPX_JAC
import problems

dev = torch.device("cuda:0")
for (builder, S, po, scheme), Bs in ((problems.BENCH_CASES[0], (20000, 40000)), (problems.BENCH_CASES[1], (9000,)), (problems.BENCH_CASES[3], (60000,))):
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    z0 = torch.tensor(mpo.initialize_solution(), device=dev)
    p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
    for B in Bs:
        g_ = torch.Generator(device=dev).manual_seed(B)
        Z = z0[None, :] + 0.03 * torch.randn((B, o.n_z), generator=g_, device=dev, dtype=torch.float64)
        lam = torch.randn((B, o.n_g), generator=g_, device=dev, dtype=torch.float64)
        sig = torch.rand(B, generator=g_, device=dev, dtype=torch.float64) + 0.5
        mk = lambda *s_: torch.full(s_, float("nan"), dtype=torch.float64, device=dev)
        f, g, q, jv, hv = mk(B), mk(B, o.n_g), mk(B, o.n_z), mk(B, o.nnz_jac), mk(B, o.nnz_hess)
        o.eval_device(31, B, Z, p, 0, lam, sig, f, g, q, jv, hv)
        o.sync()
        edge = (2 ** 31) // o.nnz_jac
        pts = sorted({0, 1, edge - 1, edge, edge + 1, B // 2, B - 2, B - 1} & set(range(B)))
        fin = all(bool(torch.isfinite(x[pts]).all()) for x in (f, g, q, jv, hv))
        for i in pts:
            idx = torch.tensor([i, 0, i], device=dev)
            f3, g3, q3, j3, h3 = mk(3), mk(3, o.n_g), mk(3, o.n_z), mk(3, o.nnz_jac), mk(3, o.nnz_hess)
            o.eval_device(31, 3, Z[idx].contiguous(), p, 0, lam[idx].contiguous(), sig[idx].contiguous(), f3, g3, q3, j3, h3)
            o.sync()
            assert torch.equal(g[i], g3[0]) and torch.equal(jv[i], j3[0]) and torch.equal(hv[i], h3[0]) and torch.equal(q[i], q3[0]) and torch.equal(g[i], g3[2]), (B, i)
            assert abs(float(f[i] - f3[0])) <= 1e-12 * max(1.0, abs(float(f3[0])))
        nan_rows = int(torch.isnan(jv[:, 0]).sum()) + int(torch.isnan(jv[:, -1]).sum()) + int(torch.isnan(g[:, -1]).sum()) + int(torch.isnan(hv[:, -1]).sum())
        print(f"{builder.__name__} {S} segments, B = {B}: jac_g array {B * o.nnz_jac * 8 / 2**30:.1f} GiB ({B * o.nnz_jac / 2**31:.2f} x 2^31 elements), points {pts} bit-identical to a batch of 3; "
              f"first / last entries of every point written: {nan_rows == 0}, finite {fin}", flush=True)
        assert nan_rows == 0 and fin
        del f, g, q, jv, hv, Z, lam, sig
        torch.cuda.empty_cache()
    o.close()
print("big batches: ok")

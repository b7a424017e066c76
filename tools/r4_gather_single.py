"""Small batches of the assembled (mpopt_adaptive) path on device pointers: two-pass kernels (mpx_pts_* + mpx_gather_kernel), wall time
per call.  A/B of the library: MPX_LIB_HIPCC_FLAGS=-DMPX_GATHER_OUT_SEARCH python tools/r4_gather_single.py"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
for name, builder, S, P in (("moon_lander 20x5", problems.moon_lander, 20, 5), ("kitchen_sink 6x4", problems.kitchen_sink, 6, 4)):
    mpo = mp.mpopt_adaptive(builder(mp, M.math), S, P, "LGR")
    o = mpo.create_nlp()[0]["oracle"]
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    for B in (1, 8, 64):
        Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
        lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev); sig = torch.ones(B, dtype=torch.float64, device=dev)
        f, g = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
        gr, jv = torch.empty(B, o.n_z, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
        hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
        line = f"{name:18s} B {B:3d}"
        for tag, mask in (("fgj", 15), ("hess", 16)):
            for _ in range(20): o.eval_device(mask, B, Z, None, 0, lam, sig, f, g, gr, jv, hv)
            o.sync(); best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(200): o.eval_device(mask, B, Z, None, 0, lam, sig, f, g, gr, jv, hv)
                o.sync(); best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
            line += f"  {tag} {best:6.2f} us"
        print(line, flush=True)
    o.close()

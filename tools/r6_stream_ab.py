"""Streamed tables (MPX_TABLES_STREAM_ABOVE, mpx_kernels.h: node_body TAB_GLB) against tables in LDS, same process, same arrays:
whole-pass time of f+g+grad_f+jac_g, of g alone and of nlp_grad for one degree per line.  Picks the default threshold.
    python tools/r6_stream_ab.py [B]"""
import os
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mpopt_amd as M
from mpopt_amd import mp
import problems

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")


def make(P, S, thr):
    os.environ["MPX_TABLES_STREAM_ABOVE"] = str(thr)
    mpo = mp.mpopt(problems.van_der_pol(mp, M.math), S, P, "CGL")
    return mpo, mpo.create_nlp()[0]["oracle"]


def timeit(fn, o, reps=30):
    for _ in range(5):
        fn()
    o.sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    for _ in range(3):
        o.timer_start()
        for _ in range(reps):
            fn()
        best = min(best, o.timer_stop() / reps)
    return best * 1e3  # us


print(f"B = {B}; us per pass: LDS tables | streamed tables")
for P in [int(x) for x in os.environ.get("PLIST", "13,20,30,40,48,56,64,80,92").split(",")]:
    S = max(1, 5000 // P)
    res = {}
    for thr in ((255, 12) if P <= 92 else (12, 12)):
        mpo, o = make(P, S, thr)
        rng = np.random.default_rng(1)
        z0 = mpo.initialize_solution()
        Z = torch.tensor(z0[None, :] + 0.05 * rng.standard_normal((B, o.n_z)), device=dev)
        p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
        lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev)
        sig = torch.ones(B, dtype=torch.float64, device=dev)
        f, g, gr = torch.empty(B, dtype=torch.float64, device=dev), torch.empty((B, o.n_g), dtype=torch.float64, device=dev), torch.empty((B, o.n_z), dtype=torch.float64, device=dev)
        jv = torch.empty((B, o.nnz_jac), dtype=torch.float64, device=dev)
        gp = torch.empty((B, o.n_p), dtype=torch.float64, device=dev)
        t_fgj = timeit(lambda: o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None), o)
        t_g = timeit(lambda: o.eval_device(2, B, Z, p, 0, None, None, None, g, None, None, None), o)
        t_gl = timeit(lambda: o.eval_grad_gamma_device(B, Z, p, lam, sig, gr, gp), o, reps=10)
        by = 8 * (2 * o.n_z + o.n_p + o.n_g + o.nnz_jac + 1) * B
        res[thr] = (t_fgj, t_g, t_gl, by / t_fgj / 1e6)
        o.close()
        del Z, jv
    a, b = res.get(255, res[12]), res[12]
    print(f"P={P:3d} S={S:4d}  fgj {a[0]:8.1f} | {b[0]:8.1f} us ({a[3]:.2f} | {b[3]:.2f} TB/s)   g {a[1]:7.1f} | {b[1]:7.1f}   nlp_grad {a[2]:7.1f} | {b[2]:7.1f}", flush=True)

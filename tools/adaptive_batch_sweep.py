"""Throughput of the assembled path (moon lander 20x5 adaptive) against the batch size: does a raw buffer that fits the caches
between the point kernels and the gather pass pay?  One JSON line per batch size."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_HESS, MPX_JAC  # noqa: E402
import problems  # noqa: E402

mpo = mp.mpopt_adaptive(problems.moon_lander(mp, M.math), 20, 5, "LGR")
o = mpo.create_nlp()[0]["oracle"]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for B in (256, 512, 1024, 2048, 4096, 8192, 16384):
    Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
    lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev)
    sig = torch.ones(B, dtype=torch.float64, device=dev)
    f, g = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
    gr, jv = torch.empty(B, o.n_z, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
    hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
    out = {"batch": B, "raw_MB_fgj": round(B * o.raw_n * 8 / 1e6, 1)}
    for tag, mask in (("fgj", MPX_F | MPX_G | MPX_GRAD | MPX_JAC), ("hess", MPX_HESS)):
        for _ in range(10):
            o.eval_device(mask, B, Z, None, 0, lam, sig, f, g, gr, jv, hv)
        o.sync()
        reps = max(20, 200000 // B)
        o.timer_start()
        for _ in range(reps):
            o.eval_device(mask, B, Z, None, 0, lam, sig, f, g, gr, jv, hv)
        ms = o.timer_stop() / reps
        out[tag + "_Mevals_per_s"] = round(B / ms / 1e3, 2)
    print(json.dumps(out), flush=True)

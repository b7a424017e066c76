"""Throughput / latency of the assembled (mpopt_adaptive) path on one GPU: batched f+g+grad_f+jac_g and hess_l
with device-resident inputs, plus single-evaluation host latency.  Prints one JSON line per case."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_HESS, MPX_JAC  # noqa: E402
import problems  # noqa: E402

CASES = [("moon_lander", problems.moon_lander, 20, 5, "LGR", 4096), ("hyper_sensitive", problems.hyper_sensitive, 40, 4, "LGR", 2048),
         ("kitchen_sink", problems.kitchen_sink, 6, 4, "LGR", 2048)]
ONLY = sys.argv[1:]
for name, builder, S, P, scheme, B in CASES:
    if ONLY and name not in ONLY:
        continue
    mpo = mp.mpopt_adaptive(builder(mp, M.math), S, P, scheme)
    t = time.time()
    o = mpo.create_nlp()[0]["oracle"]
    t_build = time.time() - t
    dev = torch.device("cuda", 0)
    z0 = mpo.initialize_solution()
    rng = np.random.default_rng(0)
    Z = torch.tensor(z0[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
    lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev)
    sig = torch.ones(B, dtype=torch.float64, device=dev)
    f, g = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
    gr, jv = torch.empty(B, o.n_z, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
    hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
    out = {"case": f"{name} {S}x{P} {scheme} adaptive", "n_z": o.n_z, "n_g": o.n_g, "nnz_jac": o.nnz_jac, "nnz_hess": o.nnz_hess, "batch": B,
           "raw_doubles_fgj": int(o.raw_n), "raw_doubles_hess": int(o.rawh_n), "gather_terms_fgj": int(len(o.fgj[1])),
           "gather_terms_hess": int(len(o.hess[1])), "build_s": round(t_build, 2)}
    for tag, mask, alg in (("fgj", MPX_F | MPX_G | MPX_GRAD | MPX_JAC, 8 * (2 * o.n_z + o.n_g + o.nnz_jac + 1)),
                           ("hess", MPX_HESS, 8 * (o.n_z + o.n_g + 1 + o.nnz_hess))):
        for _ in range(5):
            o.eval_device(mask, B, Z, None, 0, lam, sig, f, g, gr, jv, hv)
        o.sync()
        reps = 50
        o.timer_start()
        for _ in range(reps):
            o.eval_device(mask, B, Z, None, 0, lam, sig, f, g, gr, jv, hv)
        ms = o.timer_stop() / reps
        out[tag] = {"ms_per_batch": round(ms, 4), "evals_per_s": round(B / ms * 1e3), "algorithmic_GBps": round(alg * B / ms / 1e6, 1),
                    "bytes_per_eval": alg}
    z1, l1 = Z[0].cpu().numpy(), lam[0].cpu().numpy()
    for what in (["f", "g", "grad_f", "jac_g"], ["hess_l"]):
        for _ in range(20):
            o.eval(what, z1, None, lam_g=l1, sigma=1.0, pinned=True)
        t = time.perf_counter()
        for _ in range(200):
            o.eval(what, z1, None, lam_g=l1, sigma=1.0, pinned=True)
        out["latency_us_" + ("hess" if what == ["hess_l"] else "fgj")] = round((time.perf_counter() - t) / 200 * 1e6, 1)
    print(json.dumps(out), flush=True)
    o.close()

"""Assembled contexts (mpopt_adaptive): the fused persistent kernels (mpx_assembly_fused.h) against the two-pass kernels in ONE
process -- bitwise comparison of every output and HIP-event timing of both, per case and pass.  Prints one JSON line per case.
usage: python tools/r3_fused_ab.py [case ...]     (MPX_FUSE_WG=n, MPX_HIPCC_FLAGS="-DMPX_FUSE_NT=.. -DMPX_FUSE_MAX_U=.." to explore)"""
import json
import os
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_HESS, MPX_JAC  # noqa: E402
import problems  # noqa: E402

CASES = [("moon_lander", problems.moon_lander, 20, 5, "LGR", 4096), ("hyper_sensitive", problems.hyper_sensitive, 40, 4, "LGR", 2048),
         ("kitchen_sink", problems.kitchen_sink, 6, 4, "LGR", 2048), ("van_der_pol_mixed", problems.van_der_pol, 9, [2, 4, 3] * 3, "CGL", 1000)]
ONLY = sys.argv[1:]
dev = torch.device("cuda", 0)
for name, builder, S, P, scheme, B in CASES:
    if ONLY and name not in ONLY:
        continue
    mpo = mp.mpopt_adaptive(builder(mp, M.math), S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    z0 = mpo.initialize_solution()
    rng = np.random.default_rng(0)
    Z = torch.tensor(z0[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))) + 0.01 * rng.uniform(-1, 1, (B, o.n_z)), device=dev)
    lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev)
    sig = torch.tensor(rng.uniform(0.5, 1.5, B), device=dev)
    mk = lambda *s: torch.full(s, float("nan"), dtype=torch.float64, device=dev)
    out = {"case": f"{name} {S}x{P} {scheme} adaptive", "batch": B, "n_z": o.n_z, "n_g": o.n_g, "nnz_jac": o.nnz_jac, "nnz_hess": o.nnz_hess,
           "raw_doubles": [int(o.raw_n), int(o.rawh_n)]}
    for tag, mask, alg in (("fgj", MPX_F | MPX_G | MPX_GRAD | MPX_JAC, 8 * (2 * o.n_z + o.n_g + o.nnz_jac + 1)), ("fg", MPX_F | MPX_G, 8 * (o.n_z + o.n_g + 1)),
                           ("hess", MPX_HESS, 8 * (o.n_z + o.n_g + 1 + o.nnz_hess))):
        res = {}
        for variant in ("two_pass", "fused"):
            if variant == "two_pass":
                os.environ["MPX_NO_FUSE"] = "1"
            else:
                os.environ.pop("MPX_NO_FUSE", None)
            bufs = (mk(B), mk(B, o.n_g), mk(B, o.n_z), mk(B, o.nnz_jac), mk(B, o.nnz_hess))
            for _ in range(5):
                o.eval_device(mask, B, Z, None, 0, lam, sig, *bufs)
            o.sync()
            reps = 30
            o.timer_start()
            for _ in range(reps):
                o.eval_device(mask, B, Z, None, 0, lam, sig, *bufs)
            ms = o.timer_stop() / reps
            res[variant] = (ms, bufs)
        os.environ.pop("MPX_NO_FUSE", None)
        same = all(torch.equal(a, b) or (torch.isnan(a).all() and torch.isnan(b).all()) for a, b in zip(res["two_pass"][1], res["fused"][1]))
        bad = [k for k, (a, b) in enumerate(zip(res["two_pass"][1], res["fused"][1])) if not (torch.equal(a, b) or (torch.isnan(a).all() and torch.isnan(b).all()))]
        worst = max([float((a - b).abs().nan_to_num(0).max()) for a, b in zip(res["two_pass"][1], res["fused"][1])])
        out[tag] = {"two_pass_ms": round(res["two_pass"][0], 4), "fused_ms": round(res["fused"][0], 4), "bit_identical": bool(same), "arrays_differing": bad,
                    "max_abs_diff": worst, "fused_algorithmic_GBps": round(alg * B / res["fused"][0] / 1e6, 1), "two_pass_algorithmic_GBps": round(alg * B / res["two_pass"][0] / 1e6, 1)}
    print(json.dumps(out), flush=True)
    o.close()

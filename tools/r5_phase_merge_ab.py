"""In-process A/B of the all-phases launches (round 5) against one launch per phase (MPX_NO_PHASE_MERGE=1, read per call) on config 4
(two-phase Schwartz, 500 x 3 per phase): the SAME context and arrays, interleaved rounds, every oracle.
    B=4096 python tools/r5_phase_merge_ab.py"""
import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems

builder, S, P, scheme = problems.BENCH_CASES[2]
mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
o = mpo.create_nlp()[0]["oracle"]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for B in [int(b) for b in os.environ.get("B", "4096,512,1").split(",")]:
    Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
    p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
    f = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev); q = torch.empty(B, o.n_z, dtype=torch.float64, device=dev)
    jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev); hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
    lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev); sig = torch.ones(B, dtype=torch.float64, device=dev)
    for name, mask in (("nlp_f", 1), ("nlp_g", 2), ("f+grad_f", 5), ("f+g+grad_f+jac_g", 15), ("hess_l", 16)):
        args = (mask, B, Z, p, 0, lam, sig, f if mask & 1 else None, g if mask & 2 else None, q if mask & 4 else None, jv if mask & 8 else None, hv if mask & 16 else None)
        byt = 8 * B * (o.n_z + o.n_p + (1 if mask & 1 else 0) + (o.n_g if mask & 2 else 0) + (o.n_z if mask & 4 else 0) + (o.nnz_jac if mask & 8 else 0) + (o.n_g + 1 + o.nnz_hess if mask & 16 else 0))
        res, outs = {"merged": [], "per phase": []}, {}
        reps = 30 if B >= 512 else 200
        for rnd in range(7):
            for key in ("merged", "per phase"):
                if key == "per phase":
                    os.environ["MPX_NO_PHASE_MERGE"] = "1"
                for _ in range(3): o.eval_device(*args)
                o.sync(); o.timer_start()
                for _ in range(reps): o.eval_device(*args)
                res[key].append(o.timer_stop() / reps * 1e3)
                if rnd == 0: outs[key] = [a.clone() for a in (f, g, q, jv, hv)]
                os.environ.pop("MPX_NO_PHASE_MERGE", None)
        same = all(torch.equal(a, b) for a, b in zip(outs["merged"], outs["per phase"]))
        for key in res:
            med = sorted(res[key])[len(res[key]) // 2]
            print(f"B={B:5d} {name:17s} [{key:9s}] median {med:8.2f} us (whole pass)  min {min(res[key]):8.2f}  {byt / med / 1e6:5.2f} TB/s = {byt / med / 8e6:.3f} of peak   bit-equal: {same}")

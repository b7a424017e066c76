"""One evaluation point per workgroup as a COMPILE-TIME fact (-DMPX_ABL_BPB1: no software-pipeline state) against the run-time batch loop
at MPX_BPB=1 and at the library's choice: config given by CASE, masks all-four / hess_l / f+g."""
import os
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
import problems  # noqa: E402

case = int(os.environ.get("CASE", 0))
builder, S, P, scheme = problems.BENCH_CASES[case]
B = int(os.environ.get("B", 4096 if case != 1 else 512))
dev = torch.device("cuda", 0)
objs = {}
for name, fl in (("loop", ""), ("bpb1", os.environ.get("BPB1_FLAGS", "-DMPX_ABL_BPB1=1"))):
    os.environ["MPX_HIPCC_FLAGS"] = fl
    mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
    objs[name] = mpo.create_nlp()[0]["oracle"]
o = objs["loop"]
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z)))).to(dev)
p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
f = torch.empty(B, dtype=torch.float64, device=dev)
g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
gr = torch.empty(B, o.n_z, dtype=torch.float64, device=dev)
jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
jv2 = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
lam = torch.randn(B, o.n_g, dtype=torch.float64, device=dev)
sig = torch.ones(B, dtype=torch.float64, device=dev)
hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
hv2 = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
for name, mask in (("all four", 15), ("hess_l", 16), ("f+g", 3)):
    res = {}
    for rnd in range(3):
        for variant, key, bpb in (("loop, library", "loop", None), ("loop, MPX_BPB=1", "loop", "1"), ("compile-time 1", "bpb1", "1")):
            os.environ.pop("MPX_BPB", None)
            if bpb:
                os.environ["MPX_BPB"] = bpb
            ob = objs[key]
            J, H = (jv2, hv2) if key == "bpb1" else (jv, hv)
            args = (mask, B, Z, p, 0, lam if mask & 16 else None, sig if mask & 16 else None, f if mask & 1 else None, g if mask & 2 else None,
                    gr if mask & 4 else None, J if mask & 8 else None, H if mask & 16 else None)
            for _ in range(8):
                ob.eval_device(*args)
            ob.sync()
            ob.timer_start()
            for _ in range(10):
                ob.eval_device(*args)
            res.setdefault(variant, []).append(round(ob.timer_stop() / 10 * 1e3, 1))
    same = torch.equal(jv, jv2) if mask & 8 else torch.equal(hv, hv2) if mask & 16 else True
    print(f"{name:10s} {res}  identical: {same}", flush=True)

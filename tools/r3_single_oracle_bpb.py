"""Single oracles at large batches (config 2, B = 4096): evaluation points per workgroup (MPX_BPB) against the library's choice."""
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
mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
o = mpo.create_nlp()[0]["oracle"]
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z)))).to(dev)
p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
f = torch.empty(B, dtype=torch.float64, device=dev)
g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
gr = torch.empty(B, o.n_z, dtype=torch.float64, device=dev)
jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
lam = torch.randn(B, o.n_g, dtype=torch.float64, device=dev)
sig = torch.ones(B, dtype=torch.float64, device=dev)
hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
ONLY = os.environ.get("ONLY", "").split(",") if os.environ.get("ONLY") else None
for name, mask in (("f", 1), ("g", 2), ("grad_f", 4), ("f+g", 3), ("f+g+grad_f", 7), ("jac_g", 8), ("all four", 15), ("hess_l", 16)):
    if ONLY and name not in ONLY:
        continue
    row = []
    for bpb in (0, 1, 2, 4, 8, 16):
        os.environ.pop("MPX_BPB", None)
        if bpb:
            os.environ["MPX_BPB"] = str(bpb)
        args = (mask, B, Z, p, 0, lam if mask & 16 else None, sig if mask & 16 else None, f if mask & 1 else None, g if mask & 2 else None, gr if mask & 4 else None, jv if mask & 8 else None, hv if mask & 16 else None)
        for _ in range(8):
            o.eval_device(*args)
        o.sync()
        o.timer_start()
        for _ in range(10):
            o.eval_device(*args)
        row.append((bpb, round(o.timer_stop() / 10 * 1e3, 1)))
    print(f"{name:12s} us per pass by MPX_BPB (0 = library): {row}", flush=True)

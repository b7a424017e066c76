"""Same-process A/B of the packed g / grad_f staging on a mixed-degree grid (BASELINE configs[2]): MPX_NO_PACKED_G toggled per round."""
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

B = 512
builder, S, P, scheme = problems.BENCH_CASES[1]
mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
o = mpo.create_nlp()[0]["oracle"]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.01 * rng.standard_normal((B, o.n_z)), device=dev)
p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
f, g = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
gr, jv = torch.empty(B, o.n_z, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
res = {"packed": [], "direct": []}
for rnd in range(8):
    for mode in ("packed", "direct"):
        if mode == "direct":
            os.environ["MPX_NO_PACKED_G"] = "1"
        else:
            os.environ.pop("MPX_NO_PACKED_G", None)
        for _ in range(3):
            o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
        o.sync()
        o.timer_start()
        for _ in range(10):
            o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
        res[mode].append(o.timer_stop() / 10 * 1e3)
for mode, v in res.items():
    v = np.array(v)
    print(f"{mode}: median {np.median(v):8.1f} us per step  (min {v.min():.1f}, max {v.max():.1f})  {B * o.bytes_fgj / np.median(v) / 1e6:.2f} TB/s algorithmic")

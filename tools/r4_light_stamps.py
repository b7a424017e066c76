import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
builder, S, P, scheme = problems.BENCH_CASES[1]
B = 512
dev = torch.device("cuda", 0)
mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
o = mpo.create_nlp()[0]["oracle"]
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z)))).to(dev)
p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
f = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
for mask, name in ((1, "f"), (2, "g")):
    for _ in range(5):
        o.eval_device(mask, B, Z, p, 0, None, None, f if mask & 1 else None, g if mask & 2 else None, None, None, None)
    o.sync()
    print(name, flush=True)
    os.environ["MPX_LIGHT_DEBUG"] = "1"
    for _ in range(3):
        o.eval_device(mask, B, Z, p, 0, None, None, f if mask & 1 else None, g if mask & 2 else None, None, None, None)
    o.sync()
    del os.environ["MPX_LIGHT_DEBUG"]

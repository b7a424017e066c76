"""Phase stamps of one wavefront's third item in the low-degree light kernels (build the kernels with MPX_HIPCC_FLAGS=-DMPX_LIGHT_STAMPS,
run with MPX_LIGHT_DEBUG=1 set by this script).  Usage: MPX_HIPCC_FLAGS=-DMPX_LIGHT_STAMPS python tools/r4_lightlow_stamps.py [case]"""
import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
builder, S, P, scheme = problems.BENCH_CASES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
B = 4096
dev = torch.device("cuda", 0)
mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
o = mpo.create_nlp()[0]["oracle"]
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z)))).to(dev)
p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
f = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev); q = torch.empty(B, o.n_z, dtype=torch.float64, device=dev)
for mask, name in ((1, "f"), (2, "g"), (5, "f+grad_f")):
    args = (mask, B, Z, p, 0, None, None, f if mask & 1 else None, g if mask & 2 else None, q if mask & 4 else None, None, None)
    for _ in range(5):
        o.eval_device(*args)
    o.sync()
    print(name, flush=True)
    os.environ["MPX_LIGHT_DEBUG"] = "1"
    for _ in range(3):
        o.eval_device(*args)
    o.sync()
    del os.environ["MPX_LIGHT_DEBUG"]

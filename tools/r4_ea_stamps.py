"""Phase stamps of the equal-area fast kernel at config-5 sizes (second point of workgroup 0):
MPX_LIB_HIPCC_FLAGS=-DMPX_EA_STAMPS MPX_EA_DEBUG=1 python tools/r4_ea_stamps.py [B]   (diagnostics build; rebuild the library afterwards)"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp, _lib
import problems
builder, S, P, scheme = problems.BENCH_CASES[3]
mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
o = mpo.create_nlp()[0]["oracle"]
dev = torch.device("cuda:0")
n_pts = S * P
rng = np.random.default_rng(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = torch.tensor(np.abs(rng.standard_normal((B, n_pts, 1))) + 0.01, device=dev)
p0 = torch.tensor(rng.dirichlet(np.ones(S), B), device=dev)
p1 = torch.empty_like(p0)
for _ in range(4):
    o.equal_area_widths_device(0, B, n_pts, R, p0, p1, damping=0.4, p_in_per_point=1)
o.sync()

"""Equal-area kernel alone (config-5 sizes: 4000 segments, 12000 samples per point): time against the batch size -- what is per
point, what is per launch.  python tools/r4_ea_scaling.py"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
builder, S, P, scheme = problems.BENCH_CASES[3]
mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
o = mpo.create_nlp()[0]["oracle"]
dev = torch.device("cuda:0")
n_pts = S * P
rng = np.random.default_rng(1)
for B in (1, 64, 128, 256, 257, 512, 768, 1024, 2048):
    R = torch.tensor(np.abs(rng.standard_normal((B, n_pts, 1))) + 0.01, device=dev)
    p0 = torch.tensor(rng.dirichlet(np.ones(S), B), device=dev)
    p1 = torch.empty_like(p0)
    for _ in range(5): o.equal_area_widths_device(0, B, n_pts, R, p0, p1, damping=0.4, p_in_per_point=1)
    o.sync(); t0 = time.perf_counter()
    n = 200
    for _ in range(n): o.equal_area_widths_device(0, B, n_pts, R, p0, p1, damping=0.4, p_in_per_point=1)
    o.sync(); dt = (time.perf_counter() - t0) / n
    print(f"B {B:5d}  {dt * 1e6:8.2f} us per call  ({dt * 1e6 / max(1, -(-B // 256)):.2f} us per round of 256 points)")

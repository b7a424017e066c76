"""Fused assembled kernels (moon lander 20x5 adaptive): time of the first-order pass against the batch size -- fixed cost per launch
(prologue: row tables into registers, dictionaries into LDS) against cost per chunk.  python tools/r4_adaptive_scaling.py"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
mpo = mp.mpopt_adaptive(problems.moon_lander(mp, M.math), 20, 5, "LGR")
o = mpo.create_nlp()[0]["oracle"]
dev = torch.device("cuda", 0)
z0 = mpo.initialize_solution()
rng = np.random.default_rng(0)
alg = 8 * (2 * o.n_z + o.n_g + o.nnz_jac + 1)
for B in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
    Z = torch.tensor(z0[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
    f, g = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
    gr, jv = torch.empty(B, o.n_z, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
    for _ in range(5): o.eval_device(15, B, Z, None, 0, None, None, f, g, gr, jv, None)
    o.sync(); o.timer_start()
    for _ in range(50): o.eval_device(15, B, Z, None, 0, None, None, f, g, gr, jv, None)
    ms = o.timer_stop() / 50
    print(f"B {B:6d}  {ms * 1e3:8.2f} us  {alg * B / ms / 1e9:7.3f} TB/s  frac {alg * B / ms / 1e9 / 8:.3f}")

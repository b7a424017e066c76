"""Scratch performance explorer (not part of the product): FGJ / HESS throughput vs batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import mpopt_amd as M
from mpopt_amd import mp
import problems

S = int(os.environ.get("S", 1000)); P = int(os.environ.get("P", 5))
ocp = problems.moon_lander(mp, M.math)
mpo = mp.mpopt(ocp, S, P, "LGR")
nlp, bounds = mpo.create_nlp()
o = nlp["oracle"]
print("n_z", o.n_z, "n_g", o.n_g, "nnz_j", o.nnz_jac, "nnz_h", o.nnz_hess, "tiles", o.n_tiles, "bytes_fgj", o.bytes_fgj, "bytes_hess", o.bytes_hess)
dev = torch.device("cuda:0")
z0 = mpo.initialize_solution()
o.set_stream(torch.cuda.current_stream().cuda_stream)
for B in [int(b) for b in os.environ.get("BS", "1,64,512,4096").split(",")]:
    rng = np.random.default_rng(1)
    Z = torch.tensor(z0[None, :] + 0.01 * rng.standard_normal((B, o.n_z)), device=dev)
    p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
    f = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
    gr = torch.empty(B, o.n_z, dtype=torch.float64, device=dev); jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
    lam = torch.randn(B, o.n_g, dtype=torch.float64, device=dev); sig = torch.ones(B, dtype=torch.float64, device=dev)
    hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
    for mask, name, nbytes in [(15, "fgj", o.bytes_fgj), (16, "hess", o.bytes_hess), (3, "fg", 8 * (o.n_z + o.n_p + o.n_g + 1))]:
        for _ in range(3):
            o.eval_device(mask, B, Z, p, 0, lam, sig, f, g, gr, jv, hv)
        o.sync(); o.profile(True)
        K = 20
        t = time.perf_counter()
        for _ in range(K):
            o.eval_device(mask, B, Z, p, 0, lam, sig, f, g, gr, jv, hv)
        o.sync(); wall = (time.perf_counter() - t) / K
        ms, n = o.profile_read(); o.profile(False)
        kt = ms / 1e3 / K
        print(f"B={B:5d} {name:4s} wall/step {wall*1e6:9.1f} us  node-kernel {kt*1e6:9.1f} us  evals/s {B/wall:12.0f}  alg GB/s (kernel) {B*nbytes/kt/1e9:8.1f}  frac8T {B*nbytes/kt/8e12:.3f}")

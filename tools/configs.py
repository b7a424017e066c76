"""Scratch: throughput of every BASELINE.json config (fgj and hess), device-resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
dev = torch.device("cuda:0")
names = ["C2 moon lander 1000x5 LGR", "C3 vdp 2000x[3,30,3] CGL", "C4 schwartz 2x500x3 LGL", "C5 hypersens 4000x3 LGR"]
for name, (builder, S, po, scheme) in zip(names, problems.BENCH_CASES):
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    nlp, _ = mpo.create_nlp(); o = nlp["oracle"]
    for B in [int(b) for b in os.environ.get("BS", "1,512").split(",")]:
        if B * o.nnz_jac * 8 > 40e9: continue
        rng = np.random.default_rng(1)
        Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.01 * rng.standard_normal((B, o.n_z)), device=dev)
        p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
        f = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
        gr = torch.empty(B, o.n_z, dtype=torch.float64, device=dev); jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
        lam = torch.randn(B, o.n_g, dtype=torch.float64, device=dev); sig = torch.ones(B, dtype=torch.float64, device=dev)
        hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
        for mask, nm, nb in [(15, "fgj", o.bytes_fgj), (16, "hess", o.bytes_hess)]:
            for _ in range(30): o.eval_device(mask, B, Z, p, 0, lam, sig, f, g, gr, jv, hv)
            o.sync(); o.profile(True); K = 20; t = time.perf_counter()
            for _ in range(K): o.eval_device(mask, B, Z, p, 0, lam, sig, f, g, gr, jv, hv)
            o.sync(); wall = (time.perf_counter() - t) / K; ms, n = o.profile_read(); o.profile(False)
            kt = ms / 1e3 / K
            print(f"{name:28s} B={B:5d} {nm:4s} nnz={o.nnz_jac if nm=='fgj' else o.nnz_hess:8d} bytes/eval={nb:9d} wall {wall*1e6:8.1f} us node-kernels {kt*1e6:8.1f} us  {B/wall:11.0f} evals/s  {B*nb/kt/1e9:7.1f} GB/s ({B*nb/kt/8e12:.2f})")

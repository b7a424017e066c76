"""Off-node residuals (mpx_resid_<ph>_<deg>) as a batched device pass: the h-adaptive loop's sample grid (mid-points of the
collocation nodes) at the BASELINE sizes -- `resid` alone (what the width update reads) and every field.  Time per pass by HIP events, algorithmic bytes
8 (n_z + n_p + outputs) per evaluation point, fraction of 8 TB/s.    python tools/r6_resid_bench.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mpopt_amd as M
from mpopt_amd import mp
import problems

dev = torch.device("cuda:0")
for name, (builder, S, po, scheme), B in (("config5", problems.BENCH_CASES[3], 512), ("config5", problems.BENCH_CASES[3], 4096), ("config2", problems.BENCH_CASES[0], 4096),
                                          ("config3", problems.BENCH_CASES[1], 512), ("deg100", (problems.moon_lander, 50, 100, "LGR"), 512)):
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    o.set_stream(torch.cuda.current_stream().cuda_stream)
    orders = [po] * S if isinstance(po, int) else po
    taus = mpo.get_residual_grid_taus(phase=0, grid_type="mid-points")
    plan = o.residual_plan(0, taus)
    n = plan.n_pts
    g_ = torch.Generator(device=dev).manual_seed(3)
    Z = torch.tensor(mpo.initialize_solution(), device=dev)[None, :] + 0.03 * torch.randn((B, o.n_z), generator=g_, device=dev, dtype=torch.float64)
    p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
    nx, nu = ocp.nx, ocp.nu
    outs_all = dict(ti=torch.empty((B, n), dtype=torch.float64, device=dev), xi=torch.empty((B, n, nx), dtype=torch.float64, device=dev),
                    ui=torch.empty((B, n, nu), dtype=torch.float64, device=dev), dxi=torch.empty((B, n, nx), dtype=torch.float64, device=dev),
                    dui=torch.empty((B, n, nu), dtype=torch.float64, device=dev), dyn=torch.empty((B, n, nx), dtype=torch.float64, device=dev),
                    resid=torch.empty((B, n, nx), dtype=torch.float64, device=dev))
    for label, outs in (("resid alone", {"resid": outs_all["resid"]}), ("all seven fields", outs_all)):
        run = lambda: plan.eval_device(B, Z, p, 0, **outs)
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        K = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / K
        nbytes = 8 * (o.n_z + o.n_p + sum(int(np.prod(v.shape[1:])) for v in outs.values()))
        print(json.dumps({"workload": f"off-node residuals, {name}: {builder.__name__} {S} segments {scheme}, {n} sample points (phase 0), {label}", "batch": B,
                          "us_per_pass": round(us, 1), "bytes_per_eval": nbytes, "algorithmic_GBps": round(nbytes * B / us / 1e3, 1), "frac_of_8TBps": round(nbytes * B / us / 1e3 / 8000, 3)}), flush=True)
    plan.close()
    o.close()

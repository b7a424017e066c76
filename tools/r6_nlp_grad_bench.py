"""nlp_grad (grad_gamma_x, grad_gamma_p of gamma = lam_f f + lam_g^T g; mpx_node_gradl_* + mpx_gradl_finish) as a batched device
pass at the BASELINE sizes: time per pass by HIP events on the context's stream, algorithmic bytes 8 (2 n_z + n_g + 2 n_p + 1) per evaluation point
(z, lam_g, the widths and lam_f in; grad_gamma_x and grad_gamma_p out), fraction of the 8 TB/s HBM peak.  One JSON line per configuration.
    python tools/r6_nlp_grad_bench.py [config ...]      (rocprofv3 --kernel-trace --stats around it: tools/r6_nlp_grad.sh)"""
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
CASES = {"config2": (problems.BENCH_CASES[0], 4096), "config3": (problems.BENCH_CASES[1], 512), "config4": (problems.BENCH_CASES[2], 4096),
         "config5": (problems.BENCH_CASES[3], 4096), "deg100": ((problems.moon_lander, 50, 100, "LGR"), 512)}
for name in sys.argv[1:] or list(CASES):
    (builder, S, po, scheme), B = CASES[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    o.set_stream(torch.cuda.current_stream().cuda_stream)
    g_ = torch.Generator(device=dev).manual_seed(11)
    Z = torch.tensor(mpo.initialize_solution(), device=dev)[None, :] + 0.03 * torch.randn((B, o.n_z), generator=g_, device=dev, dtype=torch.float64)
    p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
    lam = torch.randn((B, o.n_g), generator=g_, device=dev, dtype=torch.float64)
    sig = torch.rand(B, generator=g_, device=dev, dtype=torch.float64) + 0.5
    gx, gp = torch.empty((B, o.n_z), dtype=torch.float64, device=dev), torch.empty((B, o.n_p), dtype=torch.float64, device=dev)
    run = lambda: o.eval_grad_gamma_device(B, Z, p, lam, sig, gx, gp)
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
    # one point against the host-pointer call (the path the parity tests check against the oracle)
    h = o.eval_grad_gamma(Z[5].cpu().numpy(), p.cpu().numpy(), lam[5].cpu().numpy(), float(sig[5]))
    assert np.array_equal(h["grad_gamma_x"], gx[5].cpu().numpy()) and np.array_equal(h["grad_gamma_p"], gp[5].cpu().numpy())
    nbytes = 8 * (2 * o.n_z + o.n_g + 2 * o.n_p + 1)
    print(json.dumps({"workload": f"nlp_grad, {name}: {builder.__name__} {S} segments {scheme}", "batch": B, "us_per_pass": round(us, 1), "evals_per_s": round(B / us * 1e6),
                      "bytes_per_eval": nbytes, "algorithmic_GBps": round(nbytes * B / us / 1e3, 1), "frac_of_8TBps": round(nbytes * B / us / 1e3 / 8000, 3),
                      "n_z": o.n_z, "n_g": o.n_g, "n_p": o.n_p}), flush=True)
    o.close()

"""config-5 loop (hypersensitive 4000x3, 512 evaluation points, 5 outer iterations): one context on one stream against two contexts of
256 points each on two streams (the equal-area update of one half -- latency-bound, one workgroup per compute unit -- can run beside
the hess_l pass of the other half -- bandwidth-bound)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
import problems  # noqa: E402

builder, S, P, scheme = problems.BENCH_CASES[3]
dev = torch.device("cuda", 0)
B = 512
rng = np.random.default_rng(1)


def make(nb, stream):
    mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    o.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.01 * rng.standard_normal((nb, o.n_z)), device=dev)
        p0 = torch.tensor(rng.dirichlet(np.ones(S), nb), device=dev)
        st = dict(o=o, Z=Z, p0=p0, pa=torch.empty_like(p0), pb=torch.empty_like(p0), lam=torch.randn(nb, o.n_g, dtype=torch.float64, device=dev),
                  sig=torch.ones(nb, dtype=torch.float64, device=dev), hv=torch.empty(nb, o.nnz_hess, dtype=torch.float64, device=dev),
                  R=torch.empty(nb, 3 * S, 1, dtype=torch.float64, device=dev), nb=nb)
    o.set_mid_resid_output(st["R"])
    stream.synchronize()
    return st


def iteration(st, it5, cur, nxt):
    o = st["o"]
    o.eval_device(16 | 1024 | (256 if it5 else 0), st["nb"], st["Z"], cur, 1, st["lam"], st["sig"], None, None, None, None, st["hv"])
    o.equal_area_widths_device(0, st["nb"], 3 * S, st["R"], cur, nxt, damping=0.4, p_in_per_point=1)


def run(sts, steps):
    for _ in range(steps):
        cur = [s["pa"] for s in sts]
        nxt = [s["pb"] for s in sts]
        for s in sts:
            s["pa"].copy_(s["p0"], non_blocking=True) if False else None
        for it5 in range(5):
            for k, s in enumerate(sts):  # interleaved issue: half 0, half 1, half 0, ...
                iteration(s, it5, cur[k] if it5 else s["p0"], nxt[k])
            cur, nxt = nxt, cur


s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
one = [make(B, s0)]
two = [make(B // 2, s1), make(B // 2, s2)]
for name, sts in (("one context, 512 points", one), ("two contexts x 256 points, two streams", two), ("one context, 512 points", one),
                  ("two contexts x 256 points, two streams", two)):
    run(sts, 5)
    torch.cuda.synchronize()
    t = time.perf_counter()
    K = 40
    run(sts, K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / K
    print(f"{name:45s} {dt * 1e3:.4f} ms per 5 iterations  = {B * 5 / dt / 1e6:.3f} M point-iterations/s", flush=True)

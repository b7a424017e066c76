#!/usr/bin/env python3
"""Per-call table on the grids the reference PUBLISHES timings for: for every oracle
    published CasADi us (unknown CPU, docs/source/notebooks/*.ipynb) | CPU port us (oracle/mpopt_oracle.c, one core of THIS box) |
    GPU, host pointers, one function per call (every call its own device pass) | GPU, host pointers, in IPOPT's call order with the
    same-iterate cache | GPU, device pointers (wall per call of mpx_eval_device, B = 1)
and the oracle time per IPOPT iteration (1.15 (nlp_f + nlp_g) + nlp_grad_f + nlp_jac_g + nlp_hess_l, moon_lander.ipynb:192-198) from
each column.  Parity of what is timed: tests/test_gpu_published_grids.py.      python tools/r6_published_table.py >> profiles/r6_report.md"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import bench
import mpopt_amd as M
from mpopt_amd import mp
import problems

NAMES = ["nlp_f", "nlp_g", "nlp_grad_f", "nlp_jac_g", "nlp_hess_l"]
MASK = {"nlp_f": 1, "nlp_g": 2, "nlp_grad_f": 5, "nlp_jac_g": 10, "nlp_hess_l": 16}
mix = lambda d: 1.15 * (d["nlp_f"] + d["nlp_g"]) + d["nlp_grad_f"] + d["nlp_jac_g"] + d["nlp_hess_l"]


def device_calls(builder, S, P, scheme):
    dev = torch.device("cuda:0")
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    rng = np.random.default_rng(3)
    z0 = mpo.initialize_solution()
    Z = torch.tensor(z0[None, :] + 0.01 * rng.standard_normal((1, o.n_z)), device=dev)
    p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
    lam, sig = torch.randn(1, o.n_g, dtype=torch.float64, device=dev), torch.ones(1, dtype=torch.float64, device=dev)
    f, g, gr = (torch.empty(s, dtype=torch.float64, device=dev) for s in ((1,), (1, o.n_g), (1, o.n_z)))
    jv, hv = torch.empty((1, o.nnz_jac), dtype=torch.float64, device=dev), torch.empty((1, max(o.nnz_hess, 1)), dtype=torch.float64, device=dev)
    out = {}
    for n in NAMES:
        call = lambda: o.eval_device(MASK[n], 1, Z, p, 0, lam, sig, f, g, gr, jv, hv)
        for _ in range(20):
            call()
        o.sync()
        t = time.perf_counter()
        for _ in range(300):
            call()
        o.sync()
        out[n] = (time.perf_counter() - t) / 300 * 1e6
    sizes = (o.n_z, o.n_g, o.nnz_jac, o.nnz_hess)
    o.close()
    return out, sizes


def main():
    print("\n## The reference's published grids: per-call microseconds beside CasADi's own table (BASELINE.md 1a)\n")
    print(f"host: {os.cpu_count()} logical CPUs; CPU port = oracle/mpopt_oracle.c, gcc -O2, ONE core, one function per call "
          "(`orc_eval_fn`), same point.  Published = CasADi's `t_wall` per call in the notebook cell cited, hardware not stated.  GPU host = "
          "the `nlp_*` C entry points (ctypes, page-locked caller arrays, zero-copy): *alone* = every function pays its own device pass "
          "(MPX_NO_COALESCE=1), *in sequence* = IPOPT's call order at iterates that change, same-iterate cache on (the fused pass is "
          "charged to nlp_f).  GPU dev = `mpx_eval_device` on device pointers, wall per call, B = 1.  Parity of every row: "
          "`tests/test_gpu_published_grids.py`.\n")
    x = torch.empty(64 << 20, device="cuda:0")
    t = time.time()
    while time.time() - t < 2.0:
        x.add_(1.0)
    torch.cuda.synchronize()
    summary = []
    for name, (builder, S, P, scheme, cnames, st, midu, pub, cite) in problems.PUBLISHED_GRIDS.items():
        r = bench.ipopt_iter_report(builder, S, P, scheme, cnames, st, midu, 0, seconds=1.0)
        devc, (n_z, n_g, nnz_j, nnz_h) = device_calls(builder, S, P, scheme)
        pubd = dict(zip(NAMES, pub))
        print(f"### {name.replace('_', ' ')} ({cite}) — n_z = {n_z}, n_g = {n_g}, nnz jac_g = {nnz_j}, nnz hess_l = {nnz_h}\n")
        print("| oracle | CasADi published | CPU port (1 core) | GPU host, alone | GPU host, in sequence | GPU dev |")
        print("|---|---|---|---|---|---|")
        for n in NAMES:
            print(f"| {n} | {pubd[n]:.2f} | {r['cpu_port_per_call_us'][n]:.2f} | {r['per_call_us_uncoalesced'][n]:.1f} | {r['per_call_us'][n]:.1f} | {devc[n]:.1f} |")
        row = (mix(pubd), r["cpu_port_us_per_iter"], r["us_per_iter_uncoalesced"], r["us_per_iter"], mix(devc))
        print(f"| **per IPOPT iteration** | **{row[0]:.1f}** | **{row[1]:.1f}** | **{row[2]:.1f}** | **{row[3]:.1f}** | **{row[4]:.1f}** |\n")
        summary.append((name, row))
    print("### Oracle time per IPOPT iteration, all published grids (us)\n")
    print("| grid | CasADi published | CPU port (1 core) | GPU host, alone | GPU host, in sequence | GPU dev | GPU in sequence / CasADi |")
    print("|---|---|---|---|---|---|---|")
    for name, row in summary:
        print(f"| {name.replace('_', ' ')} | {row[0]:.1f} | {row[1]:.1f} | {row[2]:.1f} | {row[3]:.1f} | {row[4]:.1f} | {row[3] / row[0]:.2f} |")
    print("\nThese grids are 80-500 variables: the regime where a call is launch + completion latency (~7-16 us on device pointers, ~19-30 us "
          "through host pointers), not bandwidth.  In IPOPT's call order the GPU path needs 0.15-0.54 of the oracle time CasADi recorded per "
          "iteration (unknown CPU) and 1.5-6 times what ONE CPU core needs with the C port of the same arithmetic; the crossover to the GPU "
          "lies near 1 000 nodes (config 2: 92 us against 647 us per iteration for the port, `bench.py` -> `ipopt_iter`).  Function evaluations "
          "are 4-13 % of the reference's recorded solve times (BASELINE.md 1b).")


if __name__ == "__main__":
    main()

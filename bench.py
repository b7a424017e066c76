#!/usr/bin/env python3
"""bench.py -- headline benchmark of the collocation hot path on MI355X.

Metric (BASELINE.json): NLP ``grad_f + jac_g`` evaluations per second on the 1000-segment LGR
moon-lander grid (configs[1]: n_segments=1000, poly_orders=5).  One *evaluation* = what one call
of CasADi's ``nlp_grad_f`` plus one call of ``nlp_jac_g`` produce: f, grad_f, g and the jac_g
values.  One *step* = one fused launch sequence over a batch of B evaluation points that are
already resident in HBM (``mpx_eval_device``).  ``value`` = evaluations of all ranks / wall time.

Multi-GPU (``--gpus N`` under torch.distributed.run): evaluation points are independent, so each
rank processes its own batch of B points -- no data-path collective; weak scaling.

Also on the JSON line:
  roofline      algorithmic bytes of the dominant (node) kernel / its HIP-event duration vs 8 TB/s
  cpu_baseline  the C oracle (oracle/mpopt_oracle.c, a scalar port of the reference algorithm)
                timed on one host core on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured float4 copy


def make_points(oracle, mpo, bounds, B, seed):
    """Z0 + 0.05*|Z0|*xi + 0.01*xi', clipped to the variable bounds."""
    rng = np.random.default_rng(seed)
    z0 = mpo.initialize_solution()
    Z = z0[None, :] + 0.05 * np.abs(z0)[None, :] * rng.uniform(-1, 1, (B, oracle.n_z)) + 0.01 * rng.uniform(-1, 1, (B, oracle.n_z))
    return np.minimum(np.maximum(Z, bounds["lbx"][None, :]), bounds["ubx"][None, :])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="evaluation points per GPU per step")
    ap.add_argument("--segments", type=int, default=1000)
    ap.add_argument("--degree", type=int, default=5)
    ap.add_argument("--workload", default="config2-fgj", choices=["config2-fgj", "config5-hess", "config3-fgj", "config2-hess", "adaptive-fgj"],
                    help="default: the metric's configuration (BASELINE configs[1], f+g+grad_f+jac_g).  The others are "
                         "secondary reports (configs[4]: nlp_hess_l on hypersensitive 4000x3; configs[2]: mixed-degree grid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary MPX_JAC_VARIABLE_ONLY measurement (it launches the same kernel with less work, "
                         "which would mix into a rocprofv3 per-kernel average)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--ramp-seconds", type=float, default=2.0,
                    help="untimed clock ramp before the warm-up steps: a fresh box starts in a low-power state and "
                         "runs ~18%% slower for the first few hundred milliseconds")
    args = ap.parse_args()

    from mpopt_amd import distributed as mpd

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # nccl == RCCL on ROCm.  MPX_DIST_BACKEND=gloo lets several ranks share one GPU (smoke test of the
    # multi-rank code path on a 1-GPU box); the device is then local_rank modulo the visible GPUs.
    backend = os.environ.get("MPX_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl":
        assert int(os.environ.get("LOCAL_RANK", "0")) < n_dev, "one process per GPU: LOCAL_RANK exceeds the visible GPUs"
    rank, world, local_rank = mpd.init_from_env(backend)
    if world > 1:
        import torch.distributed as dist
    dev_id = local_rank % n_dev
    dev = torch.device("cuda", dev_id)
    torch.cuda.set_device(dev)

    import mpopt_amd as M
    from mpopt_amd import mp
    from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_JAC
    import problems

    S, P, B, K, W = args.segments, args.degree, args.batch, args.steps, args.warmup
    hess_mode = args.workload.endswith("hess")
    scheme, builder, label = "LGR", problems.moon_lander, f"moon-lander OCP, n_segments={S}, poly_orders={P}, LGR (BASELINE configs[1])"
    if args.workload == "config5-hess":
        builder, S, P, scheme = problems.BENCH_CASES[3]
        label = "hypersensitive OCP, n_segments=4000, poly_orders=3, LGR (BASELINE configs[4])"
    elif args.workload == "config3-fgj":
        builder, S, P, scheme = problems.BENCH_CASES[1]
        B = min(B, 512)
        label = "Van der Pol OCP, n_segments=2000, poly_orders=[3,30,3]*, CGL (BASELINE configs[2])"
    adaptive = args.workload == "adaptive-fgj"
    if adaptive:  # widths as decision variables, assembled context (point kernels + gather)
        S, P = 20, 5
        label = "moon-lander OCP, mpopt_adaptive (segment widths as variables), n_segments=20, poly_orders=5, LGR"
    ocp = builder(mp, M.math)
    mpo = (mp.mpopt_adaptive if adaptive else mp.mpopt)(ocp, S, P, scheme, device=dev_id)
    if rank == 0 or world == 1:
        nlp, bounds = mpo.create_nlp()  # rank 0 compiles (or finds the cached code object) first
    if world > 1:
        dist.barrier()
        if rank != 0:
            nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    o.set_stream(torch.cuda.current_stream().cuda_stream)

    Zh = make_points(o, mpo, bounds, B, 20260928 + rank)
    Z = torch.tensor(Zh, device=dev)
    p = torch.tensor(np.full(max(o.n_p, 1), 1.0 / S), device=dev)
    f = torch.empty(B, dtype=torch.float64, device=dev)
    g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
    gr = torch.empty(B, o.n_z, dtype=torch.float64, device=dev)
    jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
    mask = MPX_F | MPX_G | MPX_GRAD | MPX_JAC
    if hess_mode:
        from mpopt_amd._lib import MPX_HESS

        mask = MPX_HESS
        lam = torch.tensor(np.random.default_rng(20260928 + rank).standard_normal((B, o.n_g)), device=dev)
        sig = torch.ones(B, dtype=torch.float64, device=dev)
        hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
        jv = hv

    def step():
        if hess_mode:
            o.eval_device(mask, B, Z, p, 0, lam, sig, None, None, None, None, hv)
        else:
            o.eval_device(mask, B, Z, p, 0, None, None, f, g, gr, jv, None)

    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:  # untimed; see --ramp-seconds
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(W):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    o.profile(True)
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    node_ms, n_launch = o.profile_read()
    o.profile(False)
    elapsed = mpd.max_over_ranks(elapsed, device=dev if backend == "nccl" else None)

    # secondary, opt-in mode (NOT the metric): only the (z,p)-dependent Jacobian entries are rewritten into
    # the resident buffers, which hold the grid constants from the full evaluations above
    extra = {}
    if not hess_mode and not args.no_extras and not adaptive:
        from mpopt_amd._lib import MPX_JAC_VARIABLE_ONLY

        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(K):
            o.eval_device(mask | MPX_JAC_VARIABLE_ONLY, B, Z, p, 0, None, None, f, g, gr, jv, None)
        torch.cuda.synchronize()
        extra["jac_variable_only_evals_per_s"] = world * B * K / (time.perf_counter() - t1)

    # Also NOT the metric: the same kernel on freshly allocated output buffers.  Where the arrays land in physical
    # memory moves the node kernel by +-15 % within one process (not the box, not the TLB, pure-store
    # bandwidth per allocation is flat); `value` above is whatever the first allocation gave, this records the spread.
    if not hess_mode and not args.no_extras and world == 1 and not adaptive:
        sweep, hold = [], []
        for k in range(4):
            f2, g2 = torch.empty_like(f), torch.empty_like(g)
            gr2, jv2 = torch.empty_like(gr), torch.empty_like(jv)
            for _ in range(3):
                o.eval_device(mask, B, Z, p, 0, None, None, f2, g2, gr2, jv2, None)
            torch.cuda.synchronize()
            o.profile(True)
            for _ in range(10):
                o.eval_device(mask, B, Z, p, 0, None, None, f2, g2, gr2, jv2, None)
            ms, nl = o.profile_read()
            o.profile(False)
            sweep.append(ms / max(nl, 1) * 1e3)
            if k % 2 == 0:
                hold.append((f2, g2, gr2, jv2))  # keeping some alive moves the next allocation elsewhere
            del f2, g2, gr2, jv2
            torch.cuda.empty_cache()
        del hold
        extra["placement_sweep_node_kernel_us"] = [round(v, 1) for v in sweep]

    # sanity: the timed outputs are real (finite, and f matches a host recomputation of one point)
    assert torch.isfinite(jv[0]).all() and (hess_mode or torch.isfinite(g[-1]).all())

    if rank == 0:
        n_buckets = 1 if adaptive else len(set(int(d) for d in mpo.poly_orders))
        kernel_s = node_ms / 1e3 / max(n_launch // n_buckets, 1)  # all node-kernel launches of one step
        bytes_eval = o.bytes_hess if hess_mode else o.bytes_fgj
        achieved = B * bytes_eval / kernel_s / 1e9
        out = {
            "metric": "NLP grad_f+jac_g evals/sec, 1000-seg LGR" if args.workload == "config2-fgj" else f"NLP evals/sec ({args.workload})",
            "value": world * B * K / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{label}; {'nlp_hess_l' if hess_mode else 'f+g+grad_f+jac_g'}, "
                                   f"{B} evaluation points per GPU per step, inputs resident in HBM",
                       "n_z": o.n_z, "n_g": o.n_g, "nnz_jac": o.nnz_jac, "batch_per_gpu": B,
                       "parallelism": f"independent evaluation points x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "mpx_pts_jac + mpx_gather_kernel" if adaptive else f"mpx_node_{'hess' if hess_mode else 'fgj'}_0_*",
                         "kernel_us": kernel_s * 1e6,
                         "bytes_per_eval": bytes_eval, "evals_per_launch": B,
                         "algorithmic_bytes_per_launch": B * bytes_eval},
        }
        if extra:
            out["extras"] = dict(extra, note="opt-in MPX_JAC_VARIABLE_ONLY (resident jac buffers keep the constant D / interpolation "
                                             "entries); not the metric: the headline rewrites every entry on every evaluation")
        # HBM traffic of the dominant kernel from the committed PMC passes (same workload only)
        tf = os.path.join(ROOT, "profiles", "r1_xcd", "traffic.json")
        if os.path.exists(tf):
            tr = json.load(open(tf))
            if args.workload == "config2-fgj" and tr["workload"] == {"segments": S, "degree": P, "batch": B}:
                out["roofline"]["traffic"] = tr["bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/r1_xcd/traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)"
        if world == 1 and not args.no_cpu_baseline and args.workload == "config2-fgj":
            from oracle.c_oracle import COracle

            C = COracle(["moon_lander"], S, P, "LGR")
            ns = min(B, 64)
            ph = np.full(o.n_p, 1.0 / S)
            C.time_many(Zh[:ns], ph, 1)  # touch pages
            t1 = C.time_many(Zh[:ns], ph, 2) / 2
            reps = max(1, int(args.cpu_seconds / max(t1, 1e-6)))
            tt = C.time_many(Zh[:ns], ph, reps)
            r = C.eval(Zh[0], ph)  # the CPU port and the GPU agree on the benchmarked point
            assert abs(r["f"] - float(f[0].item())) < 1e-9 * max(1.0, abs(r["f"]))
            assert np.abs(r["g"] - g[0].cpu().numpy()).max() < 1e-9
            # the same port on every host core (OpenMP over evaluation points), bounded to a few seconds
            na = min(B, 2048)
            C.time_many_all_cores(Zh[:na], ph, 1)  # thread start-up, page touching
            ta, nthr = C.time_many_all_cores(Zh[:na], ph, 3)
            ra = max(1, int(5.0 / max(ta / 3, 1e-6)))
            ta, nthr = C.time_many_all_cores(Zh[:na], ph, ra)
            out["cpu_baseline_all_cores"] = {"value": na * ra / ta, "unit": "evals/s", "cores": nthr, "kind": "port",
                                             "sample": f"{na} evaluation points x {ra} passes, OpenMP over points, {ta:.1f} s"}
            out["cpu_baseline"] = {"value": ns * reps / tt, "unit": "evals/s", "cores": 1, "kind": "port",
                                   "sample": f"{ns} of the same evaluation points x {reps} passes, oracle/mpopt_oracle.c "
                                             f"(gcc -O2, scalar, values only), {tt:.1f} s",
                                   "host_cpus": os.cpu_count()}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the collocation hot path on MI355X.

Metric (BASELINE.json): NLP ``grad_f + jac_g`` evaluations per second on the 1000-segment LGR
moon-lander grid (configs[1]: n_segments=1000, poly_orders=5).  One *evaluation* = what one call
of CasADi's ``nlp_grad_f`` plus one call of ``nlp_jac_g`` produce: f, grad_f, g and the jac_g
values.  One *step* = one fused launch sequence over a batch of B evaluation points that are
already resident in HBM (``mpx_eval_device``).  ``value`` = evaluations of all ranks / wall time.

Multi-GPU (``--gpus N``): evaluation points are independent, so each rank processes its own batch of
B points -- no data-path collective; weak scaling.  One process per GPU: under
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` the ranks are the launcher's;
a plain ``python bench.py --gpus N`` re-launches ITSELF that way (N ranks on 127.0.0.1, rank 0 prints the
one JSON line) after checking that N GPUs are visible.  The ``*-shard``
workloads instead split the SEGMENTS of every evaluation over the ranks (contiguous tile
ranges per rank, one RCCL all-gather of the owned residual / Jacobian runs per evaluation, boundary pass
on every rank; ``mpopt_amd.distributed.SegmentShardedEvaluator``): total work is fixed, strong scaling.

Also on the JSON line:
  roofline      algorithmic bytes of the dominant (node) kernel / its HIP-event duration vs 8 TB/s
  cpu_baseline  the C oracle (oracle/mpopt_oracle.c, a scalar port of the reference algorithm)
                timed on one host core on a bounded sample of the same workload (rank 0, N=1 only)
  ipopt_iter    the second half of BASELINE.json's metric: oracle wall-clock per IPOPT iteration at B=1 through
                HOST pointers (the CasADi-convention entry points nlp_f / nlp_g / nlp_grad_f / nlp_jac_g / nlp_hess_l of
                libmpx.so, values in compressed-column order, caller arrays page-locked on first sight), called in a
                solver's order at iterates that really change, with the reference's recorded call mix
                (docs/source/notebooks/moon_lander.ipynb:192-198), next to the CPU port for the same mix
  casadi        "absent", or the timings of CasADi's own nlp_* functions if `import casadi` succeeds on the box
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_MEASURED_GBS = 6290.0  # what a float4 copy sustains on this part (MI355X_MICROARCH.md: "~6.3 TB/s achievable")


def make_points(oracle, mpo, bounds, B, seed):
    """Z0 + 0.05*|Z0|*xi + 0.01*xi', clipped to the variable bounds."""
    rng = np.random.default_rng(seed)
    z0 = mpo.initialize_solution()
    Z = z0[None, :] + 0.05 * np.abs(z0)[None, :] * rng.uniform(-1, 1, (B, oracle.n_z)) + 0.01 * rng.uniform(-1, 1, (B, oracle.n_z))
    return np.minimum(np.maximum(Z, bounds["lbx"][None, :]), bounds["ubx"][None, :])


def ipopt_iter_report(builder, S, P, scheme, cnames, scale_t, midu, dev_id, seconds=0.6):
    """Oracle time per IPOPT iteration at B=1, host pointers, through the nlp_* C entry points (ctypes: ~1 us per call of overhead
    included), in the ORDER an interior-point solver makes the calls: at every new iterate x  nlp_f, nlp_g  (trial point) and then
    nlp_grad_f, nlp_jac_g (res[0] = NULL, as CasADi's IPOPT interface asks), nlp_hess_l  -- x really changes between iterates (two
    points alternate in the caller's array), so the same-iterate cache of mpx_casadi.cpp sees what it would see under IPOPT.  One
    iteration = 1.15 x (nlp_f + nlp_g) + nlp_grad_f + nlp_jac_g + nlp_hess_l (15 % rejected trial points, moon_lander.ipynb:192-198;
    round 2 left nlp_f out of the sum).  Measured with the cache on and off (MPX_NO_COALESCE), next to the CPU port."""
    import ctypes

    import mpopt_amd as M
    from mpopt_amd import mp, _lib
    from oracle.c_oracle import COracle

    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, P, scheme, device=dev_id)
    nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    L = _lib.lib()
    rng = np.random.default_rng(20260928)
    Zs = make_points(o, mpo, bounds, 2, 20260928)
    z = Zs[0].copy()  # the caller's x vector: same address on every call, new content per iterate
    p = np.full(o.n_p, 1.0 / S)
    lam, sig = rng.standard_normal(o.n_g), np.array([1.0])
    f, g, gr = np.zeros(1), np.zeros(o.n_g), np.zeros(o.n_z)
    jv, hv = np.zeros(max(o.nnz_jac, 1)), np.zeros(max(o.nnz_hess, 1))
    vp = lambda arrs: (ctypes.c_void_p * len(arrs))(*[a.ctypes.data if a is not None else None for a in arrs])
    calls = [("nlp_f", vp([z, p]), vp([f])), ("nlp_g", vp([z, p]), vp([g])), ("nlp_grad_f", vp([z, p]), vp([f, gr])),
             ("nlp_jac_g", vp([z, p]), vp([None, jv])), ("nlp_hess_l", vp([z, p, sig, lam]), vp([hv]))]
    funs = {n: getattr(L, n) for n, _, _ in calls}

    def sequence(seconds):
        for k in range(20):
            z[:] = Zs[k & 1]
            for n, a, r in calls:
                assert funs[n](a, r, None, None, 0) == 0
        acc, it, t_end = {n: [] for n, _, _ in calls}, 0, time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            it += 1
            z[:] = Zs[it & 1]  # the solver's own write of the new iterate (not timed)
            for n, a, r in calls:
                t0 = time.perf_counter()
                funs[n](a, r, None, None, 0)
                acc[n].append(time.perf_counter() - t0)
        return {n: float(np.median(v)) * 1e6 for n, v in acc.items()}, it

    mix = lambda d: 1.15 * (d["nlp_f"] + d["nlp_g"]) + d["nlp_grad_f"] + d["nlp_jac_g"] + d["nlp_hess_l"]
    os.environ["MPX_NO_COALESCE"] = "1"
    o.make_current()
    L.mpx_current_pin_buffers(1)
    plain, _ = sequence(seconds / 2)
    L.mpx_current_pin_buffers(0)
    del os.environ["MPX_NO_COALESCE"]
    o.make_current()
    L.mpx_current_pin_buffers(1)
    gpu, n_it = sequence(seconds)
    # ... and with the constants of a large Jacobian left in the caller's array (mpx_current_keep_jac_constants(1), round 6: every
    # nlp_jac_g after the first rewrites only the (z, p)-dependent quarter of the values).  A NEGATIVE result, reported beside the
    # default: in compressed-column order those entries are isolated 8-byte words, and 30 000 of them cost more PCIe packets than
    # 0.96 MB of full cache lines (config 2: nlp_jac_g 36.8 -> 43.6 us; profiles/r6_jac_constants/README.md).
    L.mpx_current_keep_jac_constants(1)
    gpu_keep, n_it2 = sequence(seconds / 2)
    n_it += n_it2 + 20
    jvar, jfull = ctypes.c_longlong(), ctypes.c_longlong()
    L.mpx_current_jac_stats(ctypes.byref(jvar), ctypes.byref(jfull))
    fused, served = ctypes.c_longlong(), ctypes.c_longlong()
    L.mpx_current_cache_stats(ctypes.byref(fused), ctypes.byref(served))
    L.mpx_current_keep_jac_constants(0)
    L.mpx_current_pin_buffers(0)
    # parity of what was timed: the CCS-ordered values against the CPU port on the same point
    C = COracle(cnames, S, P, scheme, scale_t=scale_t, midu=midu)
    c = C.eval(z, p)
    assert np.abs(c["g"] - g).max() < 1e-9 * max(1.0, np.abs(c["g"]).max()) and abs(c["f"] - f[0]) < 1e-9 * max(1.0, abs(c["f"]))
    assert np.abs(c["grad_f"] - gr).max() < 1e-9 * max(1.0, np.abs(c["grad_f"]).max())
    import scipy.sparse as sp

    perm, colind = o.ccs_perm("jac")
    jr, jc = o.jac_pattern()
    Jg = sp.coo_matrix((jv[:o.nnz_jac], (jr[perm], jc[perm])), shape=(o.n_g, o.n_z)).tocsr()
    Jc = sp.coo_matrix((c["jac_val"], (c["jac_row"], c["jac_col"])), shape=(o.n_g, o.n_z)).tocsr()
    assert abs(Jg - Jc).max() < 1e-9 * max(1.0, abs(Jc).max())
    cpu = {k: C.time_fn(k, z[None, :], p, 1.0, lam, seconds) * 1e6 for k in gpu}
    o.close()
    return {"us_per_iter": mix(gpu), "us_per_iter_keep_jac_constants": mix(gpu_keep), "us_per_iter_uncoalesced": mix(plain), "cpu_port_us_per_iter": mix(cpu),
            "speedup_vs_cpu_port": mix(cpu) / mix(gpu),
            "per_call_us": {k: round(v, 2) for k, v in gpu.items()}, "per_call_us_keep_jac_constants": {k: round(v, 2) for k, v in gpu_keep.items()},
            "per_call_us_uncoalesced": {k: round(v, 2) for k, v in plain.items()},
            "jac_constants": {"variable_only_passes": jvar.value, "full_passes": jfull.value,
                              "note": "opt-in mpx_current_keep_jac_constants(1): not faster (isolated 8-byte words over PCIe), never part of us_per_iter; small Jacobians (<= 64 KB) are served by the same-iterate cache instead"},
            "cpu_port_per_call_us": {k: round(v, 2) for k, v in cpu.items()},
            "cache": {"iterates": n_it + 20, "fused_device_passes": fused.value, "calls_served_without_a_device_pass": served.value},
            "n_z": o.n_z, "n_g": o.n_g, "nnz_jac": o.nnz_jac, "nnz_hess": o.nnz_hess}


def casadi_probe(S, P, Zs, p, g_gpu):
    """"If `import casadi` succeeds ... time the real nlp_* functions obtained from the solver and report them as the
    primary CPU reference."  CasADi is absent from the build image, so this leg has never executed; it is written against
    CasADi's documented Python API and guarded: whatever goes wrong is reported as a string, never raised.  With CasADi present it
    restates the moon-lander transcription (mpopt.py:154-237, 330-377, 415-462: defects, mid-point control rows, terminal rows,
    J = compW . q) over SX with the product's collocation tables, lets `ca.nlpsol("solver", "ipopt", nlp)` derive nlp_f / nlp_g /
    nlp_grad_f / nlp_jac_g / nlp_hess_l exactly as the reference does at mpopt.py:757, checks g against the GPU on the first
    benchmarked point and times each function on one core (>= 30 repeats, median)."""
    try:
        import casadi as ca
    except Exception:
        return "absent"
    try:
        from mpopt_amd import mp

        col = mp.Collocation([P] * S, "LGR")
        N = S * P + 1
        D = np.asarray(col.get_composite_differentiation_matrix([P] * S), float)
        W = np.asarray(col.get_composite_quadrature_weights([P] * S), float).ravel()
        r = np.asarray(col._taus_fn(P), float)
        Im = np.asarray(col.get_composite_interpolation_matrix([(r[:-1] + r[1:]) / 2.0] * S, [P] * S), float)
        X, U = ca.SX.sym("X", N, 2), ca.SX.sym("U", N, 1)
        t0, tf, w = ca.SX.sym("t0"), ca.SX.sym("tf"), ca.SX.sym("w", S)
        seg = np.concatenate([[0], np.repeat(np.arange(S), P)])
        h = ca.vertcat(*[(tf - t0) / 2.0 * w[int(s)] for s in seg])
        F = ca.horzcat(h * X[:, 1], h * (U[:, 0] - 1.5))
        G = ca.vertcat(ca.vec(ca.mtimes(ca.DM(D), X) - F), ca.mtimes(ca.DM(Im), U), X[N - 1, 0], X[N - 1, 1])
        J = ca.mtimes(ca.DM(W).T, h * U[:, 0])
        Z = ca.vertcat(ca.vec(X), ca.vec(U), t0, tf)
        solver = ca.nlpsol("solver", "ipopt", {"f": J, "x": Z, "g": G, "p": w}, {"ipopt.print_level": 0, "print_time": 0})
        out, z0 = {}, np.asarray(Zs[0], float)
        gg = np.asarray(solver.get_function("nlp_g")(z0, p)).ravel()
        out["max_abs_g_vs_gpu"] = float(np.abs(gg - g_gpu).max())
        lam = np.ones(gg.size)
        for name, args in (("nlp_f", (z0, p)), ("nlp_g", (z0, p)), ("nlp_grad_f", (z0, p)), ("nlp_jac_g", (z0, p)), ("nlp_hess_l", (z0, p, 1.0, lam))):
            fn = solver.get_function(name)
            for _ in range(5):
                fn(*args)
            ts = []
            for _ in range(30):
                t = time.perf_counter()
                fn(*args)
                ts.append(time.perf_counter() - t)
            ts.sort()
            out[name + "_us"] = {"median": ts[15] * 1e6, "p10": ts[3] * 1e6, "p90": ts[27] * 1e6}
        out["grad_f_plus_jac_g_evals_per_s"] = 1.0 / ((out["nlp_grad_f_us"]["median"] + out["nlp_jac_g_us"]["median"]) * 1e-6)
        out["version"] = ca.__version__
        return out
    except Exception as e:  # pragma: no cover
        return "present, probe failed: " + repr(e)[:300]


def segment_shard_report(dev, dev_id, rank, world, backend, B=32, K=20):
    """Secondary measurement attached to the default line when N > 1 (so that the driver's --gpus 2/4/8 runs exercise RCCL on
    the path): configs[2] (Van der Pol 2000 x [3,30,3], CGL), the segments of every evaluation sharded over the
    ranks, finished in each of the three ways of mpopt_amd.distributed.SegmentShardedEvaluator -- "allgather" (ONE
    all_gather_into_tensor of the owned runs, complete result on every rank), "root" (dist.gather of the same runs, complete
    result on rank 0) and "owner" (every rank keeps its rows / value blocks, only the tile partials are all-gathered) -- against
    the same evaluation done by one rank alone, with a bitwise comparison of what each mode promises to hold."""
    import torch.distributed as dist

    import mpopt_amd as M
    from mpopt_amd import mp, distributed as mpd
    from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_JAC
    import problems

    builder, S, P, scheme = problems.BENCH_CASES[1]
    mpo = mp.mpopt(builder(mp, M.math), S, P, scheme, device=dev_id)
    if rank == 0:
        nlp, bounds = mpo.create_nlp()
    dist.barrier()
    if rank != 0:
        nlp, bounds = mpo.create_nlp()
    o = nlp["oracle"]
    o.set_stream(torch.cuda.current_stream().cuda_stream)
    Z = torch.tensor(make_points(o, mpo, bounds, B, 20260928), device=dev)
    p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
    mk = lambda *s: torch.empty(s, dtype=torch.float64, device=dev)
    mask = MPX_F | MPX_G | MPX_GRAD | MPX_JAC
    ref = (mk(B), mk(B, o.n_g), mk(B, o.n_z), mk(B, o.nnz_jac))
    cdev = dev if backend == "nccl" else None

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        return mpd.max_over_ranks(time.perf_counter() - t0, device=cdev) / K

    t_one = timed(lambda: o.eval_device(mask, B, Z, p, 0, None, None, *ref, None))
    modes = {}
    for mode in mpd.SegmentShardedEvaluator.MODES:
        out = tuple(torch.full_like(t, float("nan")) for t in ref)
        ev = mpd.SegmentShardedEvaluator(o, rank, world, mode=mode)
        t_shard = timed(lambda: ev.eval(mask, B, Z, p, None, None, *out, None))
        if mode == "allgather":
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
        elif mode == "root":
            same = rank != 0 or all(torch.equal(a, b) for a, b in zip(out, ref))
        else:  # owner-resident: f, this rank's runs and the replicated boundary entries
            same = torch.equal(out[0], ref[0])
            for name, a, b in (("g", out[1], ref[1]), ("grad_f", out[2], ref[2]), ("jac_g", out[3], ref[3])):
                owner = np.full(a.shape[1], -1)
                for r in range(world):
                    for off, ln in ev.owned(name, r):
                        owner[off:off + ln] = r
                here = torch.tensor((owner == rank) | (owner == -1), device=dev)
                same = same and torch.equal(a[:, here], b[:, here])
        flag = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        sent, recvd = ev.exchange_doubles(mask, B)
        _, cuts = o.shard_info(mask)
        modes[mode] = {"ms_per_step": t_shard * 1e3, "evals_per_s": B / t_shard, "speedup_vs_one_rank_alone": t_one / t_shard,
                       "exchange_bytes_sent_per_rank_per_step": int(sent) * 8, "exchange_bytes_received_rank0_per_step": int(recvd) * 8 if rank == 0 else None,
                       "bit_identical_to_unsharded": bool(flag.item() == 1.0)}
        ev.close()
    o.close()
    return {"workload": "Van der Pol 2000 x [3,30,3] CGL (configs[2]), f+g+grad_f+jac_g, segments of every evaluation sharded over the ranks",
            "n_gpus": world, "batch": B, "steps": K, "backend": backend, "ms_per_step_one_rank_alone": t_one * 1e3, "tiles_per_rank": np.diff(cuts).tolist(),
            "modes": modes,
            "holds": {"allgather": "complete result on every rank", "root": "complete result on rank 0", "owner": "f + own runs + boundary entries on every rank"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: a timed region of >= 0.2 s at the default workload)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="evaluation points per GPU per step (default 4096; 512 for config 3 first-order and the config-5 loop, 16 for the shard workloads)")
    ap.add_argument("--segments", type=int, default=1000)
    ap.add_argument("--degree", type=int, default=5)
    ap.add_argument("--adaptive-grid", default="20x5", help="adaptive-* workloads: SEGMENTSxDEGREE of the mpopt_adaptive transcription (default 20x5)")
    ap.add_argument("--plain-outputs", action="store_true", help="time plain torch.empty output arrays instead of NlpFunctions.alloc_outputs")
    ap.add_argument("--workload", default="config2-fgj", choices=["config2-fgj", "config5-hess", "config3-fgj", "config3-hess", "config2-hess", "adaptive-fgj", "config3-shard", "config4-shard", "config5-loop",
                                                                      "config4-fgj", "config4-hess", "config5-fgj", "adaptive-hess"],
                    help="default: the metric's configuration (BASELINE configs[1], f+g+grad_f+jac_g).  The others are "
                         "secondary reports (configs[4]: nlp_hess_l on hypersensitive 4000x3; configs[2]: mixed-degree grid)")
    ap.add_argument("--oracles", default="f,g,grad_f,jac_g",
                    help="first-order workloads: which outputs the timed pass writes (a line search calls nlp_f / nlp_g alone); the default is "
                         "the metric's fused bundle.  Algorithmic bytes follow the selection: 8 (n_z + n_p + [1] + [n_g] + [n_z] + [nnz_jac])")
    ap.add_argument("--alloc-tries", type=int, default=16, help="candidate output allocations NlpFunctions.alloc_outputs may draw (all held until the choice is made)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary MPX_JAC_VARIABLE_ONLY measurement (it launches the same kernel with less work, "
                         "which would mix into a rocprofv3 per-kernel average)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--ramp-seconds", type=float, default=2.0,
                    help="untimed clock ramp before the warm-up steps: a fresh box starts in a low-power state and "
                         "runs ~18%% slower for the first few hundred milliseconds")
    ap.add_argument("--launch-check", action="store_true",
                    help="stop after the process group is up: every rank joins, rank 0 prints {\"launch_check\": ...}; needs no GPU with "
                         "MPX_DIST_BACKEND=gloo (what the CPU test of the N-rank launch runs)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    backend = os.environ.get("MPX_DIST_BACKEND", "nccl")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by hand (or by a driver that does not wrap it in torchrun): become the launcher -- one process
        # per GPU under torch.distributed.run on 127.0.0.1 (the container's host name may not resolve), the ranks run this file again
        # with RANK / LOCAL_RANK / WORLD_SIZE set and rank 0 prints the line.
        import socket
        import subprocess

        if backend == "nccl":  # RCCL: one GPU per rank (MPX_DIST_BACKEND=gloo lets ranks share a GPU: 1-GPU smoke tests)
            n_vis = torch.cuda.device_count()
            if n_vis < args.gpus:
                sys.exit(f"bench.py --gpus {args.gpus}: only {n_vis} GPU(s) visible (one process per GPU over RCCL)")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    from mpopt_amd import distributed as mpd

    if args.launch_check:
        rank, world, local_rank = mpd.init_from_env(backend)
        assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} rank(s)"
        seen = 1.0
        if world > 1:
            import torch.distributed as dist

            t = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1)) if backend == "nccl" else None)
            dist.all_reduce(t)
            seen = float(t.item())
            dist.barrier()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "backend": backend if world > 1 else None, "all_reduce_of_ones": seen}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # nccl == RCCL on ROCm.  MPX_DIST_BACKEND=gloo lets several ranks share one GPU (smoke test of the
    # multi-rank code path on a 1-GPU box); the device is then local_rank modulo the visible GPUs.
    n_dev = torch.cuda.device_count()
    if backend == "nccl":
        assert int(os.environ.get("LOCAL_RANK", "0")) < n_dev, "one process per GPU: LOCAL_RANK exceeds the visible GPUs"
    rank, world, local_rank = mpd.init_from_env(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} rank(s): n_gpus on the line is the number of ranks that ran"
    if world > 1:
        import torch.distributed as dist
    dev_id = local_rank % n_dev
    dev = torch.device("cuda", dev_id)
    torch.cuda.set_device(dev)
    # who took part: every rank's host / HIP device / PCI bus id through all_gather_object (the proof that an N > 1 line ran on N
    # distinct GPUs over RCCL), plus one small all_reduce on the device as the first contact of the collective library
    census = None
    if world > 1:
        census = mpd.device_census(dev_id)
        probe = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else None)
        dist.all_reduce(probe)
        census["all_reduce_of_ones"] = float(probe.item())
        assert census["all_reduce_of_ones"] == world
        if backend == "nccl":  # an N-GPU line must come from N distinct GPUs
            assert census["distinct_gpus"] == world, f"{world} ranks on {census['distinct_gpus']} distinct GPU(s): {census['ranks']}"

    import mpopt_amd as M
    from mpopt_amd import mp
    from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_JAC
    import problems

    batch_given = args.batch is not None
    S, P, B, K, W = args.segments, args.degree, args.batch if batch_given else 4096, args.steps, args.warmup
    hess_mode = args.workload.endswith("hess")
    shard = args.workload.endswith("-shard")
    scheme, builder, label = "LGR", problems.moon_lander, f"moon-lander OCP, n_segments={S}, poly_orders={P}, LGR" + (" (BASELINE configs[1])" if (S, P) == (1000, 5) else " (NOT a BASELINE configuration: --segments / --degree)")
    loop5 = args.workload == "config5-loop"
    if args.workload in ("config5-hess", "config5-loop"):
        builder, S, P, scheme = problems.BENCH_CASES[3]
        label = "hypersensitive OCP, n_segments=4000, poly_orders=3, LGR (BASELINE configs[4])"
    elif args.workload in ("config3-fgj", "config3-shard", "config3-hess"):
        builder, S, P, scheme = problems.BENCH_CASES[1]
        B = B if batch_given else min(B, 2048 if hess_mode else 512)
        label = "Van der Pol OCP, n_segments=2000, poly_orders=[3,30,3]*, CGL (BASELINE configs[2])"
    elif args.workload == "config5-fgj":
        builder, S, P, scheme = problems.BENCH_CASES[3]
        label = "hypersensitive OCP, n_segments=4000, poly_orders=3, LGR (BASELINE configs[4])"
    elif args.workload in ("config4-shard", "config4-fgj", "config4-hess"):
        builder, S, P, scheme = problems.BENCH_CASES[2]
        label = "two-phase Schwartz OCP, 500 segments per phase, poly_orders=3, LGL (BASELINE configs[3])"
    if shard:  # every rank evaluates the SAME points, each its share of the segments
        B = B if batch_given else min(B, 16)
    if loop5:  # SURVEY 8(d) config-5 protocol: widths ~ Dirichlet(1), 5 outer iterations, device resident, one context
        B = B if batch_given else min(B, 512)
        hess_mode = True
    adaptive = args.workload.startswith("adaptive")
    if adaptive:  # widths as decision variables, assembled context (point kernels + gather)
        S, P = (int(v) for v in args.adaptive_grid.lower().split("x"))
        label = f"moon-lander OCP, mpopt_adaptive (segment widths as variables), n_segments={S}, poly_orders={P}, LGR"
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

    Zh = make_points(o, mpo, bounds, B, 20260928 + (0 if shard else rank))
    ev = None
    if shard:
        ev = mpd.SegmentShardedEvaluator(o, rank, world)
    Z = torch.tensor(Zh, device=dev)
    p = torch.tensor(np.full(max(o.n_p, 1), 1.0 / S), device=dev)
    f = torch.empty(B, dtype=torch.float64, device=dev)
    g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
    gr = torch.empty(B, o.n_z, dtype=torch.float64, device=dev)
    jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
    mask = MPX_F | MPX_G | MPX_GRAD | MPX_JAC
    sel = [w for w in args.oracles.split(",") if w]
    partial_sel = sorted(sel) != ["f", "g", "grad_f", "jac_g"]
    if partial_sel:
        assert not hess_mode and not shard and not adaptive and set(sel) <= {"f", "g", "grad_f", "jac_g"}, "--oracles applies to the first-order workloads"
        mask = sum({"f": MPX_F, "g": MPX_G, "grad_f": MPX_GRAD, "jac_g": MPX_JAC}[w] for w in set(sel))
        args.no_extras = True
    if hess_mode:
        from mpopt_amd._lib import MPX_HESS

        mask = MPX_HESS
        lam = torch.tensor(np.random.default_rng(20260928 + rank).standard_normal((B, o.n_g)), device=dev)
        sig = torch.ones(B, dtype=torch.float64, device=dev)
        hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
        jv = hv
    # Output arrays placed by measurement (NlpFunctions.alloc_outputs: the fastest of up to six candidate allocations for THIS kernel and
    # THESE inputs, a one-time set-up step a caller can take as well).  --plain-outputs keeps the plain
    # torch.empty arrays above; `value_placement_median` / `frac_placement_*` below always describe plain allocations.
    placed = None
    if not args.plain_outputs and not adaptive and not shard and not loop5:
        # The metric's kernel has a known well-placed time: the algorithmic bytes of the pass at 0.95 of the MEASURED copy rate (6.29
        # TB/s, MI355X_MICROARCH.md) -- round 4's best placements ran at 0.98 of it, the slow ones at 0.89.  The search draws candidate
        # sets until one is within 3 % of that or 16 are held; the other workloads use the relative rule (8 % faster than the slowest).
        target = None
        if args.workload == "config2-fgj" and mask == (MPX_F | MPX_G | MPX_GRAD | MPX_JAC):
            target = B * o.bytes_fgj / (0.95 * HBM_MEASURED_GBS * 1e9) * 1e6
        if hess_mode:
            del hv, jv
            outs, placed = o.alloc_outputs(mask, B, Z, p, 0, lam, sig, tries=args.alloc_tries)
            hv = jv = outs[4]
        else:
            del f, g, gr, jv
            outs, placed = o.alloc_outputs(mask, B, Z, p, 0, None, None, tries=args.alloc_tries, target_us=target)
            f, g, gr, jv = outs[:4]
        o.geometry_reset()

    if loop5:
        mids = [(mpo.collocation._taus_fn(d)[:-1] + mpo.collocation._taus_fn(d)[1:]) / 2 for d in mpo.poly_orders]
        plan = o.residual_plan(0, mids)
        p0 = torch.tensor(np.random.default_rng(20260928 + rank).dirichlet(np.ones(S), B), device=dev)
        pa, pb = torch.empty_like(p0), torch.empty_like(p0)
        R = torch.empty(B, plan.n_pts, ocp.nx, dtype=torch.float64, device=dev)
        o.set_mid_resid_output(R)
        unfused_loop = bool(os.environ.get("MPX_BENCH_UNFUSED_LOOP"))

    def step():
        if loop5:
            pa.copy_(p0)
            cur, nxt = pa, pb
            for it5 in range(5):
                if unfused_loop:  # round-2 form: residual plan, then hess_l with MPX_WIDTHS_UNCHANGED (same p as the residual call)
                    plan.eval_device(B, Z, cur, p_per_point=1, resid=R)
                    o.eval_device(mask | 256, B, Z, cur, 1, lam, sig, None, None, None, None, hv)
                else:  # MPX_MID_RESID: the hess_l pass writes the mid-point residuals itself (one pass over z)
                    # (from the second iteration on the widths are the previous equal-area update's output, which left their prefix
                    # sums on the device: MPX_WIDTHS_UNCHANGED = 256, no prefix launch)
                    o.eval_device(mask | 1024 | (256 if it5 else 0), B, Z, cur, 1, lam, sig, None, None, None, None, hv)
                o.equal_area_widths_device(0, B, plan.n_pts, R, cur, nxt, damping=0.4, p_in_per_point=1)
                cur, nxt = nxt, cur
        elif shard:
            ev.eval(mask, B, Z, p, None, None, f, g, gr, jv, None)
        elif hess_mode:
            o.eval_device(mask, B, Z, p, 0, lam, sig, None, None, None, None, hv)
        else:
            o.eval_device(mask, B, Z, p, 0, None, None, f if mask & MPX_F else None, g if mask & MPX_G else None, gr if mask & MPX_GRAD else None,
                          jv if mask & MPX_JAC else None, None)

    t_ramp = time.perf_counter()
    if shard:  # collectives inside the step: every rank must issue the same number of them (no time-based loop)
        for _ in range(40):
            step()
        torch.cuda.synchronize()
    while not shard and time.perf_counter() - t_ramp < args.ramp_seconds:  # untimed; see --ramp-seconds
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
    if not hess_mode and not args.no_extras and not adaptive and not shard:
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
    if not hess_mode and not args.no_extras and world == 1 and not adaptive and not shard:
        sweep, hold = [], []
        for k in range(4):
            f2, g2 = torch.empty_like(f), torch.empty_like(g)
            gr2, jv2 = torch.empty_like(gr), torch.empty_like(jv)
            o.geometry_reset()  # new physical pages (possibly behind old addresses): let the library measure them again
            for _ in range(8):
                o.eval_device(mask, B, Z, p, 0, None, None, f2, g2, gr2, jv2, None)
            torch.cuda.synchronize()
            o.profile(True)
            for _ in range(10):
                o.eval_device(mask, B, Z, p, 0, None, None, f2, g2, gr2, jv2, None)
            ms, nl = o.profile_read()
            o.profile(False)
            sweep.append(ms / 10 * 1e3)  # all node-kernel launches of one step
            if k % 2 == 0:
                hold.append((f2, g2, gr2, jv2))  # keeping some alive moves the next allocation elsewhere
            del f2, g2, gr2, jv2
            torch.cuda.empty_cache()
        del hold
        extra["placement_sweep_node_kernel_us"] = [round(v, 1) for v in sweep]

    # sanity: the timed outputs are real (finite, and f matches a host recomputation of one point)
    if not partial_sel:
        assert torch.isfinite(jv[0]).all() and (hess_mode or torch.isfinite(g[-1]).all())

    if rank == 0:
        kernel_s = node_ms / 1e3 / K  # all node-kernel launches of one step (one event bracket per pass, mpx_profile)
        bytes_eval = o.bytes_hess if hess_mode else o.bytes_fgj
        if partial_sel:
            bytes_eval = 8 * (o.n_z + o.n_p + (1 if mask & MPX_F else 0) + (o.n_g if mask & MPX_G else 0) + (o.n_z if mask & MPX_GRAD else 0) +
                              (o.nnz_jac if mask & MPX_JAC else 0))
        if loop5:  # one step = 5 outer iterations of (residuals, hess_l, width update); the roofline object covers the whole loop
            n_pts = plan.n_pts
            bytes_eval = 5 * (o.bytes_hess + 8 * (o.n_z + 2 * o.n_p + n_pts * ocp.nx) + 8 * (n_pts * ocp.nx + 2 * o.n_p))
            kernel_s = elapsed / K
        achieved = B * bytes_eval / kernel_s / 1e9
        sweep_us = extra.get("placement_sweep_node_kernel_us")
        out = {
            "metric": "NLP grad_f+jac_g evals/sec, 1000-seg LGR" if args.workload == "config2-fgj" and not partial_sel else f"NLP evals/sec ({args.workload}{', outputs ' + '+'.join(sel) if partial_sel else ''})",
            "value": (1 if shard else world) * B * K * (5 if loop5 else 1) / elapsed,
            "unit": "point-iterations/s (one = residuals at the mid-points + nlp_hess_l + equal-area width update)" if loop5 else "evals/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if shard else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{label}; {'h-adaptive loop, widths ~ Dirichlet(1), 5 outer iterations per step (SURVEY 8(d)), ' if loop5 else ''}"
                                   f"{'nlp_hess_l' if hess_mode else '+'.join(sel)}, "
                                   f"{B} evaluation points per GPU per step, inputs resident in HBM",
                       "n_z": o.n_z, "n_g": o.n_g, "nnz_jac": o.nnz_jac, "batch_per_gpu": B,
                       "parallelism": (f"segments of every evaluation sharded over {world} rank(s), one all-gather of the owned runs per "
                                       f"evaluation pass ({os.environ.get('MPX_DIST_BACKEND', 'nccl')})" if shard
                                       else f"independent evaluation points x{world}")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if not (shard and world > 1) else None, "traffic": None,
                         "frac_vs_measured_peak": achieved / HBM_MEASURED_GBS if not (shard and world > 1) else None,
                         "measured_peak": HBM_MEASURED_GBS,
                         "kernel": ("whole loop: mpx_node_hess_0_3 (with the mid-point residuals, MPX_MID_RESID) + mpx_boundary_hess + mpx_equal_area_kernel (wall time of the step)" if loop5
                                    else ("mpx_asml_hes (lane per evaluation point, the tables of the pass as generated code)" if hess_mode and o.batched_plan()[1] and not os.environ.get("MPX_NO_LANES") and B >= 64
                                          else "mpx_pts_* + mpx_gather_kernel (MPX_NO_FUSE)" if os.environ.get("MPX_NO_FUSE") else f"mpx_asm_{'hes' if hess_mode else 'fgj'} (fused point + gather pass)") if adaptive
                                    else "mpx_node_hessn_* (node-ordered tiles of the mixed-degree grid)" if hess_mode and isinstance(P, (list, tuple)) and len(set(P)) > 1
                                    else f"mpx_light_{'fgq' if mask & MPX_GRAD else 'fg'}_0_{o.light_plan()[0]} (matrix cores)" if partial_sel and not mask & MPX_JAC and o.light_plan()[0] and not os.environ.get("MPX_NO_LIGHT")
                                    else f"mpx_node_{'hess' if hess_mode else 'fgj' if mask & (MPX_GRAD | MPX_JAC) else 'fg'}_0_*"),
                         "kernel_us": kernel_s * 1e6,
                         "bytes_per_eval": bytes_eval, "evals_per_launch": B,
                         "algorithmic_bytes_per_launch": B * bytes_eval},
        }
        if hess_mode and not loop5:
            out["roofline"]["note"] = ("SURVEY 8(d) byte model: it charges all n_g multipliers although the kernel reads only those of rows with second "
                                       "derivatives, and a working set this small is partly Infinity-Cache resident -- read `frac` together with "
                                       "`frac_by_traffic` (PMC bytes actually moved over this run's kernel time) where the line carries it")
        # what the numbers are pinned to: never read the CPU ratio as "vs CasADi", nor the parity as "vs CasADi's AD"
        gold = sorted(fn for fn in os.listdir(os.path.join(ROOT, "tests", "golden")) if fn.endswith(".npz"))
        out["parity"] = {"goldens": f"{len(gold)} files under tests/golden/ (outputs of the imported reference: tables from CollocationRoots / Collocation, "
                                    "NLP vectors incl. grad_gamma_x / grad_gamma_p from mpopt.create_nlp())",
                         "generator": "tests/golden/make_golden.py over tests/golden/casadi_shim.py (sympy stand-in for CasADi, which is absent): CasADi's own AD / "
                                      "evaluation order is unpinned",
                         "degree>10 tables": "mpmath (50 digits): the reference's 'numerical' back-end is 4e-9 off at degree 20 and 4e-4 at degree 30",
                         "tolerance": "1e-10 per entry, one floor per entry class (tests/helpers.py: assert_by_class); indices exact",
                         "full_size_checker": "oracle/mpopt_oracle.c (hand-derived derivatives) on configs 2-5 incl. nlp_grad"}
        out["outputs"] = ("NlpFunctions.alloc_outputs: the fastest of up to %d candidate allocations by measured node-kernel time, one-time set-up "
                          "(candidates, us per pass: %s; %d drawn, stopped by %s%s); value_placement_median / frac_placement_* are plain torch.empty allocations"
                          % (args.alloc_tries, placed["node_us_per_pass"], placed["tries_used"], placed["stopped_by"],
                             ", target %.1f us = the pass at 0.95 of the measured copy rate" % placed["target_us"] if placed["target_us"] else "")
                          if placed else "plain torch.empty allocations")
        if placed:
            out["alloc_outputs"] = {"tries_used": placed["tries_used"], "max_tries": args.alloc_tries, "stopped_by": placed["stopped_by"], "target_us": placed["target_us"],
                                    "candidates_node_us_per_pass": placed["node_us_per_pass"], "kept": placed["kept"]}
        if sweep_us:  # the same kernel on plain allocations: four fresh ones + (the timed one | the four candidates of alloc_outputs)
            plain_us = sweep_us + (placed["node_us_per_pass"] if placed else [kernel_s * 1e6])
            fr = sorted(B * bytes_eval / (us * 1e-6) / 1e9 / HBM_PEAK_GBS for us in plain_us)
            out["roofline"].update(frac_placement_median=fr[len(fr) // 2], frac_placement_min=fr[0], frac_placement_max=fr[-1])
            # what a caller typically gets: the step with the MEDIAN node-kernel time over the five placements (the rest of the
            # step -- prefix, boundary, launch gaps -- as measured in the timed region)
            ks = sorted(plain_us)
            step_med = elapsed / K + (ks[len(ks) // 2] * 1e-6 - kernel_s)
            out["value_placement_median"] = world * B / step_med
        if extra:
            out["extras"] = dict(extra, note="opt-in MPX_JAC_VARIABLE_ONLY (resident jac buffers keep the constant D / interpolation "
                                             "entries); not the metric: the headline rewrites every entry on every evaluation")
        # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of
        # this same command, newest round first), and the fraction of peak that traffic amounts to over THIS run's kernel time
        def _first(*rel):
            for r_ in rel:
                p_ = os.path.join(ROOT, "profiles", r_, "traffic.json")
                if os.path.exists(p_):
                    return p_, json.load(open(p_))
            return None, None
        # (newest round first; every secondary workload has a PMC pass since round 5: tools/r6_evidence.sh -> profiles/r6_final/)
        cfgn = args.workload[6] if args.workload.startswith("config") else ""
        tsrc = {"config2-fgj": (("r6_final/headline", "r5_final/headline", "r4_final/headline", "r3_final/headline", "r2_headline"), 4096),
                "config3-fgj": (("r6_final/c3_fgj", "r5_final/c3_fgj", "r4_final/c3_fgj", "r3_final/c3_fgj", "r2_c3_spans"), 512),
                "config3-hess": (("r6_final/c3_hess", "r5_final/c3_hess", "r4_final/c3_hess", "r3_final/c3_hess"), 2048),
                "config5-hess": (("r6_final/c5_hess", "r5_final/c5_hess", "r2_config5_hess"), 4096), "config2-hess": (("r6_final/c2_hess", "r5_final/c2_hess", "r2_config2_hess"), 4096),
                "config4-fgj": (("r6_final/c4_fgj", "r5_final/c4_fgj",), 4096), "config4-hess": (("r6_final/c4_hess", "r5_final/c4_hess",), 4096), "config5-fgj": (("r6_final/c5_fgj", "r5_final/c5_fgj",), 4096),
                "adaptive-fgj": (("r6_final/adaptive_fgj", "r5_final/adaptive_fgj", "r4_final/adaptive", "r3_final/adaptive", "r3_adaptive2"), 4096),
                "adaptive-hess": (("r6_lanes/adaptive_hess_20x5", "r5_lanes/adaptive_hess") if not os.environ.get("MPX_NO_LANES") else ("r5_final/adaptive_hess",), 4096),
                "config5-loop": (("r6_final/config5_loop", "r5_final/config5_loop", "r4_final/config5_loop", "r3_final/config5_loop"), 512)}.get(args.workload)
        if partial_sel:  # the light passes (no Jacobian values): one PMC pass per configuration and selection
            seln = "_".join(w for w in ("f", "g", "grad_f") if w in sel)
            tsrc = None
            if not mask & MPX_JAC and cfgn in "2345" and cfgn:
                tsrc = ((f"r6_final/c{cfgn}_light_{seln}", f"r5_final/c{cfgn}_light_{seln}") + ((f"r4_c3_fg/after_{seln}",) if cfgn == "3" else ()) + ((f"r4_final/c2_light_{seln}",) if cfgn == "2" else ()),
                        512 if cfgn == "3" else 4096)
        if tsrc and B == tsrc[1]:
            tfp, tr = _first(*tsrc[0])
            wl = tr.get("workload") if tr else None
            if tr and (not isinstance(wl, dict) or wl == {"segments": S, "degree": P, "batch": B}) and (not partial_sel or wl == args.workload):
                tb = tr.get("bytes_per_launch", tr.get("bytes_per_pass"))
                if tb is None and "bytes_per_outer_iteration" in tr:  # the config-5 loop: one step = 5 outer iterations
                    tb = 5 * tr["bytes_per_outer_iteration"]
                out["roofline"]["traffic"] = tb
                out["roofline"]["frac_by_traffic"] = tb / kernel_s / 1e9 / HBM_PEAK_GBS
                out["roofline"]["traffic_source"] = (os.path.relpath(tfp, ROOT) + " (" + str(tr.get("kernel", "the timed kernels")) +
                                                     "; rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; separate passes; 2 x FETCH + WRITE)")
        if world == 1 and not args.no_cpu_baseline and args.workload == "config2-fgj" and not partial_sel:
            from oracle.c_oracle import COracle

            C = COracle(["moon_lander"], S, P, "LGR")
            ns = min(B, 64)
            ph = np.full(o.n_p, 1.0 / S)
            # >= 5 warm-ups, >= 30 repeats, median + p10 / p90 (perf_counter around every repeat); one repeat = a few
            # passes over the sample so that it lasts ~0.3 s
            for _ in range(5):
                C.time_many(Zh[:ns], ph, 1)
            t1 = C.time_many(Zh[:ns], ph, 2) / 2
            n_rep = 35
            inner = max(1, int(args.cpu_seconds / n_rep / max(t1, 1e-6)))
            rates = sorted(ns * inner / C.time_many(Zh[:ns], ph, inner) for _ in range(n_rep))
            tt, reps = sum(ns * inner / r_ for r_ in rates), inner * n_rep
            cpu_med, cpu_p10, cpu_p90 = rates[n_rep // 2], rates[int(0.1 * n_rep)], rates[int(0.9 * n_rep)]
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
            cfg2 = (problems.moon_lander, S, P, "LGR", ["moon_lander"], 1.0, [1])
            cfg1 = (problems.moon_lander, 20, 3, "LGR", ["moon_lander"], 1.0, [1])
            out["ipopt_iter"] = dict(ipopt_iter_report(*cfg2, dev_id), unit="us", config=f"moon lander {S}x{P} LGR (configs[1]), B=1, host pointers",
                                     call_mix="per iterate, in order: nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l; one iteration = 1.15 (nlp_f + nlp_g) + nlp_grad_f "
                                              "+ nlp_jac_g + nlp_hess_l (moon_lander.ipynb:192-198); medians of the per-call times")
            out["ipopt_iter_config0"] = dict(ipopt_iter_report(*cfg1, dev_id, seconds=0.3), unit="us",
                                             config="moon lander 20x3 LGR (configs[0]), B=1, host pointers")
            cfg5 = (problems.hyper_sensitive, 4000, 3, "LGR", ["hyper_sensitive"], 1e-3, [0])
            out["ipopt_iter_config4"] = dict(ipopt_iter_report(*cfg5, dev_id, seconds=0.3), unit="us",
                                             config="hyper-sensitive 4000x3 LGR (configs[4]), B=1, host pointers")
            out["cpu_baseline"] = {"value": cpu_med, "unit": "evals/s", "cores": 1, "kind": "port", "p10": cpu_p10, "p90": cpu_p90, "repeats": n_rep,
                                   "sample": f"{ns} of the same evaluation points x {inner} passes per repeat, {n_rep} repeats (median; p10 / p90 beside it), "
                                             f"oracle/mpopt_oracle.c (gcc -O2, scalar, values only), {tt:.1f} s",
                                   "host_cpus": os.cpu_count(),
                                   "vs_casadi_published": "the port is NOT CasADi: at moon lander 10x6 it takes 2.0 us per f+g+grad_f+jac_g against the 64.5 us CasADi's "
                                                          "recorded nlp_f + nlp_g + nlp_grad_f + nlp_jac_g sum to (docs/source/notebooks/moon_lander.ipynb:203-210; "
                                                          "profiles/r3_report.md anchor row), i.e. about 30x faster per evaluation -- the GPU / port ratio "
                                                          "understates GPU / CasADi by that factor"}
            out["casadi"] = casadi_probe(S, P, Zh[:ns], ph, g[0].cpu().numpy())
    # Secondary measurement on N > 1 lines (RCCL on the path).  It runs AFTER the headline is final and under a
    # watchdog: whatever happens in it -- an exception, a collective that never returns -- rank 0 still prints its ONE line.
    if world > 1 and args.workload == "config2-fgj" and not args.no_extras:
        import threading

        def bail():
            if rank == 0:
                out["segment_shard"] = {"error": "no result within 150 s (watchdog)"}
                out["rccl"] = census
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(150.0, bail)
        dog.daemon = True
        dog.start()
        try:
            shard_extra = segment_shard_report(dev, dev_id, rank, world, backend)
        except Exception as e:  # never lose the headline line to the secondary measurement
            shard_extra = {"error": repr(e)[:300]}
        dog.cancel()
        if rank == 0:
            out["segment_shard"] = shard_extra
    if rank == 0:
        if census is not None:
            out["rccl"] = census
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

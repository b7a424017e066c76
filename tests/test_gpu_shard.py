"""Segment-sharded evaluation under torch.distributed on the GPU: `world` processes (sharing the one GPU of the
test box, gloo rendezvous; the same code runs one process per GPU over RCCL in bench.py --workload config3-shard) each run
the node kernels of their tile range, exchange their owned runs with ONE all_gather_into_tensor and finish with the boundary
pass.  Every rank must hold the complete result, bit-identical to its own unsharded evaluation.  The same for the two other
ways to finish (gather-to-root: the root holds everything; owner-resident: a rank holds its own runs + the replicated boundary
entries and nothing else is touched), and at BASELINE full size for configs 3 and 4."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import mpopt_amd as M
    from mpopt_amd import mp, distributed as D
    from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_JAC, MPX_HESS
    import problems

    D.init_from_env(backend="gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    builder, S, po, scheme = {
        "vdp_mixed": (problems.van_der_pol, 48, [30 if s % 3 == 1 else 3 for s in range(48)], "CGL"),   # config 3's pattern
        "schwartz": (problems.two_phase_schwartz, 200, 3, "LGL"),                                        # config 4, reduced
        "kitchen_sink": (problems.kitchen_sink, 40, [2, 5, 3, 4] * 10, "LGR"),                           # parameters, DU rows, 2 phases
        "hyper_sensitive": (problems.hyper_sensitive, 700, 3, "LGR"),                                    # config 5, reduced
        "kitchen_sink_400": (problems.kitchen_sink, 400, [2, 5, 3, 4] * 100, "LGR"),                     # several tiles per bucket: corner sums cross ranks
        "dae_vdp_3_100_3": (problems.dae_vdp, 12, [3, 100, 3] * 4, "LGL"),                               # round 6: a streamed-table bucket (degree 100) between register-table buckets
        "hyper_sensitive_9x128": (problems.hyper_sensitive, 9, 128, "LGR"),                              # round 6: streamed tables, two segments per tile, 5 tiles over 3 ranks
        "config3_full": problems.BENCH_CASES[1],   # BASELINE configs[2] at full size: Van der Pol 2000 x [3,30,3], CGL
        "config4_full": problems.BENCH_CASES[2],   # BASELINE configs[3] at full size: two-phase Schwartz, 500 x 3 per phase, LGL
    }[case]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    if rank == 0:
        nlp, _ = mpo.create_nlp()  # rank 0 compiles first, the others find the cached code object
    dist.barrier()
    if rank != 0:
        nlp, _ = mpo.create_nlp()
    o = nlp["oracle"]
    o.set_stream(torch.cuda.current_stream().cuda_stream)
    B = 3
    rng = np.random.default_rng(5)  # identical inputs on every rank
    Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.05 * rng.standard_normal((B, o.n_z)), device=dev)
    w = rng.uniform(0.5, 1.5, (ocp.n_phases, S))
    p = torch.tensor((w / w.sum(axis=1, keepdims=True)).ravel(), device=dev)
    lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev)
    sig = torch.tensor(rng.uniform(0.5, 1.5, B), device=dev)

    def outputs(fill):
        mk = lambda *s: torch.full(s, fill, dtype=torch.float64, device=dev)
        return dict(f=mk(B), g=mk(B, o.n_g), grad_f=mk(B, o.n_z), jac_val=mk(B, o.nnz_jac), hess_val=mk(B, o.nnz_hess))

    ref, ref_fg = outputs(0.0), outputs(0.0)
    full = MPX_F | MPX_G | MPX_GRAD | MPX_JAC | MPX_HESS
    o.eval_device(MPX_F | MPX_G | MPX_GRAD | MPX_JAC, B, Z, p, 0, None, None, ref["f"], ref["g"], ref["grad_f"], ref["jac_val"], None)
    o.eval_device(MPX_HESS, B, Z, p, 0, lam, sig, None, None, None, None, ref["hess_val"])
    # the values-only pass: a sharded evaluation runs the node kernels (tile ranges), so the reference is the unsharded pass of the
    # same kernels; the light-pass kernels of grids with a high degree (matrix cores, mpx_light_*) give the same g bit for bit and
    # an f that rounds differently in the last place (another summation order)
    os.environ["MPX_NO_LIGHT"] = "1"
    o.eval_device(MPX_F | MPX_G, B, Z, p, 0, None, None, ref_fg["f"], ref_fg["g"], None, None, None)
    o.sync()
    del os.environ["MPX_NO_LIGHT"]
    ref_light = outputs(0.0)
    o.eval_device(MPX_F | MPX_G, B, Z, p, 0, None, None, ref_light["f"], ref_light["g"], None, None, None)
    o.sync()
    assert torch.equal(ref_light["g"], ref_fg["g"])
    assert float((ref_light["f"] - ref_fg["f"]).abs().max()) <= 1e-13 * max(1.0, float(ref_fg["f"].abs().max()))
    ev = D.SegmentShardedEvaluator(o)
    assert (ev.rank, ev.world) == (rank, world)
    _, cuts = o.shard_info(MPX_JAC)
    assert cuts[-1] == o.n_tiles and (np.diff(cuts) > 0).all(), cuts  # every rank got tiles
    out = outputs(float("nan"))
    ev.eval(full, B, Z, p, lam, sig, **out)
    o.sync()
    for k in ref:
        assert torch.equal(out[k], ref[k]), (case, rank, k, float((out[k] - ref[k]).abs().max()))
    # partial masks: only g (MODE_FG) and only jac values
    out2 = outputs(float("nan"))
    ev.eval(MPX_F | MPX_G, B, Z, p, f=out2["f"], g=out2["g"])
    ev.eval(MPX_JAC, B, Z, p, jac_val=out2["jac_val"])
    o.sync()
    for k, want in (("g", ref_fg["g"]), ("f", ref_fg["f"]), ("jac_val", ref["jac_val"])):
        assert torch.equal(out2[k], want), (case, rank, "partial mask", k, float((out2[k] - want).abs().max()))
    ev.close()
    # gather-to-root: the root holds the complete result
    ev = D.SegmentShardedEvaluator(o, mode="root", root=world - 1)
    out4 = outputs(float("nan"))
    ev.eval(full, B, Z, p, lam, sig, **out4)
    o.sync()
    if rank == world - 1:
        for k in ref:
            assert torch.equal(out4[k], ref[k]), (case, rank, "root", k)
    assert ev.exchange_doubles(MPX_JAC, B)[1] == (0 if rank != world - 1 else world * ev.exchange_doubles(MPX_JAC, B)[0])
    ev.close()
    # owner-resident: own runs + everything the boundary pass writes (owned by nobody) are there, the rest is untouched
    ev = D.SegmentShardedEvaluator(o, mode="owner")
    out5 = outputs(float("nan"))
    ev.eval(full, B, Z, p, lam, sig, **out5)
    o.sync()
    assert torch.equal(out5["f"], ref["f"])
    assert ev.exchange_doubles(MPX_JAC, B)[0] * 20 < o.shard_info(MPX_JAC)[0] * B  # the partials are a sliver of the runs
    for name, key in (("g", "g"), ("grad_f", "grad_f"), ("jac_g", "jac_val"), ("hess_l", "hess_val")):
        n = ref[key].shape[1]
        owner = np.full(n, -1)
        for r in range(world):
            for off, ln in ev.owned(name, r):
                assert (owner[off:off + ln] == -1).all()
                owner[off:off + ln] = r
        here = torch.tensor((owner == rank) | (owner == -1), device=dev)
        assert torch.equal(out5[key][:, here], ref[key][:, here]), (case, rank, "owner", name)
        assert torch.isnan(out5[key][:, ~here]).all(), (case, rank, "owner: foreign entries were written", name)
    ev.close()
    out3 = outputs(float("nan"))  # back to a plain context
    o.eval_device(MPX_F | MPX_G | MPX_GRAD | MPX_JAC, B, Z, p, 0, None, None, out3["f"], out3["g"], out3["grad_f"], out3["jac_val"], None)
    o.sync()
    assert torch.equal(out3["g"], ref["g"]) and torch.equal(out3["jac_val"], ref["jac_val"])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


@pytest.mark.parametrize("case,world", [("vdp_mixed", 2), ("schwartz", 3), ("kitchen_sink", 2), ("hyper_sensitive", 2), ("kitchen_sink_400", 3), ("dae_vdp_3_100_3", 2), ("hyper_sensitive_9x128", 3),
                                        ("config3_full", 2), ("config4_full", 2)])
def test_segment_sharded_evaluator_under_torch_distributed(case, world):
    import torch.multiprocessing as tmp

    port = _free_port()
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, "ok") for r in range(world)]

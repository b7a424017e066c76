"""N>1 paths on CPU: world_size-2 gloo processes exercise the sharding helpers and the collective
choreography of mpopt_amd/distributed.py (no GPU needed: the arrays are synthetic / oracle-made)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as tmp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import mpopt_amd as M
    from mpopt_amd import mp, distributed as D
    import problems

    r, w, lr = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    # (1) batch sharding: the bench's N>1 path -- disjoint, complete, balanced
    b, e = D.shard_range(1001, w, r)
    cover = torch.zeros(1001)
    cover[b:e] = 1
    dist.all_reduce(cover)
    assert (cover == 1).all() and abs((e - b) - 1001 / w) <= 1
    # (2) max-over-ranks timing
    assert D.max_over_ranks(1.0 + r) == float(w)
    # (3) segment sharding (mpx_shard_*): every rank holds only the runs it owns (structure from libmpx, structure-only
    #     context), ONE gloo all_gather_into_tensor of the padded exchange buffers moves them, and every rank ends up with
    #     the complete arrays -- for the jac_g pass and the hess_l pass, mixed degrees, a batch of 2 points
    from helpers import emulate_shard_exchange
    from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_JAC, MPX_HESS

    S = 24
    po = [30 if s % 3 == 1 else 3 for s in range(S)]
    ocp = problems.van_der_pol(mp, M.math)
    o = M.NlpFunctions(ocp, S, po, "CGL", with_device=False)
    o.shard_setup(w, r)
    B = 2
    rng = np.random.default_rng(7)  # same stream on every rank: the "true" full arrays
    for mask, nnz in ((MPX_F | MPX_G | MPX_GRAD | MPX_JAC, o.nnz_jac), (MPX_HESS, o.nnz_hess)):
        rank_len, cuts = o.shard_info(mask)
        tab = o.shard_table(mask)
        assert cuts[0] == 0 and cuts[-1] == o.n_tiles and (np.diff(cuts) >= 0).all()
        sizes = {0: nnz, 1: int(tab[tab[:, 1] == 1][:, 2:4].sum(axis=1).max()) if (tab[:, 1] == 1).any() else 0,
                 2: int(tab[tab[:, 1] == 2][0, 4])}
        full = {k: rng.standard_normal(B * n) for k, n in sizes.items()}
        mine = {k: np.full(B * n, np.nan) for k, n in sizes.items()}
        owned = {k: np.zeros(B * n) for k, n in sizes.items()}
        for rr, kind, off, ln, stride, dst in tab:
            assert stride == sizes[int(kind)] or int(kind) == 1
            for b_ in range(B):
                sl = slice(off + b_ * stride, off + b_ * stride + ln)
                owned[int(kind)][sl] += 1
                if rr == r:
                    mine[int(kind)][sl] = full[int(kind)][sl]
        # every tile-produced value is owned exactly once; what nobody owns belongs to the boundary pass (terminal / linking entries)
        jr, jc = (o.jac_pattern() if mask != MPX_HESS else o.hess_pattern())
        assert owned[0].max() == 1 and owned[2].max() == 1
        # (hess_l on a MIXED-DEGREE grid runs over node-ordered tiles: fewer of them than bucket tiles, and exactly the spare partial
        # slots at the end of every phase's block are nobody's; everywhere else every slot is owned)
        if mask == MPX_HESS and len(set(int(d) for d in o.poly_orders)) > 1:
            nred, tpp, nht = sizes[2] // o.n_tiles, o.n_tiles // o.ocp.n_phases, -(-o.n_nodes // 256)
            spare = np.concatenate([np.arange((ph * tpp + nht) * nred, (ph + 1) * tpp * nred) for ph in range(o.ocp.n_phases)])
            for b_ in range(B):
                assert np.array_equal(np.flatnonzero(owned[2].reshape(B, -1)[b_] == 0), spare)
        else:
            assert owned[2].min() == 1
        if 1 in sizes and sizes[1]:
            stride1 = int(tab[tab[:, 1] == 1][0, 4])
            assert owned[1].reshape(B, -1)[:, :stride1].min() == 1
        mine["rank"] = r

        def all_gather(send):
            ts, tr = torch.tensor(send), torch.empty(w * len(send), dtype=torch.float64)
            dist.all_gather_into_tensor(tr, ts)
            return tr.numpy()

        emulate_shard_exchange(tab, rank_len, w, B, mine, all_gather)
        for k in sizes:
            got, ok = mine[k], owned[k] > 0
            assert np.array_equal(got[ok], full[k][ok]), (mask, k)  # bit-exact assembly on every rank
            assert np.isnan(got[~ok]).all()
    # (4) gather-to-root of the same runs (mode "root" of SegmentShardedEvaluator): only the root receives; and the
    #     owner-resident exchange (tile partials only) through the same collective
    from mpopt_amd._lib import MPX_OWNER_RESIDENT

    mask = MPX_F | MPX_G | MPX_GRAD | MPX_JAC
    rank_len, _ = o.shard_info(mask)
    send = torch.full((rank_len * B,), float(r), dtype=torch.float64)
    parts = list(torch.empty(w, rank_len * B, dtype=torch.float64).unbind(0)) if r == 0 else None
    dist.gather(send, parts, dst=0)
    if r == 0:
        assert all(bool((parts[k] == float(k)).all()) for k in range(w))
    part_len, cuts = o.shard_info(mask | MPX_OWNER_RESIDENT)
    tabp = o.shard_table(mask)
    nred = int(tabp[tabp[:, 1] == 2][0, 3]) // max(int(cuts[1] - cuts[0]), 1)
    assert part_len >= int(np.diff(cuts).max()) * nred and part_len < rank_len  # partials only: a small fraction of the runs
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]


def _evaluator_worker(rank, world, port, q):
    """SegmentShardedEvaluator itself -- mode logic, buffer sizes, the collective of every mode, group-relative root -- at world sizes
    4 and 8 over gloo, with the device calls of the oracle replaced by a host stand-in that is driven by libmpx's REAL shard tables
    (tests/helpers.py: HostShardOracle).  Every owned run must arrive bit-exactly where the mode promises it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import mpopt_amd as M
    from mpopt_amd import mp, distributed as D
    from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_JAC, MPX_HESS
    from helpers import HostShardOracle
    import problems

    r, w, _ = D.init_from_env(backend="gloo")
    B = 2
    cases = [(problems.van_der_pol, 48, [30 if s % 3 == 1 else 3 for s in range(48)], "CGL"),   # config 3's degree pattern: hess_l by node-ordered tiles
             (problems.two_phase_schwartz, 200, [3] * 200, "LGL")]                              # config 4's shape: two phases
    FGJ = MPX_F | MPX_G | MPX_GRAD | MPX_JAC
    for builder, S, po, scheme in cases:
        struct = M.NlpFunctions(builder(mp, M.math), S, po, scheme, with_device=False)
        for mode in D.SegmentShardedEvaluator.MODES:
            root = w - 1 if mode == "root" else 0  # (a root other than rank 0)
            ho = HostShardOracle(struct, B, seed=5)
            ev = D.SegmentShardedEvaluator(ho, mode=mode, root=root)
            assert (ev.rank, ev.world, ev.backend) == (r, w, "gloo")
            nan = lambda *s_: torch.full(s_, float("nan"), dtype=torch.float64)
            z, p = torch.zeros(B, struct.n_z, dtype=torch.float64), torch.zeros(struct.n_p, dtype=torch.float64)
            f, g, gq, jv, hv = nan(B), nan(B, struct.n_g), nan(B, struct.n_z), nan(B, struct.nnz_jac), nan(B, struct.nnz_hess)
            ev.eval(FGJ | MPX_HESS, B, z, p, None, None, f, g, gq, jv, hv)
            assert [c[0] for c in ho.calls] == (["nodes", "boundary"] * 2 if (mode != "root" or r == root) else ["nodes"] * 2), ho.calls
            complete = mode == "allgather" or (mode == "root" and r == root)
            for name, arr, mask in (("jac_g", jv, FGJ), ("hess_l", hv, MPX_HESS)):
                truth = torch.tensor(ho.truth[name])
                own_any = torch.zeros(arr.shape[1], dtype=torch.bool)
                own_me = torch.zeros(arr.shape[1], dtype=torch.bool)
                for rr in range(w):
                    for off, ln in struct.shard_owned(name, rr).tolist():
                        own_any[off:off + ln] = True
                        if rr == r:
                            own_me[off:off + ln] = True
                have = own_any if complete else own_me
                assert torch.equal(arr[:, have], truth[:, have]) and arr[:, ~have].isnan().all(), (mode, name, r)
            if mode == "owner":  # the rank's own node rows of g / grad_f, stored directly
                for name, arr in (("g", g), ("grad_f", gq)):
                    m = torch.zeros(arr.shape[1], dtype=torch.bool)
                    for off, ln in struct.shard_owned(name, r).tolist():
                        m[off:off + ln] = True
                    assert torch.equal(arr[:, m], torch.tensor(ho.truth[name])[:, m]) and arr[:, ~m].isnan().all()
            # the tile partials of the LAST pass (hess_l) and f of the first: complete wherever the boundary pass ran
            if mode != "root" or r == root:
                assert not f.isnan().any()
                fs = [torch.empty_like(f) for _ in range(w)] if mode != "root" else None
                if fs is not None:
                    dist.all_gather(fs, f)
                    assert all(torch.equal(fs[0], x) for x in fs)  # same slots, same order, same bits on every rank
                    assert torch.equal(f, torch.tensor(ho._truth_kind(FGJ, 2)).sum(dim=1))
            else:
                assert f.isnan().all()
            sent, recvd = ev.exchange_doubles(FGJ, B)
            n_full, _ = struct.shard_info(FGJ)
            if mode == "owner":
                assert sent < max(n_full * B, 2)  # partials only
            else:
                assert sent == max(n_full * B, 2) and recvd == (w * sent if complete else 0)
            ev.close()
            dist.barrier()
        struct.close()
    dist.destroy_process_group()
    q.put((rank, "ok"))


@pytest.mark.parametrize("world", [4, 8])
def test_segment_sharded_evaluator_modes_over_gloo(world):
    port = _free_port()
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_evaluator_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(420)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(world)) == [(r, "ok") for r in range(world)]


def test_partition_tiles_properties():
    from mpopt_amd.distributed import partition_tiles, shard_range

    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 21, 300):
        w = rng.integers(1, 100, n)
        for world in (1, 2, 4, 8):
            parts = partition_tiles(w, world)
            assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == n
            assert all(a <= b for a, b in parts) and all(parts[k][1] == parts[k + 1][0] for k in range(world - 1))
            if n >= 4 * world:
                loads = [w[a:b].sum() for a, b in parts]
                assert max(loads) <= w.sum() / world + w.max()
    assert [shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


@pytest.mark.parametrize("case", ["vdp_mixed", "schwartz", "kitchen_sink", "dae_vdp_3_100_3", "kitchen_sink_128"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_owner_resident_ownership_tables(case, world):
    """mpx_shard_owned (structure only, no GPU): over all ranks every node row of g, every node entry of grad_f and every tile
    value of jac_val / hess_val is owned exactly once; what nobody owns is exactly what the boundary pass writes (terminal rows,
    control-slope continuity rows, event rows and their Jacobian entries; the t0 / tf / a entries of grad_f; the (t0, tf, a)
    corner and terminal entries of hess_l); the value runs equal the kind-0 runs of mpx_shard_table."""
    for p_ in (ROOT, os.path.join(ROOT, "tests")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import mpopt_amd as M
    from mpopt_amd import mp
    from mpopt_amd._lib import MPX_HESS, MPX_JAC
    import problems

    builder, S, po, scheme = {"vdp_mixed": (problems.van_der_pol, 48, [30 if s % 3 == 1 else 3 for s in range(48)], "CGL"),
                              "schwartz": (problems.two_phase_schwartz, 200, [3] * 200, "LGL"),
                              "kitchen_sink": (problems.kitchen_sink, 40, [2, 5, 3, 4] * 10, "LGR"),
                              # round 6: degrees above the LDS tables (tiles of one or two segments; a streamed bucket between low-degree ones)
                              "dae_vdp_3_100_3": (problems.dae_vdp, 12, [3, 100, 3] * 4, "LGL"),
                              "kitchen_sink_128": (problems.kitchen_sink, 10, [128] * 10, "LGR")}[case]
    ocp = builder(mp, M.math)
    o = M.NlpFunctions(ocp, S, po, scheme, with_device=False)
    o.shard_setup(world, 0)
    sizes = {"g": o.n_g, "grad_f": o.n_z, "jac_g": o.nnz_jac, "hess_l": o.nnz_hess}
    count = {k: np.zeros(n, np.int64) for k, n in sizes.items()}
    for r in range(world):
        for k in sizes:
            runs = o.shard_owned(k, r)
            assert (runs[:, 1] > 0).all() and (np.diff(runs[:, 0]) > 0).all()  # sorted, disjoint, non-empty
            assert (runs[1:, 0] >= runs[:-1, 0] + runs[:-1, 1]).all()
            for off, ln in runs:
                count[k][off:off + ln] += 1
        for mask, k in ((MPX_JAC, "jac_g"), (MPX_HESS, "hess_l")):
            tab = o.shard_table(mask)
            mine = tab[(tab[:, 0] == r) & (tab[:, 1] == 0)][:, 2:4]
            assert np.array_equal(mine, o.shard_owned(k, r))
    for k in sizes:
        assert count[k].max() == 1
    N, nx, nu, na, nph = o.n_nodes, ocp.nx, ocp.nu, ocp.na, ocp.n_phases
    # grad_f: everything but the (t0, tf, a) entries of every phase is a node entry
    nzp = o.n_z // nph
    is_node = np.ones(o.n_z, bool)
    for ph in range(nph):
        is_node[ph * nzp + (nx + nu) * N: (ph + 1) * nzp] = False
    assert np.array_equal(count["grad_f"] == 1, is_node)
    # g: the unowned rows are few (terminal, continuity, events) and the owned ones are whole node blocks
    n_unowned = int((count["g"] == 0).sum())
    assert n_unowned <= nph * (8 + nu * S) + 8 * nph and (n_unowned > 0 or case in ("vdp_mixed", "dae_vdp_3_100_3"))  # Van der Pol: no terminal / linking rows
    jr, jc = o.jac_pattern()
    assert set(np.unique(jr[count["jac_g"] == 0])) <= set(np.nonzero(count["g"] == 0)[0])  # unowned Jacobian values sit on unowned rows
    hr, hc = o.hess_pattern()
    un = count["hess_l"] == 0
    if case not in ("vdp_mixed", "dae_vdp_3_100_3"):  # (time-independent dynamics without terminal functions have no corner / terminal entries)
        loc = (hc[un] % nzp) % N  # node index of a node column
        assert un.sum() > 0 and ((~is_node[hc[un]]) | (loc == 0) | (loc == N - 1)).all()  # a global variable or a phase end
    o.shard_setup(1, 0)
    o.close()

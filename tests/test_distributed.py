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
    from oracle.mpopt_oracle import OracleNLP

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
    # (3) segment sharding: every rank fills only the entries of its tile range; a SUM all-reduce
    #     assembles exactly the full arrays (structure from libmpx, values from the numpy oracle)
    S, po = 64, 5
    ocp = problems.moon_lander(mp, M.math)
    o = M.NlpFunctions(ocp, S, [po] * S, "LGR", with_device=False)
    ranges = D.partition_tiles(o.tile_weights(), w)
    assert ranges[0][0] == 0 and ranges[-1][1] == o.n_tiles and all(ranges[k][1] == ranges[k + 1][0] for k in range(w - 1))
    O = OracleNLP(ocp, S, po, "LGR")
    z = O.initial_guess() + 0.1
    p = np.full(S, 1.0 / S)
    jr, jc = o.jac_pattern()
    Jfull = np.asarray(O.jac_g(z, p).todense())[jr, jc]
    mine = np.zeros_like(Jfull)
    owned = np.zeros(o.nnz_jac, bool)
    for t in range(*ranges[r]):
        a, b2 = o.tile_jac_range(t)
        owned[a:b2] = True
    if r == 0:  # entries outside all tiles (terminal rows, linking rows) come from the boundary pass
        tiles_end = max(o.tile_jac_range(t)[1] for t in range(o.n_tiles))
        owned[tiles_end:] = True
    mine[owned] = Jfull[owned]
    tj = torch.tensor(mine)
    D.allreduce_disjoint([tj])
    assert np.array_equal(tj.numpy(), Jfull)  # bit-exact assembly
    cnt = torch.tensor(owned.astype(np.float64))
    dist.all_reduce(cnt)
    assert (cnt == 1).all()  # every entry owned exactly once
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

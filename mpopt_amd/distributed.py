"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` ("nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).

Two ways the path shards:

* **Evaluation points** are independent: each rank evaluates its own slice of a batch
  (``shard_range``), no data-path collective.  This is what ``bench.py --gpus N`` measures.
* **Segments of one evaluation**: ranks run the node kernels on disjoint, contiguous tile ranges
  (``partition_tiles``), every output entry and every per-tile partial sum is produced by exactly one
  rank and all others hold zeros, so one SUM all-reduce per array (``allreduce_disjoint``) assembles
  the full result *exactly* (x + 0 + ... + 0); the boundary pass then runs on the assembled partials.
  Results are bit-identical to the single-GPU evaluation.  The payloads are small (<= 20 MB), i.e. the
  collective is latency-bound (~15 us): it pays only for large segment counts / batches.
"""
import os

import numpy as np

from ._lib import MPX_BOUNDARY_ONLY, MPX_F, MPX_G, MPX_GRAD, MPX_HESS, MPX_JAC


def init_from_env(backend=None):
    """(rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                kw["device_id"] = torch.device("cuda", local_rank)
            dist.init_process_group(backend, **kw)
    return rank, world, local_rank


def shard_range(n, world, rank):
    """Contiguous, balanced slice [begin, end) of n items for ``rank``."""
    base, rem = divmod(int(n), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def partition_tiles(weights, world):
    """Contiguous tile ranges, one per rank, balanced by weight (greedy prefix split).  Returns
    [(begin, end)] * world; ranges may be empty when there are fewer tiles than ranks."""
    w = np.asarray(weights, dtype=np.float64)
    total, cum = w.sum(), np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, len(w))] - target):
            k -= 1
        cuts.append(max(cuts[-1], min(k, len(w))))
    cuts.append(len(w))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the bench's elapsed time)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_disjoint(tensors):
    """SUM all-reduce of tensors whose non-zero entries are owned by exactly one rank each."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in tensors:
        if t is not None and t.numel():
            dist.all_reduce(t, op=dist.ReduceOp.SUM)


class SegmentShardedEvaluator:
    """One evaluation (or batch) split over ranks by collocation segments.

    ``oracle``: this rank's ``NlpFunctions`` (device-resident).  All tensors are torch CUDA tensors of
    full size on every rank; ``z``/``p``/``lam_g``/``sigma`` must be identical on all ranks."""

    def __init__(self, oracle, rank, world):
        self.o, self.rank, self.world = oracle, rank, world
        self.ranges = partition_tiles(oracle.tile_weights(), world)

    def eval(self, mask, batch, z, p, lam_g=None, sigma=None, f=None, g=None, grad_f=None, jac_val=None, hess_val=None):
        # the per-tile partial-sum buffer is shared by the (f,g,grad_f,jac_g) pass and the hess_l pass
        for sub in (mask & (MPX_F | MPX_G | MPX_GRAD | MPX_JAC), mask & MPX_HESS):
            if sub:
                self._eval_one(sub, batch, z, p, lam_g, sigma, f, g, grad_f, jac_val, hess_val)

    def _eval_one(self, mask, batch, z, p, lam_g, sigma, f, g, grad_f, jac_val, hess_val):
        o = self.o
        outs = [t for t, bit in ((g, MPX_G), (grad_f, MPX_GRAD), (jac_val, MPX_JAC), (hess_val, MPX_HESS)) if t is not None and mask & bit]
        for t in outs:
            t.zero_()
        ptr, cnt = o.partials(batch)
        part = _wrap_device_buffer(ptr, cnt, z.device)
        part.zero_()
        b, e = self.ranges[self.rank]
        o.set_tile_range(b, e, run_boundary=False)
        o.eval_device(mask, batch, z, p, 0, lam_g, sigma, f, g, grad_f, jac_val, hess_val)
        o.sync()
        allreduce_disjoint(outs + [part])
        o.set_tile_range(0, o.n_tiles, run_boundary=True)
        o.eval_device(mask | MPX_BOUNDARY_ONLY, batch, z, p, 0, lam_g, sigma, f, g, grad_f, jac_val, hess_val)
        o.sync()


def _wrap_device_buffer(ptr, count, device):
    """torch view (float64) of a raw device pointer owned by libmpx."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)

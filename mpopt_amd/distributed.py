"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` ("nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).

Two ways the path shards:

* **Evaluation points** are independent: each rank evaluates its own slice of a batch
  (``shard_range``), no data-path collective.  This is what ``bench.py --gpus N`` measures by default.
* **Segments of one evaluation** (``SegmentShardedEvaluator``): ranks run the node kernels on disjoint,
  contiguous tile ranges balanced by Jacobian block size (libmpx, ``mpx_shard_setup``).  What a rank owns
  afterwards is a handful of contiguous runs -- the value blocks of its tiles, its run of the packed
  g / grad_f staging block, its per-tile partial sums -- which libmpx packs into one exchange buffer
  (``mpx_shard_pack``).  ONE all-gather of the padded buffers (``all_gather_into_tensor``: RCCL over xGMI)
  moves every rank's runs to every rank, ``mpx_shard_unpack`` scatters them into place and the boundary
  pass finishes reductions, terminal and event rows on every rank.  Every entry is produced by exactly
  one rank and reductions keep their fixed order, so the result is bit-identical to the single-GPU
  evaluation for any rank count.  The payload of one evaluation is small (<= 20 MB), i.e. the collective
  is latency-bound: sharding segments pays for large grids / memory capacity, not for throughput.
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
            import datetime

            # a mismatch between ranks should fail in minutes, not after the default half hour
            kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("MPX_DIST_TIMEOUT", "300")))}
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


class SegmentShardedEvaluator:
    """One evaluation (or a small batch) split over the ranks of ``group`` by collocation segments.

    ``oracle``: this rank's ``NlpFunctions`` with a device; its stream must be torch's current stream
    (``oracle.set_stream(torch.cuda.current_stream().cuda_stream)``) so that kernels and collectives are
    ordered.  Inputs ``z`` / ``p`` / ``lam_g`` / ``sigma`` are full-size torch tensors on the device,
    identical on all ranks; the outputs are full-size on every rank when ``eval`` returns (asynchronously
    on the stream with the nccl backend).  With the gloo backend the exchange buffers are staged through
    host memory (tests: several ranks sharing one GPU)."""

    def __init__(self, oracle, rank=None, world=None, group=None):
        import torch.distributed as dist

        self.o, self.group = oracle, group
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        oracle.shard_setup(self.world, self.rank)
        self._buf = {}
        self.backend = dist.get_backend(group) if (dist.is_initialized() and self.world > 1) else None

    def close(self):
        """Leave sharded mode (the oracle evaluates all tiles again)."""
        self.o.shard_setup(1, 0)

    def _buffers(self, mask, batch, device):
        import torch

        key = (1 if mask & MPX_HESS else 0, int(batch))
        if key not in self._buf:
            n, _ = self.o.shard_info(mask)
            send = torch.empty(max(n * batch, 2), dtype=torch.float64, device=device)
            recv = torch.empty(self.world * max(n * batch, 2), dtype=torch.float64, device=device)
            self._buf[key] = (send, recv)
        return self._buf[key]

    def eval(self, mask, batch, z, p, lam_g=None, sigma=None, f=None, g=None, grad_f=None, jac_val=None, hess_val=None, p_per_point=0):
        # the per-tile partial-sum buffer is shared by the (f,g,grad_f,jac_g) pass and the hess_l pass
        for sub in (mask & (MPX_F | MPX_G | MPX_GRAD | MPX_JAC), mask & MPX_HESS):
            if sub:
                self._eval_one(sub, batch, z, p, p_per_point, lam_g, sigma, f, g, grad_f, jac_val, hess_val)

    def _eval_one(self, mask, batch, z, p, ppp, lam_g, sigma, f, g, grad_f, jac_val, hess_val):
        import torch.distributed as dist

        o = self.o
        if self.world == 1:
            o.eval_device(mask, batch, z, p, ppp, lam_g, sigma, f, g, grad_f, jac_val, hess_val)
            return
        vals = hess_val if mask & MPX_HESS else (jac_val if mask & MPX_JAC else None)
        send, recv = self._buffers(mask, batch, z.device)
        o.eval_device(mask, batch, z, p, ppp, lam_g, sigma, f, g, grad_f, jac_val, hess_val)  # node kernels of this rank's tiles
        o.shard_pack(mask, batch, vals, send)
        if self.backend == "gloo" and send.is_cuda:
            o.sync()
            hs, hr = send.cpu(), recv.cpu()
            dist.all_gather_into_tensor(hr, hs, group=self.group)
            recv.copy_(hr)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        o.shard_unpack(mask, batch, recv, vals)
        o.eval_device(mask | MPX_BOUNDARY_ONLY, batch, z, p, ppp, lam_g, sigma, f, g, grad_f, jac_val, hess_val)


def _wrap_device_buffer(ptr, count, device):
    """torch view (float64) of a raw device pointer owned by libmpx."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)

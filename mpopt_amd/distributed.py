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

from ._lib import MPX_BOUNDARY_ONLY, MPX_F, MPX_G, MPX_GRAD, MPX_HESS, MPX_JAC, MPX_OWNER_RESIDENT


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

    ``oracle``: this rank's ``NlpFunctions`` with a device; every ``eval`` puts it on torch's current stream
    (``oracle.set_stream(torch.cuda.current_stream().cuda_stream)``) so that kernels and RCCL collectives are
    ordered.  Inputs ``z`` / ``p`` / ``lam_g`` / ``sigma`` are full-size torch tensors on the device,
    identical on all ranks.  ``mode`` (include/mpx.h, mpx_shard_setup):

    * ``"allgather"`` -- ONE ``all_gather_into_tensor`` of every rank's owned runs; every rank holds the complete
      result when ``eval`` returns (north_star's "RCCL all-gather of residual / Jacobian blocks").
    * ``"root"`` -- the same runs collected by rank ``root`` only (``dist.gather``): the root holds the complete
      result, the others send ``rank_len * batch`` doubles and receive nothing.
    * ``"owner"`` -- owner-resident: every rank keeps the rows / value blocks of its tiles in its own output arrays
      (``owned(which)`` lists them as (offset, length) runs) and only the per-tile partial sums travel (one
      all-gather of a few KB per point); f and everything the boundary pass writes is replicated.  The mode for
      a consumer that is itself distributed, and the only one whose exchange does not grow with the grid.

    Every entry is produced by exactly one rank and the reductions keep their fixed order: results are bit-identical
    to the unsharded evaluation in all three modes.  With the gloo backend the exchange buffers are staged through
    host memory (tests: several ranks sharing one GPU)."""

    MODES = ("allgather", "root", "owner")

    def __init__(self, oracle, rank=None, world=None, group=None, mode="allgather", root=0):
        """``root``: the rank WITHIN ``group`` (not a global rank) that holds the result in mode "root"."""
        import torch.distributed as dist

        if mode not in self.MODES:
            raise ValueError(f"mode must be one of {self.MODES}")
        self.o, self.group, self.mode, self.root = oracle, group, mode, int(root)
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        oracle.shard_setup(self.world, self.rank)
        self._buf = {}
        self.backend = dist.get_backend(group) if (dist.is_initialized() and self.world > 1) else None
        self._flag = MPX_OWNER_RESIDENT if (mode == "owner" and self.world > 1) else 0

    def close(self):
        """Leave sharded mode (the oracle evaluates all tiles again)."""
        self.o.shard_setup(1, 0)

    def owned(self, which, rank=None):
        """(offset, length) runs of output ``which`` ("g", "grad_f", "jac_g", "hess_l") owned by ``rank`` (default: this one)."""
        return self.o.shard_owned(which, self.rank if rank is None else rank)

    def exchange_doubles(self, mask, batch):
        """(sent, received) doubles per rank and pass: what one ``_eval_one`` moves through the collective."""
        n, _ = self.o.shard_info(mask | self._flag)
        n = max(n * batch, 2)
        if self.mode == "root":
            return n, (self.world * n if self.rank == self.root else 0)
        return n, self.world * n

    def _buffers(self, mask, batch, device):
        import torch

        key = (1 if mask & MPX_HESS else 0, int(batch))
        if key not in self._buf:
            n, _ = self.o.shard_info(mask | self._flag)
            send = torch.empty(max(n * batch, 2), dtype=torch.float64, device=device)
            need_recv = self.mode != "root" or self.rank == self.root
            recv = torch.empty(self.world * max(n * batch, 2), dtype=torch.float64, device=device) if need_recv else None
            self._buf[key] = (send, recv)
        return self._buf[key]

    def eval(self, mask, batch, z, p, lam_g=None, sigma=None, f=None, g=None, grad_f=None, jac_val=None, hess_val=None, p_per_point=0):
        # the per-tile partial-sum buffer is shared by the (f,g,grad_f,jac_g) pass and the hess_l pass
        for sub in (mask & (MPX_F | MPX_G | MPX_GRAD | MPX_JAC), mask & MPX_HESS):
            if sub:
                self._eval_one(sub, batch, z, p, p_per_point, lam_g, sigma, f, g, grad_f, jac_val, hess_val)

    def _collect(self, send, recv):
        """The collective of the mode: recv[world][len(send)] <- every rank's send (on the root only in mode "root")."""
        import torch.distributed as dist

        host = self.backend == "gloo" and send.is_cuda
        if host:
            self.o.sync()
            hs = send.cpu()
            hr = recv.cpu() if recv is not None else None
        else:
            hs, hr = send, recv
        if self.mode == "root":  # ``root`` is a rank of the GROUP; dist.gather's dst is a global rank
            parts = list(hr.view(self.world, -1).unbind(0)) if self.rank == self.root else None
            dst = dist.get_global_rank(self.group, self.root) if self.group is not None else self.root
            dist.gather(hs, parts, dst=dst, group=self.group)
        else:
            dist.all_gather_into_tensor(hr, hs, group=self.group)
        if host and recv is not None:
            recv.copy_(hr)

    def _eval_one(self, mask, batch, z, p, ppp, lam_g, sigma, f, g, grad_f, jac_val, hess_val):
        o, fl = self.o, self._flag
        if self.world == 1:
            o.eval_device(mask, batch, z, p, ppp, lam_g, sigma, f, g, grad_f, jac_val, hess_val)
            return
        vals = hess_val if mask & MPX_HESS else (jac_val if mask & MPX_JAC else None)
        send, recv = self._buffers(mask, batch, z.device)
        if z.is_cuda and self.backend != "gloo":
            # RCCL orders its collectives against torch's CURRENT stream: the pack / unpack kernels must be on it too
            import torch

            o.set_stream(torch.cuda.current_stream(z.device).cuda_stream)
        o.eval_device(mask | fl, batch, z, p, ppp, lam_g, sigma, f, g, grad_f, jac_val, hess_val)  # node kernels of this rank's tiles
        o.shard_pack(mask | fl, batch, vals, send)
        self._collect(send, recv)
        if self.mode == "root" and self.rank != self.root:
            return  # this rank's share is with the root; its own arrays stay partial
        o.shard_unpack(mask | fl, batch, recv, vals)
        o.eval_device(mask | fl | MPX_BOUNDARY_ONLY, batch, z, p, ppp, lam_g, sigma, f, g, grad_f, jac_val, hess_val)


def device_census(device_index, group=None):
    """What proves that the ranks of a multi-GPU run sat on distinct GPUs: every rank's (rank, host, HIP device index, PCI bus
    id, device name, uuid if torch reports one), collected with ``all_gather_object``.  Returns a dict for a bench line:
    backend, world, the per-rank list and the number of distinct (host, PCI bus id) pairs."""
    import ctypes
    import socket

    import torch
    import torch.distributed as dist

    from . import _lib

    buf = ctypes.create_string_buffer(64)
    rc = _lib.lib().mpx_device_pci_bus_id(int(device_index), buf, 64)
    me = {"rank": dist.get_rank(group) if dist.is_initialized() else 0, "host": socket.gethostname(), "device": int(device_index),
          "pci_bus_id": buf.value.decode() if rc == 0 else None}
    try:
        pr = torch.cuda.get_device_properties(int(device_index))
        me["name"] = pr.name
        if getattr(pr, "uuid", None) is not None:
            me["uuid"] = str(pr.uuid)
    except Exception:  # pragma: no cover
        pass
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        allr = [None] * dist.get_world_size(group)
        dist.all_gather_object(allr, me, group=group)
        backend = dist.get_backend(group)
    else:
        allr, backend = [me], None
    distinct = len({(r["host"], r["pci_bus_id"]) for r in allr})
    return {"backend": backend, "world": len(allr), "distinct_gpus": distinct, "ranks": allr}


def _wrap_device_buffer(ptr, count, device):
    """torch view (float64) of a raw device pointer owned by libmpx."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device=device)

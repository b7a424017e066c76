"""RCCL sanity on a 1-GPU box: a world-size-1 "nccl" process group created exactly like mpopt_amd.distributed.init_from_env does it
(device_id, timeout), then the collectives the segment-sharded evaluator and bench.py use, on the stream libmpx launches on."""
import datetime
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from mpopt_amd import distributed as mpd  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
send = torch.arange(1 << 20, dtype=torch.float64, device=dev)
recv = torch.empty_like(send)
dist.all_gather_into_tensor(recv, send)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(recv, send)
t = mpd.max_over_ranks(1.25, device=dev)
assert t == 1.25, t
flag = torch.tensor([1.0], dtype=torch.float64, device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
assert float(flag.item()) == 1.0
print("rccl ok:", dist.get_backend(), torch.cuda.get_device_name(0), "nccl version", torch.cuda.nccl.version())
dist.destroy_process_group()

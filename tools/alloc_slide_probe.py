"""ONE big device allocation, the output arrays carved out of it at different offsets: does the node-kernel time depend on the
offset (i.e. on address bits a library could choose by over-allocating and sliding), or only on the allocation?"""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
import problems  # noqa: E402

dev = torch.device("cuda", 0)
B = 4096
mpo = mp.mpopt(problems.moon_lander(mp, M.math), 1000, 5, "LGR")
o = mpo.create_nlp()[0]["oracle"]
p = torch.tensor(np.full(o.n_p, 1e-3), device=dev)
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z)))).to(dev)
f = torch.empty(B, dtype=torch.float64, device=dev)
need = B * (o.n_g + o.n_z + o.nnz_jac)


def measure(arena, off):
    g = arena[off:off + B * o.n_g].view(B, o.n_g)
    gr = arena[off + B * o.n_g:off + B * (o.n_g + o.n_z)].view(B, o.n_z)
    jv = arena[off + B * (o.n_g + o.n_z):off + need].view(B, o.nnz_jac)
    o.geometry_reset()
    for _ in range(6):
        o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
    o.sync()
    o.profile(True)
    for _ in range(8):
        o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
    ms, n = o.profile_read()
    o.profile(False)
    return round(ms / 8 * 1e3, 1)


for trial in range(3):
    arena = torch.empty(need + (12 << 27), dtype=torch.float64, device=dev)  # 12 GB of slack
    row = []
    for off_mb in (0, 2, 64, 256, 1024, 1026, 2048, 4096, 6144, 8192, 12288):
        row.append((off_mb, measure(arena, off_mb << 17)))
    print("arena", trial, hex(arena.data_ptr()), row, flush=True)
    hold = arena if trial == 0 else None
    del arena
    torch.cuda.empty_cache()

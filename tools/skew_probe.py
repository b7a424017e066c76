"""Do the RELATIVE offsets of z / g / grad_f / jac matter?  All arrays carved from one arena; the skew between consecutive
arrays is varied (config 2, B=4096).  Back-to-back allocations of equal 2 MB-rounded size put corresponding elements of the
concurrently accessed streams at offsets that are multiples of a large power of two."""
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
Zh = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z))))
arena = torch.empty(int(9 * 2 ** 30 // 8), dtype=torch.float64, device=dev)
base = arena.data_ptr()
sizes = {"Z": B * o.n_z, "g": B * o.n_g, "gr": B * o.n_z, "jv": B * o.nnz_jac, "f": B}


def carve(skew_bytes, align=2 << 20):
    off, out = 0, {}
    for k, (name, n) in enumerate(sizes.items()):
        start = (off + align - 1) // align * align + k * skew_bytes
        out[name] = arena[start // 8:start // 8 + n]
        off = start + n * 8
    return out


for skew in [0, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, 0, (5 << 20) + 8192, 256, 0]:
    bufs = carve(skew)
    bufs["Z"].view(B, o.n_z).copy_(Zh)
    for _ in range(3):
        o.eval_device(15, B, bufs["Z"], p, 0, None, None, bufs["f"], bufs["g"], bufs["gr"], bufs["jv"], None)
    o.sync()
    o.profile(True)
    for _ in range(10):
        o.eval_device(15, B, bufs["Z"], p, 0, None, None, bufs["f"], bufs["g"], bufs["gr"], bufs["jv"], None)
    ms, n = o.profile_read()
    o.profile(False)
    offs = [(bufs[k].data_ptr() - base) for k in ("Z", "g", "gr", "jv")]
    print(f"skew {skew:9d} B: node kernel {ms / n * 1e3:8.1f} us   offsets(MB) {[round(v / 2**20, 3) for v in offs]}", flush=True)

"""Which array's placement moves the node kernel?  Re-allocate one array at a time (others fixed), config 2, B=4096."""
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
bufs = {"Z": Zh.to(dev), "f": torch.empty(B, dtype=torch.float64, device=dev), "g": torch.empty(B, o.n_g, dtype=torch.float64, device=dev),
        "gr": torch.empty(B, o.n_z, dtype=torch.float64, device=dev), "jv": torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)}


def measure():
    for _ in range(3):
        o.eval_device(15, B, bufs["Z"], p, 0, None, None, bufs["f"], bufs["g"], bufs["gr"], bufs["jv"], None)
    o.sync()
    o.profile(True)
    for _ in range(10):
        o.eval_device(15, B, bufs["Z"], p, 0, None, None, bufs["f"], bufs["g"], bufs["gr"], bufs["jv"], None)
    ms, n = o.profile_read()
    o.profile(False)
    return ms / n * 1e3


print("initial", round(measure(), 1), {k: hex(v.data_ptr()) for k, v in bufs.items()}, flush=True)
junk = []
for name in ("Z", "g", "gr", "jv", "Z", "g", "gr", "jv"):
    for k in range(3):
        old = bufs[name]
        new = torch.empty_like(old)
        if name == "Z":
            new.copy_(old)
        bufs[name] = new
        junk.append(old) if k % 2 == 0 else None
        del old
        torch.cuda.empty_cache()
        print(f"re-allocated {name:3s} -> {hex(new.data_ptr())}: {measure():8.1f} us", flush=True)

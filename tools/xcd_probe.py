"""Does the node kernel's speed depend on gcd(number of tiles, 8 XCDs)?  Moon lander, degree 5, B=4096."""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_JAC  # noqa: E402
import problems  # noqa: E402

B = 4096
dev = torch.device("cuda", 0)
for S in [int(a) for a in sys.argv[1:]] or [51 * 15, 51 * 16, 51 * 23, 51 * 24, 51 * 31, 51 * 32]:
    mpo = mp.mpopt(problems.moon_lander(mp, M.math), S, 5, "LGR")
    o = mpo.create_nlp()[0]["oracle"]
    z = torch.tensor(np.tile(mpo.initialize_solution(), (B, 1)), device=dev)
    p = torch.full((o.n_p,), 1.0 / S, dtype=torch.float64, device=dev)
    f, g = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
    gr, jv = torch.empty(B, o.n_z, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
    mask = MPX_F | MPX_G | MPX_GRAD | MPX_JAC
    for _ in range(5):
        o.eval_device(mask, B, z, p, 0, None, None, f, g, gr, jv, None)
    o.sync()
    o.profile(True)
    for _ in range(20):
        o.eval_device(mask, B, z, p, 0, None, None, f, g, gr, jv, None)
    ms, n = o.profile_read()
    us = ms / n * 1e3
    print(f"S={S:5d} tiles={o.n_tiles:3d} node kernel {us:8.1f} us  {o.bytes_fgj * B / us / 1e6:7.1f} GB/s algorithmic", flush=True)
    o.close()

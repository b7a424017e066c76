"""Race hunt (round 6): the same evaluation N times in a row at the BASELINE sizes, every output compared bit for bit with the first pass.  A missing
LDS / memory synchronisation in a kernel shows up as a value that differs once in many launches; the parity tests run each case a handful of times.
    python tools/r6_repeat_stress.py [repetitions]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import mpopt_amd as M
from mpopt_amd import mp
from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_HESS, MPX_JAC
import problems

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
CASES = [("config2", problems.BENCH_CASES[0], 4096, False), ("config3", problems.BENCH_CASES[1], 512, False), ("config4", problems.BENCH_CASES[2], 4096, False),
         ("config5", problems.BENCH_CASES[3], 4096, False), ("config3 + parameter + path row", problems.FULL_EXTRA_CASES[0], 256, False),
         ("time dependent 4000x3", problems.FULL_EXTRA_CASES[1], 1024, False), ("moon lander 50x100", (problems.moon_lander, 50, 100, "LGR"), 512, False),
         ("kitchen sink 2 x [95, 71]", (problems.kitchen_sink, 2, [95, 71], "LGR"), 512, False),
         ("adaptive 20x5", (problems.moon_lander, 20, 5, "LGR"), 4096, True), ("adaptive 100x3", (problems.moon_lander, 100, 3, "LGR"), 2048, True),
         ("adaptive kitchen sink 20x3", (problems.kitchen_sink, 20, 3, "LGR"), 1024, True)]
total = 0
for name, (builder, S, po, scheme), B, adaptive in CASES:
    ocp = builder(mp, M.math)
    mpo = (mp.mpopt_adaptive if adaptive else mp.mpopt)(ocp, S, po, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    g_ = torch.Generator(device=dev).manual_seed(7)
    z0 = torch.tensor(mpo.initialize_solution(), device=dev)
    Z = z0[None, :] + 0.03 * torch.randn((B, o.n_z), generator=g_, device=dev, dtype=torch.float64)
    p = None if adaptive else torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
    lam = torch.randn((B, o.n_g), generator=g_, device=dev, dtype=torch.float64)
    sig = torch.rand(B, generator=g_, device=dev, dtype=torch.float64) + 0.5
    for mask in (MPX_F | MPX_G | MPX_GRAD | MPX_JAC, MPX_F | MPX_G, MPX_F | MPX_GRAD, MPX_HESS):
        def run():
            mk = lambda *s_: torch.full(s_, float("nan"), dtype=torch.float64, device=dev)
            outs = [mk(B) if mask & MPX_F else None, mk(B, o.n_g) if mask & MPX_G else None, mk(B, o.n_z) if mask & MPX_GRAD else None,
                    mk(B, o.nnz_jac) if mask & MPX_JAC else None, mk(B, o.nnz_hess) if mask & MPX_HESS else None]
            o.eval_device(mask, B, Z, p, 0, lam if mask & MPX_HESS else None, sig if mask & MPX_HESS else None, *outs)
            o.sync()
            return [x for x in outs if x is not None]
        ref = run()
        assert all(bool(torch.isfinite(x).all()) for x in ref)
        bad = 0
        for r in range(N):
            got = run()
            if not all(torch.equal(a, b) for a, b in zip(ref, got)):
                bad += 1
                print("DIFFERENT", name, "mask", mask, "repetition", r, flush=True)
            del got
        total += N
        print(f"{name}, B = {B}, mask {mask}: {N} repetitions, {bad} different from the first", flush=True)
        assert bad == 0
        del ref
    o.close()
    torch.cuda.empty_cache()
print(f"repeat stress: {total} passes, all bit-identical to their first")

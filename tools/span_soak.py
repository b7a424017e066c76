"""One-off soak of the row-span scheme: random mixed-degree grids, row spans against the unpack pass (bitwise) for several masks
and batch sizes.  usage: python tools/span_soak.py <first seed> <count>"""
import os
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
from mpopt_amd._lib import MPX_F, MPX_G, MPX_GRAD, MPX_JAC  # noqa: E402
import problems  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
bad = spans = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    degs = rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 13, 16, 20, 30], size=int(rng.integers(2, 4)), replace=False)
    S = int(rng.integers(2, 400))
    runs = rng.integers(1, int(rng.integers(2, 40)), size=S)
    po = np.repeat(rng.choice(degs, size=S), runs)[:S].tolist()
    builder = [problems.van_der_pol, problems.dae_vdp, problems.kitchen_sink, problems.two_phase_schwartz, problems.hyper_sensitive][seed % 5]
    scheme = ["LGR", "LGL", "CGL"][seed % 3]
    ocp = builder(mp, M.math)
    os.environ.pop("MPX_NO_ABSORB", None)
    mpo = mp.mpopt(ocp, S, po, scheme)
    oa = mpo.create_nlp()[0]["oracle"]
    os.environ["MPX_NO_ABSORB"] = "1"
    ob = mp.mpopt(ocp, S, po, scheme).create_nlp()[0]["oracle"]
    os.environ.pop("MPX_NO_ABSORB")
    has = bool(oa.tile_spans()[1].any())
    spans += has
    ok = True
    for B in (1, 5):
        Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.05 * rng.standard_normal((B, oa.n_z)), device=dev)
        w = rng.uniform(0.4, 1.6, (ocp.n_phases, S))
        p = torch.tensor((w / w.sum(axis=1, keepdims=True)).ravel(), device=dev)
        for mask in (MPX_F | MPX_G | MPX_GRAD | MPX_JAC, MPX_F | MPX_G, MPX_GRAD):
            got = []
            for o in (oa, ob):
                mk = lambda *s: torch.full(s, float("nan"), dtype=torch.float64, device=dev)
                f, g, gr, jv = mk(B), mk(B, o.n_g), mk(B, o.n_z), mk(B, o.nnz_jac)
                o.eval_device(mask, B, Z, p, 0, None, None, f if mask & MPX_F else None, g if mask & MPX_G else None,
                              gr if mask & MPX_GRAD else None, jv if mask & MPX_JAC else None, None)
                o.sync()
                got.append((f, g, gr, jv))
            for x, y in zip(*got):
                if not (torch.equal(x.isnan(), y.isnan()) and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))):
                    ok = False
            if mask & MPX_G and got[0][1].isnan().any():
                ok = False
    print(seed, builder.__name__, scheme, "S", S, "degrees", sorted(set(po)), "N", sum(po) + 1, "spans", has, "OK" if ok else "MISMATCH", flush=True)
    bad += not ok
    oa.close(), ob.close()
print("grids", count, "with spans", spans, "mismatches", bad)

"""hess_l pass of the config-5 loop (hypersensitive 4000x3 LGR, B = 512, widths per point): plain, and with the mid-point residuals
(MPX_MID_RESID), over kernel build variants in ONE process.  usage: VARIANTS="|-DMPX_ABL_MID_NOSTORE|-DMPX_ABL_MID_NOCOMPUTE" python tools/r3_midres_ab.py"""
import os
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
import problems  # noqa: E402

variants = os.environ.get("VARIANTS", "|-DMPX_ABL_MID_NOSTORE|-DMPX_ABL_MID_NOCOMPUTE").split("|")
B = int(os.environ.get("B", 512))
builder, S, P, scheme = problems.BENCH_CASES[3]
dev = torch.device("cuda", 0)
objs = []
for v in variants:
    os.environ["MPX_HIPCC_FLAGS"] = v
    mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
    objs.append(mpo.create_nlp()[0]["oracle"])
o = objs[0]
rng = np.random.default_rng(1)
Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.01 * rng.standard_normal((B, o.n_z)), device=dev)
p = torch.tensor(rng.dirichlet(np.ones(S), B), device=dev)
lam = torch.randn(B, o.n_g, dtype=torch.float64, device=dev)
sig = torch.ones(B, dtype=torch.float64, device=dev)
hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
R = torch.empty(B, 3 * S, 1, dtype=torch.float64, device=dev)
bpbs = [int(x) for x in os.environ.get("BPBS", "0").split(",")]  # 0: the library's choice (MPX_BPB unset)
res = {(v, m, q): [] for v in variants for m in ("hess", "hess+mid") for q in bpbs}
for ob in objs:
    ob.set_mid_resid_output(R)
for rnd in range(6):
    for v, ob in zip(variants, objs):
        for name, mask, q in [(nm, mk, q) for nm, mk in (("hess", 16 | 256), ("hess+mid", 16 | 256 | 1024)) for q in bpbs]:
            os.environ.pop("MPX_BPB", None)
            if q:
                os.environ["MPX_BPB"] = str(q)
            ob.eval_device(16, B, Z, p, 1, lam, sig, None, None, None, None, hv)  # prefix sums of p
            for _ in range(3):
                ob.eval_device(mask, B, Z, p, 1, lam, sig, None, None, None, None, hv)
            ob.sync()
            ob.timer_start()
            for _ in range(20):
                ob.eval_device(mask, B, Z, p, 1, lam, sig, None, None, None, None, hv)
            res[(v, name, q)].append(ob.timer_stop() / 20 * 1e3)
for (v, name, q), t in res.items():
    print(f"{v or '(default)':40s} {name:9s} bpb {q} median {sorted(t)[len(t) // 2]:7.1f} us   min {min(t):7.1f}")

"""Scratch: interleaved A/B of kernel build variants in ONE process (guide rule 24).
usage: VARIANTS="-DMPX_NO_VEC|" python tools/ab.py     (each variant = extra hipcc flags)"""
import os, sys, time
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
S, P, B = int(os.environ.get("S", 1000)), int(os.environ.get("P", 5)), int(os.environ.get("B", 4096))
mask = int(os.environ.get("MASK", 15))
variants = os.environ.get("VARIANTS", "|-DMPX_NO_VEC").split("|")
dev = torch.device("cuda:0")
objs = []
for v in variants:
    os.environ["MPX_HIPCC_FLAGS"] = v
    if "CASE" in os.environ:
        builder, S, P, scheme = problems.BENCH_CASES[int(os.environ["CASE"])]
    else:
        builder, scheme = problems.moon_lander, "LGR"
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, P, scheme)
    nlp, _ = mpo.create_nlp()
    objs.append(nlp["oracle"])
o = objs[0]
rng = np.random.default_rng(1)
Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.01 * rng.standard_normal((B, o.n_z)), device=dev)
p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
f = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
gr = torch.empty(B, o.n_z, dtype=torch.float64, device=dev); jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
lam = torch.randn(B, o.n_g, dtype=torch.float64, device=dev); sig = torch.ones(B, dtype=torch.float64, device=dev)
hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
res = {v: [] for v in variants}
for rnd in range(int(os.environ.get("ROUNDS", 6))):
    for v, ob in zip(variants, objs):
        for _ in range(2):
            ob.eval_device(mask, B, Z, p, 0, lam, sig, f, g, gr, jv, hv)
        ob.sync(); ob.profile(True)
        for _ in range(10):
            ob.eval_device(mask, B, Z, p, 0, lam, sig, f, g, gr, jv, hv)
        ms, n = ob.profile_read(); ob.profile(False)
        res[v].append(ms / 10 * 1e3)
nbytes = o.bytes_fgj if mask & 12 else (o.bytes_hess if mask & 16 else 8 * (o.n_z + o.n_p + o.n_g + 1))
for v in variants:
    a = np.array(res[v])
    print(f"variant {v!r:28s} node kernel us: median {np.median(a):8.1f} min {a.min():8.1f} max {a.max():8.1f}   {B*nbytes/np.median(a)/1e3:7.1f} GB/s")

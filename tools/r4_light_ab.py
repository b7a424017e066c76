"""In-process A/B of kernel build flags on the light passes (f, g, f+grad_f alone) of a BASELINE configuration: one context per flag
set, the SAME arrays, interleaved rounds.   CASE=0..3 B=4096 python tools/r4_light_ab.py "" "-DMPX_LIGHT_XCD_BLOCKED=0" ..."""
import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
flags = sys.argv[1:] or [""]
builder, S, P, scheme = problems.BENCH_CASES[int(os.environ.get("CASE", 0))]
B = int(os.environ.get("B", 4096))
ctx = []
percu = []
for fl in flags:  # (tokens SEGS=n / PERCU=n inside a flag string: MPX_LIGHT_SEGS at context creation, MPX_LIGHT_PER_CU around that context's calls)
    toks = fl.split()
    os.environ["MPX_HIPCC_FLAGS"] = " ".join(t for t in toks if not t.startswith(("SEGS=", "PERCU=")))
    for t in toks:
        if t.startswith("SEGS="): os.environ["MPX_LIGHT_SEGS"] = t[5:]
    percu.append(next((t[6:] for t in toks if t.startswith("PERCU=")), None))
    mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
    ctx.append((mpo, mpo.create_nlp()[0]["oracle"]))
    os.environ.pop("MPX_LIGHT_SEGS", None)
os.environ.pop("MPX_HIPCC_FLAGS")
mpo, o = ctx[0]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
f = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev); q = torch.empty(B, o.n_z, dtype=torch.float64, device=dev)
print(builder.__name__, S, "B", B, "plan", o.light_plan())
lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev); sig = torch.ones(B, dtype=torch.float64, device=dev)
heavy = os.environ.get("HEAVY")  # HEAVY=1: the passes with the Jacobian / Hessian values instead of the light ones
jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev) if heavy else None
hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev) if heavy else None
for name, mask in ((("f+g+grad_f+jac_g", 15), ("hess_l", 16)) if heavy else (("f", 1), ("g", 2), ("f+grad_f", 5))):
    args = (mask, B, Z, p, 0, lam, sig, f if mask & 1 else None, g if mask & 2 else None, q if mask & 4 else None, jv if mask & 8 else None, hv if mask & 16 else None)
    byt = 8 * B * (o.n_z + (o.n_g if mask & 2 else 0) + (o.n_z if mask & 4 else 0) + (o.nnz_jac if mask & 8 else 0) + (o.n_g + o.nnz_hess if mask & 16 else 0))
    res, outs = [[] for _ in ctx], []
    for rnd in range(6):
        for k, (_, ok) in enumerate(ctx):
            os.environ.pop("MPX_LIGHT_PER_CU", None)
            if percu[k]: os.environ["MPX_LIGHT_PER_CU"] = percu[k]
            for _ in range(3): ok.eval_device(*args)
            ok.sync(); ok.timer_start()
            for _ in range(30): ok.eval_device(*args)
            res[k].append(ok.timer_stop() / 30 * 1e3)
            if rnd == 0: outs.append((g.clone(), q.clone(), f.clone()) + ((jv.clone(), hv.clone()) if heavy else ()))
    for k, fl in enumerate(flags):
        same = all(torch.equal(a, b) for a, b in zip(outs[k], outs[0]))
        med = sorted(res[k])[len(res[k]) // 2]
        print(f"{name:9s} [{fl or 'default':36s}] median {med:7.2f} us (whole pass)  min {min(res[k]):7.2f}  {byt / med / 1e6:5.2f} TB/s  bit-equal to first: {same}")

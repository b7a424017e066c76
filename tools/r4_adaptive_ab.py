"""In-process A/B of build flags of the fused assembled kernels (moon lander 20x5 adaptive, first-order pass and hess_l, B = 4096):
one context per flag set, the SAME input / output arrays, interleaved timing rounds.
python tools/r4_adaptive_ab.py "" "-DMPX_FUSE_XCD_BLOCKED=0" ..."""
import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
flags = sys.argv[1:] or [""]
B = int(os.environ.get("B", 4096))
ctx = []
for fl in flags:
    os.environ["MPX_HIPCC_FLAGS"] = fl
    mpo = mp.mpopt_adaptive(problems.moon_lander(mp, M.math), 20, 5, "LGR")
    ctx.append((mpo, mpo.create_nlp()[0]["oracle"]))
os.environ.pop("MPX_HIPCC_FLAGS")
mpo, o = ctx[0]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev); sig = torch.ones(B, dtype=torch.float64, device=dev)
f, g = torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
gr, jv = torch.empty(B, o.n_z, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
ref = None
for tag, mask, alg in (("fgj", 15, 8 * (2 * o.n_z + o.n_g + o.nnz_jac + 1)), ("hess", 16, 8 * (o.n_z + o.n_g + 1 + o.nnz_hess))):
    res = [[] for _ in ctx]
    outs = []
    for rnd in range(6):
        for k, (_, ok) in enumerate(ctx):
            for _ in range(3): ok.eval_device(mask, B, Z, None, 0, lam, sig, f, g, gr, jv, hv)
            ok.sync(); ok.timer_start()
            for _ in range(30): ok.eval_device(mask, B, Z, None, 0, lam, sig, f, g, gr, jv, hv)
            res[k].append(ok.timer_stop() / 30 * 1e3)
            if rnd == 0: outs.append((jv.clone(), g.clone()) if mask == 15 else (hv.clone(),))
    for k, fl in enumerate(flags):
        same = all(torch.equal(a, b) for a, b in zip(outs[k], outs[0]))
        med = sorted(res[k])[len(res[k]) // 2]
        print(f"{tag:5s} [{fl or 'default':40s}] median {med:7.2f} us  min {min(res[k]):7.2f}  frac(median) {alg * B / med / 1e3 / 8e3:.3f}  bit-equal to first: {same}")

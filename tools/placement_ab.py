"""Scratch: is there a launch geometry / workgroup mapping that is fast on EVERY physical placement of the output buffers?
Config 2, B=4096.  Repeatedly re-allocates the outputs (torch.empty after empty_cache(): same virtual addresses, new physical
pages), classifies the allocation with the default build and then times the variants on the SAME buffers."""
import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems

S, P, B = 1000, 5, int(os.environ.get("B", 4096))
dev = torch.device("cuda:0")
variants = os.environ.get("VARIANTS", "|-DMPX_MAP_NATURAL").split("|")
bpbs = [int(x) for x in os.environ.get("BPBS", "0,2,8").split(",")]
objs = []
for v in variants:
    os.environ["MPX_HIPCC_FLAGS"] = v
    mpo = mp.mpopt(problems.moon_lander(mp, M.math), S, P, "LGR")
    objs.append(mpo.create_nlp()[0]["oracle"])
o = objs[0]
rng = np.random.default_rng(1)
Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.01 * rng.standard_normal((B, o.n_z)), device=dev)
p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)


def timeit(ob, outs, n=10):
    f, g, gr, jv = outs
    for _ in range(2):
        ob.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
    ob.sync(); ob.profile(True)
    for _ in range(n):
        ob.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
    ms, nl = ob.profile_read(); ob.profile(False)
    return ms / nl * 1e3


hold = []
for k in range(int(os.environ.get("ALLOCS", 8))):
    outs = [torch.empty(s, dtype=torch.float64, device=dev) for s in (B, B * o.n_g, B * o.n_z, B * o.nnz_jac)]
    line = [f"alloc {k}"]
    for v, ob in zip(variants, objs):
        for bpb in bpbs:
            if bpb:
                os.environ["MPX_BPB"] = str(bpb)
            else:
                os.environ.pop("MPX_BPB", None)
            line.append(f"{v or 'default'}/bpb{bpb or 'auto'} {timeit(ob, outs):7.1f}")
    os.environ.pop("MPX_BPB", None)
    print(" | ".join(line), flush=True)
    if k % 3 == 1:
        hold.append(outs)
    del outs
    torch.cuda.empty_cache()

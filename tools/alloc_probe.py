"""Does the node kernel's speed depend on how its output buffers were allocated?  (B sweep showed 0.58 -> 0.81 of peak from
B=8192 to B=32768.)  Config 2; variants: plain torch.empty per array vs slices of one big arena; correctness spot checks."""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
import problems  # noqa: E402

dev = torch.device("cuda", 0)
mpo = mp.mpopt(problems.moon_lander(mp, M.math), 1000, 5, "LGR")
o = mpo.create_nlp()[0]["oracle"]
p = torch.tensor(np.full(o.n_p, 1e-3), device=dev)
rng = np.random.default_rng(0)
z0 = mpo.initialize_solution()


def run(B, arena_gb=0, tag="", keep=False):
    Z = torch.tensor(z0[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (min(B, 64), o.n_z))), device=dev).repeat((B + 63) // 64, 1)[:B].contiguous()
    sizes = [B, B * o.n_g, B * o.n_z, B * o.nnz_jac]
    if arena_gb:
        arena = torch.empty(int(arena_gb * 2 ** 30 // 8), dtype=torch.float64, device=dev)
        outs, off = [], 0
        for s in sizes:
            outs.append(arena[off:off + s])
            off += (s + 511) // 512 * 512
    else:
        outs = [torch.empty(s, dtype=torch.float64, device=dev) for s in sizes]
    f, g, gr, jv = outs
    for _ in range(3):
        o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
    o.sync()
    o.profile(True)
    for _ in range(10):
        o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
    ms, n = o.profile_read()
    o.profile(False)
    us = ms / n * 1e3
    # spot check: batch entries equal a single evaluation of the same point, bit for bit
    ok = True
    for b in (0, B // 3, B - 1):
        one = o.eval(["g", "jac_g"], Z[b].cpu().numpy(), p.cpu().numpy())
        ok &= np.array_equal(one["jac_g"], jv.view(B, -1)[b].cpu().numpy()) and np.array_equal(one["g"], g.view(B, -1)[b].cpu().numpy())
    print("   ptrs", " ".join(f"{t.data_ptr():#x}" for t in (Z, f, g, gr, jv)), flush=True)
    print(f"B={B:6d} {tag:22s} node kernel {us:9.1f} us  {B * o.bytes_fgj / us / 1e6:5.2f} TB/s  frac {B * o.bytes_fgj / us / 1e6 / 8:.3f}  correct={ok}", flush=True)
    if not keep:
        del outs, f, g, gr, jv
        torch.cuda.empty_cache()


for k in range(10):
    run(4096, 0, f"torch.empty #{k}", keep=(k % 3 == 1))  # keeping some alive shifts the later allocations

"""Does the KIND of allocation of the output arrays move the headline kernel?  torch (caching allocator -> hipMalloc), raw hipMalloc,
hipExtMallocWithFlags(contiguous / uncached / fine-grained): three fresh allocations of g, grad_f, jac_g each, node-kernel time of
config 2 at B = 4096 on every one (z stays where it is)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
import problems  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
B = 4096
mpo = mp.mpopt(problems.moon_lander(mp, M.math), 1000, 5, "LGR")
o = mpo.create_nlp()[0]["oracle"]
p = torch.tensor(np.full(o.n_p, 1e-3), device=dev)
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z)))).to(dev)
f = torch.empty(B, dtype=torch.float64, device=dev)


class Raw:
    def __init__(self, n, flags):
        self.ptr = ctypes.c_void_p()
        rc = hip.hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(n * 8)) if flags is None else \
            hip.hipExtMallocWithFlags(ctypes.byref(self.ptr), ctypes.c_size_t(n * 8), ctypes.c_uint(flags))
        if rc:
            raise RuntimeError(f"allocation failed: hip error {rc}")
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (self.ptr.value, False), "version": 2}

    def free(self):
        hip.hipFree(self.ptr)


def measure(g, gr, jv):
    for _ in range(3):
        o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
    o.sync()
    o.profile(True)
    for _ in range(10):
        o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
    ms, n = o.profile_read()
    o.profile(False)
    return ms / n * 1e3


sizes = (B * o.n_g, B * o.n_z, B * o.nnz_jac)
for rnd in range(2):
    for kind, flags in (("torch", "torch"), ("hipMalloc", None), ("contiguous", 0x4), ("uncached", 0x3), ("fine-grained", 0x1)):
        times = []
        for k in range(3):
            try:
                if flags == "torch":
                    raws, ts = [], [torch.empty(n, dtype=torch.float64, device=dev) for n in sizes]
                else:
                    raws = [Raw(n, flags) for n in sizes]
                    ts = [torch.as_tensor(r, device=dev) for r in raws]
                times.append(round(measure(*ts), 1))
                del ts
                for r in raws:
                    r.free()
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                times.append(repr(e)[:60])
        print(f"round {rnd} {kind:13s} {times}", flush=True)

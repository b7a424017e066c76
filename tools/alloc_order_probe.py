"""Is the speed of an output placement a property of WHERE in the device memory it lies?  Allocation order experiments on the
headline kernel (config 2, B = 4096): node-kernel us per pass into sets of output arrays allocated in different orders."""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import mpopt_amd as M  # noqa: E402
from mpopt_amd import mp  # noqa: E402
import problems  # noqa: E402

dev = torch.device("cuda", 0)
B = 4096
mpo = mp.mpopt(problems.moon_lander(mp, M.math), 1000, 5, "LGR")
o = mpo.create_nlp()[0]["oracle"]
p = torch.tensor(np.full(o.n_p, 1e-3), device=dev)
rng = np.random.default_rng(0)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z)))).to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"


def new_set():
    return [torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B, o.n_g, dtype=torch.float64, device=dev),
            torch.empty(B, o.n_z, dtype=torch.float64, device=dev), torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)]


def measure(s):
    o.geometry_reset()
    for _ in range(6):
        o.eval_device(15, B, Z, p, 0, None, None, *s, None)
    o.sync()
    o.profile(True)
    for _ in range(8):
        o.eval_device(15, B, Z, p, 0, None, None, *s, None)
    ms, n = o.profile_read()
    o.profile(False)
    return round(ms / 8 * 1e3, 1), hex(s[3].data_ptr())


if mode == "dummy_first":
    dummy = torch.empty(int(sys.argv[2]) << 27, dtype=torch.float64, device=dev)  # GB held before anything else
    print("held", dummy.numel() * 8 / 2**30, "GB at", hex(dummy.data_ptr()))
sets = []
for k in range(5):
    sets.append(new_set())
    print("allocate set", k, measure(sets[-1]), flush=True)
print("again set 0", measure(sets[0]), "set 3", measure(sets[3]))
sets[0] = None
torch.cuda.empty_cache()
sets[0] = new_set()
print("set 0 freed and allocated again", measure(sets[0]))
sets[2] = None
torch.cuda.empty_cache()
sets[2] = new_set()
print("set 2 freed and allocated again", measure(sets[2]))

"""Where a request of the resident kernel spends its time (code object built with -DMPX_RES_STAMPS, MPX_RES_DEBUG=1)."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import mpopt_amd as M
from mpopt_amd import mp
import problems
for case in ((problems.moon_lander, 20, 3, "LGR"), problems.BENCH_CASES[0]):
    builder, S, P, scheme = case
    mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    z, p = mpo.initialize_solution(), np.full(o.n_p, 1.0 / S)
    lam = np.ones(o.n_g)
    print(builder.__name__, S, flush=True)
    for what in (["f", "g", "grad_f"], ["hess_l"], ["jac_g"]):
        for _ in range(1100):
            o.eval(what, z, p, lam_g=lam, sigma=1.0, pinned=True)
    o.close()

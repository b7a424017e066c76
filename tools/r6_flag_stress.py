"""Stress of the single-evaluation hand-off (round 6): B = 1 through host pointers -- the kernels read z from and write every output into the
caller's page-locked arrays over PCIe and raise a completion flag the host spins on (mpx_host.cpp) -- N times in a row on changing points, every
result compared bit for bit with the same point of ONE batched evaluation (copies in and out, stream synchronisation).  An ordering bug between
the output writes and the flag would show as a stale value once in many thousand calls.    python tools/r6_flag_stress.py [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import mpopt_amd as M
from mpopt_amd import mp
import problems

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for builder, S, P, scheme, npts in ((problems.moon_lander, 20, 3, "LGR", 1024), (problems.kitchen_sink, 6, 4, "LGR", 512), (problems.moon_lander, 1000, 5, "LGR", 64),
                                     (problems.van_der_pol, 1, 25, "LGR", 512)):
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    rng = np.random.default_rng(3)
    Z = mpo.initialize_solution()[None, :] + 0.05 * rng.standard_normal((npts, o.n_z))
    p = np.full(o.n_p, 1.0 / S)
    lam, sig = rng.standard_normal((npts, o.n_g)), rng.uniform(0.5, 1.5, npts)
    ref = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], Z, p, lam_g=lam, sigma=sig)
    n_calls, bad, t0 = 0, 0, time.time()
    for r in range(rounds):
        order = rng.permutation(npts)
        for i in order:
            a = o.eval(["f", "g", "grad_f", "jac_g"], Z[i], p)
            h = o.eval(["hess_l"], Z[i], p, lam_g=lam[i], sigma=sig[i])
            n_calls += 2
            if not (np.array_equal(a["g"], ref["g"][i]) and np.array_equal(a["jac_g"], ref["jac_g"][i]) and np.array_equal(a["grad_f"], ref["grad_f"][i])
                    and a["f"] == ref["f"][i] and np.array_equal(h["hess_l"], ref["hess_l"][i])):
                bad += 1
                if bad < 5:
                    print("MISMATCH", builder.__name__, S, P, "round", r, "point", i, flush=True)
    print(f"{builder.__name__} {S}x{P} {scheme}: {n_calls} single evaluations against one batch of {npts}: {bad} mismatches, {1e6 * (time.time() - t0) / n_calls:.1f} us per call incl. Python", flush=True)
    assert bad == 0
    o.close()
print("flag stress: ok")

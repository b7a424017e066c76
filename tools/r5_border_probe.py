"""Whose rounding is it?  The (t0, tf, a) border entries of jac_g / hess_l of the explicitly time-dependent problem differ between
the GPU and the C oracle by up to 6e-11 of the class floor (GPUTEST r4).  This probe evaluates the same points three ways -- GPU,
oracle/mpopt_oracle.c (binary64), the same source in 80-bit long double (the arbiter) -- and prints, per entry class, the worst
|gpu - ld|, |c - ld|, |gpu - c| over the class floor, and the ten worst border entries with their node, column and values.
    python tools/r5_border_probe.py [case ...]        (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scipy.sparse as sp

import mpopt_amd as M
from mpopt_amd import mp
from helpers import border_columns, entry_errors, hess_classes, jac_classes
from oracle.c_oracle import COracle
import test_gpu_parity as T

cases = sys.argv[1:] or ["time_dependent_2000_mixed_CGL", "time_dependent_4000x3_LGR"]
for name in cases:
    (builder, S, po, scheme), cnames, st, midu = T.FULL[name]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, po, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    kw = dict(scale_x=ocp.scale_x, scale_u=ocp.scale_u, scale_a=ocp.scale_a if ocp.na else None, scale_t=st, midu=midu)
    C, L = COracle(cnames, S, po, scheme, **kw), COracle(cnames, S, po, scheme, long_double=True, **kw)
    z, p, lam, sig = T.random_point(o, mpo, None, 23, S, ocp.n_phases)
    Z = np.stack([z, mpo.initialize_solution(), z * 1.01])
    r = o.eval(["f", "g", "grad_f", "jac_g", "hess_l"], Z, p, lam_g=lam, sigma=sig)
    jr, jc = o.jac_pattern()
    hr, hc = o.hess_pattern()
    N, nx, nu = o.n_nodes, ocp.nx, ocp.nu
    for b in range(3):
        al = lambda c: np.asarray(sp.coo_matrix((c["jac_val"], (c["jac_row"], c["jac_col"])), shape=(o.n_g, o.n_z)).tocsr()[jr, jc]).ravel()
        cj, lj, cj2 = al(C.eval(Z[b], p)), al(L.eval(Z[b], p)), al(C.eval(Z[(b + 2) % 3], p))
        ch = np.asarray(C.hess_matrix(Z[b], p, sig, lam)[hr, hc]).ravel()
        lh = np.asarray(L.hess_matrix(Z[b], p, sig, lam)[hr, hc]).ravel()
        for what, g_, c_, l_, cls, rows, cols in (("jac_g", r["jac_g"][b], cj, lj, jac_classes(o, jr, jc, cj, cj2), jr, jc),
                                                  ("hess_l", r["hess_l"][b], ch, lh, hess_classes(o, hr, hc), hr, hc)):
            for cn, m in cls.items():
                if not m.any():
                    continue
                _, fl = entry_errors(c_[m], c_[m])
                den = np.maximum(np.abs(l_[m]), fl)
                e_gl, e_cl, e_gc = np.abs(g_[m] - l_[m]) / den, np.abs(c_[m] - l_[m]) / den, np.abs(g_[m] - c_[m]) / den
                print(f"{name}[{b}] {what} [{cn}]: gpu-ld {e_gl.max():.2e}  c-ld {e_cl.max():.2e}  gpu-c {e_gc.max():.2e}  (floor {fl:.2e}, n={m.sum()})")
                if "order" in cn and b == 0:
                    idx = np.flatnonzero(m)[np.argsort(-e_gl)[:10]]
                    for k in idx:
                        rr, cc = int(rows[k]), int(cols[k])
                        print(f"    entry ({rr}, {cc})  row%N={rr % N} row//N={rr // N} col%N={cc % N} col//N={cc // N}: gpu {g_[k]!r} c {c_[k]!r} ld {l_[k]!r}")
    o.close()

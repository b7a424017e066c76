import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
ocp=problems.moon_lander(mp,M.math); mpo=mp.mpopt(ocp,1000,5,"LGR"); nlp,b=mpo.create_nlp(); o=nlp['oracle']
z=mpo.initialize_solution(); p=np.full(o.n_p,1e-3); lam=np.ones(o.n_g)
for what in (["f"],["g"],["f","grad_f"],["g","jac_g"],["hess_l"],["f","g","grad_f","jac_g"]):
    for pinned in (False, True):
        for _ in range(10): r=o.eval(what,z,p,lam_g=lam,sigma=1.0,pinned=pinned)
        t=time.perf_counter()
        for _ in range(200): r=o.eval(what,z,p,lam_g=lam,sigma=1.0,pinned=pinned)
        print(what, 'pinned' if pinned else 'pageable', f'{(time.perf_counter()-t)/200*1e6:.1f} us')
r1=o.eval(["g","jac_g"],z,p); r2=o.eval(["g","jac_g"],z,p,pinned=True)
print('equal', all(np.array_equal(r1[k],r2[k]) for k in r1))

# the CasADi-convention entry points (what nlpsol would call): includes the conversion to compressed-column order
import ctypes
from mpopt_amd import _lib
o.make_current()
L = _lib.lib()
jv, hv, g = np.zeros(o.nnz_jac), np.zeros(o.nnz_hess), np.zeros(o.n_g)
sig = np.array([1.0])


def call(fn, ins, outs, n=200):
    arg = (ctypes.c_void_p * len(ins))(*[a.ctypes.data for a in ins])
    res = (ctypes.c_void_p * len(outs))(*[a.ctypes.data for a in outs])
    f = getattr(L, fn)
    for _ in range(10):
        f(arg, res, None, None, 0)
    t = time.perf_counter()
    for _ in range(n):
        f(arg, res, None, None, 0)
    return (time.perf_counter() - t) / n * 1e6


print("nlp_jac_g (C entry point, CCS order)", f'{call("nlp_jac_g", [z, p], [g, jv]):.1f} us')
print("nlp_hess_l (C entry point, CCS order)", f'{call("nlp_hess_l", [z, p, sig, lam], [hv]):.1f} us')
L.mpx_current_pin_buffers(1)
print("nlp_jac_g, caller buffers page-locked on first sight", f'{call("nlp_jac_g", [z, p], [g, jv]):.1f} us')
print("nlp_hess_l, caller buffers page-locked on first sight", f'{call("nlp_hess_l", [z, p, sig, lam], [hv]):.1f} us')
L.mpx_current_pin_buffers(0)
o.close()

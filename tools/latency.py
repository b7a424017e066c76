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

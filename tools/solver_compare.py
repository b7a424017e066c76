"""Scratch: the stand-in NLP drivers side by side on the test problems (SI=ipm,trust-constr,auto; WS=1 warm-starts mpopt_adaptive;
PL=1 iteration log; arguments filter the case names).  Every oracle value comes from the GPU."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import mpopt_amd as M
from mpopt_amd import mp
import problems
mp.mpopt._MUTE_=True
cases=[("moon 20x3",problems.moon_lander,20,3,"LGR",mp.mpopt),("vdp 1x15 LGR",problems.van_der_pol,1,15,"LGR",mp.mpopt),("vdp 4x3 CGL",problems.van_der_pol,4,3,"CGL",mp.mpopt),
       ("schwartz 1x15",problems.two_phase_schwartz,1,15,"LGL",mp.mpopt),("schwartz 4x3",problems.two_phase_schwartz,4,3,"LGL",mp.mpopt),("hyper 15x15",problems.hyper_sensitive,15,15,"LGR",mp.mpopt),
       ("hyper 5x3",problems.hyper_sensitive,5,3,"LGR",mp.mpopt),("analytic",problems.analytic_solution,1,5,"LGR",mp.mpopt),("moon adaptive 3x2",problems.moon_lander,3,[2]*3,"LGR",mp.mpopt_adaptive),("moon adaptive 3x3",problems.moon_lander,3,[3]*3,"LGR",mp.mpopt_adaptive),("hyper adaptive 5x4",problems.hyper_sensitive,5,[4]*5,"LGR",mp.mpopt_adaptive),("vdp adaptive 3x4",problems.van_der_pol,3,[4]*3,"LGR",mp.mpopt_adaptive),("kitchen 3x3",problems.kitchen_sink,3,3,"LGR",mp.mpopt),("dae 3x4",problems.dae_vdp,3,4,"CGL",mp.mpopt),("generic2",problems.generic_two_phase,2,[2,3],"LGR",mp.mpopt)]
only=sys.argv[1:]
for name,b,S,P,sc,cls in cases:
    if only and not any(o in name for o in only): continue
    for standin in tuple(__import__("os").environ.get("SI","ipm,trust-constr").split(",")):
        mpo=cls(b(mp,M.math),S,P,sc)
        t=time.time()
        kw={"mpopt_options":{"warm_start_fixed_width":bool(int(__import__("os").environ.get("WS","0")))}} if cls is mp.mpopt_adaptive else {}
        try:
            sol=mpo.solve(nlp_solver_options={"standin":standin,"ipopt.print_level":int(__import__("os").environ.get("PL","0")),"ipopt.max_iter":int(__import__("os").environ.get("MI","2000"))}, **kw); st=mpo.nlp_solver.stats
            x=np.asarray(sol["x"]).ravel()
            g=mpo.oracle.eval(["g"],x,getattr(mpo,"_nlp_sw_params",None) if cls is mp.mpopt else None)["g"]
            viol=max(np.maximum(mpo.Gmin-g,0).max(),np.maximum(g-mpo.Gmax,0).max(),np.maximum(mpo.Zmin-x,0).max(),np.maximum(x-mpo.Zmax,0).max())
            print(f"{name:18s} {standin:12s} f={float(sol['f']): .8f} it={st['iter_count']:5d} ok={st['success']} {st['return_status'][:34]:34s} viol={viol:.1e} {time.time()-t:.2f}s",flush=True)
        except Exception as e:
            print(name,standin,"EXC",repr(e)[:200],flush=True)

"""Compatibility sweep (build container only: reads /root/reference/examples): every example script of the reference is executed
with ``from mpopt import mp`` bound to mpopt_amd.mp and ``import casadi as ca`` bound to the mpopt_amd.math spellings, up to the
point where it creates an optimizer; the OCP it defined is then validated and traced (structure-only context, small grid).
Nothing of the reference is copied: the scripts are read and executed where they lie."""
import sys, glob, re, os, traceback, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MPLBACKEND"]="Agg"
import numpy as np
import mpopt_amd as M
from mpopt_amd import mp
import matplotlib; matplotlib.use("Agg")
# a stand-in module for `import casadi as ca` providing the math spellings; `from mpopt import mp` -> our mp
ca = types.ModuleType("casadi")
for k in dir(M.math):
    if not k.startswith("_"): setattr(ca, k, getattr(M.math, k))
sys.modules["casadi"] = ca
mpopt_pkg = types.ModuleType("mpopt"); mpopt_pkg.mp = mp; sys.modules["mpopt"] = mpopt_pkg; sys.modules["mpopt.mp"] = mp
ctx = types.ModuleType("context"); ctx.mpopt = mpopt_pkg; sys.modules["context"] = ctx
results = {}
class Stop(Exception): pass
REF = os.environ.get('MPX_REFERENCE_DIR', '/root/reference')
for f in sorted(glob.glob(REF + '/examples/**/*.py', recursive=True)):
    name = f.split('/examples/')[1]
    src = open(f).read()
    created = []
    # intercept the optimizer constructors: record (ocp, grid) and stop before any solve
    def make(cls_name):
        def ctor(ocp, *a, **k):
            created.append((cls_name, ocp, a, k)); raise Stop()
        return ctor
    mp2 = types.ModuleType("mp2")
    for k in dir(mp):
        setattr(mp2, k, getattr(mp, k))
    for c in ("mpopt","mpopt_h_adaptive","mpopt_adaptive","mpopt_ph_adaptive"): setattr(mp2, c, make(c))
    def solve_stub(ocp, *a, **k): created.append(("solve", ocp, a, k)); raise Stop()
    mp2.solve = solve_stub
    mpopt_pkg.mp = mp2; sys.modules["mpopt.mp"] = mp2
    g = {"__name__": "__main__", "__file__": f}
    try:
        exec(compile(src, f, "exec"), g)
        results[name] = "ran to the end without creating an optimizer"
        continue
    except Stop:
        pass
    except BaseException as e:
        results[name] = f"definition failed: {type(e).__name__}: {str(e)[:90]}"
        continue
    kind, ocp, a, k = created[0]
    try:
        S = k.get("n_segments", a[0] if len(a) > 0 else 1)
        P = k.get("poly_orders", a[1] if len(a) > 1 else 9)
        sc = k.get("scheme", a[2] if len(a) > 2 else "LGR")
        S = min(int(S), 3)
        P = ([min(int(p), 4) for p in np.atleast_1d(P)][:S] + [3] * S)[:S] if not isinstance(P, (int, np.integer)) else [min(int(P), 4)] * S
        ocp.validate()
        o = M.NlpFunctions(ocp, S, P, sc if sc in ("LGR","LGL","CGL") else "LGR", with_device=False)
        results[name] = f"OK ({kind}; phases {ocp.n_phases}, nx {ocp.nx}, nu {ocp.nu}, na {ocp.na}; nnz_jac {o.nnz_jac} on a {S}x{P[0]} grid)"
    except BaseException as e:
        results[name] = f"trace failed: {type(e).__name__}: {str(e)[:120]}"
if __name__ == "__main__":
    for k, v in results.items():
        print(f"{k:62s} {v}")
    print(sum(v.startswith("OK") for v in results.values()), "of", len(results), "trace")

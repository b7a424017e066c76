"""Single evaluations through host pointers (the regime IPOPT drives): latency of the light passes with the span kernels (default)
and with the node kernels (MPX_NO_LIGHT=1), configs 2 and 3.  python tools/r4_single_eval_light.py"""
import os, sys, time
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import mpopt_amd as M
from mpopt_amd import mp
import problems
for case in (0, 1, 3):
    builder, S, P, scheme = problems.BENCH_CASES[case]
    mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    z = mpo.initialize_solution(); p = np.full(o.n_p, 1.0 / S)
    print(builder.__name__, S, "plan", o.light_plan())
    for what in (["f"], ["g"], ["f", "g", "grad_f"]):
        row = []
        for env in (None, "1"):
            if env: os.environ["MPX_NO_LIGHT"] = env
            else: os.environ.pop("MPX_NO_LIGHT", None)
            for _ in range(30): o.eval(what, z, p, pinned=True)
            ts = []
            for _ in range(7):
                t = time.perf_counter()
                for _ in range(100): o.eval(what, z, p, pinned=True)
                ts.append((time.perf_counter() - t) / 100 * 1e6)
            row.append(sorted(ts)[3])
        os.environ.pop("MPX_NO_LIGHT", None)
        print(f"  {'+'.join(what):12s} span kernels {row[0]:7.1f} us   node kernels {row[1]:7.1f} us")
    o.close()

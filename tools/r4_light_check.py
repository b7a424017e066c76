import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
for case in (problems.BENCH_CASES[1], (problems.kitchen_sink, 12, [20, 3, 20, 5] * 3, "LGR"), (problems.dae_vdp, 9, 17, "LGL")):
    builder, S, P, scheme = case
    mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    rng = np.random.default_rng(0)
    for B in (1, 5, 16, 37):
        Z = mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z))) + 0.01 * rng.uniform(-1, 1, (B, o.n_z))
        w = rng.uniform(0.5, 1.5, (mpo._ocp.n_phases, S)); p = (w / w.sum(1, keepdims=True)).ravel()
        res = {}
        for tag, env in (("light", None), ("node", "1")):
            if env: os.environ["MPX_NO_LIGHT"] = env
            else: os.environ.pop("MPX_NO_LIGHT", None)
            res[tag] = (o.eval(["f", "g"], Z, p), o.eval(["g"], Z, p), o.eval(["f", "grad_f"], Z, p), o.eval(["f", "g", "grad_f"], Z, p), o.eval(["f"], Z, p))
        os.environ.pop("MPX_NO_LIGHT", None)
        a, b = res["light"], res["node"]
        nzp = o.n_z // mpo._ocp.n_phases
        node = np.ones(o.n_z, bool)
        for ph in range(mpo._ocp.n_phases):  # the (t0, tf, a) entries of grad_f are sums over all nodes: they round like f
            node[ph * nzp + (mpo._ocp.nx + mpo._ocp.nu) * o.n_nodes:(ph + 1) * nzp] = False
        for r_ in (a, b):
            for d in r_:
                if "grad_f" in d:
                    d["grad_f"] = d["grad_f"][..., node]
        print(builder.__name__, "B", B, "g bit-equal", np.array_equal(a[0]["g"], b[0]["g"]), np.array_equal(a[1]["g"], b[1]["g"]), np.array_equal(a[3]["g"], b[3]["g"]),
              "grad_f bit-equal", np.array_equal(a[2]["grad_f"], b[2]["grad_f"]), np.array_equal(a[3]["grad_f"], b[3]["grad_f"]),
              "f rel diff", float(np.abs(a[0]["f"] - b[0]["f"]).max() / np.abs(b[0]["f"]).max()), float(np.abs(a[2]["f"] - b[2]["f"]).max()), float(np.abs(a[4]["f"] - b[4]["f"]).max()),
              "light f consistent", np.array_equal(a[0]["f"], a[4]["f"]), np.array_equal(a[0]["f"], a[3]["f"]))
    o.close()

"""Low-degree light kernels (mpx_lightlow_*) against the node kernels (MPX_NO_LIGHT=1): bit equality of g / node grad_f, and the
pass times of both at B=4096 (BASELINE configs 1, 2, 4, 5).  Usage: python tools/r4_lightlow_check.py [time]"""
import os, sys, time
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
cases = [(problems.moon_lander, 20, 3, "LGR"), (problems.moon_lander, 1000, 5, "LGR"), (problems.two_phase_schwartz, 500, 3, "LGL"),
         (problems.hyper_sensitive, 4000, 3, "LGR"), (problems.kitchen_sink, 40, 5, "LGR"), (problems.dae_vdp, 37, 12, "CGL"), (problems.time_dependent, 400, 3, "LGR")]
for case in cases:
    builder, S, P, scheme = case
    mpo = mp.mpopt(builder(mp, M.math), S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    print(builder.__name__, S, P, "plan", o.light_plan(), "notes", o.notes())
    rng = np.random.default_rng(0)
    for B in (1, 5, 37):
        Z = mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z))) + 0.01 * rng.uniform(-1, 1, (B, o.n_z))
        w = rng.uniform(0.5, 1.5, (mpo._ocp.n_phases, S)); p = (w / w.sum(1, keepdims=True)).ravel()
        res = {}
        for tag, env in (("light", None), ("node", "1")):
            if env: os.environ["MPX_NO_LIGHT"] = env
            else: os.environ.pop("MPX_NO_LIGHT", None)
            res[tag] = (o.eval(["f", "g"], Z, p), o.eval(["g"], Z, p), o.eval(["f", "grad_f"], Z, p), o.eval(["f", "g", "grad_f"], Z, p), o.eval(["f"], Z, p))
        os.environ.pop("MPX_NO_LIGHT", None)
        a, b = res["light"], res["node"]
        node = np.ones(o.n_z, bool)
        from helpers import border_columns
        node[border_columns(o)] = False
        print("  B", B, "g bit-equal", np.array_equal(a[0]["g"], b[0]["g"]), np.array_equal(a[1]["g"], b[1]["g"]), np.array_equal(a[3]["g"], b[3]["g"]),
              "grad_f(node) bit-equal", np.array_equal(a[2]["grad_f"][..., node], b[2]["grad_f"][..., node]), np.array_equal(a[3]["grad_f"][..., node], b[3]["grad_f"][..., node]),
              "border rel", float(np.abs(a[3]["grad_f"][..., ~node] - b[3]["grad_f"][..., ~node]).max() / max(1e-300, np.abs(b[3]["grad_f"][..., ~node]).max())),
              "f rel diff", float(np.abs(a[0]["f"] - b[0]["f"]).max() / np.abs(b[0]["f"]).max()),
              "light f consistent", np.array_equal(a[0]["f"], a[4]["f"]), np.array_equal(a[0]["f"], a[3]["f"]), np.array_equal(a[0]["f"], a[2]["f"]))
    if len(sys.argv) > 1 and S >= 400:
        B = 4096
        dev = torch.device("cuda:0")
        Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
        pt = torch.tensor(p, device=dev)
        f = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty((B, o.n_g), dtype=torch.float64, device=dev); q = torch.empty((B, o.n_z), dtype=torch.float64, device=dev)
        for tag, env in (("light", None), ("node", "1")):
            if env: os.environ["MPX_NO_LIGHT"] = env
            else: os.environ.pop("MPX_NO_LIGHT", None)
            MK = {"f": 1, "g": 2, "grad_f": 4}
            for name, kw in (("f", dict(f=f)), ("g", dict(g=g)), ("f+g", dict(f=f, g=g)), ("f+grad_f", dict(f=f, grad_f=q))):
                mask = sum(MK[k] for k in kw)
                for _ in range(12): o.eval_device(mask, B, Z, pt, **kw)
                o.sync(); t0 = time.perf_counter()
                for _ in range(20): o.eval_device(mask, B, Z, pt, **kw)
                o.sync(); dt = (time.perf_counter() - t0) / 20
                byt = 8 * B * (o.n_z + (o.n_g if "g" in kw else 0) + (o.n_z if "grad_f" in kw else 0))
                print(f"  {tag:5s} {name:9s} {dt * 1e6:8.1f} us  {byt / dt / 1e12:.2f} TB/s")
        os.environ.pop("MPX_NO_LIGHT", None)
    o.close()

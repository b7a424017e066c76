"""One-off soak of the light-pass kernels of single-degree grids (mpx_lightlow_* / mpx_lightlows_*): random and edge-case grids
(degree 1 ... 12, node counts around the chunk / span boundaries), span kernels against the node kernels (MPX_NO_LIGHT=1): g and the
node entries of grad_f bit for bit, f to rounding; short spans against long spans bit for bit; a batch against its single evaluations.
python tools/r4_light_soak.py [seed] [n_random]"""
import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import mpopt_amd as M
from mpopt_amd import mp
import problems
from helpers import border_columns
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_random = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(seed)
builders = [problems.van_der_pol, problems.moon_lander, problems.dae_vdp, problems.time_dependent, problems.hyper_sensitive, problems.two_phase_schwartz]
cases = []
for P in (1, 2, 3, 5, 8, 12):  # node counts N = S P + 1 at and around 64, 512 / 768 (span lengths) and their multiples
    for N_target in (64, 65, 513, 769, 1537):
        S = max(1, (N_target - 1 + P - 1) // P)
        cases.append((builders[(P + N_target) % len(builders)], S, P))
for _ in range(n_random):
    cases.append((builders[int(rng.integers(len(builders)))], int(rng.integers(1, 400)), int(rng.integers(1, 13))))
bad = 0
for k, (builder, S, P) in enumerate(cases):
    scheme = ("LGR", "LGL", "CGL")[k % 3]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, P, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    plan = o.light_plan()
    if plan[0] == 0:
        print(f"[{k}] {builder.__name__} S={S} P={P} {scheme}: no plan (node kernels)"); o.close(); continue
    node = np.ones(o.n_z, bool); node[border_columns(o)] = False
    B = max(2, -(-1100 // plan[1])) if k % 2 else 5  # every second case large enough for the long spans
    Z = mpo.initialize_solution()[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))) + 0.02 * rng.uniform(-1, 1, (B, o.n_z))
    w = rng.uniform(0.5, 1.5, (ocp.n_phases, S)); p = (w / w.sum(1, keepdims=True)).ravel()
    a = o.eval(["f", "g", "grad_f"], Z, p)
    os.environ["MPX_NO_LIGHT"] = "1"; b = o.eval(["f", "g", "grad_f"], Z, p); del os.environ["MPX_NO_LIGHT"]
    os.environ["MPX_LIGHT_LONG_SPANS"] = "1"; c = o.eval(["f", "g", "grad_f"], Z[:2], p); del os.environ["MPX_LIGHT_LONG_SPANS"]
    one = o.eval(["f", "g", "grad_f"], Z[1], p)
    ok = (np.array_equal(a["g"], b["g"]) and np.array_equal(a["grad_f"][:, node], b["grad_f"][:, node])
          and np.abs(a["f"] - b["f"]).max() <= 1e-12 * max(1.0, np.abs(b["f"]).max())
          and np.abs(a["grad_f"][:, ~node] - b["grad_f"][:, ~node]).max() <= 1e-11 * max(1.0, np.abs(b["grad_f"][:, ~node]).max())
          and all(np.array_equal(c[q], a[q][:2]) for q in ("f", "g", "grad_f")) and all(np.array_equal(np.asarray(one[q]), a[q][1]) for q in ("f", "g", "grad_f")))
    bad += not ok
    print(f"[{k}] {builder.__name__} S={S} P={P} {scheme} N={o.n_nodes} plan={plan} B={B}: {'ok' if ok else 'MISMATCH'}", flush=True)
    o.close()
print(f"{len(cases)} grids, {bad} mismatches")

"""One-off soak of the matrix-core light kernels (mpx_light_*): random grids with ONE degree in 13 ... 31 and up to six degrees <= 12 in
random runs, span kernels against the node kernels (MPX_NO_LIGHT=1): g and the node entries of grad_f bit for bit, f / border entries
to rounding; a batch against its single evaluations bit for bit.  python tools/r4_light_mfma_soak.py [seed] [n]"""
import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import mpopt_amd as M
from mpopt_amd import mp
import problems
from helpers import border_columns
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
builders = [problems.van_der_pol, problems.dae_vdp, problems.time_dependent, problems.kitchen_sink, problems.moon_lander]
bad = planned = 0
for k in range(n):
    builder = builders[k % len(builders)]
    hi = int(rng.integers(13, 32))
    lows = [int(x) for x in rng.choice(np.arange(1, 13), size=int(rng.integers(0, 5)), replace=False)]
    S = int(rng.integers(2, 160))
    orders = []
    while len(orders) < S:
        d = hi if (not lows or rng.random() < 0.4) else int(rng.choice(lows))
        orders += [d] * int(rng.integers(1, 6))
    orders = orders[:S]
    if hi not in orders:
        orders[int(rng.integers(S))] = hi
    scheme = ("LGR", "LGL", "CGL")[k % 3]
    ocp = builder(mp, M.math)
    mpo = mp.mpopt(ocp, S, orders, scheme)
    o = mpo.create_nlp()[0]["oracle"]
    plan = o.light_plan()
    if plan[0] <= 12:
        print(f"[{k}] {builder.__name__} S={S} hi={hi} lows={sorted(set(orders) - {hi})}: plan {plan} (not the matrix-core kernels)"); o.close(); continue
    planned += 1
    node = np.ones(o.n_z, bool); node[border_columns(o)] = False
    B = int(rng.choice([1, 3, 17, 40]))
    Z = mpo.initialize_solution()[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))) + 0.02 * rng.uniform(-1, 1, (B, o.n_z))
    w = rng.uniform(0.5, 1.5, (ocp.n_phases, S)); p = (w / w.sum(1, keepdims=True)).ravel()
    a = o.eval(["f", "g", "grad_f"], Z, p)
    os.environ["MPX_NO_LIGHT"] = "1"; b = o.eval(["f", "g", "grad_f"], Z, p); del os.environ["MPX_NO_LIGHT"]
    one = o.eval(["f", "g", "grad_f"], Z[B - 1], p)
    ok = (np.array_equal(a["g"], b["g"]) and np.array_equal(a["grad_f"][:, node], b["grad_f"][:, node])
          and np.abs(a["f"] - b["f"]).max() <= 1e-12 * max(1.0, np.abs(b["f"]).max())
          and np.abs(a["grad_f"][:, ~node] - b["grad_f"][:, ~node]).max() <= 1e-11 * max(1.0, np.abs(b["grad_f"][:, ~node]).max())
          and all(np.array_equal(np.asarray(one[q]), a[q][B - 1]) for q in ("f", "g", "grad_f")))
    bad += not ok
    print(f"[{k}] {builder.__name__} S={S} hi={hi} lows={sorted(set(orders) - {hi})} {scheme} N={o.n_nodes} plan={plan} B={B}: {'ok' if ok else 'MISMATCH'}", flush=True)
    o.close()
print(f"{n} grids, {planned} on the matrix-core kernels, {bad} mismatches")

"""config-5 loop (hypersensitive 4000x3, 5 outer iterations, B points): where do the three kernel boundaries of an iteration go?
(a) the bench's call sequence: hess_l + mid-point residuals (node pass, boundary pass) -> equal-area update;
(b) node pass -> equal-area update -> boundary pass (MPX_BOUNDARY_ONLY), ordered;
(c) as (b) with the boundary pass launched WITHOUT a barrier against the equal-area kernel (hipExtAnyOrderLaunch): the two are
    independent (the boundary pass reads the tile partials and writes corner entries of hess_val, the update reads the residuals and
    writes the widths), the next node pass waits for both.
Checks that every variant gives the same hess_val and widths bit for bit.   B=512 python tools/r5_loop_anyorder.py"""
import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems

builder, S, P, scheme = problems.BENCH_CASES[3]
B = int(os.environ.get("B", 512))
dev = torch.device("cuda", 0)
ocp = builder(mp, M.math)
mpo = mp.mpopt(ocp, S, P, scheme)
o = mpo.create_nlp()[0]["oracle"]
rng = np.random.default_rng(20260928)
Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.01 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev); sig = torch.ones(B, dtype=torch.float64, device=dev)
hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
mids = [(mpo.collocation._taus_fn(d)[:-1] + mpo.collocation._taus_fn(d)[1:]) / 2 for d in mpo.poly_orders]
plan = o.residual_plan(0, mids)
p0 = torch.tensor(rng.dirichlet(np.ones(S), B), device=dev)
pa, pb = torch.empty_like(p0), torch.empty_like(p0)
R = torch.empty(B, plan.n_pts, ocp.nx, dtype=torch.float64, device=dev)
o.set_mid_resid_output(R)
HESS, MID, UNCH, BONLY = 16, 1024, 256, 32


def loop(variant):
    pa.copy_(p0)
    cur, nxt = pa, pb
    for it in range(5):
        m = HESS | MID | (UNCH if it else 0)
        if variant == "a":
            o.eval_device(m, B, Z, cur, 1, lam, sig, None, None, None, None, hv)
            o.equal_area_widths_device(0, B, plan.n_pts, R, cur, nxt, damping=0.4, p_in_per_point=1)
        else:
            o.set_tile_range(0, o.n_tiles, run_boundary=False)
            o.eval_device(m, B, Z, cur, 1, lam, sig, None, None, None, None, hv)
            o.equal_area_widths_device(0, B, plan.n_pts, R, cur, nxt, damping=0.4, p_in_per_point=1)
            o.set_tile_range(0, o.n_tiles, run_boundary=True)
            if variant == "c":
                os.environ["MPX_BOUNDARY_ANYORDER"] = "1"
            o.eval_device(HESS | BONLY | UNCH, B, Z, cur, 1, lam, sig, None, None, None, None, hv)
            os.environ.pop("MPX_BOUNDARY_ANYORDER", None)
        cur, nxt = nxt, cur
    return cur


res, outs = {k: [] for k in "abc"}, {}
for rnd in range(7):
    for v in "abc":
        for _ in range(3): loop(v)
        o.sync(); o.timer_start()
        for _ in range(20): w = loop(v)
        res[v].append(o.timer_stop() / 20 * 1e3)
        if rnd == 0: outs[v] = (hv.clone(), w.clone())
for v, name in (("a", "bench sequence"), ("b", "boundary pass behind the update, ordered"), ("c", "boundary pass behind the update, any-order launch")):
    med = sorted(res[v])[3]
    same = all(torch.equal(x, y) for x, y in zip(outs[v], outs["a"]))
    print(f"B={B} [{name:50s}] median {med:8.1f} us per 5 iterations  min {min(res[v]):8.1f}   bit-equal to the bench sequence: {same}")

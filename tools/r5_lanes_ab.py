"""hess_l of an assembled context, three ways in ONE process on the same arrays (the host reads the switches per call): the
lane-per-evaluation-point kernel (mpx_asml_hes, round 5), the fused kernel (mpx_asm_hes, MPX_NO_LANES=1), the two-pass kernels
(MPX_NO_LANES=1 MPX_NO_FUSE=1).  Bit-equality of all three, median / min time per pass, fraction of the 8 TB/s peak by algorithmic bytes.
python tools/r5_lanes_ab.py [problem=moon_lander S=20 P=5] ; B="512 1024 4096 4133 16384"; FLAGS="-DX=1;-DY=2": more contexts whose code
objects are built with these MPX_HIPCC_FLAGS (lanes kernel only), e.g. the ablations -DMPX_LANE_ABL=1|2|4."""
import os, sys
os.environ.setdefault("MPX_ENV_DYNAMIC", "1")  # this tool switches libmpx's knobs inside one process (include/mpx.h: mpx_env_dynamic)
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
name = sys.argv[1] if len(sys.argv) > 1 else "moon_lander"
S, P = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20, 5)
mpo = mp.mpopt_adaptive(getattr(problems, name)(mp, M.math), S, P, "LGR")
o = mpo.create_nlp()[0]["oracle"]
FGJ = os.environ.get("PASS", "hes") == "fgj"  # PASS=fgj: the first-order pass (f, g, grad_f, jac_g together) instead of hess_l
pl = o.lanes_plan_fgj if FGJ else o.lanes_plan
print(f"{name} {S}x{P}: n_z {o.n_z} n_g {o.n_g} nnz_jac {o.nnz_jac} nnz_hess {o.nnz_hess}; lanes plan ({'fgj' if FGJ else 'hes'}): " + ("none" if pl is None else f"{len(pl.groups)} groups, {pl.n_tasks} tasks + {pl.halo_tasks} halo, tile rows {pl.tile_rows}, {len(pl.global_rows)} global rows / {len(pl.sid)} scratch slots") + f"; code object: {o.batched_plan()}")
dev = torch.device("cuda", 0)
VARIANTS = [("lanes", {}, o), ("fused", {"MPX_NO_LANES": "1"}, o), ("two-pass", {"MPX_NO_LANES": "1", "MPX_NO_FUSE": "1"}, o)]
for od in [x for x in os.environ.get("ORDERS", "").split() if x]:  # ORDERS="0 1": the lanes kernel with MPX_LANES_ORDER=...
    VARIANTS.append((f"lanes order {od}", {"MPX_LANES_ORDER": od}, o))
keep = []
for fl in [x for x in os.environ.get("FLAGS", "").split(";") if x]:
    if fl.startswith(("CHUNK=", "MIN_TASKS=")):  # (generation-time switches, not compiler flags: MPX_LANES_CHUNK, MPX_LANES_MIN_TASKS)
        key = "MPX_LANES_" + fl.split("=")[0]
        os.environ[key] = fl.split("=")[1]
        m2 = mp.mpopt_adaptive(getattr(problems, name)(mp, M.math), S, P, "LGR")
        keep.append(m2)
        VARIANTS.append(("lanes " + fl, {}, m2.create_nlp()[0]["oracle"]))
        VARIANTS[-1][2].batched_plan()
        print(fl, "->", len(VARIANTS[-1][2].lanes_plan.groups), "groups,", VARIANTS[-1][2].lanes_plan.halo_tasks, "halo tasks")
        del os.environ[key]
        continue
    os.environ["MPX_HIPCC_FLAGS"] = fl
    m2 = mp.mpopt_adaptive(getattr(problems, name)(mp, M.math), S, P, "LGR")
    keep.append(m2)
    VARIANTS.append(("lanes " + fl, {}, m2.create_nlp()[0]["oracle"]))
    VARIANTS[-1][2].batched_plan()  # (compiles and attaches the lane kernels while the flags are set)
os.environ.pop("MPX_HIPCC_FLAGS", None)
for B in [int(b) for b in os.environ.get("B", "512 1024 4096 4133 16384").split()]:
    rng = np.random.default_rng(B)
    Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
    lam = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev)
    sig = torch.tensor(rng.uniform(0.5, 1.5, B), device=dev)
    hv = torch.empty(B, (1 + o.n_g + o.n_z + o.nnz_jac) if FGJ else o.nnz_hess, dtype=torch.float64, device=dev)
    alg = 8 * (2 * o.n_z + o.n_g + o.nnz_jac + 1) if FGJ else 8 * (o.n_z + o.n_g + 1 + o.nnz_hess)
    if FGJ:  # (one allocation, four arrays: f | g | grad_f | jac_g)
        flat = hv.view(-1)
        of, og, oq, oj = flat[:B], flat[B:B + B * o.n_g].view(B, o.n_g), flat[B + B * o.n_g:B + B * (o.n_g + o.n_z)].view(B, o.n_z), flat[B + B * (o.n_g + o.n_z):].view(B, o.nnz_jac)
        call = lambda oc: oc.eval_device(15, B, Z, None, 0, None, None, of, og, oq, oj, None)
    else:
        call = lambda oc: oc.eval_device(16, B, Z, None, 0, lam, sig, None, None, None, None, hv)
    res, outs = {v[0]: [] for v in VARIANTS}, {}
    for rnd in range(5):
        for v, env, oc in VARIANTS:
            for k in ("MPX_NO_LANES", "MPX_NO_FUSE", "MPX_LANES_ORDER"):
                os.environ.pop(k, None)
            os.environ.update(env)
            hv.fill_(float("nan"))
            for _ in range(3): call(oc)
            oc.sync(); oc.timer_start()
            for _ in range(20): call(oc)
            res[v].append(oc.timer_stop() / 20 * 1e3)
            if rnd == 0: outs[v] = hv.clone()
    for v, _, _ in VARIANTS:
        same = torch.equal(outs[v], outs["two-pass"]) and not torch.isnan(outs[v]).any().item()
        med = sorted(res[v])[len(res[v]) // 2]
        print(f"B {B:6d} {v:28s} median {med:8.2f} us  min {min(res[v]):8.2f}  frac(median) {alg * B / med / 1e3 / 8e3:.3f}  bit-equal to two-pass: {same}")
if os.environ.get("ORDER_AB"):  # single evaluations (two-pass kernels) with the entries of hess_l group-major (lanes plan) and source-major (no plan)
    os.environ["MPX_NO_LANES_CODE"] = "1"
    m2 = mp.mpopt_adaptive(getattr(problems, name)(mp, M.math), S, P, "LGR")
    o2 = m2.create_nlp()[0]["oracle"]
    del os.environ["MPX_NO_LANES_CODE"]
    rng = np.random.default_rng(1)
    for B in (1, 16, 128):
        Z = torch.tensor(mpo.initialize_solution()[None, :] * (1 + 0.02 * rng.uniform(-1, 1, (B, o.n_z))), device=dev)
        lam, sig = torch.tensor(rng.standard_normal((B, o.n_g)), device=dev), torch.ones(B, dtype=torch.float64, device=dev)
        hv = torch.empty(B, o.nnz_hess, dtype=torch.float64, device=dev)
        res = {"group-major": [], "source-major": []}
        for rnd in range(5):
            for tag, oc in (("group-major", o), ("source-major", o2)):
                for _ in range(10): oc.eval_device(16, B, Z, None, 0, lam, sig, None, None, None, None, hv)
                oc.sync(); oc.timer_start()
                for _ in range(200): oc.eval_device(16, B, Z, None, 0, lam, sig, None, None, None, None, hv)
                res[tag].append(oc.timer_stop() / 200 * 1e3)
        print(f"two-pass hess_l, B = {B:4d}: " + ", ".join(f"{t} {sorted(v)[2]:.2f} us" for t, v in res.items()))

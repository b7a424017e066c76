"""In-process A/B of kernel build flags on the oracle time per IPOPT iteration (B = 1, host pointers, nlp_* symbols, the call order of
bench.py's ipopt_iter_report): one context per flag set, the same caller arrays, interleaved rounds (between processes the host side
of these calls -- a memcmp of x per call -- varies by more than the kernels do).
CASE=2|0|4 python tools/r4_ipopt_iter_ab.py "" "-DMPX_BOUND_LATE_LOADS" """
import ctypes, os, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import bench, problems
import mpopt_amd as M
from mpopt_amd import mp, _lib
flags = sys.argv[1:] or [""]
case = os.environ.get("CASE", "2")
builder, S, P = {"2": (problems.moon_lander, 1000, 5), "0": (problems.moon_lander, 20, 3), "4": (problems.hyper_sensitive, 4000, 3)}[case]
ctx = []
for fl in flags:
    os.environ["MPX_HIPCC_FLAGS"] = fl
    mpo = mp.mpopt(builder(mp, M.math), S, P, "LGR", device=0)
    nlp, bounds = mpo.create_nlp()
    ctx.append((mpo, nlp["oracle"], bounds))
os.environ.pop("MPX_HIPCC_FLAGS")
L = _lib.lib()
mpo, o, bounds = ctx[0]
rng = np.random.default_rng(20260928)
Zs = bench.make_points(o, mpo, bounds, 2, 20260928)
z = Zs[0].copy(); p = np.full(o.n_p, 1.0 / S)
lam, sig = rng.standard_normal(o.n_g), np.array([1.0])
f, g, gr = np.zeros(1), np.zeros(o.n_g), np.zeros(o.n_z)
jv, hv = np.zeros(max(o.nnz_jac, 1)), np.zeros(max(o.nnz_hess, 1))
vp = lambda arrs: (ctypes.c_void_p * len(arrs))(*[a.ctypes.data if a is not None else None for a in arrs])
calls = [("nlp_f", vp([z, p]), vp([f])), ("nlp_g", vp([z, p]), vp([g])), ("nlp_grad_f", vp([z, p]), vp([f, gr])),
         ("nlp_jac_g", vp([z, p]), vp([None, jv])), ("nlp_hess_l", vp([z, p, sig, lam]), vp([hv]))]
funs = {n: getattr(L, n) for n, _, _ in calls}
def sequence(seconds):
    for k in range(10):
        z[:] = Zs[k & 1]
        for n, a, r in calls: assert funs[n](a, r, None, None, 0) == 0
    acc, it, t_end = {n: [] for n, _, _ in calls}, 0, time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        it += 1; z[:] = Zs[it & 1]
        for n, a, r in calls:
            t0 = time.perf_counter(); funs[n](a, r, None, None, 0); acc[n].append(time.perf_counter() - t0)
    return {n: float(np.median(v)) * 1e6 for n, v in acc.items()}
mix = lambda d: 1.15 * (d["nlp_f"] + d["nlp_g"]) + d["nlp_grad_f"] + d["nlp_jac_g"] + d["nlp_hess_l"]
res = [[] for _ in ctx]; outs = []
for rnd in range(int(os.environ.get("ROUNDS", 6))):
    for k, (_, ok, _) in enumerate(ctx):
        ok.make_current(); L.mpx_current_pin_buffers(1)
        res[k].append(sequence(0.25))
        if rnd == 0:  # (the sequence ends on either of the two points: compare at the first)
            z[:] = Zs[0]
            for n, a, r in calls: funs[n](a, r, None, None, 0)
            outs.append((f.copy(), g.copy(), gr.copy(), jv.copy(), hv.copy()))
        L.mpx_current_pin_buffers(0)
for k, fl in enumerate(flags):
    med = {n: float(np.median([r[n] for r in res[k]])) for n in res[k][0]}
    same = all(np.array_equal(a, b) for a, b in zip(outs[k], outs[0]))
    if not same:
        for nm, a, b in zip(("f", "g", "grad_f", "jac", "hess"), outs[k], outs[0]):
            if not np.array_equal(a, b):
                d = np.flatnonzero(a != b)
                print(f"   {nm}: {len(d)} entries differ, first at {d[:5]}, max |diff| {np.abs(a - b).max():.3e}, values {a[d[:3]]} vs {b[d[:3]]}")
    print(f"case {case} [{fl or 'default':28s}] {mix(med):7.2f} us per iteration; per call " + " ".join(f"{n[4:]} {v:5.2f}" for n, v in med.items()) + f"  bit-equal to first: {same}")

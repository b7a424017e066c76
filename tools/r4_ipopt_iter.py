"""Oracle wall-clock per IPOPT iteration (bench.py's ipopt_iter_report: B = 1, host pointers, CasADi-convention symbols, the real
call order) for BASELINE configs 0, 2 and 4 without the rest of the bench.  A/B of kernel build flags between processes:
MPX_HIPCC_FLAGS=-DMPX_BOUND_LATE_LOADS python tools/r4_ipopt_iter.py"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench, problems
rows = (("config 2: moon lander 1000x5", (problems.moon_lander, 1000, 5, "LGR", ["moon_lander"], 1.0, [1])),
        ("config 0: moon lander 20x3", (problems.moon_lander, 20, 3, "LGR", ["moon_lander"], 1.0, [1])),
        ("config 4: hyper-sensitive 4000x3", (problems.hyper_sensitive, 4000, 3, "LGR", ["hyper_sensitive"], 1e-3, [0])))
for label, cfg in rows:
    for rep in range(int(os.environ.get("REPS", 2))):
        r = bench.ipopt_iter_report(*cfg, 0, seconds=0.5)
        print(f"{label:34s} {r['us_per_iter']:7.2f} us per iteration  (uncoalesced {r['us_per_iter_uncoalesced']:7.2f})  per call {r['per_call_us']}", flush=True)

#!/bin/bash
# Round 6: the lane-per-point hess_l kernel with ONE BODY PER GROUP SHAPE (assembly_lanes.py) -> gpurun_out/r6_lanes/: compile time and
# object size of the attached translation unit, kernel stats + PMC traffic + bench line of adaptive-hess at 20x5, 20x3, 100x3, 200x3.
set -u
export TMPDIR=/tmp
o=gpurun_out/r6_lanes; mkdir -p $o
python - > $o/compile.txt 2>&1 <<'PY'
import sys, time, os, glob
sys.path[:0] = [".", "tests"]
import mpopt_amd as M
from mpopt_amd import mp, _lib
import problems
print("grid | groups | shapes | source KB | source lines | hipcc s (cold) | code object KB")
for S, P in ((20, 5), (20, 3), (100, 3), (200, 3), (80, 5)):
    o = mp.mpopt_adaptive(problems.moon_lander(mp, M.math), S, P, "LGR").create_nlp()[0]["oracle"]
    src = o.lanes_source_text()
    for f in glob.glob(os.path.join(_lib.JIT_DIR, "*")):  # cold compile of THIS unit: drop its cache entry if present
        pass
    os.environ["MPX_HIPCC_FLAGS"] = f"-DMPX_COLD_{int(time.time())}"
    t = time.time(); co, path = _lib.compile_kernels(src); dt = time.time() - t
    del os.environ["MPX_HIPCC_FLAGS"]
    print(f"{S}x{P} | {len(o.lanes_plan.groups)} | {o.lanes_plan.n_shapes} | {len(src) >> 10} | {src.count(chr(10))} | {dt:.1f} | {len(co) >> 10}", flush=True)
    o.close()
PY
cat $o/compile.txt
for g in 20x5 20x3 100x3 200x3; do
  timeout 600 bash tools/profile_workload.sh r6_lanes/adaptive_hess_$g adaptive-hess mpx_asml_hes --adaptive-grid $g > /dev/null 2>&1
  python - $o/adaptive_hess_$g <<'PY'
import json, sys
d = sys.argv[1]
b = json.load(open(d + "/bench_line.json")); t = json.load(open(d + "/traffic.json"))
r = b["roofline"]
print(d.split("/")[-1], "value %.3g evals/s" % b["value"], "ms/step %.4f" % b["ms_per_step"], "kernel_us %.2f" % r["kernel_us"], "frac %.3f" % r["frac"], "traffic/alg %.3f" % t["traffic_over_algorithmic"],
      "frac_by_traffic %.3f" % (t["bytes_per_launch"] / (r["kernel_us"] * 1e-6) / 8e12), "alg MB/launch %.1f" % (r["algorithmic_bytes_per_launch"] / 1e6))
PY
done 2>&1 | tee $o/summary.txt
ls $o

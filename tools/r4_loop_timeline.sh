#!/bin/bash
# config-5 loop: kernel timeline of a few steps (rocprofv3 --kernel-trace), gaps between consecutive kernels of one outer iteration.
# usage: tools/r4_loop_timeline.sh <name>   -> gpurun_out/<name>/timeline.txt
set -u
out=gpurun_out/${1:-r4_loop_timeline}; mkdir -p $out; export TMPDIR=/tmp
W="--workload config5-loop --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o run -- python bench.py $W --steps 12 --warmup 3 --ramp-seconds 0.2 > $out/bench_under_rocprof.log 2>&1
f=$(find $out/trace -name '*kernel_trace.csv' | head -1)
python - "$f" > $out/timeline.txt <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: "hess" if "node_hess" in n else "bound" if "boundary" in n else "ea" if "equal_area" in n else "prefix" if "prefix" in n else "copy" if "copyBuffer" in n else n[:24]
ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ev = ev[len(ev) // 2:]  # steady state
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for (a, s0, e0), (b, s1, e1) in zip(ev, ev[1:]):
    gaps[f"{a}->{b}"].append((s1 - e0) / 1000.0)
    durs[a].append((e0 - s0) / 1000.0)
med = lambda v: sorted(v)[len(v) // 2]
print("kernel durations (us, median):", {k: round(med(v), 2) for k, v in durs.items()})
print("gaps end->start (us, median, count):", {k: (round(med(v), 2), len(v)) for k, v in gaps.items()})
it = [s for (a, s, e) in ev if a == "hess"]
per = [(b - a) / 1000.0 for a, b in zip(it, it[1:])]
print("hess start to next hess start (us): median", round(med(per), 2), "min", round(min(per), 2))
PY
cat $out/timeline.txt; rm -rf $out/trace

#!/bin/bash
# PMC traffic of the assembled (mpopt_adaptive) path after the round-3 fused kernels: FETCH_SIZE / WRITE_SIZE of every kernel of one
# f+g+grad_f+jac_g pass at B=4096 (moon lander 20x5): mpx_asm_fgj alone (fused) -- and, with MPX_NO_FUSE=1, mpx_pts_jac + mpx_gather_kernel.
set -u
out=${1:-gpurun_out/r3_adaptive2}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python bench.py --workload adaptive-fgj --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_line.json
MPX_NO_FUSE=1 timeout 300 python bench.py --workload adaptive-fgj --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_line_two_pass.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o run -- python bench.py --workload adaptive-fgj --no-cpu-baseline --no-extras --steps 5 --warmup 1 --ramp-seconds 0.2 > $out/pmc_$c.log 2>&1
  f=$(find $out/pmc_$c -name '*counter_collection.csv' | head -1)
  grep -E 'Counter_Name|mpx_asm|mpx_pts_jac|mpx_gather' "$f" | head -80 > $out/pmc_$(echo $c | tr A-Z a-z).csv; rm -rf $out/pmc_$c
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python bench.py --workload adaptive-fgj --no-cpu-baseline --no-extras --steps 50 --warmup 5 --ramp-seconds 0.5 > $out/trace.log 2>&1
f=$(find $out/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats.csv; rm -rf $out/trace
python - "$out" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]; tot = {}
for c in ("fetch_size", "write_size"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/pmc_{c}.csv")):
        per[r["Kernel_Name"].split("(")[0][:24]].append(float(r["Counter_Value"]))
    tot[c] = {k: sorted(v)[len(v)//2] for k, v in per.items()}
b = json.load(open(f"{out}/bench_line.json")); b2 = json.load(open(f"{out}/bench_line_two_pass.json"))
alg = b["roofline"]["algorithmic_bytes_per_launch"]
traffic = sum((2*tot["fetch_size"].get(k,0) + tot["write_size"].get(k,0))*1024 for k in set(tot["fetch_size"])|set(tot["write_size"]))
d = {"per_kernel_KB": tot, "bytes_per_pass": traffic, "algorithmic_bytes_per_pass": alg, "traffic_over_algorithmic": traffic/alg,
     "fused_evals_per_s": b["value"], "fused_frac": b["roofline"]["frac"], "two_pass_evals_per_s": b2["value"], "two_pass_frac": b2["roofline"]["frac"],
     "note": "2 x FETCH_SIZE + WRITE_SIZE (KB = 1024 B; FETCH doubled per MI355X_MICROARCH.md), median over the B=4096 launches of one f+g+grad_f+jac_g pass"}
json.dump(d, open(f"{out}/traffic.json","w"), indent=1); print(json.dumps(d))
PY

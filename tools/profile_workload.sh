#!/bin/bash
# Kernel stats + PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the dominant node kernel of any bench workload.
# usage: tools/profile_workload.sh <name> <workload> <kernel-name-prefix> [extra bench.py arguments]      -> gpurun_out/<name>/
set -u
name=$1; wl=$2; kern=$3; shift 3; xa="$*"
out=gpurun_out/$name; mkdir -p $out; export TMPDIR=/tmp
python bench.py --workload $wl --no-cpu-baseline --no-extras $xa 2>/dev/null | tail -1 > $out/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python bench.py --workload $wl --no-cpu-baseline --no-extras $xa > $out/bench_under_rocprof.log 2>&1
cp $(find $out/trace -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv; rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o run -- python bench.py --workload $wl --no-cpu-baseline --no-extras $xa --steps 5 --warmup 1 --ramp-seconds 0.2 > $out/pmc_$c.log 2>&1
  f=$(find $out/pmc_$c -name '*counter_collection.csv' | head -1)
  grep -E "Counter_Name|$kern" "$f" | head -60 > $out/pmc_$(echo $c | tr A-Z a-z).csv; rm -rf $out/pmc_$c
done
python - "$out" "$wl" "$kern" <<'PY'
import csv, json, sys
out, wl, kern = sys.argv[1:4]
vals = {}
for c in ("fetch_size", "write_size"):
    rows = [r for r in csv.DictReader(open(f"{out}/pmc_{c}.csv")) if r["Kernel_Name"].startswith(kern)]
    big = max(int(r["Grid_Size"]) for r in rows)
    v = sorted(float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == big)
    vals[c] = v[len(v) // 2]
b = json.load(open(f"{out}/bench_line.json"))
alg = b["roofline"]["algorithmic_bytes_per_launch"]
traffic = (2 * vals["fetch_size"] + vals["write_size"]) * 1024
d = {"workload": wl, "kernel": kern, "FETCH_SIZE_KB": vals["fetch_size"], "WRITE_SIZE_KB": vals["write_size"], "bytes_per_launch": traffic,
     "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg, "kernel_us_in_run": b["roofline"]["kernel_us"],
     "traffic_GBps": traffic / (b["roofline"]["kernel_us"] * 1e-6) / 1e9,
     "note": "2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, median over the launches of the largest grid; KB = 1024 B; FETCH doubled per MI355X_MICROARCH.md"}
json.dump(d, open(f"{out}/traffic.json", "w"), indent=1); print(json.dumps(d))
PY
head -4 $out/kernel_stats.csv | cut -c1-150

#!/bin/bash
# Profile the headline bench on the GPU box: rocprofv3 kernel stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE),
# summaries under gpurun_out/<name>/.   usage: tools/profile_bench.sh <name> [bench args...]
set -u
name=${1:-prof}; shift || true
out=gpurun_out/$name
mkdir -p $out
export TMPDIR=/tmp
python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $out/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python bench.py --no-cpu-baseline --no-extras "$@" > $out/bench_under_rocprof.log 2>&1
grep "^{\"metric" $out/bench_under_rocprof.log > $out/bench_line_under_rocprof.json
cp $(find $out/trace -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o run -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 --ramp-seconds 0.2 "$@" > $out/pmc_$c.log 2>&1
  f=$(find $out/pmc_$c -name '*counter_collection.csv' | head -1)
  grep -E 'Counter_Name|mpx_node' "$f" | head -40 > $out/pmc_$(echo $c | tr A-Z a-z).csv
done
python - "$out" <<'PY'
import csv, json, sys
out = sys.argv[1]
vals = {}
for c in ("fetch_size", "write_size"):
    rows = list(csv.DictReader(open(f"{out}/pmc_{c}.csv")))
    v = sorted(float(r["Counter_Value"]) for r in rows)
    vals[c], kernel, grid = v[len(v) // 2], rows[0]["Kernel_Name"], rows[0]["Grid_Size"]
d = {"kernel": kernel, "grid_size": grid, "workload": {"segments": 1000, "degree": 5, "batch": 4096}, "FETCH_SIZE_KB": vals["fetch_size"], "WRITE_SIZE_KB": vals["write_size"],
     "bytes_per_launch": (2 * vals["fetch_size"] + vals["write_size"]) * 1024,
     "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (median over the profiled launches); FETCH_SIZE doubled per "
             "MI355X_MICROARCH.md (gfx950 counts 128-B read requests at 64 B); KB = 1024 B"}
json.dump(d, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(d))
PY
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
head -4 $out/kernel_stats.csv | cut -c1-160
cat $out/bench_line.json | cut -c1-400

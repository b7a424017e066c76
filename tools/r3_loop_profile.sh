#!/bin/bash
# config-5 loop (bench.py --workload config5-loop): kernel stats, PMC traffic per outer iteration (separate FETCH_SIZE / WRITE_SIZE
# passes, summed over the kernels of one iteration), bench lines with the fast and the generic equal-area kernel, phase stamps.
# usage: tools/r3_loop_profile.sh <name>      -> gpurun_out/<name>/
set -u
out=gpurun_out/${1:-r3_config5_loop}; mkdir -p $out; export TMPDIR=/tmp
W="--workload config5-loop --no-cpu-baseline --no-extras"
python bench.py $W 2>/dev/null | tail -1 > $out/bench_line.json
MPX_EA_GENERIC=1 python bench.py $W 2>/dev/null | tail -1 > $out/bench_line_generic_equal_area_kernel.json
MPX_BENCH_UNFUSED_LOOP=1 MPX_EA_GENERIC=1 python bench.py $W 2>/dev/null | tail -1 > $out/bench_line_round2_call_sequence.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o run -- python bench.py $W --steps 40 --warmup 5 > $out/bench_under_rocprof.log 2>&1
cp $(find $out/trace -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv; rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o run -- python bench.py $W --steps 3 --warmup 1 --ramp-seconds 0.2 > $out/pmc_$c.log 2>&1
  f=$(find $out/pmc_$c -name '*counter_collection.csv' | head -1)
  grep -E "Counter_Name|mpx_node_hess|mpx_boundary_hess|equal_area|mpx_prefix" "$f" | head -400 > $out/pmc_$(echo $c | tr A-Z a-z).csv; rm -rf $out/pmc_$c
done
python - "$out" <<'PY'
import collections, csv, json, sys
out = sys.argv[1]
tot = {}
for c in ("fetch_size", "write_size"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{out}/pmc_{c}.csv")):
        n = r["Kernel_Name"]
        key = "equal_area" if "equal_area" in n else "prefix" if "prefix" in n else n.split("(")[0]
        per[key].append(float(r["Counter_Value"]))
    tot[c] = {k: sorted(v)[len(v) // 2] for k, v in per.items()}
b = json.load(open(f"{out}/bench_line.json"))
alg = b["roofline"]["algorithmic_bytes_per_launch"] / 5  # one step = 5 outer iterations
it = ("mpx_node_hess_0_3", "mpx_boundary_hess", "equal_area")
traffic = sum((2 * tot["fetch_size"].get(k, 0) + tot["write_size"].get(k, 0)) * 1024 for k in it)
d = {"per_kernel_KB": tot, "bytes_per_outer_iteration": traffic, "algorithmic_bytes_per_outer_iteration": alg, "traffic_over_algorithmic": traffic / alg,
     "note": "2 x FETCH_SIZE + WRITE_SIZE (KB = 1024 B; FETCH doubled per MI355X_MICROARCH.md), medians per kernel over one run (B = 512), summed over "
             "mpx_node_hess_0_3 (with MPX_MID_RESID) + mpx_boundary_hess + mpx_equal_area_fast_kernel of one outer iteration"}
json.dump(d, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(d)[:600])
for f in ("bench_line", "bench_line_generic_equal_area_kernel", "bench_line_round2_call_sequence"):
    x = json.load(open(f"{out}/{f}.json")); print(f, round(x["value"]), round(x["roofline"]["frac"], 4))
PY
MPX_LIB_HIPCC_FLAGS=-DMPX_EA_STAMPS python -c "
from mpopt_amd import _lib
_lib.build_library(force=True)"
# (the same flags on the run: the library is rebuilt whenever MPX_LIB_HIPCC_FLAGS changes)
MPX_LIB_HIPCC_FLAGS=-DMPX_EA_STAMPS MPX_EA_DEBUG=1 timeout 300 python bench.py $W --steps 3 --warmup 1 2>&1 | grep -A1 "equal_area phases" | tail -4 > $out/phase_stamps.txt; cat $out/phase_stamps.txt
python -c "
from mpopt_amd import _lib
_lib.build_library(force=True)"
head -8 $out/kernel_stats.csv | cut -c1-160

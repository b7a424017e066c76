#!/bin/bash
# Counter evidence for the assembled (mpopt_adaptive) path, VERDICT r2 next-2: what bounds mpx_gather_kernel / mpx_pts_jac?
# Separate rocprofv3 --pmc passes (SQ: 8 slots, TCC: 4, TCP separate), moon lander 20x5 adaptive, B = 4096.
# usage: tools/r3_adaptive_counters.sh [outdir]   (run on the GPU box, from the repo root)
set -u
out=${1:-gpurun_out/r3_adaptive}; mkdir -p $out; export TMPDIR=/tmp
run="python bench.py --workload adaptive-fgj --no-cpu-baseline --no-extras --steps 5 --warmup 1 --ramp-seconds 0.2"
timeout 300 python bench.py --workload adaptive-fgj --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_line.json
n=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$n -o run -- $run > $out/p$n.log 2>&1
  f=$(find $out/p$n -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then grep -E 'Counter_Name|mpx_pts_jac|mpx_gather' "$f" > $out/pmc_$n.csv; else echo "pass $n ($set): no counter file" >> $out/errors.txt; tail -5 $out/p$n.log >> $out/errors.txt; fi
  rm -rf $out/p$n
done
python - "$out" <<'PY'
import csv, json, sys, collections, glob
out = sys.argv[1]; per = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sorted(glob.glob(f"{out}/pmc_*.csv")):
    for r in csv.DictReader(open(fn)):
        if int(r["Grid_Size"]) > 100000:  # the B = 4096 launches
            per[r["Kernel_Name"].split("(")[0][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: {c: sorted(v)[len(v) // 2] for c, v in d.items()} for k, d in per.items()}
json.dump(med, open(f"{out}/counters.json", "w"), indent=1, sort_keys=True); print(json.dumps(med, indent=1, sort_keys=True))
PY

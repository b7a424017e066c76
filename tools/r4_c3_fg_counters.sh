#!/bin/bash
# Round 4: what bounds the g-only pass of config 3 (mpx_node_fg_0_30)?  SQ / LDS counters in separate rocprofv3 --pmc passes.
# usage: tools/r4_c3_fg_counters.sh [outdir] [extra bench args]   (GPU box, repo root)
set -u
out=${1:-gpurun_out/r4_c3_fg/counters}; shift; xa="$*"; mkdir -p $out; export TMPDIR=/tmp
run="python bench.py --workload config3-fgj --oracles g --no-cpu-baseline --no-extras --steps 5 --warmup 1 --ramp-seconds 0.2 $xa"
n=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$n -o run -- $run > $out/p$n.log 2>&1
  f=$(find $out/p$n -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then grep -E 'Counter_Name|mpx_node_fg' "$f" > $out/pmc_$n.csv; else echo "pass $n ($set): no counter file" >> $out/errors.txt; tail -5 $out/p$n.log >> $out/errors.txt; fi
  rm -rf $out/p$n
done
python - "$out" <<'PY'
import csv, json, sys, collections, glob
out = sys.argv[1]; per = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sorted(glob.glob(f"{out}/pmc_*.csv")):
    for r in csv.DictReader(open(fn)):
        per[r["Kernel_Name"].split("(")[0][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: {c: sorted(v)[len(v) // 2] for c, v in d.items()} for k, d in per.items()}
json.dump(med, open(f"{out}/counters.json", "w"), indent=1, sort_keys=True); print(json.dumps(med, indent=1, sort_keys=True))
PY

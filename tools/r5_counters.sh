#!/bin/bash
# Round 5: issue / wait / memory-request counters of ONE kernel of any bench workload, in separate rocprofv3 --pmc passes (8 SQ
# slots, 4 TCC slots per pass; never combined with trace domains).
# usage: tools/r5_counters.sh <outdir> <kernel name prefix> <bench.py arguments ...>      (GPU box, repo root)
set -u
out=$1; kern=$2; shift 2; xa="$*"; mkdir -p $out; export TMPDIR=/tmp
run="python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 --ramp-seconds 0.2 $xa"
n=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$n -o run -- $run > $out/p$n.log 2>&1
  f=$(find $out/p$n -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|$kern" "$f" > $out/pmc_$n.csv; else echo "pass $n ($set): no counter file" >> $out/errors.txt; tail -3 $out/p$n.log >> $out/errors.txt; fi
  rm -rf $out/p$n $out/p$n.log
done
python - "$out" <<'PY'
import csv, json, sys, collections, glob
out = sys.argv[1]; per = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sorted(glob.glob(f"{out}/pmc_*.csv")):
    rows = list(csv.DictReader(open(fn)))
    if not rows: continue
    big = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if int(r["Grid_Size"]) == big:
            per[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: {c: sorted(v)[len(v) // 2] for c, v in d.items()} for k, d in per.items()}
for k, d in med.items():
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        d["derived"] = {x: round(d[x] / wc, 4) for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if x in d}
        d["derived"]["note"] = "fractions of SQ_WAVE_CYCLES (quad-cycles summed over wavefronts)"
json.dump(med, open(f"{out}/counters.json", "w"), indent=1, sort_keys=True); print(json.dumps(med, indent=1, sort_keys=True))
PY

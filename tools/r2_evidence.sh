#!/bin/bash
# Round-2 evidence pass (one gpurun call): counters for the degree-30 bucket of config 3 and per-channel TCC write
# counters of the headline kernel on several buffer placements.  Everything lands under gpurun_out/r2_ev/.
set -u
out=gpurun_out/r2_ev
mkdir -p $out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
C3="python bench.py --workload config3-fgj --no-cpu-baseline --no-extras --steps 5 --warmup 1 --ramp-seconds 0.2"

# ---- config 3: kernel stats + PMC passes (one counter group per pass) ------------------------------------------
python bench.py --workload config3-fgj --no-cpu-baseline --no-extras > $out/c3_bench_line.json 2> $out/c3_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_trace -o run -- python bench.py --workload config3-fgj --no-cpu-baseline --no-extras > $out/c3_under_rocprof.log 2>&1
cp $(find $out/c3_trace -name '*kernel_stats.csv' | head -1) $out/c3_kernel_stats.csv
rm -rf $out/c3_trace
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o run -- $C3 > $out/pmc_$name.log 2>&1
  local f=$(find $out/pmc_$name -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then grep -E 'Counter_Name|mpx_node_fgj' "$f" > $out/c3_pmc_$name.csv; fi
  rm -rf $out/pmc_$name
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT
pass sq3 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
pass tcc TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum

# ---- headline kernel: per-channel write counters over ten buffer placements (json keeps the instance dimension) ----
rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL --output-format json csv -d $out/chan -o run -- python tools/alloc_probe.py > $out/chan_probe.log 2>&1
python tools/chan_summary.py $out/chan > $out/chan_summary.txt 2>&1
rm -rf $out/chan

# ---- latency baseline (single evaluations through host pointers) ---------------------------------------------------
python tools/latency.py > $out/latency_before.txt 2>&1
ls -la $out

#!/bin/bash
# Evidence for the lane-per-point hess_l kernel (mpx_asml_hes) -> gpurun_out/r5_lanes/: kernel stats + PMC traffic of the adaptive-hess
# workload with it and with the fused kernel (MPX_NO_LANES=1), bench lines at B = 4096 and 16384, the counters of tools/r5_counters.sh.
set -u
export TMPDIR=/tmp
o=gpurun_out/r5_lanes; mkdir -p $o
timeout 500 bash tools/profile_workload.sh r5_lanes/adaptive_hess adaptive-hess mpx_asml_hes > /dev/null 2>&1
MPX_NO_LANES=1 timeout 500 bash tools/profile_workload.sh r5_lanes/adaptive_hess_fused adaptive-hess mpx_asm_hes > /dev/null 2>&1
for b in 4096 16384; do
  timeout 300 python bench.py --workload adaptive-hess --no-cpu-baseline --no-extras --batch $b 2>/dev/null | tail -1 > $o/bench_line_adaptive-hess_B$b.json
  MPX_NO_LANES=1 timeout 300 python bench.py --workload adaptive-hess --no-cpu-baseline --no-extras --batch $b 2>/dev/null | tail -1 > $o/bench_line_adaptive-hess_B${b}_fused.json
done
timeout 900 bash tools/r5_counters.sh gpurun_out/r5_lanes/counters mpx_asml_hes --workload adaptive-hess > $o/counters.log 2>&1
ls -R $o | head -40

#!/bin/bash
# Late round 6 (after the high-degree light kernels): the lines of r6_final that changed + the full GPU suite + the report.
set -u
export TMPDIR=/tmp
o=gpurun_out/r6_final; mkdir -p $o
pw() { timeout 500 bash tools/profile_workload.sh "$@" > /dev/null 2>&1; }
pw r6_final/deg100_light_g config2-fgj mpx_lighthigh_fg_0_100 --segments 50 --degree 100 --batch 512 --oracles g
pw r6_final/deg255_light_g config2-fgj mpx_lighthigh_fg_0_255 --segments 20 --degree 255 --batch 512 --oracles g
pw r6_final/deg100_light_f_grad_f config2-fgj mpx_lighthigh_fgq_0_100 --segments 50 --degree 100 --batch 512 --oracles f,grad_f
( time timeout 900 python bench.py > $o/bench_line_default.json 2> $o/bench_default.err ) 2> $o/bench_default_time.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $o/gpu_tests_full_suite.log 2>&1
tail -3 $o/gpu_tests_full_suite.log
bash tools/r6_report.sh

#!/bin/bash
# Round-6 evidence in one GPU call -> gpurun_out/r6_final/: the headline five times in a row (alloc_outputs: tries_used, spread), its
# profile (kernel stats + PMC traffic), one bench line per secondary workload WITH PMC traffic (so that every line carries
# frac_by_traffic), the light passes of configs 2 / 3 / 4, config-5 loop, assembled passes, a self-launched 2-rank line over gloo.
set -u
export TMPDIR=/tmp
o=gpurun_out/r6_final; mkdir -p $o
for k in 1 2 3 4 5; do timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $o/bench_line_default_run$k.json; done
timeout 600 bash tools/profile_bench.sh r6_final/headline > $o/headline.log 2>&1
pw() { timeout 500 bash tools/profile_workload.sh "$@" > /dev/null 2>&1; }
pw r6_final/c2_hess config2-hess mpx_node_hess
pw r6_final/c5_hess config5-hess mpx_node_hess
pw r6_final/c5_fgj config5-fgj mpx_node_fgj
pw r6_final/c4_fgj config4-fgj mpx_node_fgj
pw r6_final/c4_hess config4-hess mpx_node_hess
pw r6_final/c3_fgj config3-fgj mpx_node_fgj_0_30
pw r6_final/c3_hess config3-hess mpx_node_hessn
pw r6_final/adaptive_fgj adaptive-fgj mpx_asm_fgj
pw r6_final/adaptive_hess adaptive-hess mpx_asm_hes
for x in f g f,grad_f; do
  n=$(echo $x | tr , _)
  pw r6_final/c2_light_$n config2-fgj mpx_lightlow --oracles $x
  pw r6_final/c3_light_$n config3-fgj mpx_light --oracles $x
  pw r6_final/c4_light_$n config4-fgj mpx_lightlow --oracles $x
  pw r6_final/c5_light_$n config5-fgj mpx_lightlow --oracles $x
done
timeout 900 bash tools/r3_loop_profile.sh r6_final/config5_loop > $o/config5_loop.log 2>&1
for b in 2048 4096; do timeout 300 python bench.py --workload config5-loop --no-cpu-baseline --no-extras --batch $b 2>/dev/null | tail -1 > $o/bench_line_config5-loop_B$b.json; done
# degrees above the LDS tables (streamed tables, round 6): the headline problem on 50 x 100 and 20 x 255
pw r6_final/deg100_fgj config2-fgj mpx_node_fgj_0_100 --segments 50 --degree 100 --batch 512
pw r6_final/deg255_fgj config2-fgj mpx_node_fgj_0_255 --segments 20 --degree 255 --batch 512
pw r6_final/deg100_light_g config2-fgj mpx_node_fg_0_100 --segments 50 --degree 100 --batch 512 --oracles g
MPX_DIST_BACKEND=gloo timeout 500 python bench.py --gpus 2 --steps 20 --warmup 5 --batch 1024 2>/dev/null | tail -1 > $o/bench_line_2ranks_gloo_self_launched.json
( time timeout 900 python bench.py > $o/bench_line_default.json 2> $o/bench_default.err ) 2> $o/bench_default_time.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $o/gpu_tests_full_suite.log 2>&1
find $o -name '*.log' -size +200k ! -name 'gpu_tests_full_suite.log' -delete
ls $o | head -80

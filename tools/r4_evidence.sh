#!/bin/bash
# Round-4 evidence in one GPU call: headline profile (kernel stats + PMC traffic), config-3 profiles, light passes of configs 2 / 3 / 5,
# one bench line per secondary workload, the config-5 loop profile (+ lines at larger batches), the assembled path (bench line + PMC
# traffic + in-process A/B of the round's switches), a self-launched 2-rank bench line over gloo, the default bench line.
# Everything lands under gpurun_out/r4_final/.
set -u
export TMPDIR=/tmp
o=gpurun_out/r4_final; mkdir -p $o
timeout 600 bash tools/profile_bench.sh r4_final/headline > $o/headline.log 2>&1
timeout 600 bash tools/profile_workload.sh r4_final/c3_hess config3-hess mpx_node_hessn > $o/c3_hess.log 2>&1
timeout 600 bash tools/profile_workload.sh r4_final/c3_fgj config3-fgj mpx_node_fgj_0_30 > $o/c3_fgj.log 2>&1
for x in f g f,grad_f; do
  n=$(echo $x | tr , _)
  timeout 400 bash tools/profile_workload.sh r4_final/c2_light_$n config2-fgj mpx_lightlow --oracles $x > $o/c2_light_$n.log 2>&1
  timeout 400 bash tools/profile_workload.sh r4_final/c3_light_$n config3-fgj mpx_light --oracles $x > $o/c3_light_$n.log 2>&1
done
for w in config2-hess config5-hess config3-fgj config3-hess config5-loop adaptive-fgj; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $o/bench_line_$w.json
done
for b in 2048 4096; do timeout 300 python bench.py --workload config5-loop --no-cpu-baseline --no-extras --batch $b 2>/dev/null | tail -1 > $o/bench_line_config5-loop_B$b.json; done
timeout 900 bash tools/r3_loop_profile.sh r4_final/config5_loop > $o/config5_loop.log 2>&1
timeout 600 bash tools/r3_adaptive_pmc.sh gpurun_out/r4_final/adaptive > $o/adaptive.log 2>&1
timeout 600 python tools/r4_adaptive_ab.py "" "-DMPX_FUSE_NO_SET_CONSTS" "-DMPX_FUSE_XCD_BLOCKED=0" "-DMPX_FUSE_Z_LATE=1" "-DMPX_FUSE_PAIR_ROWS=1" 2>&1 | grep -v amdgpu.ids > $o/adaptive_ab.txt
timeout 600 python tools/r4_lightlow_check.py time 2>&1 | grep -v amdgpu.ids > $o/lightlow_check.txt
MPX_DIST_BACKEND=gloo timeout 500 python bench.py --gpus 2 --steps 20 --warmup 5 --batch 1024 2>/dev/null | tail -1 > $o/bench_line_2ranks_gloo_self_launched.json
timeout 600 python bench.py > $o/bench_line_default.json 2> $o/bench_default.err
ls -la $o | head -60

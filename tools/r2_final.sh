#!/bin/bash
# Round-2 final measurements (one gpurun call): headline profile (kernel stats + PMC traffic), config 3 and config-5-loop kernel
# stats, bench lines of every workload, 2-rank gloo lines of the multi-rank paths.  Output: gpurun_out/r2_final/
set -u
export TMPDIR=/tmp
out=gpurun_out/r2_final; mkdir -p $out
bash tools/profile_bench.sh r2_final/headline > $out/headline_profile.log 2>&1
timeout 600 python bench.py > $out/bench_line_default.json 2> $out/bench_default.err
for w in config3-fgj config5-hess config5-loop adaptive-fgj; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_line_$w.json
done
for w in config3-fgj config5-loop; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -o run -- python bench.py --workload $w --no-cpu-baseline --no-extras > $out/under_rocprof_$w.log 2>&1
  cp $(find $out/trace_$w -name '*kernel_stats.csv' | head -1) $out/kernel_stats_$w.csv; rm -rf $out/trace_$w
  grep '^{' $out/under_rocprof_$w.log > $out/bench_line_under_rocprof_$w.json
done
MPX_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 2>/dev/null | tail -1 > $out/bench_line_2ranks_1gpu_gloo.json
python tools/latency.py > $out/latency.txt 2>&1
ls -la $out | head -40

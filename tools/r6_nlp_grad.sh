#!/bin/bash
# nlp_grad as a batched pass: bench lines + rocprofv3 kernel stats per configuration -> gpurun_out/<name>/ (copy to profiles/)
# usage: tools/r6_nlp_grad.sh <name> [config ...]
set -u
export TMPDIR=/tmp
name=${1:-r6_nlp_grad}; shift || true
cfgs=${*:-config2 config3 config4 config5 deg100}
O=gpurun_out/$name; mkdir -p $O
timeout 300 python tools/r6_nlp_grad_bench.py $cfgs > $O/bench_lines.jsonl 2> $O/bench.err; cat $O/bench_lines.jsonl
for c in $cfgs; do
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$c -o run -- python tools/r6_nlp_grad_bench.py $c > $O/under_rocprof_$c.log 2>&1
  f=$(find $O/trace_$c -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$c.csv; rm -rf $O/trace_$c
  echo "== $c"; head -5 $O/kernel_stats_$c.csv | cut -c1-170
done

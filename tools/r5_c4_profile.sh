#!/bin/bash
# Round 5: config 4 (two-phase Schwartz, 500 x 3 per phase) before / after "all phases in one launch": kernel stats + PMC traffic of the
# heavy passes and of nlp_g alone, with MPX_NO_PHASE_MERGE=1 (one launch per phase: round 4's form) and without.   -> gpurun_out/r5_c4/
set -u
export TMPDIR=/tmp
for tag in before after; do
  if [ $tag = before ]; then export MPX_NO_PHASE_MERGE=1; else unset MPX_NO_PHASE_MERGE; fi
  bash tools/profile_workload.sh r5_c4/${tag}_fgj config4-fgj mpx_node_fgj --plain-outputs > /dev/null 2>&1
  bash tools/profile_workload.sh r5_c4/${tag}_hess config4-hess mpx_node_hess --plain-outputs > /dev/null 2>&1
  bash tools/profile_workload.sh r5_c4/${tag}_light_g config4-fgj mpx_lightlow --oracles g --plain-outputs > /dev/null 2>&1
done
unset MPX_NO_PHASE_MERGE
for d in gpurun_out/r5_c4/*/; do rm -f $d/*.log; echo "$d: $(python -c "import json;b=json.load(open('$d/bench_line.json'));t=json.load(open('$d/traffic.json'));print('step us %.1f  frac %.3f  traffic/alg %.3f' % (b['ms_per_step']*1000, b['roofline']['frac'], t['traffic_over_algorithmic']))")"; head -3 $d/kernel_stats.csv | cut -c1-120; done

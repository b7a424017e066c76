#!/bin/bash
# Round 6: light passes of single-degree grids of degree >= 32 on the matrix cores (mpx_lighthigh_*) against the node kernels in f / g mode
# (MPX_NO_LIGHT=1) -> gpurun_out/r6_light_high/: bench lines (HIP-event kernel time, roofline) and one rocprof kernel-stats + PMC pass.
set -u
export TMPDIR=/tmp
o=gpurun_out/r6_light_high; mkdir -p $o
line() { python - "$1" "$2" <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); r = b["roofline"]
print(f"{sys.argv[2]:44s} value {b['value']:.4g} evals/s  step {b['ms_per_step']*1000:8.1f} us  kernel {r['kernel_us']:8.1f} us  frac {r['frac']:.3f}")
PY
}
for g in "50 100" "20 255" "120 40" "75 64"; do
  set -- $g
  for x in f g f,grad_f; do
    n=$(echo $x | tr , _)
    for nl in 0 1; do
      f=$o/line_${1}x${2}_${n}_nolight$nl.json
      if [ $nl = 1 ]; then export MPX_NO_LIGHT=1; else unset MPX_NO_LIGHT; fi
      timeout 300 python bench.py --segments $1 --degree $2 --batch 512 --oracles $x --no-cpu-baseline 2>/dev/null | tail -1 > $f
      line $f "${1}x${2} $x $( [ $nl = 1 ] && echo '(node kernels)' || echo '(matrix cores)')"
    done
  done
done 2>&1 | tee $o/summary.txt
unset MPX_NO_LIGHT
timeout 600 bash tools/profile_workload.sh r6_light_high/deg100_g config2-fgj mpx_lighthigh_fg_0_100 --segments 50 --degree 100 --batch 512 --oracles g > /dev/null 2>&1
cat $o/deg100_g/traffic.json; head -3 $o/deg100_g/kernel_stats.csv | cut -c1-160

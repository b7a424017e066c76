#!/bin/bash
# A/B of the fused assembled kernels under different compile-time settings: tools/r3_fused_sweep.sh out.txt "flags1" "flags2" ...
out=$1; shift; mkdir -p "$(dirname $out)"; : > $out
for fl in "$@"; do
  echo "== $fl" >> $out
  MPX_HIPCC_FLAGS="$fl" timeout 300 python tools/r3_fused_ab.py moon_lander hyper_sensitive 2>&1 | grep '^{' | python -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line)
    print(d['case'][:20], {k: (d[k]['two_pass_ms'], d[k]['fused_ms'], d[k]['bit_identical']) for k in ('fgj', 'fg', 'hess')})
" >> $out
done
cat $out

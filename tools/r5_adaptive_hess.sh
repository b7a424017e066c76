#!/bin/bash
# Round 5: the assembled (mpopt_adaptive) hess_l pass -- mpx_asm_hes, 0.24 of peak in round 4 with no evidence why.  Bench line,
# phase stamps of one workgroup (MPX_FUSE_DEBUG / -DMPX_FUSE_PT_STAMPS), issue / wait / request counters, in-process A/B of switches.
# usage: tools/r5_adaptive_hess.sh [outdir]     (GPU box, repo root)
set -u
out=${1:-gpurun_out/r5_adaptive_hess}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python bench.py --workload adaptive-hess --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $out/bench_line.json
for fl in "" "-DMPX_FUSE_PT_STAMPS=1"; do
  echo "== stamps, flags: '$fl'" >> $out/stamps.txt
  MPX_HIPCC_FLAGS="$fl" MPX_FUSE_DEBUG=1 MPX_FUSE_PT_STAMPS=1 timeout 300 python bench.py --workload adaptive-hess --no-cpu-baseline --no-extras --steps 3 --warmup 1 --ramp-seconds 0.1 2>&1 | grep -A1 "fused mode 2" | tail -4 >> $out/stamps.txt
done
bash tools/r5_counters.sh $out/counters mpx_asm_hes --workload adaptive-hess > $out/counters.log 2>&1
timeout 900 python tools/r4_adaptive_ab.py "" "-DMPX_FUSE_MROW_HES=1" "-DMPX_FUSE_MIN_WAVES=2 -DMPX_FUSE_MROW_HES=1" ${EXTRA_AB:-} 2>&1 | grep "^hess\|^fgj" > $out/ab.txt
cat $out/stamps.txt $out/ab.txt

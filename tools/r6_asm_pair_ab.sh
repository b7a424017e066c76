#!/bin/bash
# Round 6, VERDICT r5 item 4: assembled first-order pass with 16-byte stores by construction.  A/B of the entry order of jac_g
# (MPX_ASM_CLASS_MAJOR: single-term rows first) x the fused kernel's row pairing (-DMPX_FUSE_PAIR_ROWS=1), bench adaptive-fgj, and
# the bit-identity tests under the pairing.  -> gpurun_out/r6_asm_pair/
set -u
o=gpurun_out/r6_asm_pair; mkdir -p $o
for cm in 0 1; do for pr in 0 1; do
  for rep in 1 2; do
  MPX_ASM_CLASS_MAJOR=$cm MPX_HIPCC_FLAGS="-DMPX_FUSE_PAIR_ROWS=$pr" timeout 600 python bench.py --workload adaptive-fgj --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $o/line_cm${cm}_pr${pr}_$rep.json
  python - $o/line_cm${cm}_pr${pr}_$rep.json $cm $pr <<'PY'
import json, sys
b = json.load(open(sys.argv[1])); r = b["roofline"]
print(f"class_major={sys.argv[2]} pair_rows={sys.argv[3]}: ms/step {b['ms_per_step']:.4f} kernel_us {r['kernel_us']:.2f} frac {r['frac']:.3f} value {b['value']:.3g}")
PY
  done
done; done 2>&1 | tee $o/summary.txt
MPX_HIPCC_FLAGS="-DMPX_FUSE_PAIR_ROWS=1" timeout 900 python -m pytest tests/test_gpu_adaptive.py -q -x -k "fused or golden or batch or exact_ad" 2>&1 | tail -3 | tee $o/tests_pair_rows.txt

#!/bin/bash
# A/B of build flags on the high-degree light kernels: nlp_g at 50x100 and 20x255, B = 512
for fl in "" "-DMPX_ABL_LH_NO_MFMA" "-DMPX_ABL_LH_NO_STORE" "-DMPX_ABL_LH_NO_MFMA -DMPX_ABL_LH_NO_STORE"; do
  for g in "50 100" "20 255"; do set -- $g
    MPX_HIPCC_FLAGS="$fl" timeout 300 python bench.py --segments $1 --degree $2 --batch 512 --oracles g --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('[$fl] $1 x $2 g: kernel %.1f us' % b['roofline']['kernel_us'])"
  done
done

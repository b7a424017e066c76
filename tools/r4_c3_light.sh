#!/bin/bash
# Round 4: the light passes of config 3 (nlp_g / nlp_f / nlp_grad_f alone, what a line search calls): kernel stats + PMC traffic
# of mpx_node_fg_0_30 / mpx_node_fgj_0_30, and the MPX_BPB sweep.        -> gpurun_out/r4_c3_fg/
set -u
cd "$(dirname "$0")/.."
for o in g f grad_f; do
  bash tools/profile_workload.sh r4_c3_fg/$o config3-fgj mpx_node_fg --oracles $o
done
CASE=1 ONLY=f,g,grad_f,f+g python tools/r3_single_oracle_bpb.py > gpurun_out/r4_c3_fg/bpb_sweep.txt 2>&1
tail -5 gpurun_out/r4_c3_fg/bpb_sweep.txt

#!/bin/bash
# Round 4: the light passes of config 3 (nlp_g / nlp_f / nlp_grad_f alone, what a line search calls): kernel stats + PMC traffic
# of the dominant kernel and the MPX_BPB sweep.   usage: tools/r4_c3_light.sh <tag> <kernel prefix>   -> gpurun_out/r4_c3_fg/<tag>_*
set -u
cd "$(dirname "$0")/.."
tag=${1:-after}; kern=${2:-mpx_light}
for o in g f grad_f; do
  bash tools/profile_workload.sh r4_c3_fg/${tag}_$o config3-fgj $kern --oracles $o
done
CASE=1 ONLY=f,g,grad_f,f+g python tools/r3_single_oracle_bpb.py > gpurun_out/r4_c3_fg/${tag}_bpb_sweep.txt 2>&1
tail -5 gpurun_out/r4_c3_fg/${tag}_bpb_sweep.txt

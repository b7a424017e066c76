#!/bin/bash
# The per-problem code objects (mpopt_amd/_jit_cache, git-ignored, travels with the tree) that build() does not know about are compiled by
# the GPU tests on the box (hipcc, ~130 translation units, 6 of the suite's 9.5 minutes).  This runs the suite once on a GPU box, copies what
# it compiled back and merges it into the in-tree cache: the next run of the suite compiles nothing (352 tests: 9m25 -> 3m32, round 6).
#   tools/harvest_jit_cache.sh          (from the repository root; one gpurun call)
set -e
cd "$(dirname "$0")/.."
/usr/local/graft/bin/gpurun --timeout 1500 -- 'touch /tmp/start_marker; python -m pytest tests -m gpu -x -q --durations=30 > gpurun_out/gpu_suite_harvest.log 2>&1; tail -3 gpurun_out/gpu_suite_harvest.log; mkdir -p gpurun_out/jit_new; find mpopt_amd/_jit_cache -type f -newer /tmp/start_marker -exec cp {} gpurun_out/jit_new/ \; ; ls gpurun_out/jit_new | wc -l'
if [ -d gpurun_out/jit_new ]; then cp -n gpurun_out/jit_new/* mpopt_amd/_jit_cache/ 2>/dev/null || true; rm -rf gpurun_out/jit_new; fi
ls mpopt_amd/_jit_cache | wc -l

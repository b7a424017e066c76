#!/bin/bash
# equal-area kernel: tests of the config-5 loop, bench line, phase stamps (diagnostics build), generic-kernel A/B
set -u
o=gpurun_out/r3_ea; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_config5_loop.py -x -q > $o/tests.log 2>&1; tail -3 $o/tests.log
for i in 1 2; do timeout 300 python bench.py --workload config5-loop --no-cpu-baseline 2>/dev/null | tail -1 > $o/bench_line_$i.json; done
MPX_EA_GENERIC=1 timeout 300 python bench.py --workload config5-loop --no-cpu-baseline 2>/dev/null | tail -1 > $o/bench_line_generic.json
python - <<'PY'
import json
for f in ("bench_line_1","bench_line_2","bench_line_generic"):
    d=json.loads(open(f"gpurun_out/r3_ea/{f}.json").read()); print(f, round(d["value"]), round(d["roofline"]["frac"],4), round(d["ms_per_step"],4))
PY
MPX_LIB_HIPCC_FLAGS=-DMPX_EA_STAMPS python -c "
from mpopt_amd import _lib
_lib.build_library(force=True)"
MPX_EA_DEBUG=1 timeout 300 python bench.py --workload config5-loop --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -A1 "equal_area phases" | tail -6 > $o/stamps.txt; cat $o/stamps.txt
python -c "
from mpopt_amd import _lib
_lib.build_library(force=True)"

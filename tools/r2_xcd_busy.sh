#!/bin/bash
# per-XCD busy cycles of the headline node kernel (rocprofv3 json keeps the XCC / SE dimensions) + the full GPU test tier
set -u
out=gpurun_out/r2_xcd
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format json -d $out/busy -o run -- python tools/alloc_probe.py > $out/busy_probe.log 2>&1
python tools/chan_summary.py $out/busy > $out/busy_summary.txt 2>&1
mv $out/chan_records.json $out/busy_records.json 2>/dev/null
rm -rf $out/busy
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -5 $out/pytest_gpu.log

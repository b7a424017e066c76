#!/bin/bash
# profiles/r6_report.md: every oracle x configuration x batch size (tools/report.py) + the per-call table on the reference's published grids
set -u
mkdir -p gpurun_out
timeout 2400 python tools/report.py > gpurun_out/r6_report.md 2> gpurun_out/r6_report.err
timeout 900 python tools/r6_published_table.py >> gpurun_out/r6_report.md 2>> gpurun_out/r6_report.err
tail -5 gpurun_out/r6_report.err; wc -l gpurun_out/r6_report.md

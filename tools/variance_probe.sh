#!/bin/bash
# Scratch: correlate kernel time with the box's clock / power / partition state (run on the GPU box).
rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -E "GPU\[0\]" | head -4
rocm-smi --showmaxpower --showpowercap 2>/dev/null | grep -E "GPU\[0\]" | head -3
( python bench.py --no-cpu-baseline --steps 400 --ramp-seconds 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_us', round(d['roofline']['kernel_us'],1), 'frac', round(d['roofline']['frac'],3))" ) &
sleep 9
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "GPU\[0\]" | grep -E "sclk|mclk|fclk|socclk|Power|junction|memory" | head -9
wait
./tools/store_bw | grep -E "tile 16B plain|Memset" | head -2

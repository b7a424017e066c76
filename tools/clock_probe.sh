#!/bin/bash
# Which clocks does the headline kernel run under on this box?  bench.py in the background, rocm-smi while it runs.
(python bench.py --no-cpu-baseline --no-extras --steps 6000 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), round(d['roofline']['frac'],3), d['outputs'][100:200])") &
sleep 5
for i in 1 2; do rocm-smi --showclocks --showtemp --showpower 2>&1 | grep -E "fclk|mclk|sclk|junction|memory|Power" ; sleep 1.5; done
wait

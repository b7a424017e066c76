#!/bin/bash
# kernel time next to HBM temperature / clocks / power on this box (collect over several boxes to correlate)
t() { rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "GPU\[0\]" | grep -E -i "memory|junction|mclk|fclk|sclk|socclk|Average Graphics|Current Socket" | sed "s/GPU\[0\]\s*: //" | tr "\n" ";"; echo; }
echo "idle: $(t)"
python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_us', round(d['roofline']['kernel_us'],1), 'frac', round(d['roofline']['frac'],3))" &
sleep 14
echo "load: $(t)"
wait

MPX_EA_DEBUG=1 timeout 300 python bench.py --workload config5-loop --steps 2 --warmup 1 --ramp-seconds 0.1 2>&1 | grep "equal_area phases" | tail -5

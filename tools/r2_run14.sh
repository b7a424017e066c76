mkdir -p gpurun_out/r2_m
MPX_EA_DEBUG=1 timeout 300 python bench.py --workload config5-loop --steps 2 --warmup 1 --ramp-seconds 0.1 2>&1 | grep "equal_area phases" | tail -2
timeout 900 python -m pytest tests/test_gpu_config5_loop.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python bench.py --workload config5-loop --steps 20 2>/dev/null | tail -1 > gpurun_out/r2_m/loop5_bench_line.json; cut -c1-330 gpurun_out/r2_m/loop5_bench_line.json

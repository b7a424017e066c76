mkdir -p gpurun_out/r2_c
timeout 900 python -m pytest tests/test_gpu_shard.py -x -q > gpurun_out/r2_c/shard_test.log 2>&1; tail -15 gpurun_out/r2_c/shard_test.log
timeout 600 python tools/placement_ab.py > gpurun_out/r2_c/placement_ab.txt 2>&1; cat gpurun_out/r2_c/placement_ab.txt | tail -12

mkdir -p gpurun_out/r2_d
timeout 900 python -m pytest tests/test_gpu_shard.py -x -q > gpurun_out/r2_d/shard_test.log 2>&1; tail -15 gpurun_out/r2_d/shard_test.log
./tools/zc_probe > gpurun_out/r2_d/zc_probe.txt 2>&1; cat gpurun_out/r2_d/zc_probe.txt

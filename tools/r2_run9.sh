mkdir -p gpurun_out/r2_i
timeout 900 python -m pytest tests/test_gpu_config5_loop.py tests/test_residuals.py tests/test_hadaptive.py -m gpu -x -q > gpurun_out/r2_i/pytest.log 2>&1; tail -8 gpurun_out/r2_i/pytest.log
timeout 300 python bench.py --workload config5-loop --steps 20 > gpurun_out/r2_i/loop5.json 2> gpurun_out/r2_i/loop5.err; tail -2 gpurun_out/r2_i/loop5.err; cat gpurun_out/r2_i/loop5.json | cut -c1-1500
timeout 300 python bench.py --workload config5-hess --steps 20 2>/dev/null | tail -1 | cut -c1-300

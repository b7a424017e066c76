mkdir -p gpurun_out/r2_j
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_config5_loop.py -m gpu -x -q > gpurun_out/r2_j/pytest.log 2>&1; tail -3 gpurun_out/r2_j/pytest.log
timeout 300 python bench.py --workload config5-loop --steps 20 2>/dev/null | tail -1 | cut -c1-330
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_j/trace -o run -- python bench.py --workload config5-loop --steps 20 > gpurun_out/r2_j/under_rocprof.log 2>&1
cp $(find gpurun_out/r2_j/trace -name '*kernel_stats.csv' | head -1) gpurun_out/r2_j/loop5_kernel_stats.csv; rm -rf gpurun_out/r2_j/trace
head -8 gpurun_out/r2_j/loop5_kernel_stats.csv | cut -c1-140

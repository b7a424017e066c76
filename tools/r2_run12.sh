mkdir -p gpurun_out/r2_l
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_config5_loop.py -m gpu -x -q > gpurun_out/r2_l/pytest.log 2>&1; tail -3 gpurun_out/r2_l/pytest.log
timeout 300 python bench.py --workload config5-loop --steps 20 2>/dev/null | tail -1 > gpurun_out/r2_l/loop5_bench_line.json; cut -c1-330 gpurun_out/r2_l/loop5_bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_l/trace -o run -- python bench.py --workload config5-loop --steps 20 > gpurun_out/r2_l/under_rocprof.log 2>&1
cp $(find gpurun_out/r2_l/trace -name '*kernel_stats.csv' | head -1) gpurun_out/r2_l/loop5_kernel_stats.csv; rm -rf gpurun_out/r2_l/trace
head -6 gpurun_out/r2_l/loop5_kernel_stats.csv | cut -c1-60,100-200

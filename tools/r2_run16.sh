export TMPDIR=/tmp
CASE=1 B=512 ROUNDS=4 VARIANTS="|-DMPX_MIN_WAVES_HIGH=1|-DMPX_MIN_WAVES_HIGH=3|-DMPX_MIN_WAVES_HIGH=5|-DMPX_MIN_WAVES_HIGH=6" timeout 900 python tools/ab.py 2>&1 | grep variant
mkdir -p gpurun_out/r2_o
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_o/trace -o run -- python bench.py --workload config3-fgj --no-cpu-baseline --no-extras > gpurun_out/r2_o/c3_under_rocprof.log 2>&1
cp $(find gpurun_out/r2_o/trace -name '*kernel_stats.csv' | head -1) gpurun_out/r2_o/c3_kernel_stats.csv; rm -rf gpurun_out/r2_o/trace
head -6 gpurun_out/r2_o/c3_kernel_stats.csv | cut -c1-50,95-200

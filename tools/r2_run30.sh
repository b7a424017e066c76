export TMPDIR=/tmp
mkdir -p gpurun_out/r2_u
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_shard.py tests/test_residuals.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --workload config3-fgj --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3 direct', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'])"
MPX_STAGE_ALL=1 timeout 300 python bench.py --workload config3-fgj --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3 stage_all', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'])"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_u/trace -o run -- python bench.py --workload config3-fgj --no-cpu-baseline --no-extras > gpurun_out/r2_u/c3_under_rocprof.log 2>&1
cp $(find gpurun_out/r2_u/trace -name '*kernel_stats.csv' | head -1) gpurun_out/r2_u/c3_kernel_stats.csv; rm -rf gpurun_out/r2_u/trace
head -5 gpurun_out/r2_u/c3_kernel_stats.csv | cut -c1-50,95-200

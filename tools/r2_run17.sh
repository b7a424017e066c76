export TMPDIR=/tmp
mkdir -p gpurun_out/r2_p
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -m gpu -x -q 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_p/trace -o run -- python bench.py --workload config3-fgj --no-cpu-baseline --no-extras > gpurun_out/r2_p/c3_under_rocprof.log 2>&1
cp $(find gpurun_out/r2_p/trace -name '*kernel_stats.csv' | head -1) gpurun_out/r2_p/c3_kernel_stats.csv; rm -rf gpurun_out/r2_p/trace
head -5 gpurun_out/r2_p/c3_kernel_stats.csv | cut -c1-50,95-200
grep '^{' gpurun_out/r2_p/c3_under_rocprof.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'])"

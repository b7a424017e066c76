mkdir -p gpurun_out/r2_n
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --workload config3-fgj --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_us'])"; done
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['roofline']['frac'], d['roofline'].get('frac_placement_median'), d['extras']['placement_sweep_node_kernel_us'])"

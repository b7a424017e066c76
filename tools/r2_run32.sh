mkdir -p gpurun_out/r2_w
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_host.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tuned', round(d['value']), r['frac'], r.get('frac_placement_median'), r.get('frac_placement_min'), r.get('frac_placement_max'), d['extras']['placement_sweep_node_kernel_us'])"
MPX_NO_TUNE=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bpb1 ', round(d['value']), r['frac'], r.get('frac_placement_median'), r.get('frac_placement_min'), r.get('frac_placement_max'), d['extras']['placement_sweep_node_kernel_us'])"
done
timeout 300 python bench.py --workload config3-fgj --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3 tuned', round(d['value']), d['roofline']['frac'])"
MPX_NO_TUNE=1 timeout 300 python bench.py --workload config3-fgj --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C3 bpb1', round(d['value']), d['roofline']['frac'])"

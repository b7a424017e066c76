mkdir -p gpurun_out/r2_r
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_r/pytest_gpu.log 2>&1; tail -4 gpurun_out/r2_r/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r2_r/bench_line.json 2>gpurun_out/r2_r/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_r/bench_line.json'))
print('value', d['value'], 'frac', d['roofline']['frac'], 'median', d['roofline'].get('frac_placement_median'), d['roofline'].get('frac_placement_min'), d['extras']['placement_sweep_node_kernel_us'])
print('ipopt', d['ipopt_iter']['us_per_iter'], d['ipopt_iter']['per_call_us'], d['ipopt_iter']['cpu_port_us_per_iter'])
print('ipopt0', d['ipopt_iter_config0']['us_per_iter'], d['ipopt_iter_config0']['per_call_us'])
PY
MPX_NO_FOLD=1 timeout 300 python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, problems
r=bench.ipopt_iter_report(problems.moon_lander,1000,5,"LGR",["moon_lander"],1.0,[1],0,seconds=0.4); print('nofold c2', r['us_per_iter'], r['per_call_us'])
r=bench.ipopt_iter_report(problems.moon_lander,20,3,"LGR",["moon_lander"],1.0,[1],0,seconds=0.3); print('nofold c0', r['us_per_iter'], r['per_call_us'])
PY

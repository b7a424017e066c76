mkdir -p gpurun_out/r2_t
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t/pytest_gpu.log 2>&1; tail -3 gpurun_out/r2_t/pytest_gpu.log
MPX_LAT_DEBUG=1 timeout 300 python - <<'PY' 2>&1 | grep -E "c0|c2|zero-copy" | tail -6
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, problems
r=bench.ipopt_iter_report(problems.moon_lander,20,3,"LGR",["moon_lander"],1.0,[1],0,seconds=0.3); print('c0', r['us_per_iter'], r['per_call_us'])
r=bench.ipopt_iter_report(problems.moon_lander,1000,5,"LGR",["moon_lander"],1.0,[1],0,seconds=0.3); print('c2', r['us_per_iter'], r['per_call_us'])
PY
MPX_NO_FOLD=1 timeout 300 python - <<'PY' 2>&1 | grep -E "c0|c2" | tail -6
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, problems
r=bench.ipopt_iter_report(problems.moon_lander,20,3,"LGR",["moon_lander"],1.0,[1],0,seconds=0.3); print('nofold c0', r['us_per_iter'], r['per_call_us'])
r=bench.ipopt_iter_report(problems.moon_lander,1000,5,"LGR",["moon_lander"],1.0,[1],0,seconds=0.3); print('nofold c2', r['us_per_iter'], r['per_call_us'])
PY

MPX_LAT_DEBUG=1 timeout 300 python - <<'PY' 2>&1 | grep -E "zero-copy|c0|c2" | tail -14
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, problems
r=bench.ipopt_iter_report(problems.moon_lander,20,3,"LGR",["moon_lander"],1.0,[1],0,seconds=0.3); print('c0', r['us_per_iter'], r['per_call_us'])
r=bench.ipopt_iter_report(problems.moon_lander,1000,5,"LGR",["moon_lander"],1.0,[1],0,seconds=0.3); print('c2', r['us_per_iter'], r['per_call_us'])
PY

mkdir -p gpurun_out/r2_s
for w in config2-fgj config3-fgj config5-hess config2-hess config5-loop adaptive-fgj; do
timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 30 2>/dev/null | tail -1 > gpurun_out/r2_s/$w.json
python - <<PY
import json
d=json.load(open('gpurun_out/r2_s/$w.json'))
print('$w', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'frac', d['roofline']['frac'], 'kernel_us', round(d['roofline']['kernel_us'],1), d['roofline'].get('frac_placement_median'))
PY
done
BS=1,64,512 timeout 600 python tools/configs.py > gpurun_out/r2_s/configs.txt 2>&1; tail -30 gpurun_out/r2_s/configs.txt

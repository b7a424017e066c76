mkdir -p gpurun_out/r5f
for wl in config5-fgj config4-fgj; do
  for b in tuned 1 2 4 8 16; do
    if [ $b = tuned ]; then e=""; else e="MPX_BPB=$b"; fi
    r=$(env $e python bench.py --workload $wl --plain-outputs --no-cpu-baseline --no-extras --steps 60 --warmup 10 --ramp-seconds 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step']*1000,1), round(d['roofline']['kernel_us'],1), round(d['roofline']['frac'],3))")
    echo "$wl bpb=$b: step_us kernel_us frac = $r" >> gpurun_out/r5f/bpb_sweep.txt
  done
done
cat gpurun_out/r5f/bpb_sweep.txt

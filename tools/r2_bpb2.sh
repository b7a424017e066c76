for w in config3-fgj config5-hess config2-hess; do for b in 0 1 2; do
  if [ $b = 0 ]; then unset MPX_BPB; else export MPX_BPB=$b; fi
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extras --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json,os; d=json.loads(sys.stdin.read()); print('$w bpb=${b}', round(d['value']), round(d['roofline']['kernel_us'],1), round(d['roofline']['frac'],3))"
done; done

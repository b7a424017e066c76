#!/bin/bash
# Same-box A/B of the nlp_grad pass: the tree of the previous commit (a git worktree under _old_tree/, built beforehand) against this one, interleaved twice.
O=gpurun_out/r6_nlp_grad_ab; mkdir -p $O
for r in 1 2; do
  (cd _old_tree && timeout 300 python tools/r6_nlp_grad_bench.py 2>/dev/null) > $O/before_run$r.jsonl
  timeout 300 python tools/r6_nlp_grad_bench.py 2>/dev/null > $O/after_run$r.jsonl
done
python - <<'PY'
import json
O="gpurun_out/r6_nlp_grad_ab"
rd=lambda f:[json.loads(l) for l in open(f) if l.strip()]
for r in (1,2):
    b,a=rd(f"{O}/before_run{r}.jsonl"),rd(f"{O}/after_run{r}.jsonl")
    for x,y in zip(b,a):
        print(f"run {r}: {x['workload'][:48]:50s} B={x['batch']:5d}  {x['us_per_pass']:8.1f} -> {y['us_per_pass']:8.1f} us   frac {x['frac_of_8TBps']:.3f} -> {y['frac_of_8TBps']:.3f}")
PY

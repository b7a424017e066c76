#!/bin/bash
# HBM traffic of the nlp_grad pass (configs[1], B = 4096): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes -> gpurun_out/r6_nlp_grad_pmc/traffic.json
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_nlp_grad_pmc; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 280 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o run -- python tools/r6_nlp_grad_bench.py config2 > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc_$c -name '*counter_collection.csv' | head -1)
  grep -E "Counter_Name|mpx_node_gradl|mpx_gradl_finish" "$f" | head -400 > $O/pmc_$(echo $c | tr A-Z a-z).csv; rm -rf $O/pmc_$c
done
python - "$O" <<'PY'
import csv, json, sys
out = sys.argv[1]
res = {}
for kern in ("mpx_node_gradl_0_5", "mpx_gradl_finish"):
    vals = {}
    for c in ("fetch_size", "write_size"):
        rows = [r for r in csv.DictReader(open(f"{out}/pmc_{c}.csv")) if r["Kernel_Name"].startswith(kern)]
        big = max(int(r["Grid_Size"]) for r in rows)
        v = sorted(float(r["Counter_Value"]) for r in rows if int(r["Grid_Size"]) == big)
        vals[c] = v[len(v) // 2]
    res[kern] = {"FETCH_SIZE_KB": vals["fetch_size"], "WRITE_SIZE_KB": vals["write_size"], "bytes_per_launch": (2 * vals["fetch_size"] + vals["write_size"]) * 1024}
alg = 376120 * 4096
tot = sum(v["bytes_per_launch"] for v in res.values())
d = {"workload": "nlp_grad, configs[1] moon lander 1000x5, B = 4096", "kernels": res, "bytes_per_pass": tot, "algorithmic_bytes_per_pass": alg, "traffic_over_algorithmic": tot / alg,
     "note": "2 x FETCH_SIZE + WRITE_SIZE per kernel (median over the launches of the largest grid), separate rocprofv3 --pmc passes; KB = 1024 B; FETCH doubled per MI355X_MICROARCH.md"}
json.dump(d, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(d))
PY

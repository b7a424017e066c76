#!/bin/bash
# A/B of the assembled-path point kernels on one GPU: workgroup-count threshold above which a lane takes MPX_PTS_UNROLL evaluation
# points together (MPX_PTS_MIN_WG), and the unroll factor itself (MPX_HIPCC_FLAGS=-DMPX_PTS_UNROLL=n overrides the generated one).
mkdir -p gpurun_out/asm_ab
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 400 python tools/adaptive_bench.py moon_lander hyper_sensitive > gpurun_out/asm_ab/$tag.json 2> gpurun_out/asm_ab/$tag.err
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for l in open(f"gpurun_out/asm_ab/{tag}.json"):
    try: d = json.loads(l)
    except Exception: continue
    print(tag, d["case"].split()[0], "fgj", d["fgj"]["ms_per_batch"], "hess", d["hess"]["ms_per_batch"], "lat", d["latency_us_fgj"], d["latency_us_hess"])
PY
}
for rep in 1 2; do
run never MPX_PTS_MIN_WG=1000000000
run t16384 MPX_PTS_MIN_WG=16384
run t4096 MPX_PTS_MIN_WG=4096
run t1024 MPX_PTS_MIN_WG=1024
done

#!/bin/bash
# A/B of the assembled-path kernels on one GPU: gather unroll (MPX_GATHER_U), points per lane in the point kernels
# (MPX_PTS_UNROLL at JIT time + MPX_PTS_BPB on the host), set descriptors by value (MPX_NO_SETS_V switches it off).
mkdir -p gpurun_out/asm_ab
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 400 python tools/adaptive_bench.py > gpurun_out/asm_ab/$tag.json 2> gpurun_out/asm_ab/$tag.err
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for l in open(f"gpurun_out/asm_ab/{tag}.json"):
    try: d = json.loads(l)
    except Exception: continue
    print(tag, d["case"].split()[0], "fgj", d["fgj"]["ms_per_batch"], "hess", d["hess"]["ms_per_batch"], "lat", d["latency_us_fgj"], d["latency_us_hess"], "build_s", d["build_s"])
PY
}
run base X=1
run p4 MPX_HIPCC_FLAGS=-DMPX_PTS_UNROLL=4 MPX_PTS_BPB=4
run p8 MPX_HIPCC_FLAGS=-DMPX_PTS_UNROLL=8 MPX_PTS_BPB=8
run p16 MPX_HIPCC_FLAGS=-DMPX_PTS_UNROLL=16 MPX_PTS_BPB=16
run p8b16 MPX_HIPCC_FLAGS=-DMPX_PTS_UNROLL=8 MPX_PTS_BPB=16
run base2 X=1

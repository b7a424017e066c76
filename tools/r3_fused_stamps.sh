#!/bin/bash
# Phase stamps of the fused assembled kernels (chunk phases + one point task): MPX_FUSE_DEBUG=1 / -DMPX_FUSE_PT_STAMPS on the bench workload
for fl in "-DMPX_FUSE_PT_STAMPS=1" ${EXTRA_VARIANTS:-}; do
  echo "== $fl"
  MPX_HIPCC_FLAGS="$fl" MPX_FUSE_DEBUG=1 MPX_FUSE_PT_STAMPS=1 timeout 300 python bench.py --workload adaptive-fgj --no-cpu-baseline --no-extras --steps 3 --warmup 1 --ramp-seconds 0.1 2>&1 | grep -A1 "fused mode" | tail -4
done

#!/bin/bash
# A/B of MPX_TILE (lanes per workgroup tile) on the GPU box: rebuilds library + kernels in place for each value.
for t in 256 512 256 512 1024; do
  sed -i "s/^#define MPX_TILE [0-9]*/#define MPX_TILE $t/" mpopt_amd/csrc/mpx_device.h
  python -c "from mpopt_amd import _lib; _lib.build_library(force=True)" 2>/dev/null
  python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MPX_TILE=$t', round(d['value']), round(d['roofline']['kernel_us'],1), round(d['roofline']['frac'],3))"
done
sed -i "s/^#define MPX_TILE [0-9]*/#define MPX_TILE 256/" mpopt_amd/csrc/mpx_device.h

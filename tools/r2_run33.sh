timeout 900 python -m pytest tests/test_gpu_adaptive.py -m gpu -x -q 2>&1 | tail -3
echo direct; timeout 300 python tools/adaptive_bench.py 2>&1 | grep case | cut -c1-45,280-700
echo legacy; MPX_ASM_NO_DIRECT=1 timeout 300 python tools/adaptive_bench.py 2>&1 | grep case | cut -c1-45,280-700

mkdir -p gpurun_out/r2_e
python tools/latency.py > gpurun_out/r2_e/latency_zc.txt 2>&1; cat gpurun_out/r2_e/latency_zc.txt
MPX_NO_ZERO_COPY=1 python tools/latency.py > gpurun_out/r2_e/latency_nozc.txt 2>&1; grep -v amdgpu.ids gpurun_out/r2_e/latency_nozc.txt | head -14
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_e/pytest_gpu.log 2>&1; tail -5 gpurun_out/r2_e/pytest_gpu.log

set -u
export TMPDIR=/tmp
pw() { timeout 500 bash tools/profile_workload.sh "$@" > /dev/null 2>&1; }
pw r6_final/deg100_light_g config2-fgj mpx_lighthigh_fg_0_100 --segments 50 --degree 100 --batch 512 --oracles g
pw r6_final/deg255_light_g config2-fgj mpx_lighthigh_fg_0_255 --segments 20 --degree 255 --batch 512 --oracles g
pw r6_final/deg100_light_f_grad_f config2-fgj mpx_lighthigh_fgq_0_100 --segments 50 --degree 100 --batch 512 --oracles f,grad_f
bash tools/r6_light_high.sh 2>&1 | grep "matrix cores\|node kernels"
bash tools/r5_counters.sh gpurun_out/r6_light_high/counters mpx_lighthigh_fg_0_100 --segments 50 --degree 100 --batch 512 --oracles g 2>&1 | grep "MFMA\|GRBM_GUI\|SQ_WAIT_ANY\"\|SQ_WAVE_CYCLES"
timeout 1500 python -m pytest tests/test_gpu_high_degree.py tests/test_gpu_parity.py -x -q 2>&1 | tail -1

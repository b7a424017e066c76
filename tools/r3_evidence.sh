#!/bin/bash
# Round-3 evidence in one GPU call: headline profile (kernel stats + PMC traffic), config-3 profiles (hess_l over node-ordered tiles,
# f+g+grad_f+jac_g), the per-oracle x config x batch report, one bench line per secondary workload, the RCCL smoke test and a
# 2-rank bench line over gloo (three sharding modes + rccl census), the config-5 loop profile, the assembled-path PMC traffic, the
# single-oracle geometry table.  Everything lands under gpurun_out/r3_final/.
set -u
export TMPDIR=/tmp
o=gpurun_out/r3_final; mkdir -p $o
timeout 600 bash tools/profile_bench.sh r3_final/headline > $o/headline.log 2>&1
timeout 600 bash tools/profile_workload.sh r3_final/c3_hess config3-hess mpx_node_hessn > $o/c3_hess.log 2>&1
timeout 600 bash tools/profile_workload.sh r3_final/c3_fgj config3-fgj mpx_node_fgj_0_30 > $o/c3_fgj.log 2>&1
for w in config2-hess config5-hess config3-fgj config3-hess config5-loop adaptive-fgj; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $o/bench_line_$w.json
done
timeout 900 python tools/report.py > $o/r3_report.md 2> $o/report.err
timeout 900 bash tools/r3_loop_profile.sh r3_final/config5_loop > $o/config5_loop.log 2>&1
timeout 600 bash tools/r3_adaptive_pmc.sh gpurun_out/r3_final/adaptive > $o/adaptive.log 2>&1
timeout 300 python tools/r3_single_oracle_bpb.py > $o/single_oracles_bpb.txt 2>&1
timeout 120 python tools/rccl_smoke.py > $o/rccl_smoke.txt 2>&1
MPX_DIST_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 --batch 1024 2>/dev/null | tail -1 > $o/bench_line_2ranks_gloo.json
timeout 600 python bench.py > $o/bench_line_default.json 2> $o/bench_default.err
ls -la $o | head -40

mkdir -p gpurun_out/r2_g
for w in config3-shard config4-shard; do
MPX_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --workload $w --steps 20 > gpurun_out/r2_g/shard2_${w}_gloo.log 2>&1; tail -1 gpurun_out/r2_g/shard2_${w}_gloo.log | cut -c1-900
timeout 300 python bench.py --workload $w --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
done

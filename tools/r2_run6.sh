mkdir -p gpurun_out/r2_f
python bench.py > gpurun_out/r2_f/bench_line.json 2> gpurun_out/r2_f/bench.err; tail -3 gpurun_out/r2_f/bench.err; cat gpurun_out/r2_f/bench_line.json
python bench.py --workload config3-shard --steps 20 > gpurun_out/r2_f/shard1.json 2>&1; tail -1 gpurun_out/r2_f/shard1.json | cut -c1-600
MPX_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload config3-shard --steps 20 > gpurun_out/r2_f/shard2_gloo.json 2>&1; tail -1 gpurun_out/r2_f/shard2_gloo.json | cut -c1-600
MPX_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --workload config4-shard --steps 20 > gpurun_out/r2_f/shard2_c4_gloo.json 2>&1; tail -1 gpurun_out/r2_f/shard2_c4_gloo.json | cut -c1-600

MPX_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], json.dumps(d.get('segment_shard'))[:400])"
timeout 300 python bench.py --workload config2-hess --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-100
timeout 300 python bench.py --steps 10 2>/dev/null | tail -1 | cut -c1-200

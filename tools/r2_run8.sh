mkdir -p gpurun_out/r2_h
MPX_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 10 --warmup 2 > gpurun_out/r2_h/n2_gloo.log 2>&1; tail -1 gpurun_out/r2_h/n2_gloo.log | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print(d['value'], d['n_gpus'], json.dumps(d.get('segment_shard')))
except Exception as e: print('ERR', e, l[-800:])"

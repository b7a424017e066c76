export TMPDIR=/tmp
BS=4096 timeout 600 python tools/configs.py 2>&1 | grep -v amdgpu
mkdir -p gpurun_out/r2_v
cat > /tmp/c4.py <<'PY'
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import mpopt_amd as M
from mpopt_amd import mp
import problems
dev = torch.device("cuda:0")
builder, S, po, scheme = problems.BENCH_CASES[2]
mpo = mp.mpopt(builder(mp, M.math), S, po, scheme); o = mpo.create_nlp()[0]["oracle"]
B = 4096
Z = torch.tensor(mpo.initialize_solution()[None, :] + 0.01 * np.random.default_rng(1).standard_normal((B, o.n_z)), device=dev)
p = torch.tensor(np.full(o.n_p, 1.0 / S), device=dev)
f = torch.empty(B, dtype=torch.float64, device=dev); g = torch.empty(B, o.n_g, dtype=torch.float64, device=dev)
gr = torch.empty(B, o.n_z, dtype=torch.float64, device=dev); jv = torch.empty(B, o.nnz_jac, dtype=torch.float64, device=dev)
for _ in range(200): o.eval_device(15, B, Z, p, 0, None, None, f, g, gr, jv, None)
o.sync()
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_v/trace -o run -- python /tmp/c4.py > /dev/null 2>&1
cp $(find gpurun_out/r2_v/trace -name '*kernel_stats.csv' | head -1) gpurun_out/r2_v/c4_kernel_stats.csv; rm -rf gpurun_out/r2_v/trace
head -6 gpurun_out/r2_v/c4_kernel_stats.csv | cut -c1-110

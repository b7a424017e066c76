"""``python bench.py --gpus N`` launches its N ranks itself (one process per GPU under torch.distributed.run on 127.0.0.1) when it
is not already running under a launcher; rank 0 prints the ONE JSON line with ``n_gpus: N``.  SURVEY 8(e): segments / evaluation
points shard because a node reads only its own segment (reference mpopt.py:189-198)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_gpus_2_spawns_two_ranks_and_rank0_prints_one_line():
    """No GPU needed: the launch itself (argument handling, re-exec under torch.distributed.run, rendezvous on 127.0.0.1, a real
    all-reduce over gloo, one line from rank 0) up to the point where the benchmark would touch the device."""
    r, lines = run_bench(["--gpus", "2", "--launch-check"], {"MPX_DIST_BACKEND": "gloo"}, 300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "n_gpus": 2, "backend": "gloo", "all_reduce_of_ones": 2.0}


@pytest.mark.parametrize("n", [4, 8])
def test_gpus_4_and_8_launch_over_gloo(n):
    """The driver's round-end scaling run is `--gpus 1, 2, 4, 8`: the self-launch, the rendezvous and one real all-reduce with 4
    and 8 ranks (gloo, no GPU) -- the widest launch the bench will see on an 8-GPU node."""
    r, lines = run_bench(["--gpus", str(n), "--launch-check"], {"MPX_DIST_BACKEND": "gloo"}, 600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0]) == {"launch_check": True, "n_gpus": n, "backend": "gloo", "all_reduce_of_ones": float(n)}


def test_gpus_1_stays_a_single_process():
    r, lines = run_bench(["--gpus", "1", "--launch-check"], {}, 120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(lines[0])["n_gpus"] == 1


def test_rccl_launch_refuses_more_ranks_than_gpus():
    """Over RCCL a rank needs its own GPU: asking for more than are visible fails before anything is launched."""
    import torch

    n = torch.cuda.device_count()
    r, lines = run_bench(["--gpus", str(max(n, 1) + 1), "--launch-check"], {"MPX_DIST_BACKEND": "nccl"}, 120)
    assert r.returncode != 0 and not lines and "visible" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_gpus_2_runs_the_benchmark_on_one_gpu_over_gloo():
    """For real, on the test GPU: two ranks (sharing the device, MPX_DIST_BACKEND=gloo) run the headline workload for a few steps;
    the line says n_gpus = 2, carries the rank census and the segment-shard object of the N > 1 lines."""
    r, lines = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "256", "--ramp-seconds", "0.2", "--no-cpu-baseline"],
                         {"MPX_DIST_BACKEND": "gloo"}, 900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3 and out["value"] > 0
    assert out["rccl"]["world"] == 2 and out["rccl"]["all_reduce_of_ones"] == 2.0 and len(out["rccl"]["ranks"]) == 2
    modes = out["segment_shard"]["modes"]
    assert set(modes) == {"allgather", "root", "owner"} and all(m["bit_identical_to_unsharded"] for m in modes.values())

"""B=1 call latency through nlp_* (bench.ipopt_iter_report) -- run before / after `rocm-smi --setperflevel high` to see how much of
the single-evaluation floor is the GPU's idle clock state."""
import json
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import bench
import problems

for tag, cfg in (("configs[0]", (problems.moon_lander, 20, 3, "LGR", ["moon_lander"], 1.0, [1])),
                 ("configs[1]", (problems.moon_lander, 1000, 5, "LGR", ["moon_lander"], 1.0, [1]))):
    r = bench.ipopt_iter_report(*cfg, 0, seconds=0.5)
    print(json.dumps({"case": tag, "us_per_iter": round(r["us_per_iter"], 1), "per_call_us": r["per_call_us"]}), flush=True)

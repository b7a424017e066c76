"""profiles/r6_final/README.md from the files tools/r6_evidence.sh left there.  python tools/r6_final_readme.py"""
import csv, glob, json, os, re
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r6_final") + "/"
bl = lambda f: json.load(open(R + f))
runs = [bl(f"bench_line_default_run{k}.json") for k in range(1, 6)]
head = bl("bench_line_default.json")
rows = []
LABEL = {"headline": "headline: config 2 f+g+grad_f+jac_g", "c2_hess": "config 2 hess_l", "c3_fgj": "config 3 f+g+grad_f+jac_g (B=512)", "c3_hess": "config 3 hess_l (B=2048)",
         "c4_fgj": "config 4 f+g+grad_f+jac_g (all phases in one launch)", "c4_hess": "config 4 hess_l", "c5_fgj": "config 5 f+g+grad_f+jac_g", "c5_hess": "config 5 hess_l",
         "adaptive_fgj": "mpopt_adaptive 20x5 f+g+grad_f+jac_g", "adaptive_hess": "mpopt_adaptive 20x5 hess_l",
         "deg100_fgj": "moon lander 50 x degree 100, f+g+grad_f+jac_g (B=512; streamed tables)", "deg255_fgj": "moon lander 20 x degree 255, f+g+grad_f+jac_g (B=512; streamed tables)",
         "deg100_light_g": "moon lander 50 x degree 100, nlp_g alone (B=512; matrix cores, evaluation points as a matrix dimension -- bound by the FP64 matrix pipes, not HBM)",
         "deg255_light_g": "moon lander 20 x degree 255, nlp_g alone (B=512; the same)", "deg100_light_f_grad_f": "moon lander 50 x degree 100, nlp_f + nlp_grad_f alone (B=512)"}
for c in "2345":
    for sel, nm in (("f", "nlp_f"), ("g", "nlp_g"), ("f_grad_f", "nlp_f + nlp_grad_f")):
        LABEL[f"c{c}_light_{sel}"] = f"config {c} {nm} alone" + (" (B=512)" if c == "3" else "")
for d in ["headline", "c2_hess", "c5_fgj", "c5_hess", "c4_fgj", "c4_hess", "c3_fgj", "c3_hess", "adaptive_fgj", "adaptive_hess", "deg100_fgj", "deg255_fgj", "deg100_light_g", "deg255_light_g", "deg100_light_f_grad_f"] + [f"c{c}_light_{s}" for c in "2345" for s in ("f", "g", "f_grad_f")]:
    if not os.path.exists(R + d + "/traffic.json"):
        continue
    t = json.load(open(R + d + "/traffic.json")); b = json.load(open(R + d + "/bench_line.json")); r = b["roofline"]
    ks = [k for k in csv.DictReader(open(R + d + "/kernel_stats.csv")) if k["Name"].startswith("mpx_")]
    tb = t.get("bytes_per_launch")
    kern = ks[0]["Name"] + f" {float(ks[0]['AverageNs']) / 1000:.1f}" if ks else "—"
    rows.append(f"| {LABEL.get(d, d)} | {b['value']:.4g} | {b['ms_per_step'] * 1000:.1f} | `{kern}` | {r['frac']:.3f} | {tb / r['algorithmic_bytes_per_launch']:.3f} | "
                f"{tb / (r['kernel_us'] * 1e-6) / 8e12:.3f} | `{d}/` |")
loop = [(f, bl(f)) for f in ("config5_loop/bench_line.json", "bench_line_config5-loop_B2048.json", "bench_line_config5-loop_B4096.json") if os.path.exists(R + f)]
lt = json.load(open(R + "config5_loop/traffic.json"))
m = re.search(r"(\d+) passed", open(R + "gpu_tests_full_suite.log").read())
txt = f"""# r6_final — round-6 evidence (`tools/r6_evidence.sh`, one GPU call; MI355X, ROCm 7.2)

Every directory: `bench_line.json` (the bench line of the workload), `kernel_stats.csv` (`rocprofv3 --kernel-trace --stats` of the same
command), `pmc_fetch_size.csv` / `pmc_write_size.csv` (separate `rocprofv3 --pmc` passes), `traffic.json` (2 × FETCH_SIZE + WRITE_SIZE per
launch against the algorithmic bytes, corrections per MI355X_MICROARCH.md).  `gpu_tests_full_suite.log`: `pytest tests -m gpu` of the
tree ({m.group(1) if m else '?'} passed) with the per-entry parity summary by entry class.  (This file: `tools/r6_final_readme.py`.)
Every line carries `frac_by_traffic` next to `frac`.  New this round: the three lines on degrees 100 / 255 (streamed tables, DESIGN.md section 4).

## The headline five times in a row (one box, five processes; `bench_line_default_run1..5.json`)

| run | evals/s | roofline.frac | alloc_outputs: candidates (node-kernel µs per pass) → kept | stopped by |
|---|---|---|---|---|
""" + "\n".join(f"| {k + 1} | {d['value']:.4g} | {d['roofline']['frac']:.3f} | {d['alloc_outputs']['candidates_node_us_per_pass']} → #{d['alloc_outputs']['kept']} | {d['alloc_outputs']['stopped_by']} (target {d['alloc_outputs']['target_us']} µs) |" for k, d in enumerate(runs)) + f"""

Spread {100 * (max(d['value'] for d in runs) / min(d['value'] for d in runs) - 1):.1f} % over the five runs (`alloc_outputs` draws candidate allocations until one runs the pass within 3 % of its time
at 0.95 of the measured copy rate, 910.5 µs; DESIGN.md section 5).  `bench_line_default.json` = the plain `python bench.py` (driver's command;
{open(R + 'bench_default_time.txt').read().split()[1] if os.path.exists(R + 'bench_default_time.txt') else '?'} wall incl. the CPU baseline and the three `ipopt_iter` reports): {head['value']:.4g} evals/s, frac {head['roofline']['frac']:.3f}.

## Every workload, with the PMC traffic of its dominant kernel

| workload (B = 4096 unless noted) | value (evals/s) | µs per step | dominant kernel, average µs under rocprof | frac (algorithmic bytes / 8 TB/s) | PMC traffic / algorithmic | frac by traffic | dir |
|---|---|---|---|---|---|---|---|
""" + "\n".join(rows) + f"""

## Config-5 loop (`config5_loop/`: kernel stats, per-kernel PMC traffic of one outer iteration, three call sequences, phase stamps)

| batch | point-iterations/s | µs per step (5 outer iterations) | frac |
|---|---|---|---|
""" + "\n".join(f"| {d['config']['batch_per_gpu']} | {d['value']:.4g} | {d['ms_per_step'] * 1000:.1f} | {d['roofline']['frac']:.3f} |" for f, d in loop) + f"""

PMC traffic of one outer iteration / algorithmic bytes: {lt['traffic_over_algorithmic']:.3f}.

`bench_line_2ranks_gloo_self_launched.json`: `python bench.py --gpus 2` launching its own two ranks (gloo, both on the one GPU of the box).
"""
open(R + "README.md", "w").write(txt)
print(txt)

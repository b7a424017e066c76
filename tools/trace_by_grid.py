"""Median duration per (kernel, grid y) from a rocprofv3 kernel_trace.csv under a directory."""
import collections
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
d = collections.defaultdict(list)
for r in rows:
    d[(r["Kernel_Name"][:34], r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v = sorted(v)
    print(f"{k[0]:34s} grid {k[1]:>8s} x {k[2]:>6s}  n {len(v):5d}  median {v[len(v) // 2] / 1e3:9.2f} us")

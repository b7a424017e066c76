"""Per-run averages of PMC counters for the big node-kernel dispatches of tools/alloc_probe.py (13 launches per run: 3 warm + 10)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("mpx_node_fgj") and int(r["Grid_Size"]) > 1000000]
d = collections.OrderedDict()
for r in rows:
    d.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
vals = list(d.values())
names = sorted(vals[0])
print("run  " + "  ".join(f"{n[-28:]:>28s}" for n in names))
for k in range(0, len(vals), 13):
    grp = vals[k + 3:k + 13]
    if grp:
        print(f"{k // 13:3d}  " + "  ".join(f"{sum(v[n] for v in grp) / len(grp):28.4e}" for n in names))

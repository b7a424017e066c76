"""Print the head of a rocprofv3 kernel_stats.csv found under a directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(f'{r["Name"][:44]:44s} calls {r["Calls"]:>6s} total_ns {r["TotalDurationNs"]:>12s} avg_ns {float(r["AverageNs"]):12.1f} {float(r["Percentage"]):6.2f}%')

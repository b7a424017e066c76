"""Per-instance (XCD x TCC channel) PMC values of the big node-kernel dispatches from a rocprofv3 JSON output.
Writes <dir>/../chan_records.json (filtered, small) and prints a per-run table: for every counter the sum, the max
instance and max/mean over instances.  The schema of the JSON is probed defensively (printed when it is unexpected)."""
import collections
import glob
import json
import os
import sys

d = sys.argv[1]
files = glob.glob(d + "/**/*results.json", recursive=True) + glob.glob(d + "/**/*.json", recursive=True)
if not files:
    print("no json under", d)
    sys.exit(0)
J = json.load(open(files[0]))
tool = J["rocprofiler-sdk-tool"][0] if "rocprofiler-sdk-tool" in J else J
print("top-level keys:", list(tool.keys()))
counters = tool.get("counters", [])
print("n counters defs:", len(counters), "sample:", json.dumps(counters[:1])[:600])
cname = {}
for c in counters:
    cid = c.get("id", {})
    cid = cid.get("handle", cid) if isinstance(cid, dict) else cid
    cname[cid] = c.get("name")
ksym = {}
for k in tool.get("kernel_symbols", []):
    ksym[k.get("kernel_id")] = k.get("formatted_kernel_name") or k.get("kernel_name")
cc = None
for sect in ("callback_records", "buffer_records"):
    s = tool.get(sect, {})
    if isinstance(s, dict) and s.get("counter_collection"):
        cc = s["counter_collection"]
        print("counter_collection in", sect, "n =", len(cc))
        break
if not cc:
    print("no counter_collection records; sections:", {k: (list(v.keys()) if isinstance(v, dict) else type(v).__name__) for k, v in tool.items()})
    sys.exit(0)
print("sample record:", json.dumps(cc[0])[:1500])
keep = []
for r in cc:
    info = r.get("dispatch_data", {}).get("dispatch_info", {})
    name = ksym.get(info.get("kernel_id"), "")
    gs = info.get("grid_size", {})
    gx = gs.get("x", 0) * max(gs.get("y", 1), 1) if isinstance(gs, dict) else 0
    if "mpx_node_fgj" in str(name) and gx > 1000000:
        keep.append({"dispatch_id": info.get("dispatch_id"), "records": r.get("records", [])})
json.dump({"counters": cname, "dispatches": keep}, open(os.path.join(os.path.dirname(d.rstrip("/")), "chan_records.json"), "w"))
print("kept", len(keep), "big node-kernel dispatches")
# table: runs of 13 launches (3 warm + 10), as tools/alloc_probe.py issues them
for run in range(0, len(keep), 13):
    grp = keep[run + 3:run + 13]
    if not grp:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for disp in grp:
        inst = collections.Counter()
        for rec in disp["records"]:
            cid = rec.get("counter_id", {})
            cid = cid.get("handle", cid) if isinstance(cid, dict) else cid
            key = rec.get("id", inst[cid])  # record id encodes the dimension position when present
            inst[cid] += 1
            agg[cid][key if not isinstance(key, dict) else json.dumps(key)] += rec.get("value", 0.0) / len(grp)
    line = [f"run {run // 13:2d}"]
    for cid, per in agg.items():
        v = list(per.values())
        tot, mx, mean = sum(v), max(v), sum(v) / len(v)
        line.append(f"{cname.get(cid, cid)}: n_inst={len(v)} sum={tot:.4e} max={mx:.3e} max/mean={mx / mean if mean else 0:.2f}")
    print(" | ".join(line))

"""Average duration per (kernel, grid) of a rocprofv3 kernel_trace.csv: python tools/ktrace_shapes.py <trace.csv> <substring>[,<substring>...] [runs]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
subs = sys.argv[2].split(",")
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"]
    if any(s in nm for s in subs):
        short = nm.split("(")[0].split("::")[-1][:28]
        d[(short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(k, "calls/run", len(v) // runs, "avg us", round(sum(v) / len(v) / 1e3, 1), "ms/run", round(sum(v) / 1e6 / runs, 3))

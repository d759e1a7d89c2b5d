"""Average PMC counter values per launch for kernels whose name contains a substring (rocprofv3 counter_collection.csv)."""
import collections, csv, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in agg.items():
    print(f"{k:32s} {v[0] / v[1]:16.1f}  launches {v[1]}")

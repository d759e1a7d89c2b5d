"""Top rows of a rocprofv3 kernel_stats.csv found under a directory: python tools/kstats.py <dir> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])

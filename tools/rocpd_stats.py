"""Per-kernel statistics from a rocprofv3 `_results.db` (rocpd sqlite output of `--kernel-trace`): calls, average / median
duration, share of GPU time -- plus, with --step N, the per-kernel median over the LAST N dispatches of each name (the timed
decode steps at the end of a bench run) and the gaps between consecutive dispatches.
usage: python tools/rocpd_stats.py <results.db> [--last N] [--csv out.csv]"""
import argparse, re, sqlite3, statistics, sys

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--last", type=int, default=0, help="only the last N dispatches of the run (e.g. one decode step = 185)")
ap.add_argument("--csv", default=None)
args = ap.parse_args()
db = sqlite3.connect(args.db)
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
if args.last:
    rows = rows[-args.last:]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*\)$", "", n)
    return n[:90]


by = {}
for name, s, e in rows:
    by.setdefault(short(name), []).append((e - s) / 1e3)
total = sum(sum(v) for v in by.values())
gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1)]
out = []
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    out.append((n, len(v), sum(v), sum(v) / len(v), statistics.median(v), min(v), 100 * sum(v) / total))
print(f"{'kernel':90s} {'calls':>7s} {'total_us':>11s} {'avg_us':>8s} {'med_us':>8s} {'min_us':>8s} {'%':>6s}")
for r in out[:40]:
    print(f"{r[0]:90s} {r[1]:7d} {r[2]:11.1f} {r[3]:8.2f} {r[4]:8.2f} {r[5]:8.2f} {r[6]:6.2f}")
if gaps:
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print(f"dispatches {len(rows)}  span {span:.1f} us  kernel time {total:.1f} us  gaps: median {statistics.median(gaps):.2f} us, sum {sum(g for g in gaps if g > 0):.1f} us")
if args.csv:
    with open(args.csv, "w") as f:
        f.write("kernel,calls,total_us,avg_us,median_us,min_us,percent\n")
        for r in out:
            f.write(f"\"{r[0]}\",{r[1]},{r[2]:.1f},{r[3]:.3f},{r[4]:.3f},{r[5]:.3f},{r[6]:.2f}\n")

#!/usr/bin/env python3
"""Print a per-kernel summary of a rocprofv3 --kernel-trace --stats CSV directory (used to make profiles/*.txt)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}\n# total kernel time {tot/1e6:.2f} ms")
print(f"{'kernel':92s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{r['Name'][:92]:92s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.1f}")

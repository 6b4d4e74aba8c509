#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 --stats kernel_stats.csv (path or directory)."""
import csv, glob, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else sorted(glob.glob(p + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: total kernel time {tot/1e6:.3f} ms")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):5.1f}%")

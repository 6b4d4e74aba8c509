#!/usr/bin/env python3
"""Per-launch durations of the kernels whose name contains a pattern, from a rocprofv3 --kernel-trace csv directory:
   trace_kernel.py <dir> <pattern> [max rows]  ->  grid size, duration (us), in launch order, and the distinct (grid, mean) groups."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups = collections.defaultdict(list)
for r in rows:
    groups[(r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in groups.items():
    v2 = sorted(v)
    print(f"grid {k[0]:>8s} wg {k[1]:>4s}: n={len(v):4d} mean {sum(v)/len(v):8.1f} us  median {v2[len(v2)//2]:8.1f}  min {v2[0]:8.1f}  max {v2[-1]:8.1f}")

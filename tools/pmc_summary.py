#!/usr/bin/env python3
"""Average each PMC counter per dispatch of kernels whose name contains a substring."""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sub in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")

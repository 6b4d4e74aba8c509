#!/usr/bin/env python3
"""One replayed iteration out of a rocprofv3 kernel trace, as a timeline: kernel, queue, start offset, duration (us) -- and
the gaps in which no kernel of any queue runs.  usage: timeline.py <dir or kernel_trace.csv> [iteration index from the end]"""
import csv, glob, re, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else sorted(glob.glob(p + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# an iteration starts at the normal_ (noise) kernel that follows an rmsprop kernel
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:62]
opt = [i for i, r in enumerate(rows) if "rmsprop" in r["Kernel_Name"]]
# iterations: every second rmsprop (D then G) ends one
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
end = opt[-1 - 2 * k]
begin = opt[-1 - 2 * (k + 1)] + 1
seg = rows[begin:end + 1]
t0 = int(seg[0]["Start_Timestamp"])
busy_end = t0
gap = 0.0
print(f"{f}: iteration of {len(seg)} launches")
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = max(0, s - busy_end)
    gap += g
    busy_end = max(busy_end, e)
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} q{r['Queue_Id']:>2s} {'gap %5.1f' % (g / 1e3) if g > 500 else '         '} {short(r['Kernel_Name'])}")
print(f"span {(busy_end - t0) / 1e3:.1f} us, idle gaps {gap / 1e3:.1f} us")

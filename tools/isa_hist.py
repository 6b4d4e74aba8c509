#!/usr/bin/env python3
"""Instruction histogram of a kernel's hottest loop in hipcc -S output.

usage: isa_hist.py file.s <substring of the mangled kernel name> [depth]
Counts opcodes between the first and last line tagged `Depth=<depth>` (default: the deepest loop).
"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
depths = [int(m.group(1)) for l in body for m in [re.search(r"Depth=(\d+)", l)] if m]
depth = int(sys.argv[3]) if len(sys.argv) > 3 else max(depths)
tag = f"Depth={depth}"
idx = [i for i, l in enumerate(body) if tag in l]
# the loop body runs to the back-edge branch after the last tagged block
last = idx[-1]
while last + 1 < len(body) and not body[last + 1].startswith(".LBB") and "s_endpgm" not in body[last + 1]:
    last += 1
seg = body[idx[0]:last + 1]
ops = collections.Counter()
for l in seg:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    ops[t.split()[0]] += 1
tot = sum(ops.values())
print(f"{key}: kernel lines {len(body)}, loop depth {depth}: {tot} instructions, {ops.get('v_mfma_f32_32x32x16_f16', 0) + ops.get('v_mfma_f32_32x32x16_bf16', 0)} MFMA")
for k, v in ops.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 40):
    print(f"{v:6d} {k}")
spill = sum(1 for l in body if "scratch_" in l)
print(f"scratch instructions in kernel: {spill}")

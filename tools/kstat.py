#!/usr/bin/env python3
"""Per-kernel resource table of a HIP source: registers, scratch, LDS, instruction count (hipcc -S, gfx950).

usage: kstat.py file.hip [extra hipcc flags ...]   (run from anywhere; include paths as the library build uses them)
"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1])
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) if False else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fd, out = tempfile.mkstemp(suffix=".s")
os.close(fd)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", os.path.join(root, "include"),
       "-I", os.path.join(root, "mpgan_amd", "csrc"), "--cuda-device-only", "-S", src, "-o", out] + sys.argv[2:]
subprocess.check_call(cmd)
txt = open(out).read()
# instruction counts per function body
counts = {}
for m in re.finditer(r"^(_Z\w+):\n(.*?)s_endpgm", txt, re.S | re.M):
    body = m.group(2).split("\n")
    n = sum(1 for l in body if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":"))
    counts[m.group(1)] = (n, sum(1 for l in body if "scratch_" in l), sum(1 for l in body if "v_mfma" in l))
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, blk = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\S+)", blk) or [None, "?"])[1]
    n, sc, mf = counts.get(name, (0, 0, 0))
    dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem)[:110]
    print(f"{dem:110s} vgpr {g('next_free_vgpr'):>4s} accum_off {g('accum_offset'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s} instr {n:6d} scratch_ops {sc:4d} mfma {mf:5d}")
os.remove(out)

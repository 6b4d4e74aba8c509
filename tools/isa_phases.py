#!/usr/bin/env python3
"""Instruction mix of a kernel's ISA (hipcc -save-temps .s) per run of N MFMAs: where spills, register moves and
waits sit relative to the matrix work.  usage: isa_phases.py file.s [kernel-substring] [mfmas-per-row]"""
import sys
path, key, step = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ""), int(sys.argv[3]) if len(sys.argv) > 3 else 60
lines = open(path).read().split("\n")
i = 0
while i < len(lines):
    l = lines[i]
    if l.endswith(":") is False and ": " in l and l.startswith("_Z") and key in l.split(":")[0]:
        name = l.split(":")[0]
        body = []
        i += 1
        while i < len(lines) and "s_endpgm" not in lines[i]:
            t = lines[i].strip()
            if t and not t.startswith((".", ";")) and not t.endswith(":"):
                body.append(t.split()[0])
            i += 1
        print(name[:90], len(body), "instructions")
        cls = lambda op: ("mfma" if op.startswith("v_mfma") else "sc_ld" if op.startswith("scratch_load") else "sc_st" if op.startswith("scratch_store")
                          else "acc" if op.startswith("v_accvgpr") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_")
                          else "vmem" if op.startswith(("buffer_", "global_")) else "wait" if op.startswith("s_waitcnt") else "salu")
        row, nm = {}, 0
        for op in body:
            c = cls(op)
            if c == "mfma":
                nm += 1
                if nm % step == 0:
                    print(f"  mfma {nm:5d}:", " ".join(f"{k}={v}" for k, v in sorted(row.items())))
                    row = {}
            else:
                row[c] = row.get(c, 0) + 1
        print(f"  tail {nm:5d}:", " ".join(f"{k}={v}" for k, v in sorted(row.items())))
    i += 1

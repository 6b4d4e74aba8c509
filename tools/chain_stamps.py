#!/usr/bin/env python3
"""Phase timing of mpg_chain from a diagnostic build (s_memtime stamps of workgroup 0):
   MPG_HIPCC_FLAGS="-fno-slp-vectorize -DMPG_CHSTAMP" python -c "from mpgan_amd import _lib; _lib.build(force=True)"
   python tools/chain_stamps.py            (then rebuild without the flag)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import ops, _lib
exec(open(os.path.join(os.path.dirname(__file__), "kchain.py")).read().split("def timeit")[0])
lib = C.CDLL(_lib.LIBPATH)
general = bool(os.environ.get("MPG_CHAIN_GENERAL"))
if general:
    names = ["start", "staged", "sync0", "L0 mfma", "L0 epi", "L0 sync", "L1 mfma", "L1 epi", "L1 sync", "L2 mfma", "L2 epi", "L2 sync"]
    nw, ns, get = 8, 16, lib.mpg_debug_chain_stamps
else:   # chain2.hip: per layer  A = first tile's k loop, B = second tile's k loop (+ the first tile's epilogue), epi = exposed epilogue
    names = ["start", "staged", "sync0"] + [f"L{l} {x}" for l in range(3) for x in ("A", "B", "epi", "sync", "-")] + ["biases", "rows asked", "tiles asked"]
    nw, ns, get = 4, 24, lib.mpg_debug_chain2_stamps
for label, fn in (("a|c", t_ac), ("fn forward p=1/2", lambda: t_fn(128)), ("fn backward p=1/2", lambda: t_bw(128))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (nw * ns))()
    assert get(buf) == 0
    print(label, "(s_memtime ticks, about one per shader clock here; per wave, relative to the wave's start)")
    for w in range(nw):
        st = [buf[w * ns + i] for i in range(len(names))]
        print("  wave", w, " ".join(f"{names[i]}={st[i] - st[0]}" for i in range(1, len(names)) if st[i] and names[i][-1] != "-"))

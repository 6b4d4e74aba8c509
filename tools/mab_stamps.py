#!/usr/bin/env python3
"""Phase timing of mpg_mab_fwd from a diagnostic build (s_memtime stamps of workgroup 0):
   MPG_HIPCC_FLAGS="-fno-slp-vectorize -DMPG_MABSTAMP" python -c "from mpgan_amd import _lib; _lib.build(force=True)"
   python tools/mab_stamps.py            (then rebuild without the flag)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import _lib
from mpgan_amd.gapt import SAB
B = int(os.environ.get("MAB_B", "512"))
blk = SAB(embed_dim=64, ff_layers=[], final_linear=False, num_heads=4, layer_norm=False, dropout_p=0.0,
          linear_args={"leaky_relu_alpha": 0.2, "dropout_p": 0.0, "batch_norm": False, "spectral_norm": False}).cuda()
x = torch.randn(B, 30, 64, device="cuda")
with torch.no_grad():
    for _ in range(3):
        blk(x, None)
xg = x.clone().requires_grad_(True)
for _ in range(3):
    blk(xg, None).sum().backward()
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIBPATH)
buf = (C.c_ulonglong * 64)()
assert lib.mpg_debug_mab_stamps(buf) == 0
split = os.environ.get("MPG_MAB_SPLIT", "1") != "0" and B <= 512
if split:   # two waves per jet (mab_fwd2_kernel / mab_bwd2_kernel): waves 0, 1 = the first jet's pair
    names = ["start", "weights in LDS", "q k v", "attention", "o exchanged", "z", "z exchanged", "stored"]
    bnames = ["start", "weights in LDS", "du", "du exchanged", "dza exchanged", "attention", "dq dk dv exchanged", "stored"]
    n = 8
else:
    names = ["start", "weights in LDS", "rows loaded", "attention done", "out-projection done", "stored"]
    bnames = ["start", "weights in LDS", "rows loaded", "feed-forward half done", "attention + input gradients done", "stored"]
    n = 6
for w in range(4):
    st = [buf[w * 8 + i] for i in range(n)]
    print("forward  wave", w, " ".join(f"{names[i]}={st[i] - st[0]}" for i in range(1, n) if st[i]))
for w in range(4):
    st = [buf[32 + w * 8 + i] for i in range(n)]
    print("backward wave", w, " ".join(f"{bnames[i]}={st[i] - st[0]}" for i in range(1, n) if st[i]))

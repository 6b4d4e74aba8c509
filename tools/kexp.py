#!/usr/bin/env python3
"""Timing experiments on edge_fwd (debug knobs in skip_masked bits, sender-chunk override)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import ops
from mpgan_amd.mpgan import MPLayer
B, N, F = 256, 30, 32
dev = "cuda"
torch.manual_seed(0)
layer = MPLayer(F, [96, 160, 192], [256, 256], 32, dropout_p=0.0).to(dev)
x = torch.randn(B, N, F, device=dev) * 0.5
n = torch.clamp((torch.randn(B, device=dev) * 0.15 + 0.8) * N, 1, N).round()
mask = (torch.arange(N, device=dev)[None, :] < n[:, None]).float().unsqueeze(2)
orig_sc = ops._sender_chunks
for name, skipbits, sc in (("base", 1, None), ("noskip", 0, None), ("nofill", 3, None), ("w2lo_from_lds", 5, None),
                           ("nofill+w2lolds", 7, None), ("SC=1", 1, 1), ("SC=4", 1, 4), ("SC=1 noskip", 0, 1)):
    ops.OPTIONS["skip_masked"] = skipbits
    ops._sender_chunks = (lambda B, N, sc=sc: sc) if sc else orig_sc
    with torch.no_grad():
        for _ in range(3):
            layer(x, True, mask)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            layer(x, True, mask)
        e1.record()
        torch.cuda.synchronize()
    print(f"{name:18s} MPLayer fwd {e0.elapsed_time(e1)/10*1e3:8.1f} us")

#!/usr/bin/env python3
"""Time MPLayer fwd+bwd pieces at B=256,N=30 (events), with/without weight grads."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd.mpgan import MPLayer
B, N, F = 256, 30, 32
dev = "cuda"
torch.manual_seed(0)
for p in (0.0, 0.5):
    layer = MPLayer(F, [96, 160, 192], [256, 256], 32, dropout_p=p).to(dev)
    x = (torch.randn(B, N, F, device=dev) * 0.5).requires_grad_(True)
    n = torch.clamp((torch.randn(B, device=dev) * 0.15 + 0.8) * N, 1, N).round()
    mask = (torch.arange(N, device=dev)[None, :] < n[:, None]).float().unsqueeze(2)
    g = torch.randn(B, N, 32, device=dev)
    for needw in (True, False):
        for q in layer.parameters():
            q.requires_grad_(needw)
        def fb():
            y = layer(x, True, mask)
            y.backward(g)
        for _ in range(3): fb()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fb()
        e1.record(); torch.cuda.synchronize()
        print(f"p={p} need_w={needw}: fwd+bwd {e0.elapsed_time(e1)/10*1e3:8.1f} us")

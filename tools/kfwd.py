#!/usr/bin/env python3
"""Run only MPLayer forwards (for PMC profiling of edge_fwd_kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd.mpgan import MPLayer
B, N, F = 256, 30, 32
dev = "cuda"
torch.manual_seed(0)
layer = MPLayer(F, [96, 160, 192], [256, 256], 32, dropout_p=0.0).to(dev)
x = torch.randn(B, N, F, device=dev) * 0.5
n = torch.clamp((torch.randn(B, device=dev) * 0.15 + 0.8) * N, 1, N).round()
mask = (torch.arange(N, device=dev)[None, :] < n[:, None]).float().unsqueeze(2)
with torch.no_grad():
    for _ in range(6):
        layer(x, True, mask)
torch.cuda.synchronize()

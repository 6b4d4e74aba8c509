#!/usr/bin/env python3
"""Micro-benchmark of the fused MPLayer kernels at BASELINE config 2 (B=256, N=30).
Usage: python tools/kbench.py [B] [N]   -- prints per-kernel times (HIP events, 20 launches)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import ops
from mpgan_amd.mpgan import MPLayer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = "cuda"
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def flops(F):
    return B * N * N * 2 * (2 * F * 96 + 96 * 160 + 160 * 192)


for F, out, p in ((32, 32, 0.0), (32, 32, 0.5), (3, 32, 0.5)):
    layer = MPLayer(F, [96, 160, 192], [256, 256], out, dropout_p=p).to(dev)
    x = (torch.randn(B, N, F, device=dev) * 0.5).requires_grad_(True)
    n = torch.clamp((torch.randn(B, device=dev) * 0.15 + 0.8) * N, 1, N).round()
    mask = (torch.arange(N, device=dev)[None, :] < n[:, None]).float().unsqueeze(2)
    g = torch.randn(B, N, out, device=dev)
    with torch.no_grad():
        t = timeit(lambda: layer(x, True, mask))
    print(f"F={F} p={p}: MPLayer fwd (all kernels) {t:8.1f} us")

    def fb():
        y = layer(x, True, mask)
        y.backward(g)
    t = timeit(fb, 10)
    print(f"F={F} p={p}: MPLayer fwd+bwd (all kernels) {t:8.1f} us   edge algorithmic GFLOP fwd {flops(F)/1e9:.1f}")

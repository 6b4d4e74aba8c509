#!/usr/bin/env python3
"""MPLayer with its edge-feature / conditioning options on the two routes: the fused kernels (a scalar per edge and option,
coordinate differences folded into the node terms) against the un-fused route (edge matrix as the reference builds it, layer by
layer on the HIP GEMM).  One training-mode forward + backward of a single layer at the headline shape (B = 256, N = 30,
32 features), jets/s = B / time.  Usage: python tools/bench_variants.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd.mpgan import MPLayer
from mpgan_amd.data import synthetic_jets

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
data, labels = synthetic_jets(B, N, seed=4)
mask = (data[..., 3:4] + 0.5).to(dev)
labels = labels.to(dev)
njp = mask.mean(1)
torch.manual_seed(0)
x = (torch.randn(B, N, 32, device=dev) * 0.5).requires_grad_(True)
up = torch.randn(B, N, 32, device=dev)
cases = [("default", {}), ("delta_r (pos_diffs)", dict(pos_diffs=True)),
         ("delta_coords + delta_r", dict(pos_diffs=True, all_ef=False, delta_coords=True, delta_r=True)),
         ("clabels + mask_fne_np", dict(clabels=1, mask_fne_np=True))]
for name, kw in cases:
    res = {}
    for route in ("fused", "edges"):
        torch.manual_seed(1)
        layer = MPLayer(32, [96, 160, 192], [256, 256], 32, **kw).to(dev)
        assert layer.fused
        if route == "edges":
            layer.fused = False

        def step():
            x.grad = None
            layer.zero_grad(set_to_none=True)
            y = layer(x, True, mask, labels, njp)
            (y * up).sum().backward()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 10
        for _ in range(R):
            step()
        torch.cuda.synchronize()
        res[route] = (time.perf_counter() - t0) / R
    print(f"{name:28s} B={B} N={N}: fused {res['fused'] * 1e3:8.3f} ms ({B / res['fused']:10.0f} jets/s)   un-fused {res['edges'] * 1e3:8.3f} ms "
          f"({B / res['edges']:10.0f} jets/s)   x{res['edges'] / res['fused']:.1f}")

#!/usr/bin/env python3
"""Many iterations of the G+D step on fresh synthetic batches (hipGraph replay): losses and parameters must stay finite and the
losses must stay in the range a least-squares GAN lives in.  A robustness check for the GPU box, not a test of learning."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import train, data

for model, B, steps, N in (("mpgan", 256, 400, 30), ("gapt", 512, 1000, 30), ("mpgan", 16, 200, 150), ("gapt", 64, 200, 150)):
    torch.manual_seed(4)
    G, D = train.default_mpgan(N) if model == "mpgan" else train.default_gapt(N)
    lr_g, lr_d = (1e-5, 3e-5) if model == "mpgan" else train.LR_GAPT
    ts = train.TrainStep(G, D, B, N, latent=32 if model == "mpgan" else 64, lr_disc=lr_d, lr_gen=lr_g)
    lo, hi = [1e9, 1e9], [-1e9, -1e9]
    for it in range(steps):
        x, lab = data.synthetic_jets(B, N, seed=1000 + it)
        x, lab = x.cuda(), lab.cuda()
        ts.set_batch(x, lab)
        ts.step()
        if it % 50 == 49 or it == steps - 1:
            d, g = float(ts.D_loss), float(ts.G_loss)
            assert d == d and g == g and abs(d) < 10 and abs(g) < 10, (model, it, d, g)
            lo, hi = [min(lo[0], d), min(lo[1], g)], [max(hi[0], d), max(hi[1], g)]
    torch.cuda.synchronize()
    ts.check_range()
    for net in (G, D):
        for k, p in net.named_parameters():
            assert torch.isfinite(p).all(), (model, k)
    print(f"{model} N={N}: {steps} iterations at B={B}: D loss in [{lo[0]:.3f}, {hi[0]:.3f}], G loss in [{lo[1]:.3f}, {hi[1]:.3f}], parameters finite", flush=True)

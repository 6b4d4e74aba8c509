#!/usr/bin/env python3
"""Run-to-run determinism of the captured iteration: the same weights, data, noise and seed, replayed from scratch several times, must
end at the same bits -- a race inside a kernel (an early read of a staged buffer, a miscounted wait) shows up here as a difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import train, data, ops

def run(model, B, steps, dropout, N=30):
    torch.manual_seed(4)
    G, D = (train.default_mpgan(N, disc_dropout=dropout) if model == "mpgan" else train.default_gapt(N, disc_dropout=dropout))
    lr_g, lr_d = (1e-5, 3e-5) if model == "mpgan" else train.LR_GAPT
    ops.set_seed(1234, "cuda")
    import itertools
    ops.dev_state(torch.device("cuda:0")).tags = itertools.count(31)   # (dropout site tags are handed out per process: same ones every run)
    ts = train.TrainStep(G, D, B, N, latent=32 if model == "mpgan" else 64, lr_disc=lr_d, lr_gen=lr_g)
    x, lab = data.synthetic_jets(B, N, seed=77)
    ts.set_batch(x.cuda(), lab.cuda())
    for _ in range(steps):
        ts.step()
    torch.cuda.synchronize()
    return ts.fD.flat.clone(), ts.fG.flat.clone(), float(ts.D_loss), float(ts.G_loss)

# (N = 150: the sender-chunked launches, whose last-arriving workgroup adds the chunks' slabs -- in chunk order whoever it is --, and the
# large-set attention blocks, whose waves hand rows to each other through memory behind one barrier)
for model, B, steps, N in (("gapt", 512, 300, 30), ("gapt", 63, 100, 30), ("mpgan", 256, 60, 30), ("mpgan", 16, 40, 150), ("gapt", 32, 60, 150)):
    ref = run(model, B, steps, 0.5, N)
    for rep in range(3):
        got = run(model, B, steps, 0.5, N)
        assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and ref[2:] == got[2:], (model, B, N, rep)
    print(f"{model} B={B} N={N}: {steps} replayed iterations, 4 runs from scratch: identical bits (D loss {ref[2]:.4f}, G loss {ref[3]:.4f})", flush=True)

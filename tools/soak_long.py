import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import train, data
for (N, B, steps) in ((30, 256, 3000), (150, 16, 300), (30, 64, 500)):
    torch.manual_seed(4)
    G, D = train.default_mpgan(N)
    ts = train.TrainStep(G, D, B, N, latent=32, lr_disc=3e-5, lr_gen=1e-5)
    batches = [tuple(t.cuda() for t in data.synthetic_jets(B, N, seed=2000 + i)) for i in range(8)]
    for it in range(steps):
        ts.set_batch(*batches[it % 8])
        ts.step()
        if it % 250 == 249 or it == steps - 1:
            d, g = float(ts.D_loss), float(ts.G_loss)
            assert d == d and g == g and abs(d) < 10 and abs(g) < 10, (N, B, it, d, g)
    torch.cuda.synchronize()
    ts.check_range()
    assert all(bool(torch.isfinite(p).all()) for net in (G, D) for p in net.parameters())
    print(f"N={N} B={B}: {steps} iterations, losses D {float(ts.D_loss):.3f} G {float(ts.G_loss):.3f}, parameters finite, range guard clear")

#!/usr/bin/env python3
"""Count the device kernels one eager G+D iteration launches, by aten op (which PyTorch glue is left)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from mpgan_amd import train
from mpgan_amd.data import synthetic_jets
dev = torch.device("cuda:0")
G, D = (train.default_gapt if os.environ.get("OPC_GAPT") else train.default_mpgan)(30, device=dev)
BATCH = 512 if os.environ.get("OPC_GAPT") else 256
ts = train.TrainStep(G, D, BATCH, 30, latent=64 if os.environ.get("OPC_GAPT") else 32, use_graphs=False)
data, labels = synthetic_jets(BATCH, 30, seed=1, dist="gluon")
ts.set_batch(data.to(dev), labels.to(dev))
for _ in range(3): ts.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    ts.step()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[2])
for k, c, t in rows[:60]:
    print(f"{k[:70]:70s} n={c:4d} dev_us={t:9.1f}")
ours = lambda k: any(s in k for s in ("_kernel", "mpg_")) and "at::" not in k and "aten" not in k
n_ours = sum(c for k, c, _ in rows if ours(k)); n_other = sum(c for k, c, _ in rows if not ours(k))
t_ours = sum(t for k, _, t in rows if ours(k)); t_other = sum(t for k, _, t in rows if not ours(k))
print(f"library kernels: {n_ours} launches, {t_ours:.0f} us;  other (ATen) kernels: {n_other} launches, {t_other:.0f} us")

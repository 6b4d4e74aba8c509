#!/usr/bin/env python3
"""Time mpg_chain on the shapes MPLayer uses (B*N = 7680 rows): a|c projection, fn forward, fn input-gradient chain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpgan_amd import ops
V, F, out = int(os.environ.get("KCHAIN_V", "7680")), 32, 32
dev = "cuda"
torch.manual_seed(0)
W1 = torch.randn(96, 2 * F, device=dev) * 0.1; W2 = torch.randn(160, 96, device=dev) * 0.1; W3 = torch.randn(192, 160, device=dev) * 0.1
V1 = torch.randn(256, 192 + F, device=dev) * 0.05; V2 = torch.randn(256, 256, device=dev) * 0.05; V3 = torch.randn(out, 256, device=dev) * 0.05
b = torch.randn(256, device=dev) * 0.1
pk = ops.PackedMPLayer((W1, W2, W3, V1, V2, V3), F, out, 2.0, True).ensure()
x = torch.randn(V, F, device=dev); agg = torch.randn(V, 192, device=dev); gy = torch.randn(V, out, device=dev)
ac = torch.empty(V, 192, device=dev); h1 = torch.empty(V, 256, device=dev); h2 = torch.empty(V, 256, device=dev); y = torch.empty(V, out, device=dev)
dz3 = torch.empty(V, out, device=dev); dz2 = torch.empty(V, 256, device=dev); dz1 = torch.empty(V, 256, device=dev); dh0 = torch.empty(V, 224, device=dev)
seed = ops.seed_tensor(dev)

def t_ac(): ops.chain(V, [dict(img=pk.ptr("W1S"), K=F, N=192, bias=b, nbias=96, out=ac)], A=x, lda=F, K1=F, f16=True)
def t_fn(thr):
    ops.chain(V, [dict(img=pk.ptr("V1"), K=224, N=256, bias=b, act=True, drop=(8, thr, 2.0), out=h1),
                  dict(img=pk.ptr("V2"), K=256, N=256, bias=b, act=True, drop=(9, thr, 2.0), out=h2),
                  dict(img=pk.ptr("V3"), K=256, N=out, bias=b, drop=(10, thr, 2.0), out=y)],
              A=agg, lda=192, K1=192, A2=x, lda2=F, seed_t=seed, f16=True)
def t_bw(thr):
    ops.chain(V, [dict(img=pk.ptr("V3T"), K=out, N=256, gate=(h2, True, 9, thr, 2.0), out=dz2),
                  dict(img=pk.ptr("V2T"), K=256, N=256, gate=(h1, True, 8, thr, 2.0), out=dz1),
                  dict(img=pk.ptr("V1T"), K=256, N=224, out=dh0)],
              A=gy, lda=out, K1=out, in_gate=(10, thr, 2.0), in_out=dz3 if thr else None, seed_t=seed, f16=False)
def t_one(K, N):
    ops.chain(V, [dict(img=pk.ptr("V2"), K=K, N=N, bias=b, act=True, out=h2)], A=h1, lda=256, K1=K, f16=True)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"a|c projection      {timeit(t_ac):7.1f} us")
for thr in (0, 128, 77):
    print(f"fn forward  thr={thr:3d} {timeit(lambda: t_fn(thr)):7.1f} us")
    print(f"fn backward thr={thr:3d} {timeit(lambda: t_bw(thr)):7.1f} us")
print(f"one layer 256->256  {timeit(lambda: t_one(256, 256)):7.1f} us")
print(f"one layer 256->32   {timeit(lambda: t_one(256, 32)):7.1f} us")

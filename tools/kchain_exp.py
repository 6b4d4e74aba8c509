#!/usr/bin/env python3
"""Upper bounds for mpg_chain at the MPLayer shapes: what the stores of the intermediate layers and the gate loads cost
(the same launches without them -- results are then incomplete, timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kchain.py")).read().split("print(f\"a|c projection")[0])

def fn_nostore(thr):
    ops.chain(V, [dict(img=pk.ptr("V1"), K=224, N=256, bias=b, act=True, drop=(8, thr, 2.0)),
                  dict(img=pk.ptr("V2"), K=256, N=256, bias=b, act=True, drop=(9, thr, 2.0)),
                  dict(img=pk.ptr("V3"), K=256, N=out, bias=b, drop=(10, thr, 2.0), out=y)],
              A=agg, lda=192, K1=192, A2=x, lda2=F, seed_t=seed, f16=True)
def bw_nostore(thr):
    ops.chain(V, [dict(img=pk.ptr("V3T"), K=out, N=256, gate=(h2, True, 9, thr, 2.0)),
                  dict(img=pk.ptr("V2T"), K=256, N=256, gate=(h1, True, 8, thr, 2.0)),
                  dict(img=pk.ptr("V1T"), K=256, N=224, out=dh0)],
              A=gy, lda=out, K1=out, in_gate=(10, thr, 2.0), seed_t=seed, f16=False)
def bw_nogate(thr):
    ops.chain(V, [dict(img=pk.ptr("V3T"), K=out, N=256, out=dz2),
                  dict(img=pk.ptr("V2T"), K=256, N=256, out=dz1),
                  dict(img=pk.ptr("V1T"), K=256, N=224, out=dh0)],
              A=gy, lda=out, K1=out, in_gate=(10, thr, 2.0), seed_t=seed, f16=False)
for thr in (0, 128):
    print(f"thr={thr}: fn forward {timeit(lambda: t_fn(thr)):6.1f} us, without the h1/h2 stores {timeit(lambda: fn_nostore(thr)):6.1f} us")
    print(f"thr={thr}: fn backward {timeit(lambda: t_bw(thr)):6.1f} us, without the dz2/dz1 stores {timeit(lambda: bw_nostore(thr)):6.1f} us, without gates {timeit(lambda: bw_nogate(thr)):6.1f} us")

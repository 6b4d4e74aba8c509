"""Debug: the kink decisions tapped from the fused MPLayer calls of a discriminator pass with dropout, against the fp64 oracle's own
pre-activation signs, layer by layer (fraction of disagreeing elements among unmasked senders and kept elements)."""
import sys, itertools, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import train_ref as T, mpgan_ref as M
from mpgan_amd import train, ops
from conftest import hip_signs_from
sys.path.insert(0, "tests")
import test_gpu_train as TT
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 30
dev = torch.device("cuda", 0)
G, D = train.default_mpgan(N, disc_dropout=0.5)
sdD = T.init_state_dict(T.mpgan_param_shapes(False), 42, torch.float64)
D.load_state_dict({k: v.float() for k, v in sdD.items()})
data, labels = T.synthetic_batch(2 * B, N, seed=21)
st = ops.dev_state(dev)
st.tags = itertools.count(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ops.set_seed(0x5EED0000 + B)
st.tag_log, st.sign_tap = [], []
D.train()
x = data.cuda().requires_grad_(True)
out = D(x, labels.cuda())
out.sum().backward()
torch.cuda.synchronize()
log, taps = st.tag_log, st.sign_tap
st.tag_log = st.sign_tap = None
print("log", log, "taps", [t["B"] for t in taps])
ent = [e for e in log if e[2] > 0]
widths = {"e0": 96, "e1": 160, "e2": 192, "n0": 256, "n1": 256, "n2": 32}
sites = {"e0": ops.TAG_E0, "e1": ops.TAG_E1, "e2": ops.TAG_E2, "n0": ops.TAG_N0, "n1": ops.TAG_N1, "n2": ops.TAG_N2}
mask = data[:, :, -1:].double() + 0.5
xx = data[:, :, :-1].double()
for l in range(2):
    tag = ent[l][1]
    k = {}
    for s, wdt in widths.items():
        rows = 2 * B * N * N if s.startswith("e") else 2 * B * N
        m = ops.dropout_mask(rows, wdt, tag + sites[s], 128, dev).cpu()
        k[s] = m.reshape(2 * B, N, N, wdt) if s.startswith("e") else m.reshape(2 * B, N, wdt)
    probe = []
    y = M.mplayer_forward(sdD, f"mp_layers.{l}", xx, mask, True, 0.2, 0.5, k, probe=probe)
    t = taps[l]
    sg = hip_signs_from(t["ac"], t["stE2"], t["sign3"], t["h1"], t["h2"], t["B"], t["N"])
    names = ["fe1", "fe2", "fe3", "fn1", "fn2"]
    keepof = {"fe1": None, "fe2": k["e1"], "fe3": None, "fn1": k["n0"], "fn2": k["n1"]}   # (fe1 from a|c, fe3 sign words: before dropout)
    for nm, pr in zip(names, probe):
        ref = pr < 0
        d = sg[nm] != ref
        if keepof[nm] is not None:
            d = d & (keepof[nm] != 0)
        if nm.startswith("fe"):
            d = d & (mask.reshape(2 * B, 1, N, 1) != 0)
        small = (pr.abs() < 1e-4 * pr.abs().max())
        print(f"layer {l} {nm}: disagree {int(d.sum())} of {d.numel()}  (of them within 1e-4 of zero: {int((d & small).sum())})")
    print("  y err", float((t and 0) or 0))
    xx = y.detach()

"""Debug: as diag_signs.py, but the discriminator pass is TrainStep._seg_D's (real + generated jets, generator ahead, hand-offs)."""
import sys, itertools, torch, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import train_ref as T, mpgan_ref as M
from mpgan_amd import train, ops
from conftest import hip_signs_from
import test_gpu_train as TT
B, N = int(sys.argv[1]), 30
tag0 = int(sys.argv[2])
dev = torch.device("cuda", 0)
G, D = train.default_mpgan(N, disc_dropout=0.5)
sdG = T.init_state_dict(T.mpgan_param_shapes(True), 41, torch.float64)
sdD = T.init_state_dict(T.mpgan_param_shapes(False), 42, torch.float64)
G.load_state_dict({k: v.float() for k, v in sdG.items()}); D.load_state_dict({k: v.float() for k, v in sdD.items()})
data, labels = T.synthetic_batch(B, N, seed=21)
gen = torch.Generator().manual_seed(9)
nD, nG = torch.randn(B, N, 32, generator=gen) * 0.2, torch.randn(B, N, 32, generator=gen) * 0.2
ts = train.TrainStep(G, D, B, N, latent=32, use_graphs=False, lr_disc=0.0, lr_gen=train.LR["g"][1])
ts.set_batch(data.cuda(), labels.cuda()); ts.fixed_noise = (nD.cuda(), nG.cuda())
st = ops.dev_state(dev)
st.tags = itertools.count(tag0)
ops.set_seed(0x5EED0000 + B)
st.tag_log, st.sign_tap = [], []
ts._seg_D()
torch.cuda.synchronize()
log, taps = st.tag_log, st.sign_tap
st.tag_log = st.sign_tap = None
print("log", log, "taps", [t["B"] for t in taps])
kD = TT._site_masks(log, 2 * B, N, "mpgan", dev)
d2 = [t for t in taps if t["B"] == 2 * B]
with torch.no_grad():
    fake = T._fwd_G("mpgan", sdG, nD.double(), labels.double(), N, {})
x = torch.cat([data.double(), fake], 0)
mask = x[:, :, -1:] + 0.5
xx = x[:, :, :-1]
# the batch as the launches saw it
print("mask equal", bool(torch.equal(ts._mask2.cpu().double().reshape(2 * B, N, 1), mask)), " x3 err", float((ts._x3.cpu().double() - xx).abs().max()))
for l in range(2):
    probe = []
    y = M.mplayer_forward(sdD, f"mp_layers.{l}", xx, mask, True, 0.2, 0.5, kD["layers"][l], probe=probe)
    t = d2[l]
    sg = hip_signs_from(t["ac"], t["stE2"], t["sign3"], t["h1"], t["h2"], t["B"], t["N"])
    keepof = {"fe1": None, "fe2": kD["layers"][l]["e1"], "fe3": None, "fn1": kD["layers"][l]["n0"], "fn2": kD["layers"][l]["n1"]}
    for nm, pr in zip(["fe1", "fe2", "fe3", "fn1", "fn2"], probe):
        d = sg[nm] != (pr < 0)
        if keepof[nm] is not None:
            d = d & (keepof[nm] != 0)
        if nm.startswith("fe"):
            d = d & (mask.reshape(2 * B, 1, N, 1) != 0)
        small = pr.abs() < 1e-4 * pr.abs().max()
        print(f"layer {l} {nm}: disagree {int(d.sum())} of {d.numel()} (within 1e-4 of zero: {int((d & small).sum())}); real half {int(d[:B].sum())}")
    xx = y.detach()

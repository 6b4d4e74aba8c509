"""Diagnostic (not a test): per-parameter gradient errors of the HIP path against the fp64 oracle, with the fp32 oracle's
own errors beside them, for (a) one MPLayer at B = 256 with slope 1 (no kinks) and (b) the first train_D + train_G
iteration of the reference golden (B = 8).  ``python tests/probe_precision.py`` on the GPU box; numbers quoted in DESIGN.md."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_golden, rel_err  # noqa: E402


def layer_case():
    import test_gpu_mplayer as M
    for alpha in (1.0,):
        errs, frac, margin = M._run_case(256, 30, 32, 32, True, True, seed=7, alpha=alpha)
        print(f"MPLayer B=256 slope {alpha}: " + "  ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    for case in ((4, 30, 32, 32, True, True), (4, 30, 32, 3, True, True)):
        errs, _, _ = M._run_case(*case, seed=1, alpha=1.0)
        print(f"MPLayer {case} slope 1: " + "  ".join(f"{k} {v:.1e}" for k, v in errs.items()))


def train_case(B_syn=0):
    from oracle import train_ref as T
    from mpgan_amd import train
    g = load_golden("train_step_mpgan.npz")
    if B_syn:   # synthetic batch of another size (the golden's B = 8 otherwise)
        data, labels = T.synthetic_batch(B_syn, 30, seed=3)
        gen = torch.Generator().manual_seed(9)
        g = {"data": data.numpy(), "labels": labels.numpy(), "noise_D": (torch.randn(B_syn, 30, 32, generator=gen) * 0.2).numpy(),
             "noise_G": (torch.randn(B_syn, 30, 32, generator=gen) * 0.2).numpy()}
    B, N = g["data"].shape[:2]
    G, D = train.default_mpgan(N, disc_dropout=0.0)
    G.load_state_dict(T.init_state_dict(T.mpgan_param_shapes(True), 41, torch.float32))
    D.load_state_dict(T.init_state_dict(T.mpgan_param_shapes(False), 42, torch.float32))
    ts = train.TrainStep(G, D, B, N, lr_disc=0.0, lr_gen=0.0, use_graphs=False)
    ts.set_batch(torch.from_numpy(g["data"]).float().cuda(), torch.from_numpy(g["labels"]).float().cuda())
    ts.fixed_noise = (torch.from_numpy(g["noise_D"]).float().cuda(), torch.from_numpy(g["noise_G"]).float().cuda())
    ts._seg_D()
    gD = {k: p.grad.double().cpu().numpy().copy() for k, p in D.named_parameters()}
    ts._seg_G()
    gG = {k: p.grad.double().cpu().numpy().copy() for k, p in G.named_parameters()}
    runs = {}
    for dt in (torch.float64, torch.float32):
        sdD = T.init_state_dict(T.mpgan_param_shapes(False), 42, torch.float32)
        sdG = T.init_state_dict(T.mpgan_param_shapes(True), 41, torch.float32)
        c = lambda a: torch.from_numpy(np.asarray(a)).to(dt)
        _, _, rD, rG = T.train_iteration("mpgan", {k: v.to(dt) for k, v in sdD.items()}, {k: v.to(dt) for k, v in sdG.items()}, {}, {},
                                         c(g["data"]), c(g["labels"]), c(g["noise_D"]), c(g["noise_G"]), 0.0, 0.0, return_grads=True)
        runs[dt] = ({k: v.double().numpy() for k, v in rD.items()}, {k: v.double().numpy() for k, v in rG.items()})
    for name, got, idx in (("D", gD, 0), ("G", gG, 1)):
        ref, ctl = runs[torch.float64][idx], runs[torch.float32][idx]
        for k in ref:
            d = got[k] - ref[k]
            print(f"train B={B} {name}.{k}: HIP {rel_err(got[k], ref[k]):.1e}  fp32 {rel_err(ctl[k], ref[k]):.1e}   rms err / rms ref "
                  f"{np.sqrt((d ** 2).mean() / (ref[k] ** 2).mean()):.1e}  max|ref| / rms ref {np.abs(ref[k]).max() / np.sqrt((ref[k] ** 2).mean()):.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        train_case(int(sys.argv[1]))
    else:
        layer_case()
        train_case()

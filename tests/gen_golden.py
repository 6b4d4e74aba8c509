#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE implementation (build container only).

Imports the reference's own ``mpgan`` / ``gapt`` / ``setup_training`` modules from
/root/reference (read-only; never copied), loads deterministic parameter values from
``oracle.init_state_dict`` into the reference nn.Modules, runs forward/backward on seeded
inputs and stores inputs + outputs as small ``.npz`` files under tests/golden/.

Weights are NOT stored: they are a pure function of (parameter name, seed) via numpy's
MT19937, so the tests rebuild the very same tensors.  Parameter gradients are stored as
summaries (sum, L2 norm, 64 sampled entries per tensor) to keep the fixtures small; input
gradients and outputs are stored in full.

Run:  python tests/gen_golden.py          (no-op with a message if /root/reference is absent)
"""
import json
import os
import sys

import numpy as np
import torch

REF = os.environ.get("MPGAN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "golden")
sys.path.insert(0, os.path.dirname(HERE))

from oracle.train_ref import (  # noqa: E402
    init_state_dict, mpgan_param_shapes, gapt_param_shapes, synthetic_batch)


def summarize(name, t):
    """sum, l2, and 64 sampled entries (indices from a RandomState keyed on the name)."""
    import zlib
    flat = t.detach().double().reshape(-1)
    rs = np.random.RandomState(zlib.crc32(name.encode()) % (2**31))
    idx = rs.randint(0, flat.numel(), size=64)
    return np.concatenate([[flat.sum().item(), flat.norm().item()], flat[idx].numpy()])


def seeded(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).normal(0, scale, size=shape))


def rand_mask(B, N, seed):
    rs = np.random.RandomState(seed)
    n = rs.randint(1, N + 1, size=B)
    m = np.zeros((B, N, 1))
    for b in range(B):
        m[b, rs.permutation(N)[: n[b]], 0] = 1.0
    return torch.from_numpy(m)


# MPLayer configurations outside the fused kernels' shapes (name, B, N, F, out, constructor keywords); the tests import
# this table so that the oracle and the product are built with the very same arguments
OPTION_CASES = [
    ("ef", 3, 9, 32, 32, dict(pos_diffs=True)),                                                   # delta_r over all features
    ("efc", 3, 8, 3, 32, dict(pos_diffs=True, all_ef=False, delta_coords=True, delta_r=True)),   # [diffs(2), dists]
    ("cl", 4, 7, 32, 32, dict(clabels=1, mask_fne_np=True)),                                      # rows tiled jet r mod B
    ("knnef", 3, 12, 32, 32, dict(pos_diffs=True, fully_connected=False, num_knn=5, self_loops=True, sum=False)),
    ("widths", 3, 9, 16, 8, dict(fe=[64, 48], fn=[40], use_mask=False)),                          # any layer widths
]
OPTION_SEEDS = {"ef": 60, "efc": 61, "cl": 62, "knnef": 63, "widths": 64}
# a whole discriminator with the conditioning options on (keywords as setup_training.setup_mpgan passes them, :1206-1293)
D_OPT = dict(
    num_particles=8, hidden_node_size=32, fe_layers=[96, 160, 192], fn_layers=[256, 256], fn1_layers=None, mp_iters=2,
    fe1_layers=None, final_activation="sigmoid", input_node_size=3, dea=True, dea_sum=True, fnd=[], mask_fnd_np=True,
    mp_args=dict(pos_diffs=True, all_ef=False, coords="polarrel", delta_coords=False, delta_r=True, int_diffs=False, clabels=1,
                 mask_fne_np=True, fully_connected=True, num_knn=20, self_loops=True, sum=True),
    mp_args_first_layer=dict(clabels=1, all_ef=False),
    linear_args=dict(leaky_relu_alpha=0.2, dropout_p=0.0, batch_norm=False, spectral_norm=False),
    mask_args=dict(mask_feat=False, mask_feat_bin=False, mask_weights=False, mask_manual=False, mask_exp=False,
                   mask_real_only=False, mask_learn=False, mask_learn_bin=True, mask_learn_sep=False, fmg=[64],
                   mask_disc_sep=False, mask_fnd_np=True, mask_c=True, mask_fne_np=True))
# the same discriminator with TWO scalars per edge (distance + one conditioning column; no particle-count columns): what the
# fused edge kernels take (mpgan_amd.ops.EDGE_SCALARS)
D_OPT2 = dict(D_OPT, mask_fnd_np=False,
              mp_args=dict(D_OPT["mp_args"], mask_fne_np=False),
              mp_args_first_layer=dict(clabels=1, all_ef=False),   # (its own dict: the reference's constructor fills it in place)
              mask_args=dict(D_OPT["mask_args"], mask_fnd_np=False, mask_fne_np=False))


def main():
    if not os.path.isdir(REF):
        print(f"reference not found at {REF}; nothing generated")
        return 0
    sys.path.insert(0, REF)
    import mpgan as rmp  # reference
    import gapt as rga  # reference
    import setup_training as st  # reference

    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ------------------------------------------------------------------ 1. MPLayer fwd/bwd
    fe, fn = [96, 160, 192], [256, 256]
    cases = [  # name, B, N, F, out, use_mask, sum
        ("g0", 4, 30, 32, 32, True, True),
        ("d0", 4, 30, 3, 32, True, True),
        ("g1", 4, 30, 32, 3, True, True),
        ("n150", 2, 150, 32, 32, True, True),
        ("mean", 3, 30, 32, 32, True, False),
        ("nomask", 3, 30, 32, 32, False, True),
        ("small", 2, 5, 32, 32, True, True),
    ]
    for dt_name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        for ci, (name, B, N, F, out, use_mask, sm) in enumerate(cases):
            if dt_name == "f32" and name not in ("g0", "d0"):
                continue
            layer = rmp.MPLayer(F, fe, fn, out, sum=sm).to(dt)
            shapes = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
            sd = init_state_dict(shapes, seed=ci, dtype=dt)
            layer.load_state_dict(sd)
            x = seeded((B, N, F), 100 + ci, 0.5).to(dt).requires_grad_(True)
            mask = rand_mask(B, N, 200 + ci).to(dt) if use_mask else None
            g = seeded((B, N, out), 300 + ci).to(dt)
            y = layer(x, use_mask, mask)
            (y * g).sum().backward()
            rec = dict(x=x.detach().numpy(), g=g.numpy(), y=y.detach().numpy(),
                       dx=x.grad.numpy(), seed=ci, sum=int(sm), out=out)
            if mask is not None:
                rec["mask"] = mask.numpy()
            for k, p in layer.named_parameters():
                rec["grad__" + k] = summarize(k, p.grad)
            np.savez_compressed(os.path.join(OUT, f"mplayer_{name}_{dt_name}.npz"), **rec)
            print("mplayer", name, dt_name, float(y.abs().max()))

    # ------------------------------------------------------------------ 1b. MPLayer on the k-nearest-neighbour graph
    for ci, (name, B, N, F, out, knn, loops, sm, use_mask) in enumerate([
            ("knn10", 3, 30, 32, 32, 10, True, True, True), ("knn5nl", 3, 30, 3, 32, 5, False, False, True),
            ("knn20u", 2, 30, 32, 32, 20, True, True, False)]):
        dt = torch.float64
        layer = rmp.MPLayer(F, fe, fn, out, sum=sm, fully_connected=False, num_knn=knn, self_loops=loops).to(dt)
        shapes = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
        layer.load_state_dict(init_state_dict(shapes, seed=80 + ci, dtype=dt))
        x = seeded((B, N, F), 180 + ci, 0.5).to(dt).requires_grad_(True)
        mask = rand_mask(B, N, 280 + ci).to(dt) if use_mask else None
        g = seeded((B, N, out), 380 + ci).to(dt)
        y = layer(x, use_mask, mask)
        (y * g).sum().backward()
        rec = dict(x=x.detach().numpy(), g=g.numpy(), y=y.detach().numpy(), dx=x.grad.numpy(), seed=80 + ci, sum=int(sm),
                   out=out, num_knn=knn, self_loops=int(loops))
        if mask is not None:
            rec["mask"] = mask.numpy()
        for k, p in layer.named_parameters():
            rec["grad__" + k] = summarize(k, p.grad)
        np.savez_compressed(os.path.join(OUT, f"mplayer_{name}_f64.npz"), **rec)
        print("mplayer", name, float(y.abs().max()))

    # ------------------------------------------------------------------ 1c. MPLayer with its non-default options
    for name, B, N, F, out, kw in OPTION_CASES:
        dt = torch.float64
        ci = OPTION_SEEDS[name]
        layer = rmp.MPLayer(F, kw.get("fe", fe), kw.get("fn", fn), out,
                            **{k: v for k, v in kw.items() if k not in ("fe", "fn", "use_mask")}).to(dt)
        shapes = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
        layer.load_state_dict(init_state_dict(shapes, seed=ci, dtype=dt))
        x = seeded((B, N, F), 100 + ci, 0.5).to(dt).requires_grad_(True)
        use_mask = kw.get("use_mask", True)
        mask = rand_mask(B, N, 200 + ci).to(dt) if use_mask else None
        labels = seeded((B, 2), 400 + ci, 1.0).to(dt)
        njp = mask.sum(1) / N if mask is not None else torch.ones(B, 1, dtype=dt)
        g = seeded((B, N, out), 300 + ci).to(dt)
        y = layer(x, use_mask, mask, labels, njp)
        (y * g).sum().backward()
        rec = dict(x=x.detach().numpy(), g=g.numpy(), y=y.detach().numpy(), dx=x.grad.numpy(), seed=ci,
                   labels=labels.numpy(), njp=njp.numpy())
        if mask is not None:
            rec["mask"] = mask.numpy()
        for k, p in layer.named_parameters():
            rec["grad__" + k] = summarize(k, p.grad)
        np.savez_compressed(os.path.join(OUT, f"mplayer_opt_{name}_f64.npz"), **rec)
        print("mplayer option case", name, float(y.abs().max()))

    # ------------------------------------------------------------------ 1d. a discriminator with the conditioning options
    dt = torch.float64
    Dopt = rmp.MPDiscriminator(**D_OPT).to(dt)
    shapes = {k: tuple(v.shape) for k, v in Dopt.state_dict().items()}
    Dopt.load_state_dict(init_state_dict(shapes, seed=70, dtype=dt))
    Dopt.eval()
    B, N = 5, D_OPT["num_particles"]
    mk = rand_mask(B, N, 270).to(dt)
    xin = torch.cat((seeded((B, N, 3), 170, 0.5).to(dt) * mk, mk - 0.5), dim=2).requires_grad_(True)
    lab = seeded((B, 2), 470, 1.0).to(dt)
    yD = Dopt(xin, lab)
    gD = seeded(tuple(yD.shape), 370).to(dt)
    (yD * gD).sum().backward()
    rec = dict(x=xin.detach().numpy(), labels=lab.numpy(), g=gD.numpy(), y=yD.detach().numpy(), dx=xin.grad.numpy(), seed=70,
               keys=np.array(list(shapes.keys())), shapes=np.array([str(v) for v in shapes.values()]))
    for k, p in Dopt.named_parameters():
        rec["grad__" + k] = summarize(k, p.grad)
    np.savez_compressed(os.path.join(OUT, "mpdisc_opt_f64.npz"), **rec)
    print("mpdisc options", float(yD.abs().max()))

    # ------------------------------------------------------------------ 1d'. the same with two scalars per edge (distance + one
    # conditioning column): the combination the fused edge kernels take -- every layer of this discriminator is fused here
    dt = torch.float64
    Dopt2 = rmp.MPDiscriminator(**D_OPT2).to(dt)
    shapes = {k: tuple(v.shape) for k, v in Dopt2.state_dict().items()}
    Dopt2.load_state_dict(init_state_dict(shapes, seed=71, dtype=dt))
    Dopt2.eval()
    B, N = 6, D_OPT2["num_particles"]
    mk = rand_mask(B, N, 271).to(dt)
    xin = torch.cat((seeded((B, N, 3), 171, 0.5).to(dt) * mk, mk - 0.5), dim=2).requires_grad_(True)
    lab = seeded((B, 2), 471, 1.0).to(dt)
    yD = Dopt2(xin, lab)
    gD = seeded(tuple(yD.shape), 371).to(dt)
    (yD * gD).sum().backward()
    rec = dict(x=xin.detach().numpy(), labels=lab.numpy(), g=gD.numpy(), y=yD.detach().numpy(), dx=xin.grad.numpy(), seed=71,
               keys=np.array(list(shapes.keys())), shapes=np.array([str(v) for v in shapes.values()]))
    for k, p in Dopt2.named_parameters():
        rec["grad__" + k] = summarize(k, p.grad)
    np.savez_compressed(os.path.join(OUT, "mpdisc_opt2_f64.npz"), **rec)
    print("mpdisc options, two scalars", float(yD.abs().max()))

    # ------------------------------------------------------------------ 1e. LinearNet with batch norm and spectral norm
    dt = torch.float64
    ln = rmp.LinearNet([24, 16], input_size=12, output_size=5, final_linear=True, batch_norm=True, spectral_norm=True,
                       dropout_p=0.0).to(dt)
    lshapes = {k: tuple(v.shape) for k, v in ln.state_dict().items() if "running" not in k and "num_batches" not in k}
    sd_ln = ln.state_dict()
    sd_ln.update(init_state_dict(lshapes, seed=75, dtype=dt))   # running statistics keep their defaults (0, 1, 0)
    ln.load_state_dict(sd_ln)
    ln.train()
    x1, x2 = seeded((40, 12), 175).to(dt), seeded((40, 12), 176).to(dt).requires_grad_(True)
    g1, g2 = seeded((40, 5), 375).to(dt), seeded((40, 5), 376).to(dt)
    y1 = ln(x1)
    (y1 * g1).sum().backward()
    ln.zero_grad()
    y2 = ln(x2)
    (y2 * g2).sum().backward()
    rec = dict(x1=x1.numpy(), x2=x2.detach().numpy(), g1=g1.numpy(), g2=g2.numpy(), y1=y1.detach().numpy(), y2=y2.detach().numpy(),
               dx2=x2.grad.numpy(), seed=75)
    for k, p in ln.named_parameters():
        if p.grad is not None:
            rec["grad__" + k] = p.grad.numpy()
    ln.eval()
    rec["y3"] = ln(x2.detach()).detach().numpy()
    for k, v in ln.state_dict().items():
        if "running" in k or "num_batches" in k or k.endswith(("weight_u", "weight_v")):
            rec["state__" + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "linearnet_bn_sn_f64.npz"), **rec)
    print("linearnet bn+sn", float(y2.abs().max()))

    # ------------------------------------------------------------------ default args / manifests
    sys.argv = ["gen_golden"]
    args = st.process_args(st.parse_args())
    manifests = {}
    G = st.setup_mpgan(args, gen=True)
    D = st.setup_mpgan(args, gen=False)
    manifests["mpgan_G"] = {k: list(v.shape) for k, v in G.state_dict().items()}
    manifests["mpgan_D"] = {k: list(v.shape) for k, v in D.state_dict().items()}
    args.model = "gapt"
    Gg = st.setup_gapt(args, gen=True)
    Dg = st.setup_gapt(args, gen=False)
    manifests["gapt_G"] = {k: list(v.shape) for k, v in Gg.state_dict().items()}
    manifests["gapt_D"] = {k: list(v.shape) for k, v in Dg.state_dict().items()}
    args.use_isab = True
    manifests["gapt_G_isab"] = {k: list(v.shape) for k, v in st.setup_gapt(args, gen=True).state_dict().items()}
    args.use_isab = False
    with open(os.path.join(OUT, "manifests.json"), "w") as f:
        json.dump(manifests, f, indent=1, sort_keys=True)
    assert manifests["mpgan_G"] == {k: list(v) for k, v in mpgan_param_shapes(True).items()}
    assert manifests["mpgan_D"] == {k: list(v) for k, v in mpgan_param_shapes(False).items()}
    assert manifests["gapt_G"] == {k: list(v) for k, v in gapt_param_shapes(True).items()}
    assert manifests["gapt_D"] == {k: list(v) for k, v in gapt_param_shapes(False).items()}

    # ------------------------------------------------------------------ 2. MPGenerator / MPDiscriminator fwd
    for dt_name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        B, N = 6, 30
        G.to(dt).eval()
        D.to(dt).eval()
        G.load_state_dict(init_state_dict(mpgan_param_shapes(True), seed=11, dtype=dt))
        D.load_state_dict(init_state_dict(mpgan_param_shapes(False), seed=12, dtype=dt))
        data, labels = synthetic_batch(B, N, seed=5, dist="uniform", dtype=dt)
        noise = seeded((B, N, 32), 17, 0.2).to(dt).requires_grad_(True)
        gout = G(noise, labels)
        gg = seeded(gout.shape, 18).to(dt)
        (gout * gg).sum().backward()
        dd = data.clone().requires_grad_(True)
        dout = D(dd, labels)
        dg = seeded(dout.shape, 19).to(dt)
        (dout * dg).sum().backward()
        rec = dict(noise=noise.detach().numpy(), labels=labels.numpy(), data=data.numpy(),
                   gout=gout.detach().numpy(), gg=gg.numpy(), dnoise=noise.grad.numpy(),
                   dout=dout.detach().numpy(), dg=dg.numpy(), ddata=dd.grad.numpy())
        for k, p in G.named_parameters():
            rec["gradG__" + k] = summarize(k, p.grad)
        for k, p in D.named_parameters():
            rec["gradD__" + k] = summarize(k, p.grad)
        np.savez_compressed(os.path.join(OUT, f"mpgan_nets_{dt_name}.npz"), **rec)
        G.zero_grad(); D.zero_grad()
        print("mpgan nets", dt_name, float(dout.mean()))

    # ------------------------------------------------------------------ 3. published weights (outputs only)
    for jets in ("g", "q", "t"):
        path = os.path.join(REF, "trained_models", f"mp_{jets}", "G_best_epoch.pt")
        if not os.path.isfile(path):
            continue
        G.to(torch.float32).eval()
        G.load_state_dict(torch.load(path, map_location="cpu"))
        _, labels = synthetic_batch(8, 30, seed=21, dist="gluon")
        noise = seeded((8, 30, 32), 22, 0.2).float()
        with torch.no_grad():
            out = G(noise, labels)
        np.savez_compressed(os.path.join(OUT, f"published_mp_{jets}.npz"), noise=noise.numpy(),
                            labels=labels.numpy(), out=out.numpy())
        print("published", jets, float(out.abs().max()))

    # ------------------------------------------------------------------ 4. GAPT blocks
    E, H = 64, 4
    sab_args = dict(embed_dim=E, ff_layers=[], final_linear=False, num_heads=H, layer_norm=False,
                    dropout_p=0.0, linear_args={"leaky_relu_alpha": 0.2, "dropout_p": 0.0})
    for dt_name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        for ci, (name, B, N, use_mask) in enumerate(
                [("m30", 4, 30, True), ("u30", 3, 30, False), ("m150", 2, 150, True)]):
            if dt_name == "f32" and name != "m30":
                continue
            mask = rand_mask(B, N, 400 + ci).to(dt) if use_mask else None
            amask = None if mask is None else rga.model._attn_mask(mask)
            x0 = seeded((B, N, E), 410 + ci, 0.5).to(dt)
            rec = dict(x=x0.numpy())
            if mask is not None:
                rec["mask"] = mask.numpy()
            blocks = {
                "sab": rga.SAB(**sab_args),
                "pma": rga.PMA(num_seeds=1, **sab_args),
                "isab": rga.ISAB(10, **sab_args),
            }
            for bname, blk in blocks.items():
                blk.to(dt)
                shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
                blk.load_state_dict(init_state_dict(shapes, seed=50 + ci, dtype=dt))
                x = x0.clone().requires_grad_(True)
                y = blk(x, amask)
                g = seeded(y.shape, 420 + ci).to(dt)
                (y * g).sum().backward()
                rec[f"{bname}_y"] = y.detach().numpy()
                rec[f"{bname}_g"] = g.numpy()
                rec[f"{bname}_dx"] = x.grad.numpy()
                for k, p in blk.named_parameters():
                    rec[f"{bname}_grad__{k}"] = summarize(k, p.grad)
            np.savez_compressed(os.path.join(OUT, f"gapt_blocks_{name}_{dt_name}.npz"), **rec)
            print("gapt blocks", name, dt_name)

        # 4b. SAB with LayerNorm (layer_norm=True: gapt/model.py:118-120, :131-136), fp64 only
        if dt_name == "f64":
            ln_args = dict(sab_args, layer_norm=True)
            B, N = 4, 30
            mask = rand_mask(B, N, 460).to(dt)
            x0 = seeded((B, N, E), 461, 0.5).to(dt)
            blk = rga.SAB(**ln_args).to(dt)
            shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
            blk.load_state_dict(init_state_dict(shapes, seed=60, dtype=dt))
            x = x0.clone().requires_grad_(True)
            y = blk(x, rga.model._attn_mask(mask))
            g = seeded(y.shape, 462).to(dt)
            (y * g).sum().backward()
            rec = dict(x=x0.numpy(), mask=mask.numpy(), y=y.detach().numpy(), g=g.numpy(), dx=x.grad.numpy(),
                       keys=np.array(sorted(shapes)), shapes=np.array([str(shapes[k]) for k in sorted(shapes)]))
            for k, p in blk.named_parameters():
                rec["grad__" + k] = summarize(k, p.grad)
            np.savez_compressed(os.path.join(OUT, "gapt_sab_layernorm_f64.npz"), **rec)
            print("gapt sab layer_norm", float(y.abs().max()))

        B, N = 6, 30
        Gg.to(dt).eval(); Dg.to(dt).eval()
        Gg.load_state_dict(init_state_dict(gapt_param_shapes(True), seed=31, dtype=dt))
        Dg.load_state_dict(init_state_dict(gapt_param_shapes(False), seed=32, dtype=dt))
        data, labels = synthetic_batch(B, N, seed=6, dist="uniform", dtype=dt)
        noise = seeded((B, N, E), 33, 0.2).to(dt)
        with torch.no_grad():
            gout = Gg(noise, labels)
            dout = Dg(data, labels)
        np.savez_compressed(os.path.join(OUT, f"gapt_nets_{dt_name}.npz"), noise=noise.numpy(),
                            labels=labels.numpy(), data=data.numpy(), gout=gout.numpy(),
                            dout=dout.reshape(B, 1).numpy())
        print("gapt nets", dt_name, float(dout.mean()))

    # ------------------------------------------------------------------ 5. one train_D + train_G step
    # train.py cannot be imported (needs jetnet): its step (train.py:398-523) is driven here
    # with the reference's own modules, torch.optim.RMSprop and MSELoss; dropout p = 0.
    mse = torch.nn.MSELoss()
    for model in ("mpgan", "gapt"):
        dt = torch.float64
        B, N = 8, 30
        sys.argv = ["gen_golden", "--model", model, "--disc-dropout", "0"]
        a = st.process_args(st.parse_args())
        if model == "mpgan":
            Gm, Dm = st.setup_mpgan(a, gen=True), st.setup_mpgan(a, gen=False)
            shG, shD, lat = mpgan_param_shapes(True), mpgan_param_shapes(False), 32
        else:
            Gm, Dm = st.setup_gapt(a, gen=True), st.setup_gapt(a, gen=False)
            shG, shD, lat = gapt_param_shapes(True), gapt_param_shapes(False), 64
        Gm.to(dt); Dm.to(dt)
        Gm.load_state_dict(init_state_dict(shG, seed=41, dtype=dt))
        Dm.load_state_dict(init_state_dict(shD, seed=42, dtype=dt))
        # the reference's default learning rates (setup_training.py:848-872); RMSprop's first
        # update is +-10*lr per parameter whatever the gradient's size, so larger values
        # saturate D after one step
        lr_d, lr_g = (3e-5, 1e-5) if model == "mpgan" else (1.5e-4, 0.5e-4)
        oD = torch.optim.RMSprop(Dm.parameters(), lr=lr_d)
        oG = torch.optim.RMSprop(Gm.parameters(), lr=lr_g)
        data, labels = synthetic_batch(B, N, seed=7, dist="uniform", dtype=dt)
        nD = seeded((B, N, lat), 43, 0.2).to(dt)
        nG = seeded((B, N, lat), 44, 0.2).to(dt)
        rec = dict(data=data.numpy(), labels=labels.numpy(), noise_D=nD.numpy(), noise_G=nG.numpy(),
                   lr_d=lr_d, lr_g=lr_g)
        for it in range(2):
            # train_D
            Dm.train(); oD.zero_grad(); Gm.eval()
            out_r = Dm(data.clone(), labels).reshape(B, 1)
            fake = Gm(nD, labels)
            out_f = Dm(fake, labels).reshape(B, 1)
            D_loss = mse(out_r, torch.ones(B, 1, dtype=dt)) + mse(out_f, torch.zeros(B, 1, dtype=dt))
            D_loss.backward(); oD.step()
            if it == 0:
                for k, p in Dm.named_parameters():
                    rec["gradD__" + k] = summarize(k, p.grad)
            # train_G
            Gm.train(); oG.zero_grad()
            fake = Gm(nG, labels)
            out = Dm(fake, labels).reshape(B, 1)
            G_loss = mse(out, torch.ones(B, 1, dtype=dt))
            G_loss.backward(); oG.step()
            if it == 0:
                for k, p in Gm.named_parameters():
                    rec["gradG__" + k] = summarize(k, p.grad)
            rec[f"D_loss{it}"] = D_loss.item()
            rec[f"G_loss{it}"] = G_loss.item()
        for k, p in Dm.named_parameters():
            rec["postD__" + k] = summarize(k, p.data)
        for k, p in Gm.named_parameters():
            rec["postG__" + k] = summarize(k, p.data)
        np.savez_compressed(os.path.join(OUT, f"train_step_{model}.npz"), **rec)
        print("train step", model, rec["D_loss0"], rec["G_loss0"], rec["D_loss1"], rec["G_loss1"])

    # ------------------------------------------------------------------ 6. gradient penalty (--gp): double backward through D
    # train.py cannot be imported (jetnet), but its ``gradient_penalty`` and ``calc_D_loss`` (train.py:286-395) are
    # self-contained: their definitions are taken out of the reference's source with ``ast`` and EXECUTED here as they
    # are, on the reference's own discriminator.  The interpolation weights they draw (torch.rand(B, 1, 1), the first
    # draw after the seed) are recorded by repeating the draw.
    import ast
    with open(os.path.join(REF, "train.py")) as f:
        tree = ast.parse(f.read())
    wanted = [n for n in tree.body
              if (isinstance(n, ast.FunctionDef) and n.name in ("gradient_penalty", "calc_D_loss", "calc_G_loss"))
              or (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") in ("bce", "mse"))]
    assert len(wanted) == 5, [getattr(n, "name", None) for n in wanted]
    from torch.autograd import Variable, grad as torch_grad
    ns = {"torch": torch, "Variable": Variable, "torch_grad": torch_grad}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), os.path.join(REF, "train.py"), "exec"), ns)
    for loss in ("w", "ls"):
        dt = torch.float64
        B, N = 8, 30
        sys.argv = ["gen_golden", "--model", "mpgan", "--disc-dropout", "0", "--loss", loss, "--gp", "10"]
        a = st.process_args(st.parse_args())
        Gm, Dm = st.setup_mpgan(a, gen=True).to(dt), st.setup_mpgan(a, gen=False).to(dt)
        Gm.load_state_dict(init_state_dict(mpgan_param_shapes(True), seed=41, dtype=dt))
        Dm.load_state_dict(init_state_dict(mpgan_param_shapes(False), seed=42, dtype=dt))
        data, labels = synthetic_batch(B, N, seed=9, dist="uniform", dtype=dt)
        nD = seeded((B, N, 32), 45, 0.2).to(dt)
        Dm.train(); Gm.eval()
        out_r = Dm(data.clone(), labels)
        fake = Gm(nD, labels)
        out_f = Dm(fake, labels)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dt)   # (the reference draws alpha and its ones in the default dtype)
        try:
            torch.manual_seed(1234)
            alpha = torch.rand(B, 1, 1)
            torch.manual_seed(1234)
            D_loss, items = ns["calc_D_loss"](loss, Dm, data, fake, out_r, out_f, B, model="mpgan", gp_lambda=a.gp)
        finally:
            torch.set_default_dtype(prev)
        Dm.zero_grad()
        D_loss.backward()
        rec = dict(data=data.numpy(), labels=labels.numpy(), noise_D=nD.numpy(), alpha=alpha.numpy(), gp_lambda=a.gp,
                   D_loss=D_loss.item(), Dr=items["Dr"], Df=items["Df"], gp=items["gp"], fake=fake.detach().numpy())
        for k, p in Dm.named_parameters():
            rec["gradD__" + k] = summarize(k, p.grad)
        np.savez_compressed(os.path.join(OUT, f"gp_step_mpgan_{loss}.npz"), **rec)
        print("gp step", loss, rec["D_loss"], rec["gp"])

    # the same for the attention discriminator: gradient_penalty calls D(interpolated) on whatever D is (train.py:301);
    # GAPT_D ends in a sigmoid for every loss (gapt/model.py:344).  nn.MultiheadAttention(need_weights=False) goes through
    # F.scaled_dot_product_attention, and torch 2.10's CPU flash kernel has no second derivative ("derivative for
    # aten::_scaled_dot_product_flash_attention_for_cpu_backward is not implemented"): the reference's functions are run
    # under torch's MATH backend of that very operator (same definition, differentiable to any order).
    from torch.nn.attention import sdpa_kernel, SDPBackend
    for loss in ("ls", "w"):
        dt = torch.float64
        B, N = 8, 30
        sys.argv = ["gen_golden", "--model", "gapt", "--disc-dropout", "0", "--loss", loss, "--gp", "10"]
        a = st.process_args(st.parse_args())
        Gm, Dm = st.setup_gapt(a, gen=True).to(dt), st.setup_gapt(a, gen=False).to(dt)
        Gm.load_state_dict(init_state_dict(gapt_param_shapes(True), seed=41, dtype=dt))
        Dm.load_state_dict(init_state_dict(gapt_param_shapes(False), seed=42, dtype=dt))
        data, labels = synthetic_batch(B, N, seed=9, dist="uniform", dtype=dt)
        nD = seeded((B, N, 64), 45, 0.2).to(dt)
        Dm.train(); Gm.eval()
        out_r = Dm(data.clone(), labels)
        fake = Gm(nD, labels)
        out_f = Dm(fake, labels)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dt)
        try:
            torch.manual_seed(1234)
            alpha = torch.rand(B, 1, 1)
            torch.manual_seed(1234)
            with sdpa_kernel(SDPBackend.MATH):
                D_loss, items = ns["calc_D_loss"](loss, Dm, data, fake, out_r, out_f, B, model="gapt", gp_lambda=a.gp)
                Dm.zero_grad()
                D_loss.backward()
        finally:
            torch.set_default_dtype(prev)
        rec = dict(data=data.numpy(), labels=labels.numpy(), noise_D=nD.numpy(), alpha=alpha.numpy(), gp_lambda=a.gp,
                   D_loss=D_loss.item(), Dr=items["Dr"], Df=items["Df"], gp=items["gp"], fake=fake.detach().numpy())
        for k, p in Dm.named_parameters():
            rec["gradD__" + k] = summarize(k, p.grad)
        np.savez_compressed(os.path.join(OUT, f"gp_step_gapt_{loss}.npz"), **rec)
        print("gp step gapt", loss, rec["D_loss"], rec["gp"])

    # ------------------------------------------------------------------ 7. the four loss branches, executed from the source
    # calc_D_loss (train.py:331-395) and calc_G_loss (:465-476) on fixed discriminator outputs [B, 1]: values and the
    # gradients with respect to the outputs.  Outputs in (0, 1) for og / ls (a sigmoid's range; two entries at the ends,
    # where BCELoss clamps its logarithm at -100), any sign and beyond +-1 for w / hinge (some hinge terms inactive).
    rs = np.random.RandomState(77)
    B = 16
    rec = {}
    for loss in ("og", "ls", "w", "hinge"):
        if loss in ("og", "ls"):
            r, f = rs.uniform(0.02, 0.98, size=(B, 1)), rs.uniform(0.02, 0.98, size=(B, 1))
            r[0, 0], f[0, 0] = 1.0, 0.0          # log(1 - 1), log(0) of the wrong-side terms never occur; these are the exact ends
            r[1, 0], f[1, 0] = 1e-50, 1.0 - 1e-17  # (rounds to 1.0 in fp64: log(1 - f) = -inf -> clamped)
        else:
            r, f = rs.normal(0, 1.5, size=(B, 1)), rs.normal(0, 1.5, size=(B, 1))
        out_r = torch.from_numpy(r).requires_grad_(True)
        out_f = torch.from_numpy(f).requires_grad_(True)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        try:
            D_loss, items = ns["calc_D_loss"](loss, None, out_r, None, out_r, out_f, B)
            gr, gf = torch.autograd.grad(D_loss, (out_r, out_f))
            out_g = torch.from_numpy(f).requires_grad_(True)
            G_loss = ns["calc_G_loss"](loss, out_g)
            (gg,) = torch.autograd.grad(G_loss, out_g)
        finally:
            torch.set_default_dtype(prev)
        rec.update({f"{loss}_out_r": r, f"{loss}_out_f": f, f"{loss}_D_loss": D_loss.item(), f"{loss}_Dr": items["Dr"],
                    f"{loss}_Df": items["Df"], f"{loss}_dD_dr": gr.numpy(), f"{loss}_dD_df": gf.numpy(),
                    f"{loss}_G_loss": G_loss.item(), f"{loss}_dG_df": gg.numpy()})
        print("loss", loss, D_loss.item(), G_loss.item())
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **rec)
    return 0


if __name__ == "__main__":
    sys.exit(main())

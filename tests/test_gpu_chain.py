"""GPU parity of the node-path entry points on their own: mpg_chain, mpg_pack_many, the grouped weight-gradient
GEMM / reduction -- against plain fp64 torch restatements of LinearNet.forward (mpgan/model.py:70-85) and of
autograd's dX = dY W, dW = dY^T X.  Tolerance: 1e-3 relative (north star), asserted at 1e-4."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TIGHT = 1e-4


def _dev():
    return torch.device("cuda:0")


def _t(rs, *shape, scale=1.0):
    return torch.from_numpy(rs.normal(size=shape) * scale).float().to(_dev())


def _lrelu(v, a):
    return torch.where(v > 0, v, a * v)


def _holder(rs, F, out, n1=256, n2=256, dscale=1.0, f16=True):
    from mpgan_amd import ops
    W = dict(W1=_t(rs, 96, 2 * F, scale=0.2), W2=_t(rs, 160, 96, scale=0.1), W3=_t(rs, 192, 160, scale=0.1),
             V1=_t(rs, n1, 192 + F, scale=0.1), V2=_t(rs, n2, n1, scale=0.1), V3=_t(rs, out, n2, scale=0.1))
    pk = ops.PackedMPLayer(tuple(W.values()), F, out, dscale, f16).ensure()
    return W, pk


@pytest.mark.parametrize("M,F,out,slabs", [(7680, 32, 32, 1), (100, 3, 32, 2), (37, 32, 5, 1), (1, 3, 3, 3)])
def test_chain_forward_three_layers(M, F, out, slabs):
    """fn = LinearNet([256,256] -> out) on [sum of agg slabs | x], ragged row counts, odd widths."""
    from mpgan_amd import ops
    rs = np.random.RandomState(M + F + out)
    W, pk = _holder(rs, F, out)
    aggp, x = _t(rs, slabs, M, 192), _t(rs, M, F)
    c1, c2, c3 = _t(rs, 256), _t(rs, 256), _t(rs, out)
    h1, h2, y = (torch.empty(M, n, device=_dev()) for n in (256, 256, out))
    # (the forward images of a PackedMPLayer hold SC_WN * W, the activations are split as SC_ACT * x: exact
    # power-of-two scales that keep the lo halves of the fp16 pairs normal -- the results must not notice)
    ops.chain(M, [dict(img=pk.ptr("V1"), K=192 + F, N=256, bias=c1, act=True, out=h1, wscale=ops.SC_WN),
                  dict(img=pk.ptr("V2"), K=256, N=256, bias=c2, act=True, out=h2, wscale=ops.SC_WN),
                  dict(img=pk.ptr("V3"), K=256, N=out, bias=c3, out=y, wscale=ops.SC_WN)],
              A=aggp, lda=192, K1=192, A2=x, lda2=F, a_slabs=slabs, a_slab_stride=M * 192, alpha=0.2, f16=True,
              ascale=ops.SC_ACT)
    inp = torch.cat([aggp.double().sum(0), x.double()], 1)
    r1 = _lrelu(inp @ W["V1"].double().t() + c1.double(), 0.2)
    r2 = _lrelu(r1 @ W["V2"].double().t() + c2.double(), 0.2)
    r3 = r2 @ W["V3"].double().t() + c3.double()
    for got, ref in ((h1, r1), (h2, r2), (y, r3)):
        assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < TIGHT


@pytest.mark.parametrize("p_drop", [0.0, 0.5, 0.3])
def test_chain_gradient_chain_with_gates(p_drop):
    """dz3 = gy * keep3 ; dz2 = (dz3 V3) * phi'(h2) keep2 ; dz1 = (dz2 V2) * phi'(h1) keep1 ; dh0 = dz1 V1 -- the keep
    masks are the kernel's own (mpg_dropout_mask), fed to the fp64 restatement."""
    from mpgan_amd import ops
    rs = np.random.RandomState(11)
    M, F, out = 333, 32, 32
    W, pk = _holder(rs, F, out)
    thr, scale = ops.drop_params(p_drop)
    ops.set_seed(99)
    seed = ops.seed_tensor(_dev())
    gy, h1, h2 = _t(rs, M, out), _t(rs, M, 256), _t(rs, M, 256)
    keep = {s: (ops.dropout_mask(M, n, 40 + s, thr).double() if thr else torch.ones(M, n, device=_dev()).double())
            for s, n in ((0, 256), (1, 256), (2, out))}
    if thr:  # dropped activations are exactly zero in the saved tensors
        h1, h2 = h1 * keep[0].float(), h2 * keep[1].float()
    dz3, dz2, dz1, dh0 = (torch.empty(M, n, device=_dev()) for n in (out, 256, 256, 192 + F))
    ops.chain(M, [dict(img=pk.ptr("V3T"), K=out, N=256, gate=(h2, True, 41, thr, scale), out=dz2),
                  dict(img=pk.ptr("V2T"), K=256, N=256, gate=(h1, True, 40, thr, scale), out=dz1),
                  dict(img=pk.ptr("V1T"), K=256, N=192 + F, out=dh0)],
              A=gy, lda=out, K1=out, in_gate=(42, thr, scale), in_out=dz3 if thr else None, alpha=0.2, seed_t=seed, f16=False)
    sc = scale if thr else 1.0
    r3 = gy.double() * keep[2] * sc
    slope = lambda h: torch.where(h.double() > 0, torch.ones_like(h.double()), 0.2 * torch.ones_like(h.double()))
    r2 = (r3 @ W["V3"].double()) * slope(h2) * keep[1] * sc
    r1 = (r2 @ W["V2"].double()) * slope(h1) * keep[0] * sc
    r0 = r1 @ W["V1"].double()
    pairs = [(dz2, r2), (dz1, r1), (dh0, r0)] + ([(dz3, r3)] if thr else [])
    for got, ref in pairs:
        assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < TIGHT


@pytest.mark.parametrize("F", [32, 3])
def test_stacked_layer1_views(F):
    """fe.net.0 on [x_i ; x_j] split as a_i + c_j: the stacked image gives a | c in one launch, its
    transpose gives dx = [da | dc] [W1a ; W1c] + resid."""
    from mpgan_amd import ops
    rs = np.random.RandomState(5 + F)
    M = 450
    W, pk = _holder(rs, F, 32)
    x, b1 = _t(rs, M, F), _t(rs, 96)
    ac = torch.empty(M, 192, device=_dev())
    ops.chain(M, [dict(img=pk.ptr("W1S"), K=F, N=192, bias=b1, nbias=96, out=ac, wscale=ops.SC_WN)], A=x, lda=F, K1=F,
              f16=True, ascale=ops.SC_ACT)
    W1 = W["W1"].double()
    ref = torch.cat([x.double() @ W1[:, :F].t() + b1.double(), x.double() @ W1[:, F:].t()], 1)
    assert rel_err(ac.cpu().numpy(), ref.cpu().numpy()) < TIGHT
    da, dc, resid = _t(rs, 2, M, 96), _t(rs, M, 96), _t(rs, M, 40)[:, 8:8 + F]
    dx = torch.empty(M, F, device=_dev())
    ops.chain(M, [dict(img=pk.ptr("W1ST"), K=192, N=F, resid=resid, out=dx)], A=da, lda=96, K1=96, a_slabs=2,
              a_slab_stride=M * 96, A2=dc, lda2=96, f16=False)
    refx = da.double().sum(0) @ W1[:, :F] + dc.double() @ W1[:, F:] + resid.double()
    assert rel_err(dx.cpu().numpy(), refx.cpu().numpy()) < TIGHT


def test_pack_many_equals_pack_weights():
    """Every image mpg_pack_many builds is bit-identical to mpg_pack_weights' (same fragment order, same split)."""
    from mpgan_amd import ops
    rs = np.random.RandomState(8)
    W, pk = _holder(rs, 32, 32, dscale=2.0)
    singles = {
        # the edge network's images (forward and transposed) carry the power-of-two operand scales on top of the
        # dropout scale and are fp16; the node network's gradient images are plain bf16
        "W2": ops.pack_weights(W["W2"], 160, 96, scale=2.0 * ops.SC_W2, f16=True),
        "W3": ops.pack_weights(W["W3"], 192, 160, scale=2.0 * ops.SC_W3, f16=True),
        "W3T": ops.pack_weights(W["W3"], 192, 160, transpose=True, scale=2.0 * ops.SC_W3, f16=True),
        "W2T": ops.pack_weights(W["W2"], 160, 96, transpose=True, scale=2.0 * ops.SC_W2, f16=True),
        "V2": ops.pack_weights(W["V2"], 256, 256, scale=ops.SC_WN, f16=True), "V1T": ops.pack_weights(W["V1"], 256, 224, transpose=True),
    }
    for k, img in singles.items():
        assert torch.equal(img.view(torch.int16), pk.img[k].view(torch.int16)), k


def test_wgrad_group_and_accumulate():
    """Eleven dW = dY^T X (+ bias sums: every even job) as one grouped launch; a second flush with accumulate doubles the result."""
    from mpgan_amd import ops
    rs = np.random.RandomState(21)
    M = 7680
    # (the node network's own shapes with and without the bias column -- folded into tile column 0 at 256 inputs, inside the last
    # tile at 224 / 192 --, ragged tiles (160 x 130), a single column (96 x 3))
    shapes = [(32, 256), (256, 256), (256, 192), (256, 32), (96, 32), (96, 3), (256, 224), (160, 130), (256, 256), (128, 128), (160, 130)]
    dys = [_t(rs, M, n) for n, _ in shapes]
    xs = [_t(rs, M, k) for _, k in shapes]
    outs = [torch.zeros(n, k + 5, device=_dev()) for n, k in shapes]
    biases = [torch.zeros(n, device=_dev()) if i % 2 == 0 else None for i, (n, _) in enumerate(shapes)]
    for rep in range(2):
        wb = ops.WgradBatch()
        for dy, x, o, b in zip(dys, xs, outs, biases):
            wb.add(dy, x, out=o, out_col0=5, bias_out=b, accumulate=rep == 1)
        wb.flush()
    for dy, x, o, b in zip(dys, xs, outs, biases):
        ref = 2 * (dy.double().t() @ x.double())
        assert rel_err(o[:, 5:].cpu().numpy(), ref.cpu().numpy()) < TIGHT
        assert float(o[:, :5].abs().max()) == 0.0
        if b is not None:
            assert rel_err(b.cpu().numpy(), (2 * dy.double().sum(0)).cpu().numpy()) < TIGHT


def test_chain_rejects_bad_arguments():
    from mpgan_amd import ops
    rs = np.random.RandomState(1)
    W, pk = _holder(rs, 32, 32)
    x = _t(rs, 64, 32)
    y = torch.empty(64, 192, device=_dev())
    with pytest.raises(RuntimeError):  # LeakyReLU slope outside [0, 1]
        ops.chain(64, [dict(img=pk.ptr("W1S"), K=32, N=192, out=y)], A=x, lda=32, K1=32, alpha=1.5, f16=True)
    with pytest.raises(RuntimeError):  # K beyond the 256 features a fragment buffer holds
        ops.chain(64, [dict(img=pk.ptr("W1S"), K=300, N=192, out=y)], A=x, lda=32, K1=300, f16=True)


def test_packed_images_follow_parameter_updates():
    """MPLayer caches its weight images; an optimizer step (in-place update seen by autograd's version counter) or a
    load_state_dict must be picked up by the next forward, refresh_packed() covers writes torch cannot see."""
    from mpgan_amd.mpgan import MPLayer
    torch.manual_seed(0)
    layer = MPLayer(32, [96, 160, 192], [256, 256], 32).to(_dev())
    x = torch.randn(2, 30, 32, device=_dev())
    y0 = layer(x).detach().clone()
    opt = torch.optim.SGD(layer.parameters(), lr=0.05)
    layer(x).pow(2).mean().backward()
    opt.step()
    y1 = layer(x).detach().clone()
    assert torch.isfinite(y1).all() and (y1 - y0).abs().max() > 1e-5   # the update is visible ...
    fresh = MPLayer(32, [96, 160, 192], [256, 256], 32).to(_dev())
    fresh.load_state_dict(layer.state_dict())      # ... and equals a freshly packed layer with the same weights
    assert torch.equal(fresh(x), y1)
    with torch.no_grad():
        for q in layer.parameters():
            q.data.mul_(0.5)                       # a write behind autograd's back (as mpg_rmsprop does) ...
    layer.refresh_packed()                         # ... needs the explicit refresh
    fresh.load_state_dict(layer.state_dict())
    assert torch.equal(fresh(x), layer(x))


def test_linearnet_batch_norm_spectral_norm_vs_reference_golden():
    """LinearNet with both normalisations (mpgan/model.py:55-83, spectral_normalization.py): two training forwards (the power
    iteration and the running statistics move), gradients of the second, then an eval forward -- against the reference's own."""
    from conftest import load_golden
    from oracle import train_ref as T
    from mpgan_amd.mpgan import LinearNet
    g = load_golden("linearnet_bn_sn_f64.npz")
    net = LinearNet([24, 16], input_size=12, output_size=5, final_linear=True, batch_norm=True, spectral_norm=True, dropout_p=0.0).cuda()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items() if "running" not in k and "num_batches" not in k}
    sd = net.state_dict()
    sd.update({k: v.cuda() for k, v in T.init_state_dict(shapes, seed=int(g["seed"]), dtype=torch.float32).items()})
    net.load_state_dict(sd)
    net.train()
    dev = lambda a: torch.from_numpy(a).float().cuda()
    y1 = net(dev(g["x1"]))
    (y1 * dev(g["g1"])).sum().backward()
    net.zero_grad()
    x2 = dev(g["x2"]).requires_grad_(True)
    y2 = net(x2)
    (y2 * dev(g["g2"])).sum().backward()
    assert rel_err(y1.detach().cpu().numpy(), g["y1"]) < TIGHT
    assert rel_err(y2.detach().cpu().numpy(), g["y2"]) < TIGHT
    assert rel_err(x2.grad.cpu().numpy(), g["dx2"]) < TIGHT
    for k, p in net.named_parameters():
        if "grad__" + k in g:
            assert rel_err(p.grad.cpu().numpy(), g["grad__" + k]) < TIGHT, k
    net.eval()
    assert rel_err(net(x2.detach()).detach().cpu().numpy(), g["y3"]) < TIGHT
    for k, v in net.state_dict().items():
        if "state__" + k in g:
            assert rel_err(v.double().cpu().numpy(), g["state__" + k]) < TIGHT, k

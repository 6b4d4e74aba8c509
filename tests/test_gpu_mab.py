"""GPU parity of the one-launch attention block (mpg_mab_fwd / mpg_mab_bwd, csrc/mab.hip) against the reference goldens,
the fp64 oracle (exact dropout masks) and the block-by-block path."""
import numpy as np
import pytest
import torch

from conftest import load_golden, summarize, rel_err

pytestmark = pytest.mark.gpu
TIGHT = 1e-4
LA = {"leaky_relu_alpha": 0.2, "dropout_p": 0.0, "batch_norm": False, "spectral_norm": False}
SAB_ARGS = dict(embed_dim=64, ff_layers=[], final_linear=False, num_heads=4, layer_norm=False, dropout_p=0.0,
                linear_args=LA)


def _blocks(args=SAB_ARGS):
    from mpgan_amd.gapt import SAB, PMA, ISAB
    from oracle import train_ref as T
    E = args["embed_dim"]
    return {
        "sab": (lambda: SAB(**args), T._mab_shapes("mab", E)),
        "pma": (lambda: PMA(num_seeds=1, **args), {"S": (1, 1, E), **T._mab_shapes("mab", E)}),
        "isab": (lambda: ISAB(10, **args), {"I": (1, 10, E), **T._mab_shapes("mab0", E), **T._mab_shapes("mab1", E)}),
    }


@pytest.fixture
def launches():
    """Counts the calls of every C-ABI entry point while the test runs."""
    from mpgan_amd import _lib
    lib = _lib.lib()
    seen = {}

    class Proxy:
        def __getattr__(self, k):
            fn = getattr(lib, k)
            if not k.startswith("mpg_"):
                return fn

            def counted(*a):
                seen[k] = seen.get(k, 0) + 1
                return fn(*a)
            return counted

    saved = _lib._lib
    _lib._lib = Proxy()
    try:
        yield seen
    finally:
        _lib._lib = saved


@pytest.mark.parametrize("name,ci", [("m30", 0), ("u30", 1), ("m150", 2)])
def test_forward_vs_reference_golden(name, ci, launches):
    """SAB / PMA / ISAB outputs of the fused launch (no autograd) against the reference's own outputs."""
    from oracle import train_ref as T
    from mpgan_amd.gapt import _attn_mask
    g = load_golden(f"gapt_blocks_{name}_f64.npz")
    mask = torch.from_numpy(g["mask"]).float().cuda() if "mask" in g else None
    for bname, (ctor, shapes) in _blocks().items():
        blk = ctor().cuda()
        blk.load_state_dict(T.init_state_dict(shapes, 50 + ci, torch.float32))
        x = torch.from_numpy(g["x"]).float().cuda()
        launches.clear()
        with torch.no_grad():
            y = blk(x, _attn_mask(mask))
        n_mab = 2 if bname == "isab" else 1
        assert launches.get("mpg_mab_fwd") == n_mab and "mpg_gemm" not in launches and "mpg_attn_fwd" not in launches, launches
        assert rel_err(y.cpu().numpy(), g[f"{bname}_y"]) < TIGHT, bname


def test_forward_exact_dropout_and_embed32():
    """Dropout at its three sites with the exact masks (mpg_dropout_mask) fed to the fp64 oracle; E = 32 / 2 heads."""
    from oracle import train_ref as T, gapt_ref as R
    from mpgan_amd import ops
    from mpgan_amd.gapt import SAB, _attn_mask
    for E, H, p in ((64, 4, 0.5), (32, 2, 0.3)):
        la = dict(LA, dropout_p=p)
        args = dict(SAB_ARGS, embed_dim=E, num_heads=H, dropout_p=p, linear_args=la)
        blk = SAB(**args).cuda().train()
        sd = T.init_state_dict(T._mab_shapes("mab", E), 7, torch.float32)
        blk.load_state_dict(sd)
        B, N = 9, 30
        gen = torch.Generator().manual_seed(3)
        x = torch.randn(B, N, E, generator=gen)
        mask = (torch.rand(B, N, 1, generator=gen) < 0.8).float()
        mask[:, 0] = 1
        with torch.no_grad():
            y = blk(x.cuda(), _attn_mask(mask.cuda()))
        tag = ops.last_tag()
        thr, _ = ops.drop_params(p)
        keeps = {k: ops.dropout_mask(B * N, E, tag + site, thr).double().cpu().reshape(B, N, E)
                 for site, k in enumerate(("a", "f", "o"))}
        sd64 = {k: v.double() for k, v in sd.items()}
        ref = R.mab_forward(sd64, "mab", x.double(), x.double(), (1 - mask[:, :, 0]).bool(), num_heads=H, p=thr / 256.0, keeps=keeps)   # (byte-mode dropout quantises p to thr / 256)
        assert rel_err(y.cpu().numpy(), ref.numpy()) < TIGHT, (E, p)
        frac = float(sum(k.mean() for k in keeps.values()) / 3)
        assert abs(frac - (1 - p)) < 0.02


@pytest.mark.parametrize("name,ci", [("m30", 0), ("u30", 1), ("m150", 2)])
def test_blocks_forward_backward_vs_reference_golden(name, ci, launches):
    """SAB / PMA / ISAB through autograd: one launch each way per block, gradients against the reference's.  m150: sets of 150
    tokens (--num-hits 150) -- SAB 150 x 150, PMA 1 x 150, ISAB 10 x 150 and 150 x 10 -- on the large-set kernels (a workgroup per
    jet, a wave per tile of 32 tokens)."""
    from oracle import train_ref as T
    from mpgan_amd.gapt import _attn_mask
    g = load_golden(f"gapt_blocks_{name}_f64.npz")
    mask = torch.from_numpy(g["mask"]).float().cuda() if "mask" in g else None
    for bname, (ctor, shapes) in _blocks().items():
        blk = ctor().cuda()
        blk.load_state_dict(T.init_state_dict(shapes, 50 + ci, torch.float32))
        x = torch.from_numpy(g["x"]).float().cuda().requires_grad_(True)
        launches.clear()
        y = blk(x, _attn_mask(mask))
        (y * torch.from_numpy(g[f"{bname}_g"]).float().cuda()).sum().backward()
        n_mab = 2 if bname == "isab" else 1
        assert launches.get("mpg_mab_fwd") == n_mab and launches.get("mpg_mab_bwd") == n_mab, launches
        assert "mpg_attn_bwd" not in launches and "mpg_gate" not in launches, launches
        assert rel_err(y.detach().cpu().numpy(), g[f"{bname}_y"]) < TIGHT, bname
        assert rel_err(x.grad.cpu().numpy(), g[f"{bname}_dx"]) < TIGHT, bname
        for k, p in blk.named_parameters():
            assert rel_err(summarize(k, p.grad), g[f"{bname}_grad__{k}"]) < TIGHT, (bname, k)


def test_backward_exact_dropout_vs_oracle():
    """Training-mode block (dropout at all three sites, p = 1/2 and 0.3; E = 64 and 32; cross attention with 10 queries)
    against autograd on the fp64 oracle fed with the very masks the kernels drew."""
    from oracle import train_ref as T, gapt_ref as R
    from mpgan_amd import ops
    from mpgan_amd.gapt import MAB
    for E, H, p, L, N in ((64, 4, 0.5, 30, 30), (32, 2, 0.3, 30, 30), (64, 4, 0.5, 10, 30),
                          # large sets (33 ... 160 tokens): self-attention at 150 and 33, both cross shapes of an ISAB, E = 32
                          (64, 4, 0.5, 150, 150), (64, 4, 0.3, 33, 33), (64, 4, 0.5, 10, 150), (64, 4, 0.5, 150, 10),
                          (32, 2, 0.5, 97, 97), (64, 4, 0.5, 1, 160)):
        la = dict(LA, dropout_p=p)
        blk = MAB(E, H, ff_layers=[], final_linear=False, layer_norm=False, dropout_p=p, linear_args=la).cuda().train()
        sd = T.init_state_dict(T._mab_shapes("mab", E), 11, torch.float32)
        blk.load_state_dict({k[len("mab."):]: v for k, v in sd.items()})
        assert blk._fused_ok(torch.empty(1, device="cuda"), L, N)
        B = 7 if N <= 32 else 3
        gen = torch.Generator().manual_seed(5)
        yk = torch.randn(B, N, E, generator=gen)
        xq = yk if L == N else torch.randn(B, L, E, generator=gen)
        ign = torch.rand(B, N, generator=gen) > 0.8
        ign[:, 0] = False
        gy = torch.randn(B, L, E, generator=gen)
        xg = xq.cuda().requires_grad_(True)
        yg = xg if L == N else yk.cuda().requires_grad_(True)
        out = blk(xg, yg, ign.cuda())
        tag = ops.last_tag()
        (out * gy.cuda()).sum().backward()
        thr, _ = ops.drop_params(p)
        keeps = {k: ops.dropout_mask(B * L, E, tag + site, thr).double().cpu().reshape(B, L, E)
                 for site, k in enumerate(("a", "f", "o"))}
        sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
        xo = xq.double().requires_grad_(True)
        yo = xo if L == N else yk.double().requires_grad_(True)
        ref = R.mab_forward(sd64, "mab", xo, yo, ign, num_heads=H, p=thr / 256.0, keeps=keeps)
        (ref * gy.double()).sum().backward()
        assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < TIGHT, (E, p, L)
        assert rel_err(xg.grad.cpu().numpy(), xo.grad.numpy()) < TIGHT, (E, p, L)
        if L != N:
            assert rel_err(yg.grad.cpu().numpy(), yo.grad.numpy()) < TIGHT, (E, p, L)
        for k, q in blk.named_parameters():
            assert rel_err(q.grad.cpu().numpy(), sd64["mab." + k].grad.numpy()) < TIGHT, (E, p, L, k)


def test_fused_equals_block_by_block():
    """The same module with the fused launches switched off (mpg_gemm / mpg_attn_* / mpg_gate): outputs and gradients agree."""
    from oracle import train_ref as T
    from mpgan_amd.gapt import MAB, ISAB, _attn_mask
    blk = ISAB(10, **SAB_ARGS).cuda()
    blk.load_state_dict(T.init_state_dict({"I": (1, 10, 64), **T._mab_shapes("mab0", 64), **T._mab_shapes("mab1", 64)}, 3, torch.float32))
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(5, 30, 64, generator=gen).cuda()
    mask = (torch.rand(5, 30, 1, generator=gen) < 0.7).float().cuda()
    mask[:, 0] = 1
    res = []
    for fused in (True, False):
        MAB.fused = fused
        try:
            blk.zero_grad()
            xx = x.clone().requires_grad_(True)
            y = blk(xx, _attn_mask(mask))
            y.square().sum().backward()
            res.append((y.detach(), xx.grad, {k: p.grad.clone() for k, p in blk.named_parameters()}))
        finally:
            MAB.fused = True
    (y1, dx1, g1), (y0, dx0, g0) = res
    assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < TIGHT
    assert rel_err(dx1.cpu().numpy(), dx0.cpu().numpy()) < TIGHT
    for k in g1:
        assert rel_err(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < TIGHT, k


def test_layer_norm_block_exact_dropout_vs_oracle(launches):
    """``layer_norm=True`` blocks on the one-launch kernels (``ops.FusedMABLayerNormFn``: norm1 / norm2 inside ``mpg_mab_fwd`` /
    ``mpg_mab_bwd``): training mode with dropout at all three sites, E = 64 and 32, self- and cross-attention, padded keys,
    random norm weights -- against autograd on the fp64 oracle fed with the very masks the kernels drew: output, input
    gradients, the six block parameters and the four norm parameters.  One launch each way."""
    from oracle import train_ref as T, gapt_ref as R
    from mpgan_amd import ops
    from mpgan_amd.gapt import MAB
    for E, H, p, L, N in ((64, 4, 0.5, 30, 30), (32, 2, 0.3, 30, 30), (64, 4, 0.0, 10, 30), (32, 2, 0.5, 7, 30),
                          # large sets: the norms inside the workgroup-per-jet kernels
                          (64, 4, 0.5, 150, 150), (64, 4, 0.3, 10, 150), (32, 2, 0.5, 70, 70)):
        la = dict(LA, dropout_p=p)
        blk = MAB(E, H, ff_layers=[], final_linear=False, layer_norm=True, dropout_p=p, linear_args=la).cuda().train()
        shapes = dict(T._mab_shapes("mab", E))
        shapes.update({"mab.norm1.weight": (E,), "mab.norm1.bias": (E,), "mab.norm2.weight": (E,), "mab.norm2.bias": (E,)})
        sd = T.init_state_dict(shapes, 13, torch.float32)
        for k in ("mab.norm1.weight", "mab.norm2.weight"):   # (around 1, as a trained norm's)
            sd[k] = 1.0 + 0.3 * sd[k] / sd[k].abs().max()
        blk.load_state_dict({k[len("mab."):]: v for k, v in sd.items()})
        B = 9 if N <= 32 else 3
        gen = torch.Generator().manual_seed(6)
        yk = torch.randn(B, N, E, generator=gen)
        xq = yk if L == N else torch.randn(B, L, E, generator=gen)
        ign = torch.rand(B, N, generator=gen) > 0.8
        ign[:, 0] = False
        gy = torch.randn(B, L, E, generator=gen)
        xg = xq.cuda().requires_grad_(True)
        yg = xg if L == N else yk.cuda().requires_grad_(True)
        launches.clear()
        out = blk(xg, yg, ign.cuda())
        tag = ops.last_tag()
        (out * gy.cuda()).sum().backward()
        assert launches.get("mpg_mab_fwd") == 1 and launches.get("mpg_mab_bwd") == 1 and "mpg_layernorm_fwd" not in launches, launches
        thr, _ = ops.drop_params(p)
        keeps = ({k: ops.dropout_mask(B * L, E, tag + site, thr).double().cpu().reshape(B, L, E)
                  for site, k in enumerate(("a", "f", "o"))} if thr else None)
        sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
        xo = xq.double().requires_grad_(True)
        yo = xo if L == N else yk.double().requires_grad_(True)
        ref = R.mab_forward(sd64, "mab", xo, yo, ign, num_heads=H, p=thr / 256.0, keeps=keeps, layer_norm=True)
        (ref * gy.double()).sum().backward()
        assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < TIGHT, (E, p, L)
        assert rel_err(xg.grad.cpu().numpy(), xo.grad.numpy()) < TIGHT, (E, p, L)
        if L != N:
            assert rel_err(yg.grad.cpu().numpy(), yo.grad.numpy()) < TIGHT, (E, p, L)
        for k, q in blk.named_parameters():
            assert rel_err(q.grad.cpu().numpy(), sd64["mab." + k].grad.numpy()) < TIGHT, (E, p, L, k)


@pytest.mark.parametrize("kind", ["pma", "isab"])
def test_layer_norm_pooling_and_induced_blocks_equal_block_by_block(kind):
    """PMA (one seed row shared by all jets) and ISAB with ``layer_norm=True``: the one-launch route against the block-by-block one
    (``MAB.fused = False``: ``mpg_layernorm_*`` launches) -- outputs, input gradient and every parameter gradient, the seed /
    inducing points and the norms' parameters among them."""
    from mpgan_amd.gapt import MAB, PMA, ISAB, _attn_mask
    torch.manual_seed(4)
    args = dict(SAB_ARGS, layer_norm=True)
    blk = (PMA(num_seeds=1, **args) if kind == "pma" else ISAB(10, **args)).cuda()
    with torch.no_grad():
        for k, q in blk.named_parameters():
            if ".norm" in k:
                q.add_(0.2 * torch.randn_like(q))
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(6, 30, 64, generator=gen).cuda()
    mask = (torch.rand(6, 30, 1, generator=gen) < 0.7).float().cuda()
    mask[:, 0] = 1
    res = []
    for fused in (True, False):
        MAB.fused = fused
        try:
            blk.zero_grad()
            xx = x.clone().requires_grad_(True)
            y = blk(xx, _attn_mask(mask))
            y.square().sum().backward()
            res.append((y.detach(), xx.grad, {k: p.grad.clone() for k, p in blk.named_parameters()}))
        finally:
            MAB.fused = True
    (y1, dx1, g1), (y0, dx0, g0) = res
    assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) < TIGHT
    assert rel_err(dx1.cpu().numpy(), dx0.cpu().numpy()) < TIGHT
    assert any(".norm" in k for k in g1)
    for k in g1:
        assert rel_err(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < TIGHT, k


def test_large_set_kernels_on_small_sets_agree_with_the_one_wave_kernels(monkeypatch):
    """``MPG_MAB_BIG=1`` sends sets of <= 32 tokens through the large-set kernels too (a workgroup per jet, a wave per tile of 32
    tokens: running maximum / sum over the key tiles in the forward; statistics pass, dS pass and the key owners' pass in the
    backward): the same block through two independent schedules -- SAB with padded keys, cross attention 10 x 30, PMA's shared
    seed row, E = 32 -- agrees to rounding in the output and in every gradient."""
    from mpgan_amd.gapt import SAB, PMA, MAB, _attn_mask
    la = dict(LA, dropout_p=0.5)
    gen = torch.Generator().manual_seed(8)
    cases = [("sab", lambda: SAB(**dict(SAB_ARGS, dropout_p=0.5, linear_args=la)), 64),
             ("pma", lambda: PMA(num_seeds=1, **SAB_ARGS), 64),
             ("sab32", lambda: SAB(**dict(SAB_ARGS, embed_dim=32, num_heads=2)), 32)]
    for name, ctor, E in cases:
        torch.manual_seed(3)
        blk = ctor().cuda().train()
        x = torch.randn(7, 30, E, generator=gen).cuda()
        mask = (torch.rand(7, 30, 1, generator=gen) < 0.7).float().cuda()
        mask[:, 0] = 1
        res = []
        for big in (True, False):
            if big:
                monkeypatch.setenv("MPG_MAB_BIG", "1")
            else:
                monkeypatch.delenv("MPG_MAB_BIG", raising=False)
            from mpgan_amd import ops
            ops.set_seed(77)
            import itertools
            ops.dev_state(x.device).tags = itertools.count(5000)   # (the same dropout sites, hence masks, in both runs)
            blk.zero_grad()
            xx = x.clone().requires_grad_(True)
            y = blk(xx, _attn_mask(mask))
            y.square().sum().backward()
            res.append((y.detach().cpu().numpy(), xx.grad.cpu().numpy(), {k: q.grad.cpu().numpy().copy() for k, q in blk.named_parameters()}))
        (y1, dx1, g1), (y0, dx0, g0) = res
        assert np.isfinite(y1).all() and np.isfinite(dx1).all(), name
        assert rel_err(y1, y0) < 1e-5, name
        assert rel_err(dx1, dx0) < 1e-4, name
        for k in g1:
            assert rel_err(g1[k], g0[k]) < 1e-4, (name, k)

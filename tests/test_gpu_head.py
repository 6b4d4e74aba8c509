"""GPU parity of the per-jet pieces around the message-passing layers (csrc/head.hip): rank mask, generator tail,
discriminator head with and without the fused loss -- against plain torch restatements of the reference lines."""
import numpy as np
import pytest
import torch

from conftest import rel_err, load_golden

pytestmark = pytest.mark.gpu


def _ref_rank_mask(x0, labels, N):
    """mpgan/model.py:689-699 as written there: double argsort."""
    n = (labels[:, -1] * N).int() - 1
    rank = x0.argsort(dim=1).argsort(dim=1)
    return (rank <= n.unsqueeze(1)).float()


@pytest.mark.parametrize("B,N", [(7, 30), (3, 150), (2, 1), (5, 33)])
def test_rank_mask(B, N):
    from mpgan_amd import ops
    g = torch.Generator(device="cuda").manual_seed(B * N)
    x = torch.randn(B, N, 8, device="cuda", generator=g)
    n = torch.randint(1, N + 1, (B,), device="cuda", generator=g)
    labels = (n.float() * np.float32(1.0 / N)).reshape(B, 1)
    m = ops.rank_mask(x[:, :, 0], labels, N)
    assert torch.equal(m, _ref_rank_mask(x[:, :, 0], labels, N))
    assert torch.equal(m.sum(1).int(), n.int())
    # ties are ordered by index
    xt = torch.zeros(2, 6, 1, device="cuda")
    lt = torch.tensor([[0.5], [1.0]], device="cuda")
    mt = ops.rank_mask(xt[:, :, 0], lt, 6)
    assert mt.tolist() == [[1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1]]


@pytest.mark.parametrize("act", ["tanh", "", "sigmoid"])
def test_gen_tail(act):
    from mpgan_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    B, N, F = 5, 30, 3
    y = torch.randn(B, N, F, device="cuda", generator=g, requires_grad=True)
    mask = (torch.rand(B, N, 1, device="cuda", generator=g) < 0.7).float()
    up = torch.randn(B, N, F + 1, device="cuda", generator=g)
    out = ops.GenTailFn.apply(y, mask, ops.ACT_CODES[act])
    (out * up).sum().backward()
    yr = y.detach().double().requires_grad_(True)
    a = torch.tanh(yr) if act == "tanh" else (torch.sigmoid(yr) if act == "sigmoid" else yr)
    ref = torch.cat((a, mask.double() - 0.5), 2)
    (ref * up.double()).sum().backward()
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-6
    assert rel_err(y.grad.cpu().numpy(), yr.grad.cpu().numpy()) < 1e-6
    # into a caller-owned strided buffer (second half of a batch)
    big = torch.zeros(2 * B, N, F + 1, device="cuda")
    with torch.no_grad():
        ops.gen_tail_into(y.detach(), mask, ops.ACT_CODES[act], big[B:])
    assert torch.equal(big[B:], out.detach()) and float(big[:B].abs().sum()) == 0.0


def _ref_head(y, mask, w, b, mean, sigmoid, keep=None):
    pooled = (y * mask).sum(1) if mask is not None else y.sum(1)
    if mean:
        pooled = pooled / (mask.sum(1) + 1e-12) if mask is not None else pooled / y.shape[1]
    z = pooled @ w.t() + b
    if keep is not None:
        z = z * keep
    return torch.sigmoid(z) if sigmoid else z


@pytest.mark.parametrize("mean,sigmoid,use_mask,N,F", [(False, True, True, 30, 32), (True, True, True, 30, 32),
                                                         (False, False, True, 150, 32), (True, False, False, 7, 32),
                                                         (False, True, False, 1, 64)])
def test_disc_head_autograd(mean, sigmoid, use_mask, N, F):
    from mpgan_amd import ops
    g = torch.Generator(device="cuda").manual_seed(N + F)
    B = 9
    y = torch.randn(B, N, F, device="cuda", generator=g).mul_(0.3).requires_grad_(True)
    mask = (torch.rand(B, N, 1, device="cuda", generator=g) < 0.7).float() if use_mask else None
    w = torch.randn(1, F, device="cuda", generator=g).mul_(0.2).requires_grad_(True)
    b = torch.randn(1, device="cuda", generator=g).requires_grad_(True)
    up = torch.randn(B, device="cuda", generator=g)
    out = ops.DiscHeadFn.apply(y, mask, w, b, mean, sigmoid, 0.0, False)
    (out * up).sum().backward()
    yr, wr, br = (t.detach().double().requires_grad_(True) for t in (y, w, b))
    ref = _ref_head(yr, None if mask is None else mask.double(), wr, br, mean, sigmoid).reshape(-1)
    (ref * up.double()).sum().backward()
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 1e-5
    for a, r in ((y.grad, yr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
        assert rel_err(a.cpu().numpy(), r.cpu().numpy()) < 1e-5


def test_disc_head_dropout_mask_is_the_helper_mask():
    """Dropout on the head's Linear output (LinearNet puts one after every layer): the keep decisions are those of
    mpg_dropout_mask for the same site, the scale 1 / (1 - p)."""
    from mpgan_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    B, N, F = 64, 30, 32
    y = torch.randn(B, N, F, device="cuda", generator=g).mul_(0.3)
    w = torch.randn(1, F, device="cuda", generator=g).mul_(0.2)
    b = torch.zeros(1, device="cuda") + 0.1
    ops.set_seed(77)
    out = ops.DiscHeadFn.apply(y, None, w, b, False, False, 0.5, True)
    tag = ops.last_tag("cuda") + ops.TAG_GENERIC
    keep = ops.dropout_mask(B, 1, tag, 128).reshape(B)
    ref = _ref_head(y.double(), None, w.double(), b.double(), False, False).reshape(-1) * keep.double() * 2.0
    assert 0.2 < float(keep.mean()) < 0.8
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("loss", ["ls", "og", "w", "hinge"])
@pytest.mark.parametrize("gen_step", [False, True])
def test_disc_head_fused_loss(loss, gen_step):
    """ops.disc_head_loss == head + calc_D_loss / calc_G_loss (oracle definitions) + autograd, in value and gradients."""
    from mpgan_amd import ops
    from oracle import train_ref as T
    g = torch.Generator(device="cuda").manual_seed(11)
    B, N, F = 8, 30, 32
    nj = B if gen_step else 2 * B
    sigmoid = loss in ("ls", "og")
    y = torch.randn(nj, N, F, device="cuda", generator=g).mul_(0.3)
    mask = (torch.rand(nj, N, 1, device="cuda", generator=g) < 0.7).float()
    w = torch.randn(1, F, device="cuda", generator=g).mul_(0.2)
    b = torch.randn(1, device="cuda", generator=g)
    loss_out = torch.zeros((), device="cuda")
    dw, db = torch.zeros(1, F, device="cuda"), torch.zeros(1, device="cuda")
    out, dy = ops.disc_head_loss(y, mask, w, b, mean=False, sigmoid=sigmoid, p_drop=0.0, training=True, loss=loss,
                                 n_real=B, gen_step=gen_step, count=B, loss_out=loss_out, wgrad=(dw, db))
    yr, wr, br = (t.double().requires_grad_(True) for t in (y, w, b))
    o = _ref_head(yr, mask.double(), wr, br, False, sigmoid)
    L = T.g_loss_ref(loss, o) if gen_step else T.d_loss_ref(loss, o[:B], o[B:])
    L.backward()
    assert abs(float(loss_out) - float(L)) < 1e-5 * max(abs(float(L)), 1e-3)
    assert rel_err(out.cpu().numpy(), o.detach().reshape(-1).cpu().numpy()) < 1e-5
    assert rel_err(dy.cpu().numpy(), yr.grad.cpu().numpy()) < 1e-5
    assert rel_err(dw.cpu().numpy(), wr.grad.cpu().numpy()) < 1e-5
    assert rel_err(db.cpu().numpy(), br.grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("loss", ["og", "ls", "w", "hinge"])
def test_disc_head_fused_loss_vs_reference_golden(loss):
    """The loss branches of ``mpg_disc_head_bwd`` against calc_D_loss / calc_G_loss EXECUTED from the reference's source
    (tests/golden/losses.npz): a head with one particle, one feature, weight 1, no activation hands the golden's
    discriminator outputs to the loss as they are -- BCELoss's clamps at the ends of (0, 1) and inactive hinge terms included."""
    from mpgan_amd import ops
    g = load_golden("losses.npz")
    r, f = g[f"{loss}_out_r"], g[f"{loss}_out_f"]
    B = r.shape[0]
    w, b = torch.ones(1, 1, device="cuda"), torch.zeros(1, device="cuda")
    for gen_step in (False, True):
        outs = f if gen_step else np.concatenate([r, f])
        y = torch.from_numpy(outs).float().cuda().reshape(-1, 1, 1)
        loss_out = torch.zeros((), device="cuda")
        out, dy = ops.disc_head_loss(y, None, w, b, mean=False, sigmoid=False, p_drop=0.0, training=True, loss=loss,
                                     n_real=B, gen_step=gen_step, count=B, loss_out=loss_out)
        want_L = float(g[f"{loss}_G_loss"] if gen_step else g[f"{loss}_D_loss"])
        want_g = g[f"{loss}_dG_df"] if gen_step else np.concatenate([g[f"{loss}_dD_dr"], g[f"{loss}_dD_df"]])
        assert abs(float(loss_out) - want_L) < 1e-5 * max(1.0, abs(want_L)), (loss, gen_step)
        assert rel_err(dy.reshape(-1).cpu().numpy(), want_g.reshape(-1)) < 1e-5, (loss, gen_step)


def test_normal_noise_stream():
    """``mpg_normal`` (the generator's input noise inside a captured iteration): moments of N(0, 0.2^2) over 2e6 values, no
    correlation between neighbours or between the two values of a Box-Muller pair, a Kolmogorov distance to the normal CDF
    at the sampling-noise level, the same values for the same (seed, site) and unrelated ones for another site or seed."""
    from mpgan_amd import ops
    import math
    dev = torch.device("cuda:0")
    ops.set_seed(2024, dev)
    n = 2_000_001   # (odd: the last pair is half used)
    a = ops.normal_noise((n,), 0.2, site=0, device=dev)
    b = ops.normal_noise((n,), 0.2, site=0, device=dev)
    c = ops.normal_noise((n,), 0.2, site=1, device=dev)
    ops.bump_seed(dev)
    d = ops.normal_noise((n,), 0.2, site=0, device=dev)
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, d)
    z = (a.double() / 0.2).cpu()
    assert bool(torch.isfinite(z).all())
    se = 1.0 / math.sqrt(n)
    assert abs(float(z.mean())) < 5 * se
    assert abs(float(z.var()) - 1.0) < 5 * math.sqrt(2.0) * se
    assert abs(float((z ** 3).mean())) < 5 * math.sqrt(15.0) * se
    assert abs(float((z ** 4).mean()) - 3.0) < 5 * math.sqrt(96.0) * se
    for x, y in ((z[:-1], z[1:]), (z[0:-1:2], z[1::2]), (z[:-2], z[2:])):
        assert abs(float((x * y).mean())) < 5 / math.sqrt(x.numel())
    for x, y in ((z, (c.double() / 0.2).cpu()), (z, (d.double() / 0.2).cpu())):
        assert abs(float((x * y).mean())) < 5 * se
    zs, _ = torch.sort(z)
    cdf = 0.5 * (1 + torch.erf(zs / math.sqrt(2.0)))
    ks = float((cdf - torch.arange(1, n + 1, dtype=torch.float64) / n).abs().max())
    assert ks < 2.0 / math.sqrt(n), ks      # (1.36 / sqrt(n) is the 5 % point of the Kolmogorov statistic)
    assert float(z.abs().max()) > 4.5       # the tails are there


def test_noise_and_masks_from_one_launch_equal_the_two_calls():
    """``ops.normal_noise_masked`` (mpg_normal_rank_mask) against ``ops.normal_noise`` + ``ops.rank_mask``: the same noise,
    the same masks, bit for bit; odd batch, labels as the trainer holds them."""
    from mpgan_amd import ops
    dev = torch.device("cuda:0")
    B, N, L = 37, 30, 32
    rs = np.random.RandomState(1)
    labels = torch.from_numpy(rs.randint(1, N + 1, size=(B, 1)) / N).float().to(dev)
    ops.set_seed(2024, dev)
    z0 = ops.normal_noise((B, N, L), 0.2, site=1, device=dev)
    m0, i0 = ops.rank_mask(z0[:, :, 0], labels, N, with_ignore=True)
    mask_rows = torch.full((2 * B, N), -1.0, device=dev)
    z1, m1, i1 = ops.normal_noise_masked((B, N, L), 0.2, labels, site=1, device=dev, mask_out=mask_rows[B:])
    assert torch.equal(z0, z1) and torch.equal(m0, m1) and torch.equal(i0, i1)
    assert torch.equal(mask_rows[B:], m0) and bool((mask_rows[:B] == -1).all())
    assert float(m0.sum(1).sub(labels[:, 0] * N).abs().max()) < 0.5


@pytest.mark.parametrize("B,N,L", [(1, 30, 64), (5, 150, 32), (3, 31, 2)])
def test_noise_and_masks_from_one_launch_other_sizes(B, N, L):
    """``mpg_normal_rank_mask`` at one jet, at 150 particles and at an odd particle count: the two-call values bit for bit; an odd
    number of values per jet is refused (pairs of values must not straddle jets)."""
    from mpgan_amd import ops, _lib
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(B)
    labels = torch.from_numpy(rs.randint(1, N + 1, size=(B, 1)) / N).float().to(dev)
    ops.set_seed(7, dev)
    z0 = ops.normal_noise((B, N, L), 0.2, site=0, device=dev)
    m0, i0 = ops.rank_mask(z0[:, :, 0], labels, N, with_ignore=True)
    z1, m1, i1 = ops.normal_noise_masked((B, N, L), 0.2, labels, site=0, device=dev)
    assert torch.equal(z0, z1) and torch.equal(m0, m1) and torch.equal(i0, i1)
    out = torch.empty(3 * 5, device=dev)
    rc = _lib.lib().mpg_normal_rank_mask(ops._p(out), 1, 3, 5, ops._p(ops.seed_tensor(dev)), 0, 0.0, 1.0, ops._p(labels), 1, ops._p(m0), None, None)
    assert rc == -1


def test_slab_sums_equal_sequential_adds():
    """``mpg_slab_sums``: [da | dc] rows from the sender chunks' da slabs and the receiver blocks' dc slabs of a sender-chunked data-gradient
    launch -- slab by slab in index order: bit-identical to the same adds written out in torch; refusals for misaligned / odd shapes."""
    import ctypes as C
    from mpgan_amd import _lib, ops
    gen = torch.Generator().manual_seed(3)
    for SA, SB, M, cols in ((3, 5, 2400, 96), (1, 2, 33, 96), (4, 1, 7, 8)):
        A = torch.randn(SA, M, cols, generator=gen).cuda()
        Bm = torch.randn(SB, M, cols, generator=gen).cuda()
        out = torch.empty(M, 2 * cols, device="cuda")
        rc = _lib.lib().mpg_slab_sums(ops._p(A), SA, M * cols, ops._p(Bm), SB, M * cols, ops._p(out), M, cols, ops._stream())
        assert rc == 0
        ra, rb = A[0].clone(), Bm[0].clone()
        for q in range(1, SA):
            ra = ra + A[q]
        for q in range(1, SB):
            rb = rb + Bm[q]
        assert torch.equal(out[:, :cols], ra) and torch.equal(out[:, cols:], rb), (SA, SB, M, cols)
    A = torch.zeros(2, 4, 6, device="cuda")
    out = torch.empty(4, 12, device="cuda")
    assert _lib.lib().mpg_slab_sums(ops._p(A), 2, 24, ops._p(A), 2, 24, ops._p(out), 4, 6, ops._stream()) == -2      # cols % 4
    A = torch.zeros(2 * 4 * 8 + 1, device="cuda")
    assert _lib.lib().mpg_slab_sums(ops._p(A, 1), 2, 32, ops._p(A, 1), 2, 32, ops._p(out), 4, 8, ops._stream()) == -5  # 16-byte alignment

"""CPU, world_size 2 over gloo: the data-parallel protocol of the product's ``TrainStep`` -- parameter broadcast,
local-batch losses, ONE flat-bucket SUM all-reduce per network between the graph segments, 1/world folded into the
optimiser step, per-rank noise / dropout seeds -- reproduces the single-process global-batch iteration.

``TrainStep`` itself runs here (its host logic is device-agnostic) on toy CPU networks; the one thing replaced is
the fused optimiser launch (``FlatParams.step`` -> the same RMSprop arithmetic in torch), because libmpgan_amd has
no CPU path.  The fused kernels under the same protocol are covered on the GPU box (tests/test_gpu_dist.py).
"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N, LAT = 6, 5


class ToyG(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(LAT, 3)

    def forward(self, noise, labels):
        x = torch.tanh(self.lin(noise))
        return torch.cat([x, torch.full_like(x[..., :1], 0.5)], 2)


class ToyD(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(4 * N, 7), torch.nn.LeakyReLU(0.2), torch.nn.Linear(7, 1))

    def forward(self, x, labels):
        return torch.sigmoid(self.net(x.reshape(x.shape[0], -1)) + labels)


def _torch_rmsprop(self, lr, gscale=1.0, zero_grad=False, advance_seed=None):
    """FlatParams.step for CPU tensors: mpg_rmsprop's arithmetic (csrc/optim.hip) in torch."""
    if advance_seed is not None:
        from mpgan_amd import ops
        advance_seed.add_(ops.SEED_STEP)
    self.last_grad = self.grad.clone()      # (what the step consumed: the buffer itself is cleared below)
    g = self.grad * gscale
    self.sq.mul_(0.99).addcmul_(g, g, value=0.01)
    self.flat.addcdiv_(g, self.sq.sqrt() + 1e-8, value=-lr)
    if zero_grad:
        self.grad.zero_()
    self._host_steps += 1
    self.lr = lr


def _inputs(B):
    g = torch.Generator().manual_seed(7)
    data = torch.randn(B, N, 4, generator=g)
    labels = torch.rand(B, 1, generator=g)
    nD = torch.randn(B, N, LAT, generator=g) * 0.2
    nG = torch.randn(B, N, LAT, generator=g) * 0.2
    return data, labels, nD, nG


def _run(world, rank, pg, steps=3, loss="ls"):
    from mpgan_amd import dist as mdist, train
    train.FlatParams.step = _torch_rmsprop
    torch.manual_seed(100 + rank)       # ranks start from different weights ...
    G, D = ToyG(), ToyD()
    if world > 1:
        mdist.broadcast_module(G, 0, pg)  # ... and are brought to rank 0's
        mdist.broadcast_module(D, 0, pg)
    Bg = 8
    B = Bg // world
    data, labels, nD, nG = _inputs(Bg)
    sl = slice(rank * B, (rank + 1) * B)
    ts = train.TrainStep(G, D, B, N, latent=LAT, lr_disc=1e-2, lr_gen=2e-2, use_graphs=False, process_group=pg,
                         world_size=world, loss=loss)
    ts.set_batch(data[sl], labels[sl])
    ts.fixed_noise = (nD[sl], nG[sl])
    for _ in range(steps):
        ts.step()
    return ts.fD.flat.clone(), ts.fG.flat.clone(), ts.fD.last_grad.clone(), float(ts.D_loss), ts


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mpgan_amd import dist as mdist
    r, w, pg = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    fD, fG, gD, dl, ts = _run(world, rank, pg)
    assert ts.fD.steps == 3 and ts.fG.steps == 3
    out[rank] = (fD, fG, gD, dl)
    assert len({mdist.rank_seed(4, r) for r in range(world)}) == world
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,port", [(2, 29517), (4, 29541)])
def test_multi_rank_trainstep_equals_global_batch(world, port):
    """world 2 and 4 (the driver's scaling points below 8; eight ranks of 1 jet each would only repeat the protocol)."""
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    d0, g0, gr0, _ = out[0]
    for r in range(1, world):
        dr, gr_, grr, _ = out[r]
        assert torch.equal(d0, dr) and torch.equal(g0, gr_)   # broadcast + identical averaged updates on every rank
        assert torch.equal(gr0, grr)                          # the all-reduced (summed) gradient buffer
    sys.path.insert(0, ROOT)
    sD, sG, sgr, sl, _ = _run(1, 0, None)
    assert torch.allclose(sD, d0, rtol=1e-4, atol=1e-6), float((sD - d0).abs().max())
    assert torch.allclose(sG, g0, rtol=1e-4, atol=1e-6), float((sG - g0).abs().max())
    # summed local-mean gradients / world = global-batch mean gradient (last iteration's D gradient)
    assert torch.allclose(sgr, gr0 / world, rtol=1e-3, atol=1e-6)
    assert abs(sum(out[r][3] for r in range(world)) / world - sl) < 1e-5   # local losses average to the global loss


def _resume_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mpgan_amd import dist as mdist, ops, train
    r, w, pg = mdist.init_from_env("gloo")
    train.FlatParams.step = _torch_rmsprop
    cpu = torch.device("cpu")
    torch.manual_seed(5)
    G, D = ToyG(), ToyD()
    ts = train.TrainStep(G, D, 4, N, latent=LAT, use_graphs=False, process_group=pg, world_size=world)
    # (the device seed belongs to the GPU path; its host logic -- what is saved, what a rank makes of it -- is device-agnostic)
    ts.fG.seed_device, ts.fG.seed_rank = cpu, rank
    ops.set_seed(ops.derived_seed(torch.initial_seed(), rank), cpu, _auto=True)
    data, labels, nD, nG = _inputs(8)
    ts.set_batch(data[rank * 4:rank * 4 + 4], labels[rank * 4:rank * 4 + 4])
    ts.fixed_noise = (nD[rank * 4:rank * 4 + 4], nG[rank * 4:rank * 4 + 4])
    ts.step(); ts.step()
    mine = ops.get_seed(cpu)                          # two iterations on, on this rank
    assert mine == (ops.derived_seed(5, rank) + 2 * ops.SEED_STEP) & 0xFFFFFFFFFFFFFFFF
    # the reference's checkpoint: ONE G_optim_<epoch>.pt, written by rank 0, read by everyone (train.py:534-535)
    box = [ts.optimizer_state_dicts() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=pg)
    sdD, sdG = box[0]
    assert sdG["param_groups"][0][train.FlatParams.SEED_RANK_KEY] == 0
    ops.set_seed(999, cpu)
    ts.load_optimizer_state_dicts(sdD, sdG)
    out[rank] = (mine, ops.get_seed(cpu), sdG["param_groups"][0][train.FlatParams.SEED_KEY])
    dist.destroy_process_group()


def test_two_rank_resume_gives_every_rank_its_own_seed():
    """A data-parallel run resumed from the single optimizer checkpoint must not put rank 0's noise / dropout seed on every
    rank: the saving rank continues its stream bit for bit, the others move the saved value by their distance in rank."""
    world = 2
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_resume_worker, args=(world, 29533, out), nprocs=world, join=True)
    (mine0, got0, saved0), (mine1, got1, saved1) = out[0], out[1]
    assert saved0 == saved1 == mine0
    assert got0 == mine0                                  # rank 0: resume == uninterrupted run
    assert got1 != got0 and got1 != 999                   # rank 1: a stream of its own
    sys.path.insert(0, ROOT)
    from mpgan_amd import ops
    assert got1 == ops.rerank_seed(saved0, 0, 1)
    assert len({ops.rerank_seed(saved0, 0, r) for r in range(8)}) == 8
    assert ops.rerank_seed(ops.rerank_seed(saved0, 0, 5), 5, 0) == saved0


def test_losses_match_oracle_definitions():
    """d_loss / g_loss (slice-free forms) == the restated calc_D_loss / calc_G_loss for every loss choice."""
    sys.path.insert(0, ROOT)
    from mpgan_amd import train
    from oracle import train_ref as T
    g = torch.Generator().manual_seed(3)
    B = 16
    for loss in train.LOSSES:
        raw = torch.randn(2 * B, 1, generator=g, dtype=torch.float64)
        out = torch.sigmoid(raw) if loss in ("ls", "og") else raw
        a = train.d_loss(loss, out, B)
        b = T.d_loss_ref(loss, out[:B], out[B:])
        assert abs(float(a) - float(b)) < 1e-12, loss
        assert abs(float(train.g_loss(loss, out[B:])) - float(T.g_loss_ref(loss, out[B:]))) < 1e-12, loss


class ToyD2(torch.nn.Module):
    """A plain-torch discriminator whose labels are optional (gradient_penalty calls D(interpolated), train.py:301)."""

    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(4 * N, 7), torch.nn.Tanh(), torch.nn.Linear(7, 1))

    def forward(self, x, labels=None):
        return torch.sigmoid(self.net(x.reshape(x.shape[0], -1)))


def test_gradient_penalty_host_logic_on_a_plain_torch_discriminator():
    """``TrainStep(gp_lambda=...)`` with a discriminator that is plain torch (twice differentiable as it is): the D step's
    gradients are those of  D_loss + gp_lambda * mean_b (||dD/dx_b|| - 1)^2  (train.py:286-324) written out directly --
    the host logic of the penalty (interpolation, detached generated jets, create_graph, what is added to the loss)."""
    sys.path.insert(0, ROOT)
    from mpgan_amd import train
    train.FlatParams.step = _torch_rmsprop
    torch.manual_seed(5)
    G, D = ToyG(), ToyD2()
    B = 6
    data, labels, nD, nG = _inputs(B)
    ts = train.TrainStep(G, D, B, N, latent=LAT, use_graphs=False, loss="w", gp_lambda=10.0, lr_disc=0.0)
    ts.set_batch(data, labels)
    ts.fixed_noise = (nD, nG)
    ts.fixed_alpha = torch.rand(B, 1, 1, generator=torch.Generator().manual_seed(1))
    ts._seg_D()
    got = ts.fD.grad.clone()
    with torch.no_grad():
        fake = G(nD, labels)
    x = (ts.fixed_alpha * data + (1 - ts.fixed_alpha) * fake).requires_grad_(True)
    prob = D(x)
    (gx,) = torch.autograd.grad(prob, x, torch.ones_like(prob), create_graph=True)
    gp = 10.0 * ((torch.sqrt((gx.reshape(B, -1) ** 2).sum(1) + 1e-12) - 1) ** 2).mean()
    base = -D(data, labels).mean() + D(fake, labels).mean()
    ref = torch.autograd.grad(base + gp, list(D.parameters()))
    ref = torch.cat([r.reshape(-1) for r in ref])
    assert abs(float(ts.GP) - float(gp)) < 1e-6 * abs(float(gp))
    assert abs(float(ts.D_loss) - float(base)) < 1e-6
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-7), float((got - ref).abs().max())


def test_loss_functions_vs_reference_golden():
    """``train.d_loss`` / ``train.g_loss`` against calc_D_loss / calc_G_loss EXECUTED from the reference's source
    (tests/golden/losses.npz): value and gradient of all four branches."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from mpgan_amd import train
    g = np.load(os.path.join(ROOT, "tests", "golden", "losses.npz"))
    for loss in train.LOSSES:
        r, f = torch.from_numpy(g[f"{loss}_out_r"]), torch.from_numpy(g[f"{loss}_out_f"])
        B = r.shape[0]
        out = torch.cat([r, f]).requires_grad_(True)
        L = train.d_loss(loss, out, B)
        (go,) = torch.autograd.grad(L, out)
        ref = float(g[f"{loss}_D_loss"])
        assert abs(float(L) - ref) < 1e-12 * max(1.0, abs(ref)), loss
        want = np.concatenate([g[f"{loss}_dD_dr"], g[f"{loss}_dD_df"]])
        assert np.abs(go.numpy() - want).max() < 1e-12 * max(1.0, np.abs(want).max()), loss
        f2 = f.clone().requires_grad_(True)
        Lg = train.g_loss(loss, f2)
        (gg,) = torch.autograd.grad(Lg, f2)
        ref = float(g[f"{loss}_G_loss"])
        assert abs(float(Lg) - ref) < 1e-12 * max(1.0, abs(ref)), loss
        assert np.abs(gg.numpy() - g[f"{loss}_dG_df"]).max() < 1e-12 * max(1.0, np.abs(g[f"{loss}_dG_df"]).max()), loss

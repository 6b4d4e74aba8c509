"""CPU, world_size 2 over gloo: the data-parallel protocol (flat gradient bucket, SUM all-reduce with
1/world folded into the update, per-rank seeds, parameter broadcast) reproduces the single-process
global-batch gradient."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mpgan_amd import dist as mdist
    from mpgan_amd.train import FlatParams
    r, w, pg = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # ranks start from different weights ...
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LeakyReLU(0.2), torch.nn.Linear(5, 1))
    mdist.broadcast_module(net, 0, pg)  # ... and are brought to rank 0's
    flat = FlatParams(net)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g)
    Y = torch.randn(8, 1, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    flat.zero_grad()
    ((net(xs) - ys) ** 2).mean().backward()  # local-batch mean loss, as every rank computes it
    scale = mdist.allreduce_sum_(flat.grad, pg, w)
    out[rank] = (flat.flat.clone(), flat.grad.clone() * scale)
    assert mdist.rank_seed(4, 0) != mdist.rank_seed(4, 1)
    dist.destroy_process_group()


def test_two_rank_gradient_equals_global_batch():
    world = 2
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, 29517, out), nprocs=world, join=True)
    p0, g0 = out[0]
    p1, g1 = out[1]
    assert torch.equal(p0, p1)            # broadcast worked
    assert torch.allclose(g0, g1)         # every rank holds the same averaged gradient
    # single-process reference on the global batch
    sys.path.insert(0, ROOT)
    from mpgan_amd.train import FlatParams
    torch.manual_seed(100)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LeakyReLU(0.2), torch.nn.Linear(5, 1))
    flat = FlatParams(net)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 6, generator=g)
    Y = torch.randn(8, 1, generator=g)
    ((net(X) - Y) ** 2).mean().backward()
    assert torch.allclose(flat.flat, p0)
    assert torch.allclose(flat.grad, g0, atol=1e-6)

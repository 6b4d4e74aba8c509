"""GPU: the data-parallel shape of the iteration on hardware.

* a world-size-1 ``nccl`` (= RCCL) group: ``TrainStep`` takes its multi-rank path -- three hipGraph segments with a
  real ``dist.all_reduce`` of the flat gradient bucket between them -- and must reproduce eager execution bit for bit;
* two ranks sharing the one GPU of the box over ``gloo`` (RCCL refuses two ranks on one device): the fused kernels
  under the full protocol (parameter broadcast, per-rank shards, SUM all-reduce, 1/world in the optimiser) against
  the single-process global-batch run.
The N = 2..8 RCCL runs themselves belong to the driver's scaling bench (``bench.py --gpus N``).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world1_nccl_group_three_segments_equal_eager():
    import torch.distributed as dist
    from test_gpu_train import _three_steps
    a = _three_steps(16, 30, use_graphs=False)
    # (the reference point without the side streams: the weight-gradient stream joins BEFORE each all-reduce -- in the
    # three-segment form and inside the one-graph form alike -- so every run below must land on these very bits)
    os.environ.update(MPG_WGRAD_SIDE="0", MPG_GEN_AHEAD="0")
    try:
        a0 = _three_steps(16, 30, use_graphs=False)
    finally:
        os.environ.pop("MPG_WGRAD_SIDE", None); os.environ.pop("MPG_GEN_AHEAD", None)
    assert torch.equal(a[0], a0[0]) and torch.equal(a[1], a0[1]) and a[2:] == a0[2:]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        # graphs + collective: every segment is a graph of its own, so the generator-ahead branch joins at the end of the D segment
        b = _three_steps(16, 30, use_graphs=True, pg=dist.group.WORLD, gen_join="seg_D")
        # eager + collective: the D all-reduce is ordered behind D's backward and its weight-gradient stream only -- the
        # generator-ahead branch stays open across it and joins where the G step takes its jets
        c = _three_steps(16, 30, use_graphs=False, pg=dist.group.WORLD, gen_join="seg_G")
        # the two all-reduces captured INSIDE the graph: the multi-rank iteration as one replay, same edge as eagerly
        os.environ["MPG_GRAPH_COLLECTIVES"] = "1"
        try:
            d = _three_steps(16, 30, use_graphs=True, pg=dist.group.WORLD, n_graphs=1, gen_join="seg_G")
        finally:
            os.environ.pop("MPG_GRAPH_COLLECTIVES", None)
    finally:
        dist.destroy_process_group()
    for r in (b, c, d):
        assert torch.equal(a[0], r[0]) and torch.equal(a[1], r[1]) and a[2:] == r[2:]


def _rank_main(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from mpgan_amd import dist as mdist, train
    from oracle import train_ref as T
    from oracle.train_ref import synthetic_batch
    torch.cuda.set_device(0)
    r, w, pg = mdist.init_from_env("gloo")
    Bg, N = 16, 30
    B = Bg // world
    G, D = train.default_mpgan(N, disc_dropout=0.0)
    if rank == 0:   # only rank 0 holds the weights the run is about; the others get them by broadcast
        G.load_state_dict(T.init_state_dict(T.mpgan_param_shapes(True), 41, torch.float32))
        D.load_state_dict(T.init_state_dict(T.mpgan_param_shapes(False), 42, torch.float32))
    mdist.broadcast_module(G, 0, pg)
    mdist.broadcast_module(D, 0, pg)
    data, labels = synthetic_batch(Bg, N, seed=3)
    gen = torch.Generator().manual_seed(5)
    nD, nG = torch.randn(Bg, N, 32, generator=gen) * 0.2, torch.randn(Bg, N, 32, generator=gen) * 0.2
    sl = slice(rank * B, (rank + 1) * B)
    ts = train.TrainStep(G, D, B, N, lr_disc=train.LR["g"][0], lr_gen=train.LR["g"][1], use_graphs=True,
                         process_group=pg, world_size=world)
    ts.set_batch(data[sl].cuda(), labels[sl].cuda())
    ts.fixed_noise = (nD[sl].cuda(), nG[sl].cuda())
    ts.capture(warmup=0)
    assert len(ts._graphs) == 3
    for _ in range(2):
        ts.step()
    torch.cuda.synchronize()
    out[rank] = (ts.fD.flat.cpu(), ts.fG.flat.cpu(), float(ts.D_loss), float(ts.G_loss))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_global_batch():
    import torch.multiprocessing as mp
    from mpgan_amd import train
    from oracle import train_ref as T
    from oracle.train_ref import synthetic_batch
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_rank_main, args=(2, 29533, out), nprocs=2, join=True)
    (d0, g0, dl0, gl0), (d1, g1, dl1, gl1) = out[0], out[1]
    assert torch.equal(d0, d1) and torch.equal(g0, g1)        # same averaged update on every rank
    # single process, global batch, same noise
    Bg, N = 16, 30
    G, D = train.default_mpgan(N, disc_dropout=0.0)
    G.load_state_dict(T.init_state_dict(T.mpgan_param_shapes(True), 41, torch.float32))
    D.load_state_dict(T.init_state_dict(T.mpgan_param_shapes(False), 42, torch.float32))
    init_D = torch.cat([p.detach().reshape(-1) for p in D.parameters()]).cpu()
    init_G = torch.cat([p.detach().reshape(-1) for p in G.parameters()]).cpu()
    data, labels = synthetic_batch(Bg, N, seed=3)
    gen = torch.Generator().manual_seed(5)
    nD, nG = torch.randn(Bg, N, 32, generator=gen) * 0.2, torch.randn(Bg, N, 32, generator=gen) * 0.2
    ts = train.TrainStep(G, D, Bg, N, lr_disc=train.LR["g"][0], lr_gen=train.LR["g"][1], use_graphs=False)
    ts.set_batch(data.cuda(), labels.cuda())
    ts.fixed_noise = (nD.cuda(), nG.cuda())
    for _ in range(2):
        ts.step()
    torch.cuda.synchronize()
    assert abs(0.5 * (dl0 + dl1) - float(ts.D_loss)) < 1e-5 and abs(0.5 * (gl0 + gl1) - float(ts.G_loss)) < 1e-5
    for ours, ref, init in ((d0, ts.fD.flat.cpu(), init_D), (g0, ts.fG.flat.cpu(), init_G)):
        step = (ref - init).abs().max()
        # summation order differs (per-rank partial sums): updates agree to a small fraction of the step, except where
        # a gradient entry is within rounding of zero (RMSprop's first steps are sign-like)
        off = ((ours - ref).abs() > 2e-2 * step).float().mean()
        assert float(off) < 0.02, float(off)


def test_bench_self_launch_two_ranks_rehearsal():
    """``python bench.py --gpus 2`` without torchrun: the launcher path the driver's scaling run takes (ranks started
    before any GPU call, process group, parameter broadcast, three graph segments with the gradient exchange between
    them, max-over-ranks timing, one JSON line with n_gpus = 2, the roofline leg with collectives off, clean teardown).
    On this one-GPU box both ranks share cuda:0 and exchange over gloo (MPGAN_BENCH_SHARE_GPU)."""
    import json
    import subprocess
    env = dict(os.environ, MPGAN_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--batch", "32"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and "roofline" in d and "cpu_baseline" not in d
    assert np.isfinite(d["losses"]["D"]) and np.isfinite(d["losses"]["G"])

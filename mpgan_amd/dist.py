"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference only offers single-process ``nn.DataParallel`` (setup_training.py:1418-1421).  Jets are
independent (no cross-jet op on the path), so the path shards by batch with exactly one exchange per
backward: a SUM all-reduce of the network's flat gradient buffer (1.4 MB), whose 1/world factor is
folded into the fused RMSprop launch.  The payload is latency-bound on xGMI (a ring step per link,
~16 us of wire time), so one flat bucket per network -- two collectives per iteration -- is the shape
to use, not per-parameter buckets.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl", device: torch.device | None = None):
    """Process group from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, group|None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, **kw)
    return rank, world, dist.group.WORLD


def broadcast_module(module: torch.nn.Module, src: int = 0, group=None):
    """One-time parameter broadcast so every rank starts from rank `src`'s weights."""
    for p in module.parameters():
        dist.broadcast(p.data, src, group=group)


def allreduce_sum_(flat_grad: torch.Tensor, group=None, world: int = 1) -> float:
    """In-place SUM all-reduce of a flat gradient buffer; returns the scale (1/world) the optimiser
    must apply to turn the sum into the global-batch mean gradient."""
    if world > 1 or group is not None:  # a group of one still goes through the collective (and its stream ordering)
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def rank_seed(base: int, rank: int) -> int:
    """Per-rank seed for noise and dropout streams (ranks must not share masks)."""
    return (base + 0x9E3779B1 * rank) & 0x7FFFFFFF

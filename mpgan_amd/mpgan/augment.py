"""Jet augmentations the reference's ``train.py`` reaches as ``mpgan.augment.augment`` (rkansal47/MPGAN
``mpgan/augment.py``; call sites ``train.py:438-441``, ``:508-510``): random 90-degree rotations, flips,
translations and scalings of (eta, phi), each mixed in per jet with probability ``p``.

Data preparation around the hot path (elementwise on [B, N, >=3]), plain torch.  The random draws are made in
the reference's order and shapes, so under the same torch RNG state the results are identical (checked against
the reference where it is importable: tests/test_reference_dropin_cpu.py).  ``args`` is the reference's
namespace: ``device``, ``num_hits``, ``aug_r90 / aug_f / aug_t / aug_s``, ``translate_ratio``,
``translate_pn_ratio``, ``scale_sd``.  Features beyond the first three (a mask column) pass through unchanged.
"""
from __future__ import annotations

import math

import torch


def _xy_rest(X: torch.Tensor):
    return X[..., :2], X[..., 2:]


def rand_mix(args, X1: torch.Tensor, X2: torch.Tensor, p: float) -> torch.Tensor:
    """Per jet: X2 with probability ``p``, else X1 (``p == 1`` returns X1 untouched, as the reference does)."""
    if p == 1:
        return X1
    assert X1.size(0) == X2.size(0), "Error: different batch sizes of rand mix data"
    take = (torch.rand(X1.size(0), 1, 1).to(args.device) < p).to(X1.dtype)
    return X1 * (1 - take) + X2 * take


def rand_flip(args, X: torch.Tensor) -> torch.Tensor:
    """Mirror eta and / or phi of a whole jet (independent signs per jet)."""
    sign = torch.round(torch.rand(X.size(0), 1, 2).to(args.device)) * 2 - 1
    xy, rest = _xy_rest(X)
    return torch.cat((xy * sign, rest), dim=2)


def rand_90_rotation(args, X: torch.Tensor) -> torch.Tensor:
    """Rotate (eta, phi) of a whole jet by a random multiple of 90 degrees."""
    angle = torch.floor(torch.rand(X.size(0), 1, 1).to(args.device) * 4) * (math.pi / 2)
    s, c = torch.sin(angle), torch.cos(angle)
    xy, rest = _xy_rest(X)
    x, y = xy[..., 0:1], xy[..., 1:2]
    return torch.cat((c * x - s * y, s * x + c * y, rest), dim=2)


def rand_translate(args, X: torch.Tensor) -> torch.Tensor:
    """Shift (eta, phi) of a whole jet by a uniform offset in +-translate_ratio / 2."""
    shift = (torch.rand(X.size(0), 1, 2).to(args.device) - 0.5) * args.translate_ratio
    xy, rest = _xy_rest(X)
    return torch.cat((xy + shift, rest), dim=2)


def rand_translate_per_node(args, X: torch.Tensor) -> torch.Tensor:
    """Shift (eta, phi) of every particle independently by a uniform offset in +-translate_pn_ratio / 2."""
    shift = (torch.rand(X.size(0), args.num_hits, 2).to(args.device) - 0.5) * args.translate_pn_ratio
    xy, rest = _xy_rest(X)
    return torch.cat((xy + shift, rest), dim=2)


def rand_scale(args, X: torch.Tensor) -> torch.Tensor:
    """Scale (eta, phi) of a whole jet by one log-normal factor (sigma = scale_sd)."""
    law = torch.distributions.log_normal.LogNormal(torch.tensor([0.0]).to(args.device),
                                                   torch.tensor([args.scale_sd]).to(args.device))
    factor = law.sample((X.size(0), 1))  # [B, 1, 1]
    xy, rest = _xy_rest(X)
    return torch.cat((xy * factor, rest), dim=2)


def augment(args, X: torch.Tensor, p: float) -> torch.Tensor:
    """Apply the enabled augmentations in the reference's order: rotation, flip, translation, scaling."""
    for flag, fn in (("aug_r90", rand_90_rotation), ("aug_f", rand_flip), ("aug_t", rand_translate),
                     ("aug_s", rand_scale)):
        if getattr(args, flag):
            X = rand_mix(args, X, fn(args, X), p)
    return X

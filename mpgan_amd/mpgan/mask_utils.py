"""``mask_manual`` -- the pT-cutoff mask column the reference's ``train.py`` imports from the ``mpgan`` package
(rkansal47/MPGAN ``mpgan/mask_utils.py:5-24``; used by ``train.gen`` at ``train.py:208-210``).

Data preparation around the hot path, elementwise on a [B, N, F] tensor: plain torch on whatever device the data
lives on.  ``args`` is the reference's namespace (``mask_real_only``, ``mask_exp``, ``device``).
"""
from __future__ import annotations

import logging

import torch

__all__ = ["mask_manual"]


def mask_manual(args, gen_data: torch.Tensor, pt_cutoff: float) -> torch.Tensor:
    """Append a mask feature (shifted by -0.5 like every mask column of the data set) to ``gen_data``:
    1 where the particle's relative pT (feature 2) exceeds ``pt_cutoff``; with ``mask_exp`` the part below the cut
    decays as exp((pT - cut) / |cut|); with ``mask_real_only`` every generated particle counts as real."""
    logging.debug("Before Mask: ")
    logging.debug(gen_data[0])
    pt = gen_data[:, :, 2:3]
    if args.mask_real_only:
        mask = torch.full_like(pt, 0.5)
    elif args.mask_exp:
        above = (pt > pt_cutoff).to(pt.dtype)
        mask = above + (1 - above) * torch.exp((pt - pt_cutoff) / abs(pt_cutoff)) - 0.5
    else:
        mask = (pt > pt_cutoff).to(pt.dtype) - 0.5
    out = torch.cat((gen_data, mask.to(gen_data.device)), dim=2)
    logging.debug("After Mask: ")
    logging.debug(out[0])
    return out

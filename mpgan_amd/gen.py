"""Forward-only generation on the fused path -- the reference's ``gen`` / ``gen_multi_batch``
(train.py:100-282) and the un-normalising epilogue of its ``gen.py`` (:85-145).

Same function names and argument meaning as the reference, so ``train.py``'s evaluation loop and ``gen.py`` can call
these instead of their own.  Differences, all results-neutral: generation runs under ``torch.no_grad()`` whenever the
caller asked for detached output (the reference builds and then drops the autograd graph, train.py:253-282; without it
the fused MPLayer skips its sign words and saves nothing for a backward), and the default chunk is sized for the
device (thousands of jets per launch keep all 256 CUs busy; the reference's default of 16 leaves 240 idle).
"""
from __future__ import annotations

import logging
from typing import Optional

import torch
from torch import Tensor

from .data import unnormalise_jets
from .mpgan.mask_utils import mask_manual


def get_gen_noise(model_args: dict, num_samples: int, num_particles: int, model: str = "mpgan", device=None,
                  noise_std: float = 0.2):
    """Generator input noise ~ N(0, noise_std) in the shape the model family takes (train.py:100-140):
    mpgan ``[n, N (+1 with mask_learn_sep), latent_node_size]`` (``[n, lfc_latent_size]`` with ``lfc``),
    gapt ``[n, N, embed_dim]``.  Returns ``(noise, None)`` like the reference (second slot: PCGAN point noise)."""
    if device is None:
        device = "cuda"
    if model in ("mpgan", "old_mpgan"):
        if model_args.get("lfc"):
            shape = (num_samples, model_args["lfc_latent_size"])
        else:
            extra = int(bool(model_args.get("mask_learn_sep")))
            shape = (num_samples, num_particles + extra, model_args["latent_node_size"])
    elif model == "gapt":
        shape = (num_samples, num_particles, model_args["embed_dim"])
    else:
        raise NotImplementedError(f"mpgan_amd generates for the mpgan and gapt model families only (got {model!r})")
    return torch.empty(shape, device=device).normal_(0.0, noise_std), None


def gen(model_args: dict, G: torch.nn.Module, num_samples: int, num_particles: int, model: str = "mpgan",
        noise: Tensor = None, labels: Tensor = None, noise_std: float = 0.2, **extra_args) -> Tensor:
    """``num_samples`` jets in one go (train.py:143-215): ``G(noise, labels)``, then the optional manual pT mask."""
    device = next(G.parameters()).device
    if labels is not None:
        assert labels.shape[0] == num_samples, "number of labels doesn't match num_samples"
        labels = labels.to(device)
    if noise is None:
        noise, _ = get_gen_noise(model_args, num_samples, num_particles, model, device, noise_std)
    gen_data = G(noise, labels)
    if extra_args.get("mask_manual"):
        gen_data = mask_manual(model_args, gen_data, extra_args["pt_cutoff"])
    logging.debug(gen_data[0, :10])
    return gen_data


def gen_multi_batch(model_args: dict, G: torch.nn.Module, batch_size: int, num_samples: int, num_particles: int,
                    out_device: str = "cpu", detach: bool = False, use_tqdm: bool = True, model: str = "mpgan",
                    noise: Tensor = None, labels: Tensor = None, noise_std: float = 0.2, **extra_args) -> Tensor:
    """``num_samples`` jets in chunks of ``batch_size`` (train.py:226-282), gathered on ``out_device``.
    ``use_tqdm`` is accepted for signature compatibility (no progress bar is drawn)."""
    assert out_device == "cuda" or out_device == "cpu", "Invalid device type"
    if labels is not None:
        assert labels.shape[0] == num_samples, "number of labels doesn't match num_samples"
        labels = torch.as_tensor(labels, dtype=torch.float32)
    chunks = []
    with torch.set_grad_enabled(torch.is_grad_enabled() and not detach):
        for start in range(0, num_samples, batch_size):
            n = min(batch_size, num_samples - start)
            out = gen(model_args, G, num_samples=n, num_particles=num_particles, model=model, noise=noise,
                      labels=None if labels is None else labels[start:start + n], noise_std=noise_std, **extra_args)
            if detach:
                out = out.detach()
            chunks.append(out.to(out_device))
    return torch.cat(chunks, dim=0) if chunks else torch.empty(0)


def generate_jets(G: torch.nn.Module, num_samples: int, num_particles: int = 30, labels: Optional[Tensor] = None,
                  jet_type: str = "g", model: str = "mpgan", model_args: Optional[dict] = None, mask: bool = True,
                  batch_size: int = 4096, noise_std: float = 0.2) -> Tensor:
    """What the reference's ``gen.py`` writes to its output file: ``[num_samples, N, 3]`` un-normalised
    (eta_rel, phi_rel, pT_rel), masked particles zeroed (gen.py:111-141).  ``labels`` = num_particles / N per jet
    (gen.py samples them from the data set's jet features; the caller supplies them here)."""
    if model_args is None:
        model_args = ({"lfc": False, "latent_node_size": getattr(G, "input_node_size", 32)} if model == "mpgan"
                      else {"embed_dim": getattr(G, "embed_dim", 64)})
    was_training = G.training
    G.eval()
    try:
        jets = gen_multi_batch(model_args, G, batch_size, num_samples, num_particles, out_device="cuda", detach=True,
                               use_tqdm=False, model=model, labels=labels, noise_std=noise_std)
    finally:
        G.train(was_training)
    return unnormalise_jets(jets, jet_type, mask=mask)

"""On-disk formats of a reference training run, so a run can be resumed by either code base.

* ``save_models``  -- ``<models_path>/{D,G}_<epoch>.pt`` (module state dicts) and ``{D,G}_optim_<epoch>.pt``
  (``torch.optim`` state dicts), train.py:526-537;
* ``load_models`` / ``load_optimizers`` -- their readers, setup_training.py:1406-1416 and :1525-1535;
* ``latest_epoch`` -- the newest epoch for which both networks were saved, setup_training.py:1140-1156;
* ``save_losses`` / ``load_losses`` -- one ``<key>.txt`` per loss or metric (``np.savetxt`` / ``np.loadtxt``),
  train.py:538-540 and setup_training.py:1540-1598.

"Optimizers" here are anything with ``state_dict()`` / ``load_state_dict()`` in ``torch.optim`` layout: a real
``torch.optim`` object or ``mpgan_amd.train.FlatParams``.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch


def _plain(module: torch.nn.Module) -> torch.nn.Module:
    return module.module if hasattr(module, "module") and isinstance(module.module, torch.nn.Module) else module


def save_models(D, G, D_optimizer, G_optimizer, models_path: str, epoch: int, multi_gpu: bool = False):
    """train.py:526-537.  ``multi_gpu`` (a DataParallel / DDP wrapper) is also detected from ``.module``."""
    os.makedirs(models_path, exist_ok=True)
    for tag, net, opt in (("D", D, D_optimizer), ("G", G, G_optimizer)):
        torch.save(_plain(net).state_dict(), os.path.join(models_path, f"{tag}_{epoch}.pt"))
        torch.save(opt.state_dict(), os.path.join(models_path, f"{tag}_optim_{epoch}.pt"))


def latest_epoch(models_path: str) -> int:
    """Newest epoch with both ``D_<e>.pt`` and ``G_<e>.pt`` present; 0 when there is none
    (setup_training.py:1140-1156: ``--start-epoch -1``)."""
    found = {"D": set(), "G": set()}
    if os.path.isdir(models_path):
        for f in os.listdir(models_path):
            stem, ext = os.path.splitext(f)
            parts = stem.split("_")
            if ext == ".pt" and len(parts) == 2 and parts[0] in found and parts[1].isdigit():
                found[parts[0]].add(int(parts[1]))
    both = found["D"] & found["G"]
    return max(both) if both else 0


def load_models(D, G, models_path: str, epoch: int, map_location=None):
    """setup_training.py:1406-1416 (state-dict files; a pickled whole module is taken over as the reference does)."""
    out = []
    for tag, net in (("D", D), ("G", G)):
        obj = torch.load(os.path.join(models_path, f"{tag}_{epoch}.pt"), map_location=map_location, weights_only=False)
        if isinstance(obj, torch.nn.Module):
            net = obj
        else:
            _plain(net).load_state_dict(obj)
        out.append(net)
    return out[0], out[1]


def load_optimizers(D_optimizer, G_optimizer, models_path: str, epoch: int, map_location=None):
    """setup_training.py:1525-1535."""
    for tag, opt in (("D", D_optimizer), ("G", G_optimizer)):
        opt.load_state_dict(torch.load(os.path.join(models_path, f"{tag}_optim_{epoch}.pt"), map_location=map_location,
                                       weights_only=False))


# metrics holding several values per evaluation.  The reference's list (setup_training.py:1551) leaves out "fpd",
# whose entries are (value, error) pairs as well (train.py:633): a history with a single evaluation then comes
# back as two scalar entries there; here it keeps its shape.
MULTI_VALUE_KEYS = ("w1p", "w1m", "w1efp", "fpd")


def loss_keys(gp: bool = False, fpnd: bool = False, fpd: bool = True, efp: bool = False) -> Tuple[List[str], List[str]]:
    """(all keys, evaluation keys) of the ``losses`` dict for a flag combination (setup_training.py:1540-1562)."""
    keys = ["D", "Dr", "Df", "G"] + (["gp"] if gp else [])
    eval_keys = ["w1p", "w1m"] + (["w1efp"] if efp else []) + (["fpnd"] if fpnd else []) + (["fpd"] if fpd else [])
    return keys + eval_keys, eval_keys


def save_losses(losses: Dict[str, Iterable], losses_path: str):
    """train.py:538-540."""
    os.makedirs(losses_path, exist_ok=True)
    for key in losses:
        np.savetxt(f"{losses_path}/{key}.txt", losses[key])


def load_losses(losses_path: str, keys: Iterable[str], eval_keys: Iterable[str] = (), start_epoch: int = 0,
                save_epochs: int = 1) -> Dict[str, list]:
    """The resume branch of ``setup_training.losses`` (:1564-1584): read ``<key>.txt``, restore the list shape,
    and cut the history at ``start_epoch`` (evaluation metrics are stored every ``save_epochs`` epochs)."""
    eval_keys = set(eval_keys)
    out = {}
    for key in keys:
        try:
            arr = np.loadtxt(f"{losses_path}/{key}.txt")
        except OSError:
            out[key] = []
            continue
        multi = key in MULTI_VALUE_KEYS
        if (arr.ndim == 1 and multi) or (arr.ndim == 0 and not multi):
            arr = np.expand_dims(arr, 0)
        hist = arr.tolist()
        out[key] = hist[: int(start_epoch / save_epochs) + 1] if key in eval_keys else hist[: start_epoch + 1]
    return out

"""Jet data in the layout the hot path consumes, without the JetNet package.

The reference feeds ``train_D`` batches ``data [B, N, 4] = (eta_rel, phi_rel, pT_rel, mask)`` normalised by
``FeaturewiseLinearBounded(feature_norms=1, feature_shifts=[0, 0, -0.5, -0.5], feature_maxes=fpnd maxes + [1])``
and ``labels [B, 1] = num_particles / N`` (``train.py:36-67``); ``gen.py:127-139`` undoes that normalisation on
generated jets.  This module holds

* ``synthetic_jets``   -- the synthetic stand-in for JetNet used by bench.py and the tests,
* ``normalise_jets`` / ``unnormalise_jets`` -- the two directions of the reference's feature normalisation,
* ``JetArrayDataset``  -- a ``torch.utils.data.Dataset`` over raw ``[n, N, 4]`` particle arrays (``.npy`` / ``.npz``
  exports of JetNet's ``particle_features``), yielding ``(data, labels)`` exactly as ``train.py:843-846`` unpacks them.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

# maxima of (eta_rel, phi_rel, pT_rel, mask) per jet type: gen.py:10-14 (= JetNet.fpnd_norm.feature_maxes + [1])
FEATURE_MAXES = {
    "g": [1.4532885551452637, 0.520724892616272, 0.8537549376487732, 1.0],
    "q": [1.6211985349655151, 0.4568111002445221, 0.8896132111549377, 1.0],
    "t": [1.4242753982543945, 0.4949831962585449, 0.8774275183677673, 1.0],
}
FEATURE_NORMS = [1.0, 1.0, 1.0, 1.0]       # gen.py:16
FEATURE_SHIFTS = [0.0, 0.0, -0.5, -0.5]    # gen.py:17, train.py:43


def synthetic_jets(B: int, N: int, seed: int = 4, dist: str = "gluon", dtype=torch.float32):
    """``(data [B,N,4], labels [B,1])`` of synthetic JetNet-like jets, already normalised.

    Multiplicity n per jet: ``gluon`` = clip(round(Normal(0.8 N, 0.15 N)), 1, N), ``uniform`` = UniformInt[1, N].
    The first n particles are real: eta_rel, phi_rel ~ clip(Normal(0, 0.15), -1, 1), pT_rel ~ Uniform(-0.5, 0.5),
    mask = +0.5; padding particles are (0, 0, -0.5, -0.5).  ``labels = float32(n) * float32(1/N)`` -- the product
    with the reciprocal makes ``int(labels * N)`` return n for every n <= 150, n / N does not."""
    rs = np.random.RandomState(seed)
    if dist == "uniform":
        n = rs.randint(1, N + 1, size=B)
    elif dist == "gluon":
        n = np.clip(np.rint(rs.normal(0.8 * N, 0.15 * N, size=B)), 1, N).astype(np.int64)
    else:
        raise ValueError(f"unknown multiplicity law {dist!r}")
    eta = np.clip(rs.normal(0, 0.15, size=(B, N)), -1, 1)
    phi = np.clip(rs.normal(0, 0.15, size=(B, N)), -1, 1)
    pt = rs.uniform(-0.5, 0.5, size=(B, N))
    real = np.arange(N)[None, :] < n[:, None]
    pad = (0.0, 0.0, -0.5, -0.5)
    data = np.stack([np.where(real, f, v) for f, v in zip((eta, phi, pt, np.full((B, N), 0.5)), pad)], axis=2)
    labels = (n.astype(np.float32) * np.float32(1.0 / N)).reshape(B, 1)
    return torch.from_numpy(data).to(dtype), torch.from_numpy(labels).to(dtype)


def normalise_jets(raw: torch.Tensor, jet_type: str = "g") -> torch.Tensor:
    """Raw ``(eta_rel, phi_rel, pT_rel[, mask])`` -> the network's input range: x / max * norm + shift per feature
    (``FeaturewiseLinearBounded`` as configured at ``train.py:41-45``)."""
    F = raw.shape[-1]
    mx = raw.new_tensor(FEATURE_MAXES[jet_type][:F])
    return raw / mx * raw.new_tensor(FEATURE_NORMS[:F]) + raw.new_tensor(FEATURE_SHIFTS[:F])


def unnormalise_jets(gen_jets: torch.Tensor, jet_type: str = "g", mask: bool = True) -> torch.Tensor:
    """The epilogue of the reference's ``gen.py:127-141`` on generator output ``[n, N, 3 (+ mask)]``: undo shift /
    norm / max on the three particle features, zero the particles whose mask feature is below 0.5 (the generator
    emits mask - 0.5, so that is "mask bit clear" exactly as gen.py tests it), clamp pT_rel at 0, drop the mask
    column.  Returns a new ``[n, N, 3]`` tensor."""
    out = gen_jets[..., :3].clone()
    for i in range(3):
        if FEATURE_SHIFTS[i]:
            out[..., i] -= FEATURE_SHIFTS[i]
        out[..., i] /= FEATURE_NORMS[i]
        out[..., i] *= FEATURE_MAXES[jet_type][i]
    if mask:
        out[~(gen_jets[..., -1] >= 0.5)] = 0
    out[..., 2].clamp_(min=0)
    return out


class JetArrayDataset(torch.utils.data.Dataset):
    """Raw JetNet-style particle arrays ``[n, N, 4] = (eta_rel, phi_rel, pT_rel, mask in {0, 1})`` served the way
    ``train.py``'s ``DataLoader`` serves JetNet: item = ``(normalised particles [N, 4], label [1])`` with
    label = num_particles / N (``jet_normalisation``: ``FeaturewiseLinear(feature_scales=1 / num_hits)``)."""

    def __init__(self, particles, jet_type: str = "g", num_particles: Optional[int] = None,
                 split: str = "train", split_fraction: Sequence[float] = (0.7, 0.3, 0.0)):
        if isinstance(particles, str):
            arr = np.load(particles)
            particles = arr[arr.files[0]] if hasattr(arr, "files") else arr
        p = torch.as_tensor(np.asarray(particles), dtype=torch.float32)
        if p.dim() != 3 or p.shape[-1] != 4:
            raise ValueError(f"expected particle features [n, N, 4], got {tuple(p.shape)}")
        if num_particles is not None:
            p = p[:, :num_particles]
        n = p.shape[0]
        cuts = np.cumsum([0] + [int(round(f * n)) for f in split_fraction])
        k = {"train": 0, "valid": 1, "test": 2, "all": None}[split]
        if k is not None:
            p = p[cuts[k]:min(cuts[k + 1], n)]
        self.num_particles = p.shape[1]
        mask = p[..., 3]
        self.jet_features = (mask.sum(1, keepdim=True) * np.float32(1.0 / self.num_particles)).float()
        self.particle_data = normalise_jets(p, jet_type)

    def __len__(self):
        return self.particle_data.shape[0]

    def __getitem__(self, i):
        return self.particle_data[i], self.jet_features[i]

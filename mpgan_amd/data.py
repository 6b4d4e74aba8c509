"""Jet data in the layout the hot path consumes, without the JetNet package.

The reference feeds ``train_D`` batches ``data [B, N, 4] = (eta_rel, phi_rel, pT_rel, mask)`` normalised by
``FeaturewiseLinearBounded(feature_norms=1, feature_shifts=[0, 0, -0.5, -0.5], feature_maxes=fpnd maxes + [1])``
and ``labels [B, 1] = num_particles / N`` (``train.py:36-67``); ``gen.py:127-139`` undoes that normalisation on
generated jets.  This module holds

* ``synthetic_jets``   -- the synthetic stand-in for JetNet used by bench.py and the tests,
* ``normalise_jets`` / ``unnormalise_jets`` -- the two directions of the reference's feature normalisation,
* ``JetArrayDataset``  -- a ``torch.utils.data.Dataset`` over raw ``[n, N, 4]`` particle arrays, yielding
  ``(data, labels)`` exactly as ``train.py:843-846`` unpacks them,
* ``read_jetnet_file`` / ``JetArrayDataset.from_jetnet_file`` -- the reader of JetNet's on-disk layout
  (``<data_dir>/<jet_type>[150].hdf5`` with the datasets ``particle_features [n, N, 4]`` = (etarel, phirel, ptrel, mask)
  and ``jet_features [n, 4]`` = (pt, eta, mass, num_particles), jetnet >= 0.2.1 as ``requirements.txt:2`` pins it; the
  same two arrays in an ``.npz`` are read alike).
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

    Multiplicity n per jet: ``gluon`` = clip(round(Normal(0.8 N, 0.15 N)), 1, N), ``uniform`` = UniformInt[1, N]; ``top`` =
    clip(round(Normal(0.95 N, 0.08 N)), 1, N) and ``quark`` = clip(round(Normal(0.7 N, 0.18 N)), 1, N) -- stand-ins for the
    other two jet types of the reference's ``--jets`` (top jets have more constituents than gluon jets, so nearly all fill the
    N leading-pT slots; light-quark jets fewer); the dataset itself is not available here, the laws only shape the work.
    The first n particles are real: eta_rel, phi_rel ~ clip(Normal(0, 0.15), -1, 1), pT_rel ~ Uniform(-0.5, 0.5),
    mask = +0.5; padding particles are (0, 0, -0.5, -0.5).  ``labels = float32(n) * float32(1/N)`` -- the product
    with the reciprocal makes ``int(labels * N)`` return n for every n <= 150, n / N does not."""
    rs = np.random.RandomState(seed)
    if dist == "uniform":
        n = rs.randint(1, N + 1, size=B)
    elif dist == "gluon":
        n = np.clip(np.rint(rs.normal(0.8 * N, 0.15 * N, size=B)), 1, N).astype(np.int64)
    elif dist == "top":
        n = np.clip(np.rint(rs.normal(0.95 * N, 0.08 * N, size=B)), 1, N).astype(np.int64)
    elif dist == "quark":
        n = np.clip(np.rint(rs.normal(0.7 * N, 0.18 * N, size=B)), 1, N).astype(np.int64)
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


JETNET_PARTICLE_FEATURES = ("etarel", "phirel", "ptrel", "mask")      # JetNet.all_particle_features
JETNET_JET_FEATURES = ("pt", "eta", "mass", "num_particles")           # JetNet.all_jet_features


def read_jetnet_file(path: str):
    """``(particle_features [n, N, 4] float32, jet_features [n, 4] float32 or None)`` of a JetNet data file.

    ``.hdf5`` / ``.h5``: the datasets ``particle_features`` and ``jet_features`` of jetnet's own files (needs ``h5py``, which
    is what jetnet itself reads them with; a clear error says so when it is missing).  ``.npz``: the same two keys
    (``jet_features`` optional).  ``.npy``: the particle array alone."""
    import os
    ext = os.path.splitext(path)[1].lower()
    if ext in (".hdf5", ".h5"):
        try:
            import h5py
        except ImportError as e:  # pragma: no cover - depends on the environment
            raise ImportError(f"reading {path} needs h5py (JetNet's files are HDF5); export the two datasets "
                              "'particle_features' and 'jet_features' to an .npz to read them without it") from e
        with h5py.File(path, "r") as f:
            pf = np.asarray(f["particle_features"], dtype=np.float32)
            jf = np.asarray(f["jet_features"], dtype=np.float32) if "jet_features" in f else None
    elif ext == ".npz":
        with np.load(path) as f:
            if "particle_features" not in f.files:
                raise KeyError(f"{path}: no 'particle_features' array (found {f.files})")
            pf = np.asarray(f["particle_features"], dtype=np.float32)
            jf = np.asarray(f["jet_features"], dtype=np.float32) if "jet_features" in f.files else None
    elif ext == ".npy":
        pf, jf = np.asarray(np.load(path), dtype=np.float32), None
    else:
        raise ValueError(f"{path}: expected .hdf5 / .h5 / .npz / .npy")
    if pf.ndim != 3 or pf.shape[-1] != len(JETNET_PARTICLE_FEATURES):
        raise ValueError(f"{path}: particle_features has shape {pf.shape}, expected [n, N, 4] = {JETNET_PARTICLE_FEATURES}")
    if jf is not None and (jf.ndim != 2 or jf.shape[0] != pf.shape[0] or jf.shape[1] != len(JETNET_JET_FEATURES)):
        raise ValueError(f"{path}: jet_features has shape {jf.shape}, expected [{pf.shape[0]}, 4] = {JETNET_JET_FEATURES}")
    return pf, jf


class JetArrayDataset(torch.utils.data.Dataset):
    """Raw JetNet-style particle arrays ``[n, N, 4] = (eta_rel, phi_rel, pT_rel, mask in {0, 1})`` served the way
    ``train.py``'s ``DataLoader`` serves JetNet: item = ``(normalised particles [N, 4], label [1])`` with
    label = num_particles / N (``jet_normalisation``: ``FeaturewiseLinear(feature_scales=1 / num_hits)``)."""

    def __init__(self, particles, jet_type: str = "g", num_particles: Optional[int] = None,
                 split: str = "train", split_fraction: Sequence[float] = (0.7, 0.3, 0.0), multiplicities=None):
        """``particles``: an array ``[n, N, 4]`` or a file ``read_jetnet_file`` understands.  ``multiplicities`` ``[n]``: the
        ``num_particles`` jet feature when it comes from a file; by default (and always when the particle axis is cut to
        ``num_particles``) it is counted from the mask column, which is what JetNet's own column holds."""
        if isinstance(particles, str):
            particles, jf = read_jetnet_file(particles)
            if multiplicities is None and jf is not None:
                multiplicities = jf[:, JETNET_JET_FEATURES.index("num_particles")]
        p = torch.as_tensor(np.asarray(particles), dtype=torch.float32)
        if p.dim() != 3 or p.shape[-1] != 4:
            raise ValueError(f"expected particle features [n, N, 4], got {tuple(p.shape)}")
        if num_particles is not None and num_particles < p.shape[1]:
            p = p[:, :num_particles]
            multiplicities = None        # the first num_particles (pT-ordered) particles: count again
        n = p.shape[0]
        cuts = np.cumsum([0] + [int(round(f * n)) for f in split_fraction])
        k = {"train": 0, "valid": 1, "test": 2, "all": None}[split]
        lo, hi = (0, n) if k is None else (int(cuts[k]), int(min(cuts[k + 1], n)))
        p = p[lo:hi]
        self.num_particles = p.shape[1]
        if multiplicities is None:
            mult = p[..., 3].sum(1, keepdim=True)
        else:
            mult = torch.as_tensor(np.asarray(multiplicities), dtype=torch.float32).reshape(-1, 1)[lo:hi]
        # jet_normalisation = FeaturewiseLinear(feature_scales = 1 / num_hits) (train.py:46): the PRODUCT with the reciprocal
        self.jet_features = (mult * np.float32(1.0 / self.num_particles)).float()
        self.particle_data = normalise_jets(p, jet_type)

    @classmethod
    def from_jetnet_file(cls, data_dir: str, jet_type: str = "g", num_particles: int = 30, split: str = "train",
                         split_fraction: Sequence[float] = (0.7, 0.3, 0.0)):
        """The dataset ``JetNet(jet_type, data_dir, num_particles, particle_features = all, jet_features = "num_particles",
        particle_normalisation, jet_normalisation, split_fraction, split)`` of ``train.py:47-64`` from jetnet's file in
        ``data_dir``: ``<jet_type>.hdf5`` (30 particles) or ``<jet_type>150.hdf5`` (more), else the same stem as ``.npz``."""
        import os
        stem = jet_type + ("150" if num_particles > 30 else "")
        for ext in (".hdf5", ".h5", ".npz"):
            path = os.path.join(data_dir, stem + ext)
            if os.path.isfile(path):
                return cls(path, jet_type=jet_type, num_particles=num_particles, split=split, split_fraction=split_fraction)
        raise FileNotFoundError(f"no {stem}.hdf5 / .h5 / .npz in {data_dir}")

    def __len__(self):
        return self.particle_data.shape[0]

    def __getitem__(self, i):
        return self.particle_data[i], self.jet_features[i]

"""CPU restatement of one G+D training iteration and of the synthetic workload (TEST ORACLE).

Follows (reference paths relative to /root/reference):
  * train_D                    train.py:398-462   (D.train, G.eval, D(real), G(noise), D(fake),
                                                   LSGAN  mse(real,1)+mse(fake,0)  :368-370,378)
  * train_G                    train.py:479-523   (G.train, D stays in train mode, mse(D(G(z)),1) :471-472)
  * get_gen_noise              train.py:100-141   (Normal(0, sd=0.2) of [B,N,latent])
  * optimizers                 setup_training.py:1511-1513  (torch.optim.RMSprop defaults)
  * data layout                train.py:41-67, gen.py:10-17  (eta_rel, phi_rel, pt_rel, mask-0.5)

``train.py`` itself cannot be imported here (needs the ``jetnet`` package), so the step is
restated from its source text; its pieces (G, D forward) are pinned by the goldens.
In train_D the reference back-propagates through G as well and throws those gradients
away at the next ``zero_grad`` (train.py:420,495): the fake batch is detached here, which
changes no result.
"""

from __future__ import annotations

import zlib
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import mpgan_ref as M
from . import gapt_ref as A

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- shapes / init
def mpgan_param_shapes(gen: bool, latent=32, hidden=32, feat=3, fe=(96, 160, 192), fn=(256, 256)):
    """State-dict manifest of the default MPGenerator / MPDiscriminator."""
    shapes = {}
    ins = [latent, hidden] if gen else [feat, hidden]
    outs = [hidden, feat] if gen else [hidden, hidden]
    for l in range(2):
        dims = [2 * ins[l]] + list(fe)
        for k in range(len(fe)):
            shapes[f"mp_layers.{l}.fe.net.{k}.weight"] = (dims[k + 1], dims[k])
            shapes[f"mp_layers.{l}.fe.net.{k}.bias"] = (dims[k + 1],)
        dims = [fe[-1] + ins[l]] + list(fn) + [outs[l]]
        for k in range(len(fn) + 1):
            shapes[f"mp_layers.{l}.fn.net.{k}.weight"] = (dims[k + 1], dims[k])
            shapes[f"mp_layers.{l}.fn.net.{k}.bias"] = (dims[k + 1],)
    if not gen:
        shapes["fnd_layer.net.0.weight"] = (1, hidden)
        shapes["fnd_layer.net.0.bias"] = (1,)
    return shapes


def _mab_shapes(prefix, E):
    return {
        f"{prefix}.attention.in_proj_weight": (3 * E, E),
        f"{prefix}.attention.in_proj_bias": (3 * E,),
        f"{prefix}.attention.out_proj.weight": (E, E),
        f"{prefix}.attention.out_proj.bias": (E,),
        f"{prefix}.ff.net.0.weight": (E, E),
        f"{prefix}.ff.net.0.bias": (E,),
    }


def gapt_param_shapes(gen: bool, E=64, feat=3, sab_layers=None):
    """State-dict manifest of the default GAPT_G / GAPT_D."""
    shapes = {}
    if sab_layers is None:
        sab_layers = 4 if gen else 2
    if not gen:
        shapes["input_embedding.net.0.weight"] = (E, feat)
        shapes["input_embedding.net.0.bias"] = (E,)
    for s in range(sab_layers):
        shapes.update(_mab_shapes(f"sabs.{s}.mab", E))
    if not gen:
        shapes["pma.S"] = (1, 1, E)
        shapes.update(_mab_shapes("pma.mab", E))
    shapes["final_fc.net.0.weight"] = (feat if gen else 1, E)
    shapes["final_fc.net.0.bias"] = (feat if gen else 1,)
    return shapes


def init_state_dict(shapes: Dict[str, tuple], seed: int = 0, dtype=torch.float32, scale: float = 1.0):
    """Deterministic, torch-version-independent parameter values (numpy MT19937 keyed on the
    parameter name): U(-b, b), b = scale/sqrt(fan_in) for matrices (and for their biases),
    matching the distribution family of nn.Linear's default init.  Used so that goldens need
    to carry inputs/outputs only -- the generator script loads these values into the
    reference modules, the tests load them into the oracle and into the HIP modules."""
    sd = {}
    fan_in = {}
    for name, shp in shapes.items():
        if len(shp) >= 2:
            fan_in[name.rsplit(".", 1)[0] if name.endswith(".weight") else name] = shp[-1]
    for name, shp in shapes.items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) % (2**31))
        base = name.rsplit(".", 1)[0]
        if name.endswith("in_proj_bias"):
            fi = shapes[name.replace("in_proj_bias", "in_proj_weight")][-1]
        else:
            fi = fan_in.get(base, shp[-1])
        b = scale / np.sqrt(fi)
        sd[name] = torch.from_numpy(rs.uniform(-b, b, size=shp)).to(dtype)
    return sd


# ----------------------------------------------------------------------------- synthetic data
def synthetic_batch(B: int, N: int, seed: int = 4, dist: str = "gluon", dtype=torch.float32):
    """Synthetic JetNet-like batch.  data [B,N,4] = (eta_rel, phi_rel,
    pt_rel, mask-0.5); labels [B,1] = float32(n) * float32(1/N) (multiply by reciprocal so that
    int(labels*N) round-trips for every n <= 150)."""
    rs = np.random.RandomState(seed)
    if dist == "uniform":
        n = rs.randint(1, N + 1, size=B)
    else:  # gluon-like: most jets are close to full
        n = np.clip(np.rint(rs.normal(0.8 * N, 0.15 * N, size=B)), 1, N).astype(np.int64)
    eta = np.clip(rs.normal(0, 0.15, size=(B, N)), -1, 1)
    phi = np.clip(rs.normal(0, 0.15, size=(B, N)), -1, 1)
    pt = rs.uniform(-0.5, 0.5, size=(B, N))
    real = (np.arange(N)[None, :] < n[:, None])
    data = np.stack(
        [np.where(real, eta, 0.0), np.where(real, phi, 0.0), np.where(real, pt, -0.5),
         np.where(real, 0.5, -0.5)], axis=2)
    labels = (n.astype(np.float32) * np.float32(1.0 / N)).reshape(B, 1)
    return torch.from_numpy(data).to(dtype), torch.from_numpy(labels).to(dtype)


# ----------------------------------------------------------------------------- optimiser
def rmsprop_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: Dict[str, Tensor],
                 lr: float, alpha: float = 0.99, eps: float = 1e-8):
    """torch.optim.RMSprop defaults (momentum 0, not centered, no weight decay):
    v = alpha v + (1-alpha) g^2 ;  p -= lr * g / (sqrt(v) + eps)."""
    for k, p in params.items():
        g = grads[k]
        v = state.setdefault(k, torch.zeros_like(p))
        v.mul_(alpha).addcmul_(g, g, value=1 - alpha)
        p.data.addcdiv_(g, v.sqrt().add_(eps), value=-lr)


# ----------------------------------------------------------------------------- one iteration
def _fwd_G(model, sdG, noise, labels, N, cfg, signs=None):
    if model == "mpgan":
        return M.mpgen_forward(sdG, noise, labels, num_particles=N, signs=signs, **cfg.get("G", {}))
    return A.gapt_g_forward(sdG, noise, labels, num_particles=N, **cfg.get("G", {}))


def _fwd_D(model, sdD, x, labels, p, keeps, cfg, signs=None):
    if model == "mpgan":
        return M.mpdisc_forward(sdD, x, labels, p=p, keeps=keeps,
                                keep_fnd=None if keeps is None else keeps.get("fnd"), signs=signs,
                                **cfg.get("D", {}))
    return A.gapt_d_forward(sdD, x, labels, p=p, keeps=keeps, **cfg.get("D", {}))


def d_loss_ref(loss: str, out_r: Tensor, out_f: Tensor) -> Tensor:
    """calc_D_loss (train.py:352-379), no label smoothing / noise / gradient penalty."""
    if loss == "ls":
        return ((out_r - 1.0) ** 2).mean() + (out_f**2).mean()
    if loss == "og":  # nn.BCELoss against ones / zeros
        bce = torch.nn.BCELoss()
        return bce(out_r, torch.ones_like(out_r)) + bce(out_f, torch.zeros_like(out_f))
    if loss == "w":
        return -out_r.mean() + out_f.mean()
    if loss == "hinge":
        return torch.relu(1.0 - out_r).mean() + torch.relu(1.0 + out_f).mean()
    raise ValueError(loss)


def gradient_penalty_ref(gp_lambda: float, sdD: Dict[str, Tensor], real: Tensor, fake: Tensor, alpha: Tensor,
                         cfg: Optional[dict] = None, p_disc: float = 0.0, keeps=None, model: str = "mpgan") -> Tensor:
    """gradient_penalty (train.py:286-324) for either discriminator (``model`` = "mpgan" / "gapt"):
    x = alpha real + (1 - alpha) fake with alpha [B, 1, 1] (:288-294); D(x) WITHOUT labels (:301);
    g = d sum(D(x)) / dx with the graph kept (:304-311); per jet || g ||_2 over all particles and features -- the
    mask column included -- with 1e-12 under the root (:316-320); gp_lambda * mean (norm - 1)^2 (:323).
    Differentiable with respect to the entries of ``sdD`` (the D step adds it to the loss, :381-384)."""
    B = real.shape[0]
    x = (alpha * real + (1 - alpha) * fake.detach()).requires_grad_(True)
    prob = _fwd_D(model, sdD, x, None, p_disc, keeps, cfg or {})
    g = torch.autograd.grad(prob, x, torch.ones_like(prob), create_graph=True, retain_graph=True)[0]
    norm = torch.sqrt((g.reshape(B, -1) ** 2).sum(1) + 1e-12)
    return gp_lambda * ((norm - 1) ** 2).mean()


def g_loss_ref(loss: str, out: Tensor) -> Tensor:
    """calc_G_loss (train.py:465-476)."""
    if loss == "ls":
        return ((out - 1.0) ** 2).mean()
    if loss == "og":
        return torch.nn.BCELoss()(out, torch.ones_like(out))
    if loss in ("w", "hinge"):
        return -out.mean()
    raise ValueError(loss)


def train_iteration(
    model: str,
    sdD: Dict[str, Tensor],
    sdG: Dict[str, Tensor],
    stD: Dict[str, Tensor],
    stG: Dict[str, Tensor],
    data: Tensor,
    labels: Tensor,
    noise_D: Tensor,
    noise_G: Tensor,
    lr_disc: float,
    lr_gen: float,
    p_disc: float = 0.0,
    keeps: Optional[Tuple] = None,
    cfg: Optional[dict] = None,
    return_grads: bool = False,
    loss: str = "ls",
    signs: Optional[Tuple] = None,
):
    """One train_D + train_G (num_critic = num_gen = 1), RMSprop; ``loss`` = ls (default) / og / w / hinge as
    ``calc_D_loss`` (train.py:331-395) and ``calc_G_loss`` (:465-476) define them (for w / hinge the caller passes
    ``cfg={"D": {"sigmoid": False}}``, setup_training.py:1250).  Parameters in
    ``sdD``/``sdG`` are updated in place.  ``keeps`` = (keeps for D(real), D(fake) in the
    D step, D(fake) in the G step) or None; each is a dict as taken by the D forward, or a
    ``RandKeeps()`` to draw Bernoulli masks.  ``signs`` (MPGAN, tests): the sign-conditioned evaluation
    (``mpgan_ref.leaky``) -- per-layer sign dicts for (D(real), D(fake) in the D step, G and D(fake) in the G step).
    Returns (D_loss, G_loss[, gradsD, gradsG])."""
    cfg = cfg or {}
    N = data.shape[1]
    kr, kf, kg = keeps if keeps is not None else (None, None, None)
    sr, sf, sgg, sgd = signs if signs is not None else (None, None, None, None)

    # ---- train_D (train.py:398-462)
    pD = {k: v.detach().requires_grad_(True) for k, v in sdD.items()}
    with torch.no_grad():
        fake = _fwd_G(model, sdG, noise_D, labels, N, cfg)
    out_r = _fwd_D(model, pD, data, labels, p_disc, kr, cfg, sr)
    out_f = _fwd_D(model, pD, fake, labels, p_disc, kf, cfg, sf)
    D_loss = d_loss_ref(loss, out_r, out_f)
    gD = dict(zip(pD.keys(), torch.autograd.grad(D_loss, list(pD.values()))))
    rmsprop_step(sdD, gD, stD, lr_disc)

    # ---- train_G (train.py:479-523)
    pG = {k: v.detach().requires_grad_(True) for k, v in sdG.items()}
    fake = _fwd_G(model, pG, noise_G, labels, N, cfg, sgg)
    out = _fwd_D(model, sdD, fake, labels, p_disc, kg, cfg, sgd)
    G_loss = g_loss_ref(loss, out)
    gG = dict(zip(pG.keys(), torch.autograd.grad(G_loss, list(pG.values()))))
    rmsprop_step(sdG, gG, stG, lr_gen)

    if return_grads:
        return float(D_loss.detach()), float(G_loss.detach()), gD, gG
    return float(D_loss.detach()), float(G_loss.detach())

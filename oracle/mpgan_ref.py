"""Functional CPU restatement of MPGAN's message-passing networks (TEST ORACLE).

Follows (reference paths relative to /root/reference):
  * LinearNet.forward                 mpgan/model.py:70-85
  * MPLayer.forward (default branch)  mpgan/model.py:206-282, _getA_fully_connected :284-317
  * MPNet.forward loop                mpgan/model.py:498-523
  * MPGenerator mask_c / final mask   mpgan/model.py:689-704, :723-752
  * MPDiscriminator mask/pool/fnd     mpgan/model.py:810-831, :833-890

Written from the closed form in SURVEY.md A.3 -- no ``repeat``/``cat`` edge tensor is
built the way the reference does it; instead the pairwise pre-activation is formed as
``W1[:, :F] x_i + W1[:, F:] x_j`` by broadcasting, which is the identity the HIP kernel
uses as well.  ``exact_concat=True`` switches to the literal concat form (used by the
golden test to show both agree with the reference).

Dropout: the reference uses torch's Bernoulli stream, which no other implementation can
reproduce bit-for-bit.  Here every dropout site takes an explicit keep-mask tensor (1 =
keep) already containing the {0,1} decision; the scale 1/(1-p) is applied here.  With
``drop=None`` dropout is the identity (p = 0 or eval mode).
"""

from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

Tensor = torch.Tensor

MPGAN_DEFAULTS = dict(
    num_particles=30,
    latent_node_size=32,
    hidden_node_size=32,
    node_feat_size=3,
    fe_layers=(96, 160, 192),
    fn_layers=(256, 256),
    mp_iters=2,
    sum_agg=True,
    alpha=0.2,
)


def leaky(t: Tensor, alpha: float, neg: Optional[Tensor] = None) -> Tensor:
    # phi(t) = max(t,0) + alpha*min(t,0); F.leaky_relu semantics (mpgan/model.py:80)
    # ``neg`` (tests only): the branch of every element decided by the caller instead of by the sign of t -- the
    # "sign-conditioned" evaluation: gradients of the function the device kernels computed, whose kink decisions within
    # rounding of zero may differ from fp64's (the decisions themselves are counted against fp32's by the tests)
    if neg is not None:
        return torch.where(neg, t * alpha, t)
    return torch.where(t > 0, t, t * alpha)


RAND = "rand"  # sentinel keep-mask: draw a fresh Bernoulli(1-p) mask (CPU-baseline timing only)


class RandKeeps(dict):
    """A keeps-dict/sequence stand-in that answers RAND (or itself, for nested lookups) to
    every key: makes the oracle draw torch Bernoulli masks like the reference's nn.Dropout."""

    def get(self, k, d=None):
        return self if (isinstance(k, str) and (k.startswith("sab") or k == "pma")) else RAND

    def __getitem__(self, k):
        return self


def _drop(t: Tensor, keep, p: float) -> Tensor:
    if keep is None or p == 0.0:
        return t
    if isinstance(keep, str):  # RAND
        keep = torch.empty_like(t).bernoulli_(1.0 - p)
    return t * keep * (1.0 / (1.0 - p))


def linearnet_forward(
    sd: Dict[str, Tensor],
    prefix: str,
    x: Tensor,
    n_layers: int,
    final_linear: bool,
    alpha: float = 0.2,
    p: float = 0.0,
    keeps: Optional[Sequence[Optional[Tensor]]] = None,
) -> Tensor:
    """``prefix`` = e.g. 'mp_layers.0.fe' ; parameters are ``{prefix}.net.{l}.weight|bias``.

    Per layer: Linear -> (LeakyReLU unless last & final_linear) -> Dropout, dropout after
    EVERY layer including a final linear one (mpgan/model.py:77-83)."""
    for l in range(n_layers):
        w = sd[f"{prefix}.net.{l}.weight"]
        b = sd[f"{prefix}.net.{l}.bias"]
        x = x @ w.t() + b
        if l != n_layers - 1 or not final_linear:
            x = leaky(x, alpha)
        x = _drop(x, None if keeps is None else keeps[l], p)
    return x


def mplayer_forward(
    sd: Dict[str, Tensor],
    prefix: str,
    x: Tensor,
    mask: Optional[Tensor] = None,
    sum_agg: bool = True,
    alpha: float = 0.2,
    p: float = 0.0,
    keeps: Optional[Dict[str, Tensor]] = None,
    n_fe: int = 3,
    n_fn: int = 3,
    exact_concat: bool = False,
    probe: Optional[list] = None,
    knn: Optional[tuple] = None,
    signs: Optional[Dict[str, Tensor]] = None,
) -> Tensor:
    """One message-passing layer, fully connected or (``knn`` = (num_knn, self_loops)) over each node's nearest
    neighbours as ``MPLayer._getA_knn`` picks them (mpgan/model.py:319-381): distances || s_j x_j - x_i + 1e-12 || with
    s_j = 1e4 for zero-masked senders, ascending sort, the ``num_knn`` entries from position 0 (self loops) or 1.

    x    [B,N,F]; mask [B,N,1] (1 real / 0 padded) or None
    keeps: optional dict of keep masks 'e0','e1','e2' with shape [B,N,N,H_l] and
           'n0','n1','n2' with shape [B,N,out_l].
    Edge row (b, i, j) = [x_i ; x_j] (receiver first) -- mpgan/model.py:294-295,315.
    Mask multiplies the SENDER axis j only (:262); mean divides by N (:267).
    signs: optional dict 'fe1','fe2','fe3' [B,N,N,H_l] / 'fn1','fn2' [B,N,*] of bool "took the negative branch" (see ``leaky``;
           fully connected form only).
    """
    B, N, F = x.shape
    k = keeps or {}
    sg = signs or {}
    w1 = sd[f"{prefix}.fe.net.0.weight"]
    b1 = sd[f"{prefix}.fe.net.0.bias"]
    if knn is not None:
        num_knn, self_loops = knn
        xs = x if mask is None else ((1 - 1e4) * mask + 1e4) * x               # (:333-335)
        dists = torch.norm(xs.unsqueeze(1) - x.unsqueeze(2) + 1e-12, dim=3)    # [B, i, j]
        first = 0 if self_loops else 1
        idx = torch.sort(dists, dim=2)[1][:, :, first:first + num_knn]         # [B, N, k]
        gather = lambda t: torch.gather(t.unsqueeze(1).expand(B, N, N, t.shape[-1]), 2,
                                        idx.unsqueeze(3).expand(B, N, num_knn, t.shape[-1]))
        xg = gather(x)                                                         # neighbours' features [B, N, k, F]
        a = x @ w1[:, :F].t() + b1
        e = a.unsqueeze(2) + xg @ w1[:, F:].t()
        if probe is not None:
            probe.append(e.detach())
        e = _drop(leaky(e, alpha), k.get("e0"), p)
        for l in range(1, n_fe):
            e = e @ sd[f"{prefix}.fe.net.{l}.weight"].t() + sd[f"{prefix}.fe.net.{l}.bias"]
            if probe is not None:
                probe.append(e.detach())
            e = _drop(leaky(e, alpha), k.get(f"e{l}"), p)
        if mask is not None:
            e = e * gather(mask)
        agg = e.sum(dim=2) if sum_agg else e.mean(dim=2)
        return _node_net(sd, prefix, agg, x, n_fn, alpha, p, k, probe)
    if exact_concat:
        xi = x.unsqueeze(2).expand(B, N, N, F)
        xj = x.unsqueeze(1).expand(B, N, N, F)
        e = torch.cat((xi, xj), dim=3) @ w1.t() + b1
    else:
        a = x @ w1[:, :F].t() + b1  # receiver term  [B,N,H1]
        c = x @ w1[:, F:].t()  # sender term    [B,N,H1]
        e = a.unsqueeze(2) + c.unsqueeze(1)  # [B,N(i),N(j),H1]
    if probe is not None:  # pre-activations, for the tests' distance-from-the-kink check
        probe.append(e.detach())
    e = _drop(leaky(e, alpha, sg.get("fe1")), k.get("e0"), p)
    for l in range(1, n_fe):
        w = sd[f"{prefix}.fe.net.{l}.weight"]
        b = sd[f"{prefix}.fe.net.{l}.bias"]
        e = e @ w.t() + b
        if probe is not None:
            probe.append(e.detach())
        e = _drop(leaky(e, alpha, sg.get(f"fe{l + 1}")), k.get(f"e{l}"), p)
    if mask is not None:
        e = e * mask.reshape(B, 1, N, 1)
    agg = e.sum(dim=2)
    if not sum_agg:
        agg = agg / N
    return _node_net(sd, prefix, agg, x, n_fn, alpha, p, k, probe, sg)


def mplayer_forward_general(
    sd: Dict[str, Tensor],
    prefix: str,
    x: Tensor,
    mask: Optional[Tensor] = None,
    labels: Optional[Tensor] = None,
    num_jet_particles: Optional[Tensor] = None,
    *,
    pos_diffs: bool = False,
    all_ef: bool = True,
    coords: str = "polarrel",
    delta_coords: bool = False,
    delta_r: bool = True,
    clabels: int = 0,
    mask_fne_np: bool = False,
    knn: Optional[tuple] = None,
    sum_agg: bool = True,
    alpha: float = 0.2,
) -> Tensor:
    """``MPLayer.forward`` with its non-default options (mpgan/model.py:206-282), any layer widths, no dropout: the edge
    tensor is materialised row by row as the reference builds it.

    * edge features (``_getA_fully_connected`` :297-313): diffs = x_j - x_i over all features (``all_ef``) or the first
      2 / 3 coordinates; dists = || diffs + 1e-12 ||; appended as [diffs, dists] (delta_r and delta_coords), [dists]
      (delta_r or all_ef) or [diffs] (delta_coords).  On the k-NN graph (:319-381) the appended column is the sorted
      distance itself (computed with zero-masked senders pushed away by 1e4).
    * ``clabels`` / ``mask_fne_np`` (:247-253, :270-276) are appended with ``t.repeat(rows / B, 1)``: ROW r of the edge
      (node) matrix receives the entry of jet  r mod B  -- not of the jet the row belongs to.  Restated as is.
    """
    B, N, F = x.shape
    nc = 3 if coords == "cartesian" else 2
    if knn is None:
        k = N
        xi = x.unsqueeze(2).expand(B, N, N, F)   # edge (b, i, j): receiver i first (:294)
        xj = x.unsqueeze(1).expand(B, N, N, F)
        parts = [xi, xj]
        if pos_diffs:
            diffs = (xj - xi) if all_ef else (xj[..., :nc] - xi[..., :nc])
            dists = torch.norm(diffs + 1e-12, dim=3, keepdim=True)
            if delta_r and delta_coords:
                parts += [diffs, dists]
            elif delta_r or all_ef:
                parts += [dists]
            elif delta_coords:
                parts += [diffs]
        edge_mask = None if mask is None else mask.unsqueeze(1)   # senders only (:262)
    else:
        k, self_loops = knn
        xs = x if mask is None else ((1 - 1e4) * mask + 1e4) * x
        dd = xs.unsqueeze(1) - x.unsqueeze(2)
        if not (all_ef or not pos_diffs):
            dd = dd[..., :nc]
        srt = torch.sort(torch.norm(dd + 1e-12, dim=3), dim=2)
        first = 0 if self_loops else 1
        idx = srt[1][:, :, first:first + k]
        gather = lambda t: torch.gather(t.unsqueeze(1).expand(B, N, N, t.shape[-1]), 2,
                                        idx.unsqueeze(3).expand(B, N, k, t.shape[-1]))
        parts = [x.unsqueeze(2).expand(B, N, k, F), gather(x)]
        if pos_diffs:
            parts.append(srt[0][:, :, first:first + k].unsqueeze(3))
        edge_mask = None if mask is None else gather(mask)
    A = torch.cat(parts, dim=3).reshape(B * N * k, -1)
    if clabels:
        A = torch.cat((A, labels[:, :clabels].repeat(N * k, 1)), dim=1)
    if mask_fne_np:
        A = torch.cat((A, num_jet_particles.repeat(N * k, 1)), dim=1)
    n_fe = sum(1 for key in sd if key.startswith(f"{prefix}.fe.net.") and key.endswith(".weight"))
    n_fn = sum(1 for key in sd if key.startswith(f"{prefix}.fn.net.") and key.endswith(".weight"))
    e = A
    for l in range(n_fe):
        e = leaky(e @ sd[f"{prefix}.fe.net.{l}.weight"].t() + sd[f"{prefix}.fe.net.{l}.bias"], alpha)
    e = e.reshape(B, N, k, -1)
    if edge_mask is not None:
        e = e * edge_mask
    agg = e.sum(dim=2) if sum_agg else e.mean(dim=2)
    h = torch.cat((agg, x), dim=2).reshape(B * N, -1)
    if clabels:
        h = torch.cat((h, labels[:, :clabels].repeat(N, 1)), dim=1)
    if mask_fne_np:
        h = torch.cat((h, num_jet_particles.repeat(N, 1)), dim=1)
    for l in range(n_fn):
        h = h @ sd[f"{prefix}.fn.net.{l}.weight"].t() + sd[f"{prefix}.fn.net.{l}.bias"]
        if l + 1 < n_fn:
            h = leaky(h, alpha)
    return h.reshape(B, N, -1)


def _node_net(sd, prefix, agg, x, n_fn, alpha, p, k, probe, sg=None):
    """fn on [agg ; x] (mpgan/model.py:268-279)."""
    sg = sg or {}
    h = torch.cat((agg, x), dim=2)
    for l in range(n_fn):
        w = sd[f"{prefix}.fn.net.{l}.weight"]
        b = sd[f"{prefix}.fn.net.{l}.bias"]
        h = h @ w.t() + b
        if l != n_fn - 1:
            if probe is not None:
                probe.append(h.detach())
            h = leaky(h, alpha, sg.get(f"fn{l + 1}"))
        h = _drop(h, k.get(f"n{l}"), p)
    return h


def gen_mask_from_labels(first_feat: Tensor, labels: Tensor, num_particles: int) -> Tensor:
    """mask_c: n = int(labels[:,-1]*N) - 1 ; mask_i = rank_i(x[:,:,0]) <= n  (mpgan/model.py:689-699)."""
    n = (labels[:, -1] * num_particles).int() - 1
    rank = first_feat.argsort(1).argsort(1)
    return (rank <= n.unsqueeze(1)).unsqueeze(2).to(first_feat.dtype)


def mpgen_forward(
    sd: Dict[str, Tensor],
    noise: Tensor,
    labels: Tensor,
    num_particles: int = 30,
    mp_iters: int = 2,
    sum_agg: bool = True,
    alpha: float = 0.2,
    p: float = 0.0,
    keeps: Optional[Sequence[Optional[Dict[str, Tensor]]]] = None,
    tanh: bool = True,
    signs: Optional[Sequence[Optional[Dict[str, Tensor]]]] = None,
) -> Tensor:
    """MPGenerator.forward, default config (mask_c, no lfc): mpgan/model.py:498-523,:689-704,:752."""
    mask = gen_mask_from_labels(noise[:, :, 0], labels, num_particles)
    x = noise
    for l in range(mp_iters):
        x = mplayer_forward(
            sd, f"mp_layers.{l}", x, mask, sum_agg, alpha, p, None if keeps is None else keeps[l],
            signs=None if signs is None else signs[l]
        )
    if tanh:
        x = torch.tanh(x)
    return torch.cat((x, mask - 0.5), dim=2)


def mpdisc_forward(
    sd: Dict[str, Tensor],
    data: Tensor,
    labels: Optional[Tensor] = None,
    mp_iters: int = 2,
    sum_agg: bool = True,
    alpha: float = 0.2,
    p: float = 0.0,
    keeps: Optional[Sequence[Optional[Dict[str, Tensor]]]] = None,
    keep_fnd: Optional[Tensor] = None,
    sigmoid: bool = True,
    signs: Optional[Sequence[Optional[Dict[str, Tensor]]]] = None,
) -> Tensor:
    """MPDiscriminator.forward, default config (mask_c, dea, dea_sum, fnd=[]):
    mask = x[...,-1:]+0.5 (:881); features = x[...,:-1] (:884); pooled = sum_i x_i*mask_i
    (:816-817, sum because dea and dea_sum); fnd Linear(+dropout) (:829); sigmoid (:537)."""
    mask = data[:, :, -1:] + 0.5
    x = data[:, :, :-1]
    for l in range(mp_iters):
        x = mplayer_forward(
            sd, f"mp_layers.{l}", x, mask, sum_agg, alpha, p, None if keeps is None else keeps[l],
            signs=None if signs is None else signs[l]
        )
    pooled = (x * mask).sum(dim=1)
    if not sum_agg:
        pooled = pooled / (mask.sum(dim=1) + 1e-12)
    out = pooled @ sd["fnd_layer.net.0.weight"].t() + sd["fnd_layer.net.0.bias"]
    out = _drop(out, keep_fnd, p)
    return torch.sigmoid(out) if sigmoid else out

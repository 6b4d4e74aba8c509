"""Functional CPU restatement of GAPT's attention blocks (TEST ORACLE).

Follows (reference paths relative to /root/reference):
  * MAB.forward          gapt/model.py:124-139  (nn.MultiheadAttention :107, batch_first)
  * SAB / PMA / ISAB     gapt/model.py:143-191
  * _attn_mask           gapt/model.py:194-202
  * GAPT_G.forward       gapt/model.py:251-274
  * GAPT_D.forward       gapt/model.py:332-344

``nn.MultiheadAttention`` is restated from its published definition:
q,k,v = rows [0:E],[E:2E],[2E:3E] of in_proj_weight; per head softmax(q k^T / sqrt(d))
with ignored keys at -inf; concat heads; out_proj.  Key names are the reference's
state-dict names.  Dropout sites take explicit keep masks (see mpgan_ref).
"""

from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .mpgan_ref import leaky, _drop, gen_mask_from_labels

Tensor = torch.Tensor


def _mha(sd, prefix, x: Tensor, y: Tensor, ignore: Optional[Tensor], num_heads: int) -> Tensor:
    """x [B,L,E] queries, y [B,S,E] keys/values, ignore [B,S] bool (True = do not attend)."""
    B, L, E = x.shape
    S = y.shape[1]
    d = E // num_heads
    w = sd[f"{prefix}.in_proj_weight"]
    b = sd[f"{prefix}.in_proj_bias"]
    q = x @ w[:E].t() + b[:E]
    k = y @ w[E : 2 * E].t() + b[E : 2 * E]
    v = y @ w[2 * E :].t() + b[2 * E :]
    q = q.reshape(B, L, num_heads, d).transpose(1, 2)  # [B,H,L,d]
    k = k.reshape(B, S, num_heads, d).transpose(1, 2)
    v = v.reshape(B, S, num_heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d)  # [B,H,L,S]
    if ignore is not None:
        ig = ignore.reshape(B, 1, 1, S)
        # a query row whose keys are ALL ignored: torch's scaled_dot_product_attention (>= 2.5, `_safe_softmax`; pinned here
        # by the 2.10 goldens) returns zero attention weights for it instead of softmax(-inf, ..., -inf) = NaN.  It happens
        # in the gradient penalty, whose interpolated jets attend only to particles real in BOTH endpoints
        # (gapt/model.py:194-202: every mask value other than exactly 1 is "ignore").
        dead = ig.all(dim=-1, keepdim=True)
        s = s.masked_fill(ig & ~dead, float("-inf"))
        pr = torch.softmax(s, dim=-1) * (~dead).to(s.dtype)
    else:
        pr = torch.softmax(s, dim=-1)
    o = (pr @ v).transpose(1, 2).reshape(B, L, E)
    return o @ sd[f"{prefix}.out_proj.weight"].t() + sd[f"{prefix}.out_proj.bias"]


def _layer_norm(sd, prefix, x):
    return torch.nn.functional.layer_norm(
        x, (x.shape[-1],), sd[f"{prefix}.weight"], sd[f"{prefix}.bias"]
    )


def mab_forward(
    sd: Dict[str, Tensor],
    prefix: str,
    x: Tensor,
    y: Tensor,
    ignore: Optional[Tensor],
    num_heads: int = 4,
    alpha: float = 0.2,
    p: float = 0.0,
    keeps: Optional[Dict[str, Tensor]] = None,
    layer_norm: bool = False,
    ff_final_linear: bool = False,
) -> Tensor:
    """z = drop([LN](x + MHA(x,y,y))); out = drop([LN](z + ff(z))), ff = Linear(E,E)->LReLU->drop.

    keeps: 'a' (after attention residual), 'f' (inside ff), 'o' (after ff residual)."""
    k = keeps or {}
    z = x + _mha(sd, f"{prefix}.attention", x, y, ignore, num_heads)
    if layer_norm:
        z = _layer_norm(sd, f"{prefix}.norm1", z)
    z = _drop(z, k.get("a"), p)
    f = z @ sd[f"{prefix}.ff.net.0.weight"].t() + sd[f"{prefix}.ff.net.0.bias"]
    if not ff_final_linear:
        f = leaky(f, alpha)
    f = _drop(f, k.get("f"), p)
    o = z + f
    if layer_norm:
        o = _layer_norm(sd, f"{prefix}.norm2", o)
    return _drop(o, k.get("o"), p)


def sab_forward(sd, prefix, x, mask, **kw):
    """SAB: every query row shares the key mask (gapt/model.py:148-154). mask [B,N,1] 1=real."""
    ignore = None if mask is None else (1 - mask[:, :, 0]).bool()
    return mab_forward(sd, f"{prefix}.mab", x, x, ignore, **kw)


def pma_forward(sd, prefix, x, mask, **kw):
    """PMA: queries are the learned seeds S repeated over the batch (gapt/model.py:170-174)."""
    ignore = None if mask is None else (1 - mask[:, :, 0]).bool()
    s = sd[f"{prefix}.S"].expand(x.shape[0], -1, -1)
    return mab_forward(sd, f"{prefix}.mab", s, x, ignore, **kw)


def isab_forward(sd, prefix, x, mask, **kw):
    """ISAB: H = MAB0(I, X, mask); out = MAB1(X, H) unmasked (gapt/model.py:187-191)."""
    ignore = None if mask is None else (1 - mask[:, :, 0]).bool()
    ind = sd[f"{prefix}.I"].expand(x.shape[0], -1, -1)
    h = mab_forward(sd, f"{prefix}.mab0", ind, x, ignore, **kw)
    return mab_forward(sd, f"{prefix}.mab1", x, h, None, **kw)


def gapt_g_forward(
    sd, noise, labels, num_particles=30, sab_layers=4, num_heads=4, alpha=0.2, use_isab=False
):
    """GAPT_G.forward (gapt/model.py:251-274), use_mask=True, dropout 0."""
    mask = gen_mask_from_labels(noise[:, :, 0], labels, num_particles)
    x = noise
    blk = isab_forward if use_isab else sab_forward
    for s in range(sab_layers):
        x = blk(sd, f"sabs.{s}", x, mask, num_heads=num_heads, alpha=alpha)
    x = torch.tanh(x @ sd["final_fc.net.0.weight"].t() + sd["final_fc.net.0.bias"])
    return torch.cat((x, mask - 0.5), dim=2)


def gapt_d_forward(
    sd,
    data,
    labels=None,
    sab_layers=2,
    num_heads=4,
    alpha=0.2,
    p=0.0,
    keeps=None,
    use_isab=False,
):
    """GAPT_D.forward (gapt/model.py:332-344): embed(Linear+LReLU+drop) -> SABs -> PMA(1 seed)
    -> squeeze -> Linear(+drop) -> sigmoid (always).
    keeps: dict with 'emb', 'sab{i}' (dict a/f/o), 'pma' (dict), 'fc'."""
    k = keeps or {}
    mask = data[..., -1:] + 0.5
    x = data[..., :-1]
    x = leaky(x @ sd["input_embedding.net.0.weight"].t() + sd["input_embedding.net.0.bias"], alpha)
    x = _drop(x, k.get("emb"), p)
    blk = isab_forward if use_isab else sab_forward
    for s in range(sab_layers):
        x = blk(sd, f"sabs.{s}", x, mask, num_heads=num_heads, alpha=alpha, p=p, keeps=k.get(f"sab{s}"))
    x = pma_forward(sd, "pma", x, mask, num_heads=num_heads, alpha=alpha, p=p, keeps=k.get("pma"))
    x = x.reshape(x.shape[0], -1)
    out = x @ sd["final_fc.net.0.weight"].t() + sd["final_fc.net.0.bias"]
    out = _drop(out, k.get("fc"), p)
    return torch.sigmoid(out)

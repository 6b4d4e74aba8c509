"""GAPT (set-transformer GAN) on the MI355X path -- drop-in for the reference's ``gapt`` package.

Same class names, constructor keywords, ``forward`` signatures and state-dict key names as
rkansal47/MPGAN ``gapt/model.py`` (LinearNet :11-89, MAB :93-139, SAB :143-154, PMA :158-174,
ISAB :178-191, GAPT_G :205-274, GAPT_D :277-344).  ``MAB.attention`` keeps the parameter layout of
``nn.MultiheadAttention`` (``in_proj_weight [3E,E]``, ``in_proj_bias``, ``out_proj.weight|bias``) so
reference checkpoints load; the arithmetic runs in libmpgan_amd.so: projections and the feed-forward
on the split-16-bit MFMA GEMM, softmax(QK^T)V in the attention-core kernel.
LayerNorm (``layer_norm=True``) runs on ``ops.LayerNormFn``; batch norm / spectral norm variants are outside the fused
path (NotImplementedError).
"""
from __future__ import annotations

from typing import Optional

import os

import torch
from torch import Tensor, nn

from .. import ops
from ..mpgan.model import LinearNet, _unsupported, _rank_mask


class _MHAParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names and default initialisation."""

    def __init__(self, embed_dim: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


def _lin(x2, W, b, row0, rows):
    """x2 @ W[row0:row0+rows].T + b[row0:row0+rows] through the fused linear op."""
    return ops.FusedLinearFn.apply(x2, W[row0:row0 + rows], None if b is None else b[row0:row0 + rows],
                                   False, 0.2, 0.0, False)


class MAB(nn.Module):
    # class-wide switch: eligible blocks run as one launch (tests compare both paths; MPG_MAB_FUSED=0: block by block from the start)
    fused = os.environ.get("MPG_MAB_FUSED", "1") != "0"

    def __init__(self, embed_dim: int, num_heads: int, ff_layers: list = [], layer_norm: bool = False,
                 dropout_p: float = 0.0, final_linear: bool = True, linear_args={}):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.attention = _MHAParams(embed_dim)
        self.ff = LinearNet(ff_layers, input_size=embed_dim, output_size=embed_dim, final_linear=final_linear,
                            **linear_args)
        self.layer_norm = layer_norm
        if layer_norm:  # (registered after ``ff`` as in the reference, so state-dict order matches)
            self.norm1 = nn.LayerNorm(embed_dim)
            self.norm2 = nn.LayerNorm(embed_dim)
        self.dropout_p = float(dropout_p)

    def forward(self, x: Tensor, y: Tensor, y_mask: Tensor = None):
        """x [B,L,E] queries, y [B,S,E] keys/values; y_mask: bool, True = ignore, of shape [B,L,S]
        (every query row carries the same key mask -- that is how SAB/PMA/ISAB build it) or [B,S]."""
        B, L, E = x.shape
        S = y.shape[1]
        att = self.attention
        ignore = self._ignore_of(y_mask, B, S)
        x2 = x.reshape(B * L, E)
        if x.is_cuda and ops.double_backward_on(x.device):
            return self._forward_dd(x, y, ignore)
        if self._fused_ok(x, L, S):
            return self._fused(x, y, ignore, B, L, S)
        if x is y:   # the packed projections go to the attention core as they are (no q/k/v slices in autograd)
            qkv = _lin(x2, att.in_proj_weight, att.in_proj_bias, 0, 3 * E)
            o = ops.FusedPackedAttnFn.apply(qkv, None, ignore, B, L, S, self.num_heads)
        else:
            q = _lin(x2, att.in_proj_weight, att.in_proj_bias, 0, E)
            kv = _lin(y.reshape(B * S, E), att.in_proj_weight, att.in_proj_bias, E, 2 * E)
            o = ops.FusedPackedAttnFn.apply(q, kv, ignore, B, L, S, self.num_heads)
        # x + out_proj(attention): the residual is added in the projection's own launch
        za = ops.FusedLinearFn.apply(o, att.out_proj.weight, att.out_proj.bias, False, 0.2, 0.0, False, x2)
        if self.layer_norm:
            za = ops.LayerNormFn.apply(za, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        z = ops.FusedDropoutFn.apply(za, self.dropout_p, self.training)
        zf = self.ff(z, resid=z)
        if self.layer_norm:
            zf = ops.LayerNormFn.apply(zf, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        out = ops.FusedDropoutFn.apply(zf, self.dropout_p, self.training)
        return out.reshape(B, L, E)


    def _forward_dd(self, x: Tensor, y: Tensor, ignore: Optional[Tensor]) -> Tensor:
        """The twice-differentiable form of the block (``ops.double_backward_route``: the gradient penalty's
        ``D(interpolated)``, train.py:301-311): the five projections as ``ops.MatMulFn`` (closed under differentiation),
        the two attention contractions per (jet, head) as broadcast products + sums, softmax / LeakyReLU / dropout / LayerNorm
        as ATen's own -- all of whose derivative formulas are differentiable again.  A query row whose keys are ALL ignored
        gets zero attention weights, as torch's scaled_dot_product_attention gives it (an interpolated jet attends only
        to particles that are real in both of its endpoints, gapt/model.py:194-202 -- possibly none)."""
        B, L, E = x.shape
        S, H = y.shape[1], self.num_heads
        d = E // H
        att = self.attention
        W, b = att.in_proj_weight, att.in_proj_bias
        mm = ops.MatMulFn.apply
        x2, y2 = x.reshape(B * L, E), y.reshape(B * S, E)
        q = (mm(x2, W[:E], "nt") + b[:E]).reshape(B, L, H, d).permute(0, 2, 1, 3)            # [B, H, L, d]
        k = (mm(y2, W[E:2 * E], "nt") + b[E:2 * E]).reshape(B, S, H, d).permute(0, 2, 1, 3)
        v = (mm(y2, W[2 * E:], "nt") + b[2 * E:]).reshape(B, S, H, d).permute(0, 2, 1, 3)
        s = (q.unsqueeze(3) * k.unsqueeze(2)).sum(-1) * (1.0 / d ** 0.5)                     # [B, H, L, S]
        if ignore is not None:
            ig = ignore.reshape(B, 1, 1, S) != 0     # (1 - mask).bool(): everything but exactly 1 is "ignore" (:194-202)
            dead = ig.all(dim=-1, keepdim=True)
            s = s.masked_fill(ig & ~dead, float("-inf"))
            pr = torch.softmax(s, dim=-1) * (~dead).to(s.dtype)
        else:
            pr = torch.softmax(s, dim=-1)
        o = (pr.unsqueeze(-1) * v.unsqueeze(2)).sum(3).permute(0, 2, 1, 3).reshape(B * L, E)
        za = x2 + mm(o, att.out_proj.weight, "nt") + att.out_proj.bias
        if self.layer_norm:
            za = torch.nn.functional.layer_norm(za, (E,), self.norm1.weight, self.norm1.bias, self.norm1.eps)
        z = torch.nn.functional.dropout(za, self.dropout_p, self.training)
        zf = self.ff(z, resid=z)                                                             # (LinearNet takes its own dd form)
        if self.layer_norm:
            zf = torch.nn.functional.layer_norm(zf, (E,), self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return torch.nn.functional.dropout(zf, self.dropout_p, self.training).reshape(B, L, E)

    @staticmethod
    def _ignore_of(y_mask, B: int, S: int):
        """The float key mask [B*S] (1 = ignore) the kernels take, from the bool / float mask the blocks pass around."""
        if y_mask is None:
            return None
        ignore = getattr(y_mask, "_mpg_ignore", None)  # float key mask prepared once per network forward
        if ignore is None or ignore.numel() != B * S:
            km = y_mask if y_mask.dim() == 2 else y_mask[:, 0, :]
            ignore = km.reshape(B * S).float().contiguous()
        return ignore

    # -- the whole block as one launch (ops.mab_forward; csrc/mab.hip) ------------------------------------------
    def _fused_ok(self, x: Tensor, L: int, S: int) -> bool:
        return (MAB.fused and x.is_cuda and len(self.ff.net) == 1 and self.ff.plain
                and ops.mab_fusable(self.embed_dim, self.num_heads, L, S))

    def _packed(self) -> "ops.PackedMAB":
        pk = self.__dict__.get("_pack")
        att = self.attention
        if pk is None or pk.params[0] is not att.in_proj_weight:
            pk = self.__dict__["_pack"] = ops.PackedMAB(att.in_proj_weight, att.out_proj.weight, self.ff.net[0].weight)
        return pk.ensure()

    def packed_sets(self):
        pk = self.__dict__.get("_pack")
        return [] if pk is None else [pk]

    def refresh_packed(self):
        """Re-pack after an update torch cannot see (``train.TrainStep``'s fused optimizer step)."""
        for pk in self.packed_sets():
            pk.refresh()

    def _fused(self, x, y, ignore, B, L, S):
        att, lin = self.attention, self.ff.net[0]
        E = self.embed_dim
        kw = dict(alpha=self.ff.leaky_relu_alpha, ff_act=not self.ff.final_linear, p_mab=self.dropout_p,
                  p_ff=self.ff.dropout_p, training=self.training)
        ln = (self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias, self.norm1.eps) if self.layer_norm else None
        if not torch.is_grad_enabled():
            # (one query row for all jets -- PMA's seed -- is read with row stride 0)
            x2 = x.reshape(1, E).contiguous().expand(B, E) if (x.shape[0] == 1 and B > 1) else x.reshape(B * L, E).contiguous()
            y2 = None if x is y else y.reshape(B * S, E).contiguous()
            out = ops.mab_forward(x2, y2, ignore, self._packed(), att.in_proj_bias, att.out_proj.bias, lin.bias,
                                  B, L, S, self.num_heads, ln=ln, **kw)[0]
            return out.reshape(B, L, E)
        if ln is not None:   # layer_norm=True: both norms inside the block's launches (ops.FusedMABLayerNormFn)
            assert self.norm1.eps == self.norm2.eps
            if x.shape[0] == 1 and B > 1:
                x = x.expand(B, L, E)     # (a shared query row: autograd sums its gradient over the jets)
            return ops.FusedMABLayerNormFn.apply(x, None if x is y else y, ignore, att.in_proj_weight, att.in_proj_bias,
                                                 att.out_proj.weight, att.out_proj.bias, lin.weight, lin.bias, ln[0], ln[1], ln[2], ln[3],
                                                 ln[4], self.num_heads, kw["alpha"], kw["ff_act"], kw["p_mab"], kw["p_ff"],
                                                 kw["training"], self._packed())
        return ops.FusedMABFn.apply(x, None if x is y else y, ignore, att.in_proj_weight, att.in_proj_bias,
                                    att.out_proj.weight, att.out_proj.bias, lin.weight, lin.bias, self.num_heads,
                                    kw["alpha"], kw["ff_act"], kw["p_mab"], kw["p_ff"], kw["training"], self._packed())


class SAB(nn.Module):
    def __init__(self, **mab_args):
        super().__init__()
        self.mab = MAB(**mab_args)

    def forward(self, x: Tensor, mask: Tensor = None):
        return self.mab(x, x, _key_mask(mask))  # mask [B,N,1] bool, True = ignore


class PMA(nn.Module):
    def __init__(self, embed_dim: int, num_seeds: int, **mab_args):
        super().__init__()
        self.S = nn.Parameter(torch.empty(1, num_seeds, embed_dim))
        nn.init.xavier_uniform_(self.S)
        self.mab = MAB(embed_dim, **mab_args)

    def forward(self, x: Tensor, mask: Tensor = None):
        if self.S.shape[1] == 1 and x.is_cuda and x.size(0) > 1 and self.mab._fused_ok(x, 1, x.shape[1]) and not ops.double_backward_on(x.device):
            # one seed: the one-launch block reads the single row for every jet (no B copies, no reduction launch for its gradient)
            return self.mab._fused(self.S, x, self.mab._ignore_of(_key_mask(mask), x.size(0), x.shape[1]), x.size(0), 1, x.shape[1])
        seeds = self.S.expand(x.size(0), -1, -1).contiguous()
        return self.mab(seeds, x, _key_mask(mask))


class ISAB(nn.Module):
    def __init__(self, num_inds, embed_dim, **mab_args):
        super().__init__()
        self.I = nn.Parameter(torch.empty(1, num_inds, embed_dim))
        self.num_inds = num_inds
        nn.init.xavier_uniform_(self.I)
        self.mab0 = MAB(embed_dim=embed_dim, **mab_args)
        self.mab1 = MAB(embed_dim=embed_dim, **mab_args)

    def forward(self, X, mask: Tensor = None):
        ind = self.I.expand(X.size(0), -1, -1).contiguous()
        H = self.mab0(ind, X, _key_mask(mask))
        return self.mab1(X, H)


def _run_sabs(sabs, x: Tensor, am) -> Tensor:
    """``for sab in sabs: x = sab(x, mask)`` (gapt/model.py:261-262, :341-342).  Runs of plain SABs whose blocks take the
    one-launch kernel go out as ONE forward launch per run (``ops.FusedSABChainFn``, up to ``MAB_CHAIN_MAX`` blocks each): a wave
    keeps its jet's rows in registers from block to block.  MPG_MAB_CHAIN=0: block by block."""
    import os
    from .._lib import MAB_CHAIN_MAX
    sabs = list(sabs)
    B, N, E = x.shape
    chainable = (x.is_cuda and os.environ.get("MPG_MAB_CHAIN", "1") != "0" and not ops.double_backward_on(x.device)
                 and B * N > 0 and N <= ops.MAB_CHAIN_TOKENS)
    i = 0
    while i < len(sabs):
        run = []
        if chainable:
            while (i + len(run) < len(sabs) and len(run) < MAB_CHAIN_MAX and isinstance(sabs[i + len(run)], SAB)
                   and sabs[i + len(run)].mab._fused_ok(x, N, N) and not sabs[i + len(run)].mab.layer_norm):
                run.append(sabs[i + len(run)].mab)
            if run and not all(_same_block_config(m, run[0]) for m in run):
                run = run[:1]
        if len(run) >= 2 and B <= 4096:      # (one jet per wave: the launch covers at most 1,024 workgroups of four)
            first = run[0]
            ignore = MAB._ignore_of(_key_mask(am), B, N)
            params = []
            for m in run:
                att, lin = m.attention, m.ff.net[0]
                params += [att.in_proj_weight, att.in_proj_bias, att.out_proj.weight, att.out_proj.bias, lin.weight, lin.bias]
            pks = [m._packed() for m in run]
            args = (first.num_heads, first.ff.leaky_relu_alpha, not first.ff.final_linear, first.dropout_p, first.ff.dropout_p,
                    first.training)
            if torch.is_grad_enabled() and (x.requires_grad or any(q.requires_grad for q in params)):
                x = ops.FusedSABChainFn.apply(x, ignore, *args, pks, *params)
            else:
                x = ops.sab_chain_forward(x, ignore, *args, pks, params)
            i += len(run)
        else:
            x = sabs[i](x, am)
            i += 1
    return x


def _same_block_config(a: "MAB", b: "MAB") -> bool:
    return (a.embed_dim == b.embed_dim and a.num_heads == b.num_heads and a.dropout_p == b.dropout_p and a.training == b.training
            and a.ff.dropout_p == b.ff.dropout_p and a.ff.leaky_relu_alpha == b.ff.leaky_relu_alpha
            and a.ff.final_linear == b.ff.final_linear)


def _attn_mask(mask: Tensor) -> Optional[Tensor]:
    """JetNet mask (1 real, 0 padded) -> attention convention (True = ignore).  The float form the attention kernels
    take ([B*N], 1 = ignore) rides along as an attribute, so that the blocks of a network do not each convert it."""
    if mask is None:
        return None
    return _ignore_mask(1 - mask)


def _ignore_mask(inv: Tensor) -> Tensor:
    """[B, N, 1] floats, 1 = ignore, as the attention mask the blocks pass around.  On the GPU the float tensor itself
    travels (the kernels read it through ``_mpg_ignore``; a bool copy would be one more launch per forward)."""
    am = inv if inv.is_cuda else inv.bool()
    am._mpg_ignore = inv.reshape(-1).contiguous()
    return am


def _key_mask(mask: Optional[Tensor]) -> Optional[Tensor]:
    """[B,N,1] attention mask -> the [B,N] key mask a MAB takes (keeping the prepared float form attached)."""
    if mask is None:
        return None
    km = mask[:, :, 0]
    ig = getattr(mask, "_mpg_ignore", None)
    if ig is not None:
        km._mpg_ignore = ig
    return km


def _sab_args(embed_dim, sab_fc_layers, num_heads, layer_norm, dropout_p, linear_args):
    return {"embed_dim": embed_dim, "ff_layers": sab_fc_layers, "final_linear": False, "num_heads": num_heads,
            "layer_norm": layer_norm, "dropout_p": dropout_p, "linear_args": linear_args}


class GAPT_G(nn.Module):
    def __init__(self, num_particles: int, output_feat_size: int, sab_layers: int = 2, num_heads: int = 4,
                 embed_dim: int = 32, sab_fc_layers: list = [], layer_norm: bool = False, dropout_p: float = 0.0,
                 final_fc_layers: list = [], use_mask: bool = True, use_isab: bool = False,
                 num_isab_nodes: int = 10, linear_args: dict = {}):
        super().__init__()
        self.num_particles, self.output_feat_size, self.use_mask = num_particles, output_feat_size, use_mask
        self.embed_dim = embed_dim
        args = _sab_args(embed_dim, sab_fc_layers, num_heads, layer_norm, dropout_p, linear_args)
        self.sabs = nn.ModuleList(ISAB(num_isab_nodes, **args) if use_isab else SAB(**args)
                                  for _ in range(sab_layers))
        self.final_fc = LinearNet(final_fc_layers, input_size=embed_dim, output_size=output_feat_size,
                                  final_linear=True, **linear_args)

    def _mask(self, x, labels):
        """(mask [B, N, 1] or None, the attention mask built from it): on the GPU one launch writes both."""
        if not self.use_mask:
            return None, None
        if x.is_cuda:
            mask, ign = ops.rank_mask(x[:, :, 0], labels, self.num_particles, with_ignore=True)
            return mask.unsqueeze(2), _ignore_mask(ign.unsqueeze(2))
        mask = _rank_mask(x[:, :, 0], labels, self.num_particles)
        return mask, _attn_mask(mask)

    def forward(self, x: Tensor, labels: Tensor = None):
        mask, am = self._mask(x, labels)
        x = _run_sabs(self.sabs, x, am)
        x = self.final_fc(x)
        if x.is_cuda:  # tanh + the (mask - 0.5) column in one launch each way
            return ops.GenTailFn.apply(x, mask, ops.ACT_CODES["tanh"])
        x = torch.tanh(x)
        return torch.cat((x, mask - 0.5), dim=2) if mask is not None else x

    def noise_mask_ok(self) -> bool:
        """The mask is a function of the input noise alone: a caller may draw both in one launch (``ops.normal_noise_masked``)
        and hand the masks in as ``premask``."""
        return bool(self.use_mask)

    def generate_parts(self, x: Tensor, labels: Tensor, feat_out: Tensor = None, mask_out: Tensor = None, ign_out: Tensor = None,
                       premask=None):
        """``forward`` without gluing the mask column on: (particle features [B, N, F] after tanh, mask [B, N, 1], 1 - mask
        [B, N]); see ``MPGenerator.generate_parts``.  ``premask``: (mask [B, N], 1 - mask) already drawn with the noise."""
        assert x.is_cuda and self.use_mask
        B = x.shape[0]
        if premask is not None:
            mask2d, ign = premask
        else:
            mask2d, ign = ops.rank_mask(x[:, :, 0], labels, self.num_particles, out=None if mask_out is None else mask_out.view(B, -1),
                                        with_ignore=True, ignore_out=None if ign_out is None else ign_out.view(B, -1))
        mask = mask2d.unsqueeze(2)
        am = _ignore_mask(ign.unsqueeze(2))
        x = _run_sabs(self.sabs, x, am)
        x = self.final_fc(x)
        if feat_out is not None:
            assert not torch.is_grad_enabled()
            return ops.gen_tail_into(x, None, ops.ACT_CODES["tanh"], feat_out), mask, ign
        return ops.GenTailFn.apply(x, None, ops.ACT_CODES["tanh"]), mask, ign

    def bridge_head(self):
        """(weight [F, E], bias, activation code) of ``final_fc`` + tanh when they can run inside ``ops.GenDiscBridgeFn``
        (one plain Linear, no dropout on it); else None."""
        fc = self.final_fc
        if len(fc.net) != 1 or not fc.plain or (fc.dropout_p and self.training):
            return None
        lin = fc.net[0]
        return lin.weight, lin.bias, ops.ACT_CODES["tanh"]

    def generate_rows(self, x: Tensor, labels: Tensor, mask_out: Tensor = None, ign_out: Tensor = None, premask=None):
        """``generate_parts`` up to the last attention block: (rows [B, N, E] that ``final_fc`` would take, mask [B, N, 1],
        1 - mask [B, N]) -- for a caller that runs ``final_fc``, the tanh and the discriminator's embedding as one launch
        (``GAPT_D.features_rows``)."""
        assert x.is_cuda and self.use_mask
        B = x.shape[0]
        if premask is not None:
            mask2d, ign = premask
        else:
            mask2d, ign = ops.rank_mask(x[:, :, 0], labels, self.num_particles, out=None if mask_out is None else mask_out.view(B, -1),
                                        with_ignore=True, ignore_out=None if ign_out is None else ign_out.view(B, -1))
        return _run_sabs(self.sabs, x, _ignore_mask(ign.unsqueeze(2))), mask2d.unsqueeze(2), ign

    def generate_into(self, x: Tensor, labels: Tensor, out: Tensor) -> Tensor:
        """``forward`` into caller-owned output rows, no gradient (``train.TrainStep``'s D step)."""
        assert not torch.is_grad_enabled() and x.is_cuda
        mask, am = self._mask(x, labels)
        x = _run_sabs(self.sabs, x, am)
        return ops.gen_tail_into(self.final_fc(x), mask, ops.ACT_CODES["tanh"], out)


class GAPT_D(nn.Module):
    def __init__(self, num_particles: int, input_feat_size: int, sab_layers: int = 2, num_heads: int = 4,
                 embed_dim: int = 32, sab_fc_layers: list = [], layer_norm: bool = False, dropout_p: float = 0.0,
                 final_fc_layers: list = [], use_mask: bool = True, use_isab: bool = False,
                 num_isab_nodes: int = 10, linear_args: dict = {}):
        super().__init__()
        self.num_particles, self.input_feat_size, self.use_mask = num_particles, input_feat_size, use_mask
        self.embed_dim = embed_dim
        args = _sab_args(embed_dim, sab_fc_layers, num_heads, layer_norm, dropout_p, linear_args)
        self.sabs = nn.ModuleList()  # registered first, as in the reference, so state-dict order matches
        self.input_embedding = LinearNet([], input_size=input_feat_size, output_size=embed_dim, **linear_args)
        for _ in range(sab_layers):
            self.sabs.append(ISAB(num_isab_nodes, **args) if use_isab else SAB(**args))
        self.pma = PMA(num_seeds=1, **args)
        self.final_fc = LinearNet(final_fc_layers, input_size=embed_dim, output_size=1, final_linear=True,
                                  **linear_args)

    def fused_head(self):
        """(weight [1, E], bias, mean?, sigmoid?, dropout p) of ``final_fc`` + sigmoid as ``ops.DiscHeadFn`` (with
        N = 1 "particles": the pooled seed), when ``final_fc`` is a single Linear; else None."""
        if len(self.final_fc.net) != 1 or not self.final_fc.plain:
            return None
        lin = self.final_fc.net[0]
        return lin.weight, lin.bias, False, True, self.final_fc.dropout_p

    def features(self, x: Tensor, labels: Tensor = None):
        """Everything up to the pooled seed: ([B, 1, E], None)."""
        am = None
        if self.use_mask:
            # mask = x[..., -1:] + 0.5 (:336); what the blocks need is 1 - mask = 0.5 - x[..., -1:]: one launch
            if x.is_cuda and ops.double_backward_on(x.device):
                # interpolated jets (gradient penalty) carry fractional mask values: the reference's own order of operations,
                # so that "exactly 1" is decided on the same roundings
                am = _ignore_mask(1 - (x.detach()[..., -1:] + 0.5))
            else:
                am = _ignore_mask(0.5 - x.detach()[..., -1:])   # (no gradient flows through the mask column: :336-338, bool mask)
            x = x[..., :-1]
        x = self.input_embedding(x)   # (a column slice of the [.., 4] rows: the GEMM takes the row stride as it is)
        x = _run_sabs(self.sabs, x, am)
        return self.pma(x, am), None

    def parts_ok(self) -> bool:
        return bool(self.use_mask)

    def features_parts(self, x3: Tensor, mask: Tensor, labels: Tensor = None, ignore: Tensor = None):
        """``features`` for callers that hold the particle features [B, N, F], the mask and 1 - mask [B, N] apart."""
        B, N = x3.shape[:2]
        inv = (1 - mask) if ignore is None else ignore
        am = _ignore_mask(inv.reshape(B, N, 1))
        x = self.input_embedding(x3)
        x = _run_sabs(self.sabs, x, am)
        return self.pma(x, am), None

    def bridge_tail(self):
        """(weight [E, F], bias, LeakyReLU alpha, dropout p) of ``input_embedding`` when it can run inside
        ``ops.GenDiscBridgeFn`` (one plain Linear); else None."""
        emb = self.input_embedding
        if len(emb.net) != 1 or not emb.plain or emb.final_linear:
            return None
        lin = emb.net[0]
        return lin.weight, lin.bias, emb.leaky_relu_alpha, emb.dropout_p

    def features_rows(self, pre: Tensor, head, feat_buf: Tensor, mask: Tensor, labels: Tensor = None, ignore: Tensor = None):
        """``features_parts`` for generated jets that are still the generator's rows ``pre`` [Bg, N, E] (``GAPT_G.generate_rows``;
        ``head`` = its ``bridge_head()``): ``final_fc``, tanh and ``input_embedding`` in one launch.  ``feat_buf``: None, or
        the [B, N, F] batch whose first B - Bg jets are real -- the generated features are written behind them."""
        W2, b2, alpha, p = self.bridge_tail()
        W1, b1, act1 = head
        B = pre.shape[0] if feat_buf is None else feat_buf.shape[0]
        N = pre.shape[1]
        inv = (1 - mask) if ignore is None else ignore
        am = _ignore_mask(inv.reshape(B, N, 1))
        _, x = ops.GenDiscBridgeFn.apply(pre, W1, b1, feat_buf, W2, b2, act1, True, alpha, p, self.training)
        x = _run_sabs(self.sabs, x, am)
        return self.pma(x, am), None

    def forward(self, x: Tensor, labels: Tensor = None):
        pooled, _ = self.features(x, labels)
        head = self.fused_head() if (pooled.is_cuda and not ops.double_backward_on(pooled.device)) else None
        if head is None:
            return torch.sigmoid(self.final_fc(pooled.squeeze()))  # .squeeze() as the reference (:344)
        w, b, mean, sigmoid, p = head
        out = ops.DiscHeadFn.apply(pooled, None, w, b, mean, sigmoid, p, self.training)
        return out.unsqueeze(1) if out.shape[0] > 1 else out   # shapes of the reference's .squeeze() path

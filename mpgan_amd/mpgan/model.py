"""MPGAN networks on the fused MI355X hot path -- drop-in for the reference's ``mpgan`` package.

Same class names, constructor keywords, ``forward`` signatures and state-dict key names as
rkansal47/MPGAN ``mpgan/model.py`` (LinearNet :11-88, MPLayer :91-384, MPNet :387-569,
MPGenerator :572-757, MPDiscriminator :760-894), so ``setup_training.setup_mpgan`` /
``train.py`` / ``gen.py`` and published checkpoints work unchanged.  The arithmetic runs in
libmpgan_amd.so (HIP, gfx950); there is no CPU fallback.

Fused (``ops.FusedMPLayerFn``: the reference's default and every published ``mp_*`` configuration): edge network
[96, 160, 192], two hidden node layers, fully connected or ``fully_connected=False`` with ``num_knn`` / ``self_loops``,
and up to ``ops.EDGE_SCALARS`` scalars per edge -- the distance column of ``pos_diffs``, ``clabels``, ``mask_fne_np`` --
plus coordinate differences (``delta_coords``, folded into the layer-1 projection); ``mask_c`` masking, ``dea`` pooling.
Every other constructor option of the reference (more edge scalars, conditioning columns on the k-NN graph, other layer
widths, batch / spectral norm) runs un-fused, layer by layer on the HIP GEMM (``MPLayer._forward_edges``); the few options
the reference itself cannot run (``int_diffs``, ``mask_learn*``, ``mask_feat_bin``) raise ``NotImplementedError`` at
construction.  ``MPLayer.fused`` says which route a layer takes.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from .. import ops


def _unsupported(**flags):
    bad = [k for k, v in flags.items() if v]
    if bad:
        raise NotImplementedError(
            "mpgan_amd: option(s) %s are outside the fused MI355X path (see DESIGN.md, scope)" % ", ".join(bad))


class SpectralNorm(nn.Module):
    """Spectral normalisation of a Linear layer as the reference wraps it (mpgan/spectral_normalization.py:11-61): the
    wrapped module keeps ``weight_bar`` (trained), ``weight_u`` / ``weight_v`` (power-iteration vectors, no gradient) and
    ``bias`` -- same state-dict keys (``module.weight_bar`` ...).  ``weight()`` runs one power iteration and
    returns  W_bar / (u' W_bar v + 1e-12)  with the gradient flowing through W_bar only, which the caller feeds to the
    fused Linear launch."""

    def __init__(self, module: nn.Linear, power_iterations: int = 1):
        super().__init__()
        self.module, self.power_iterations = module, power_iterations
        w = module.weight
        u = nn.Parameter(torch.nn.functional.normalize(torch.randn(w.shape[0]), dim=0, eps=1e-12), requires_grad=False)
        v = nn.Parameter(torch.nn.functional.normalize(torch.randn(w.shape[1]), dim=0, eps=1e-12), requires_grad=False)
        del module._parameters["weight"]
        module.register_parameter("weight_u", u)
        module.register_parameter("weight_v", v)
        module.register_parameter("weight_bar", nn.Parameter(w.data))

    @property
    def bias(self):
        return self.module.bias

    def weight(self) -> Tensor:
        mod = self.module
        u, v, w = mod.weight_u, mod.weight_v, mod.weight_bar
        # The power iteration runs OUT OF PLACE and sigma is built from the fresh vectors, so autograd never saves u / v
        # themselves (train_D runs D(real) and D(fake) before one backward: writing u in place must not touch what an
        # earlier forward saved).  The new vectors are then stored IN PLACE, at the parameters' fixed addresses: a captured
        # hipGraph replays this very copy, so the iteration keeps accumulating across replayed steps as it does across the
        # reference's eager ones (spectral_normalization.py:29-39) -- rebinding ``.data`` would run once, at capture time.
        with torch.no_grad():
            un, vn = u, v
            for _ in range(self.power_iterations):
                t = torch.mv(w.t(), un)
                vn = t / (t.norm() + 1e-12)
                t = torch.mv(w, vn)
                un = t / (t.norm() + 1e-12)
            if self.power_iterations > 0:
                u.copy_(un)
                v.copy_(vn)
            else:
                un, vn = u.clone(), v.clone()
        sigma = un.dot(w.mv(vn))
        return w / (sigma + 1e-12)


class LinearNet(nn.Module):
    """Stack of ``Linear -> LeakyReLU -> [BatchNorm1d] -> Dropout`` (last layer ``Linear -> Dropout`` when
    ``final_linear``); parameters live in ``self.net`` so keys read ``net.{i}.weight|bias`` (``bn.{i}.*`` with batch
    norm, ``net.{i}.module.weight_bar|weight_u|weight_v|bias`` with spectral norm: mpgan/model.py:55-68)."""

    def __init__(self, layers: list, input_size: int = 0, output_size: int = 0, final_linear: bool = False,
                 leaky_relu_alpha: float = 0.2, dropout_p: float = 0, batch_norm: bool = False,
                 spectral_norm: bool = False):
        super().__init__()
        widths = ([input_size] if input_size else []) + list(layers) + ([output_size] if output_size else [])
        self.final_linear = final_linear
        self.leaky_relu_alpha = leaky_relu_alpha
        self.dropout_p = float(dropout_p)
        self.batch_norm, self.spectral_norm = bool(batch_norm), bool(spectral_norm)
        self.net = nn.ModuleList(nn.Linear(i, o) for i, o in zip(widths[:-1], widths[1:]))
        if self.batch_norm:   # (one per layer, the last one unused when final_linear: as the reference registers them)
            self.bn = nn.ModuleList(nn.BatchNorm1d(o) for o in widths[1:])
        if self.spectral_norm:
            for i in range(len(self.net)):
                if i != len(self.net) - 1 or not final_linear:
                    self.net[i] = SpectralNorm(self.net[i])

    @property
    def plain(self) -> bool:
        """Neither normalisation: the form the fused kernels (and the grouped weight-gradient launches) take."""
        return not (self.batch_norm or self.spectral_norm)

    def _bn(self, i: int, x: Tensor) -> Tensor:
        bn = self.bn[i]
        if not (self.training or not bn.track_running_stats):
            return ops.batchnorm_eval(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        y, mean, var = ops.BatchNormFn.apply(x, bn.weight, bn.bias, bn.eps)
        if bn.track_running_stats:   # running statistics as nn.BatchNorm1d keeps them (unbiased variance, momentum 0.1)
            with torch.no_grad():
                M = x.shape[0]
                bn.num_batches_tracked += 1
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var, alpha=mom * M / max(M - 1, 1))
        return y

    def _forward_dd(self, x: Tensor, resid: Tensor = None) -> Tensor:
        """The twice-differentiable form (``ops.double_backward_route``): Linear as ``ops.MatMulFn`` + bias, LeakyReLU and
        dropout as ATen's own (their derivative formulas are differentiable; dropout draws from torch's generator, as
        the reference's does).  Batch norm has a first-order kernel only."""
        if self.batch_norm:
            raise NotImplementedError("LinearNet: batch norm has no double-backward route (the gradient penalty needs one)")
        last = len(self.net) - 1
        shp = x.shape
        x = x.reshape(-1, shp[-1])
        for k, lin in enumerate(self.net):
            W = lin.weight() if isinstance(lin, SpectralNorm) else lin.weight
            x = ops.MatMulFn.apply(x, W, "nt")
            if lin.bias is not None:
                x = x + lin.bias
            if not (self.final_linear and k == last):
                x = torch.nn.functional.leaky_relu(x, self.leaky_relu_alpha)
            x = torch.nn.functional.dropout(x, self.dropout_p, self.training)
        x = x.reshape(*shp[:-1], x.shape[-1])
        return x if resid is None else x + resid

    def forward(self, x: Tensor, resid: Tensor = None) -> Tensor:
        """``resid`` (not in the reference signature): added to the output, inside the last layer's launch when that
        layer has no activation -- MAB's ``x + ff(x)``."""
        if x.is_cuda and ops.double_backward_on(x.device):
            return self._forward_dd(x, resid)
        last = len(self.net) - 1
        for k, lin in enumerate(self.net):
            act = not (self.final_linear and k == last)
            fuse = resid if (k == last and not act) else None
            W = lin.weight() if isinstance(lin, SpectralNorm) else lin.weight
            if self.batch_norm and act:   # the norm sits between the activation and the dropout (:78-83)
                x = ops.FusedLinearFn.apply(x, W, lin.bias, True, self.leaky_relu_alpha, 0.0, self.training, None)
                x = self._bn(k, x.reshape(-1, x.shape[-1])).reshape(x.shape)
                x = ops.FusedDropoutFn.apply(x, self.dropout_p, self.training)
            else:
                x = ops.FusedLinearFn.apply(x, W, lin.bias, act, self.leaky_relu_alpha, self.dropout_p, self.training, fuse)
            if k == last and resid is not None and fuse is None:
                x = x + resid
        return x

    def __repr__(self):
        return f"{self.__class__.__name__}(net = {self.net})"


class MPLayer(nn.Module):
    """One message-passing iteration.  The reference's default configuration -- edge network [96, 160, 192] on
    [x_i ; x_j], two hidden node layers, fully connected or k-NN graph -- runs on ``ops.FusedMPLayerFn`` (fused edge
    kernels + chained node network).  Every other option of the reference's constructor (edge features ``pos_diffs`` /
    ``delta_r`` / ``delta_coords``, conditioning columns ``clabels`` / ``mask_fne_np``, other layer widths) takes the
    un-fused route of ``_forward_edges``: the edge matrix is materialised as the reference builds it and the two
    networks run layer by layer on the HIP GEMM -- same results, none of the fused kernels' speed."""

    def __init__(self, input_node_size: int, fe_layers: list, fn_layers: list, output_node_size: int,
                 pos_diffs: bool = False, all_ef: bool = True, coords: str = "polarrel", delta_coords: bool = False,
                 delta_r: bool = True, int_diffs: bool = False, clabels: int = 0, mask_fne_np: bool = False,
                 fully_connected: bool = True, num_knn: int = 20, self_loops: bool = True, sum: bool = True,
                 **linear_args):
        super().__init__()
        # int_diffs reserves an input column of the edge network that the reference's forward never fills (:180, :284-317)
        _unsupported(int_diffs=int_diffs)
        self.input_node_size, self.output_node_size = input_node_size, output_node_size
        self.fe_layers, self.fn_layers = list(fe_layers), list(fn_layers)
        self.sum = sum
        self.pos_diffs, self.all_ef, self.coords = bool(pos_diffs), bool(all_ef), coords
        self.delta_coords, self.delta_r = bool(delta_coords), bool(delta_r)
        self.clabels, self.mask_fne_np = int(clabels), bool(mask_fne_np)
        # fully_connected=False: every receiver aggregates over its num_knn nearest senders only (reference _getA_knn);
        # the fused kernels still walk all N senders with the neighbour sets as a per-edge 0/1 factor
        self.fully_connected, self.num_knn, self.self_loops = fully_connected, int(num_knn), bool(self_loops)
        num_ef = 0   # edge features appended to [x_i ; x_j] (:169-178)
        if self.pos_diffs:
            if self.delta_coords:
                num_ef += 3 if coords == "cartesian" else 2
            if self.delta_r or self.all_ef:
                num_ef += 1
        self.num_ef = num_ef
        extra = self.clabels + int(self.mask_fne_np)
        # What the fused kernels take besides [x_i ; x_j]: up to ops.EDGE_SCALARS scalars per edge, each times its own column
        # of fe.net.0.weight -- the distance column (delta_r / all_ef) and the conditioning columns, which the reference
        # tiles over ROWS and so are per-edge gathers (_edge_scalars) -- and coordinate differences, which are linear in
        # x_i, x_j and fold into the a | c projection (_folded_w1).  On the k-NN graph the one edge feature is the distance the
        # neighbours were ranked by (to the mask-scaled senders, mpgan/model.py:333-345, :372): the same scalar.  Not fused:
        # differences over ALL features with their own columns (all_ef + delta_coords: the reference sizes fe for 2-3 of
        # them and fails itself), more scalars than EDGE_SCALARS, other layer widths.
        nc = 3 if coords == "cartesian" else 2
        # the columns MPLayer.forward appends (:303-308): [diffs, dists] | [dists] | [diffs], diffs over nc coordinates or all features
        has_diffs = self.pos_diffs and self.delta_coords and (self.delta_r or not self.all_ef)
        self._dist_col = self.pos_diffs and (self.delta_r or self.all_ef)
        self._diff_cols = nc if (has_diffs and not self.all_ef) else 0
        self.n_es = int(self._dist_col) + extra
        consistent = num_ef == self._diff_cols + int(self._dist_col) and not (has_diffs and self.all_ef)
        self.fused = (list(self.fe_layers) == [ops.H1, ops.H2, ops.H3] and len(self.fn_layers) == 2
                      and consistent and self.n_es <= ops.EDGE_SCALARS
                      and (self.fully_connected or (not self._diff_cols and extra == 0))
                      and not (linear_args.get("batch_norm") or linear_args.get("spectral_norm")))
        self.fe = LinearNet(self.fe_layers, input_size=2 * input_node_size + num_ef + extra, final_linear=False, **linear_args)
        self.fn = LinearNet(self.fn_layers, input_size=self.fe_layers[-1] + input_node_size + extra,
                            output_size=output_node_size, final_linear=True, **linear_args)

    def _forward_edges(self, x: Tensor, use_mask: bool, mask: Tensor, labels: Tensor, num_jet_particles: Tensor) -> Tensor:
        """The un-fused route: edge matrix built row by row as ``MPLayer.forward`` / ``_getA_*`` do (mpgan/model.py:
        206-381), edge and node networks through ``LinearNet`` (one fused Linear launch per layer)."""
        B, N, F = x.shape
        nc = 3 if self.coords == "cartesian" else 2
        if self.fully_connected:
            k = N
            xi, xj = x.unsqueeze(2).expand(B, N, N, F), x.unsqueeze(1).expand(B, N, N, F)   # edge (b, i, j) = [x_i ; x_j]
            cols = [xi, xj]
            if self.pos_diffs:
                diffs = (xj - xi) if self.all_ef else (xj[..., :nc] - xi[..., :nc])
                dists = torch.norm(diffs + 1e-12, dim=3, keepdim=True)
                if self.delta_r and self.delta_coords:
                    cols += [diffs, dists]
                elif self.delta_r or self.all_ef:
                    cols += [dists]
                elif self.delta_coords:
                    cols += [diffs]
            edge_mask = mask.unsqueeze(1) if use_mask else None   # the sender's mask
        else:
            k = self.num_knn
            if k + int(not self.self_loops) > N:
                raise ValueError(f"num_knn = {k} neighbours (self_loops = {self.self_loops}) out of {N} nodes")
            xs = (((1 - 1e4) * mask + 1e4) * x) if use_mask else x   # zero-masked senders are pushed out of reach
            dd = xs.unsqueeze(1) - x.unsqueeze(2)
            if self.pos_diffs and not self.all_ef:
                dd = dd[..., :nc]
            order = torch.sort(torch.norm(dd + 1e-12, dim=3), dim=2)
            first = int(not self.self_loops)
            idx = order.indices[:, :, first:first + k]

            def neighbours(t):
                return torch.gather(t.unsqueeze(1).expand(B, N, N, t.shape[-1]), 2,
                                    idx.unsqueeze(3).expand(B, N, k, t.shape[-1]))
            cols = [x.unsqueeze(2).expand(B, N, k, F), neighbours(x)]
            if self.pos_diffs:
                cols.append(order.values[:, :, first:first + k].unsqueeze(3))
            edge_mask = neighbours(mask) if use_mask else None
        A = torch.cat(cols, dim=3).reshape(B * N * k, -1)
        # conditioning columns: ``t.repeat(rows / B, 1)`` tiles the [B, C] block, so ROW r gets the entry of jet r mod B
        # (:249, :253, :272, :276) -- reproduced as the reference computes it
        if self.clabels:
            A = torch.cat((A, labels[:, :self.clabels].repeat(N * k, 1)), dim=1)
        if self.mask_fne_np:
            A = torch.cat((A, num_jet_particles.repeat(N * k, 1)), dim=1)
        E = self.fe(A.contiguous()).reshape(B, N, k, self.fe_layers[-1])
        if edge_mask is not None:
            E = E * edge_mask
        agg = E.sum(dim=2) if self.sum else E.mean(dim=2)
        h = torch.cat((agg, x), dim=2).reshape(B * N, -1)
        if self.clabels:
            h = torch.cat((h, labels[:, :self.clabels].repeat(N, 1)), dim=1)
        if self.mask_fne_np:
            h = torch.cat((h, num_jet_particles.repeat(N, 1)), dim=1)
        return self.fn(h.contiguous()).reshape(B, N, self.output_node_size)

    def forward(self, x: Tensor, use_mask: bool = False, mask: Tensor = None, labels: Tensor = None,
                num_jet_particles: Tensor = None) -> Tensor:
        assert not (use_mask and mask is None), "need ``mask`` tensor if using ``use_mask`` option"
        assert not (self.clabels and labels is None), "need ``labels`` tensor if using ``clabels`` option"
        assert not (self.mask_fne_np and num_jet_particles is None), "need ``num_jet_particles`` tensor if using ``mask_fne_np`` option"
        if not self.fused or (x.is_cuda and ops.double_backward_on(x.device)):
            return self._forward_edges(x, use_mask, mask, labels, num_jet_particles)
        fe, fn = self.fe.net, self.fn.net
        nbr = None
        if not self.fully_connected:
            if self.num_knn + int(not self.self_loops) > x.shape[1]:
                raise ValueError(f"num_knn = {self.num_knn} neighbours (self_loops = {self.self_loops}) out of {x.shape[1]} nodes")
            with torch.no_grad():   # (with pos_diffs and not all_ef the reference ranks by the coordinates alone, :340-345)
                xr = x[..., :(3 if self.coords == "cartesian" else 2)] if (self.pos_diffs and not self.all_ef) else x
                nbr = ops.knn_sets(xr, mask if use_mask else None, self.num_knn, self.self_loops)
        es, xfn = self._edge_scalars(x, labels, num_jet_particles, mask if (use_mask and not self.fully_connected) else None)
        W1, packed = fe[0].weight, None
        if self._diff_cols:
            W1 = self._folded_w1(W1)   # (not a parameter: its images are packed for this call)
        else:
            packed = self._packed()
        # consecutive fused layers of a network hand each other the layer-1 node terms a | c (ops.LayerHandoff): the launch
        # that produces a layer's output rows also projects them for the next layer
        handoff = None
        nxt = self.__dict__.get("_next_layer")
        ac_in = getattr(x, "_mpg_ac", None)
        # (x straight out of another fused layer: this layer's backward may run that layer's input-gradient chain in its own launch)
        prev_node = x.grad_fn if (x.grad_fn is not None and type(x.grad_fn).__name__ == "FusedMPLayerFnBackward") else None
        if packed is not None and (ac_in is not None or nxt is not None or prev_node is not None):
            nx = None
            if nxt is not None and nxt.fused and not nxt._diff_cols and nxt.training == self.training and nxt.n_es == 0:
                nx = (nxt._packed(), nxt.fe.net[0].bias)
            handoff = ops.LayerHandoff(next=nx, ac_in=ac_in, prev_node=prev_node)
        y = ops.FusedMPLayerFn.apply(
            x, mask if use_mask else None,
            W1, fe[0].bias, fe[1].weight, fe[1].bias, fe[2].weight, fe[2].bias,
            fn[0].weight, fn[0].bias, fn[1].weight, fn[1].bias, fn[2].weight, fn[2].bias,
            self.sum, self.fe.leaky_relu_alpha, self.fe.dropout_p, self.training, packed, nbr, self.num_knn,
            es, self.n_es, xfn, handoff, not torch.is_grad_enabled())
        if handoff is not None and handoff.ac_out is not None:
            y._mpg_ac = handoff.ac_out
        return y

    def _folded_w1(self, W1: Tensor) -> Tensor:
        """fe.net.0.weight with the coordinate-difference columns folded into the node columns: the reference appends
        x_j[:nc] - x_i[:nc] to [x_i ; x_j] (mpgan/model.py:297-313), and W_d (x_j - x_i) is -W_d x_i + W_d x_j.  Built with
        torch operations, so the columns' gradients come out of autograd."""
        F, nc = self.input_node_size, self._diff_cols
        Wd = torch.nn.functional.pad(W1[:, 2 * F:2 * F + nc], (0, F - nc))
        return torch.cat((W1[:, :F] - Wd, W1[:, F:2 * F] + Wd, W1[:, 2 * F + nc:]), dim=1)

    def _edge_scalars(self, x: Tensor, labels: Tensor, num_jet_particles: Tensor, knn_mask: Tensor = None):
        """``(es [B, N senders, EDGE_SCALARS, N receivers] or None, xfn [B, N, F + E] or None)``: the scalars the fused edge
        kernels multiply with their own columns of fe.net.0.weight -- the distance ||x_j - x_i + 1e-12|| (mpgan/model.py:
        299-302; 4 bytes per edge instead of the edge matrix) and the conditioning columns AS THE REFERENCE TILES THEM:
        ``t.repeat(rows / B, 1)`` gives ROW r the entry of jet r mod B (:249, :253), a per-edge gather for the edge network
        and a per-node one for the node network (:272, :276).  (On the k-NN graph the reference's edge rows are (b, i, rank);
        the conditioning columns are then tiled over THOSE rows, which the fused kernels -- walking all senders -- do not
        have: clabels / mask_fne_np with fully_connected=False stay on the un-fused route.)"""
        if self.n_es == 0:
            return None, None
        B, N, F = x.shape
        cols = []
        if self._dist_col:
            xs = x if knn_mask is None else ((1 - 1e4) * knn_mask + 1e4) * x   # k-NN: zero-masked senders pushed away (:333-335)
            d = xs.unsqueeze(1) - x.unsqueeze(2)                # [B, i, j, F] = x_j - x_i
            if not self.all_ef:
                d = d[..., :(3 if self.coords == "cartesian" else 2)]
            cols.append(torch.norm(d + 1e-12, dim=3).transpose(1, 2))   # [B, j, i]
        xfn = None
        if self.clabels or self.mask_fne_np:
            erow = (torch.arange(B * N * N, device=x.device) % B).reshape(B, N, N).transpose(1, 2)   # [B, j, i]: jet of edge row (b, i, j)
            nrow = torch.arange(B * N, device=x.device) % B
            extra = []
            for t in ([labels[:, q] for q in range(self.clabels)] + ([num_jet_particles[:, 0]] if self.mask_fne_np else [])):
                cols.append(t.to(x.dtype)[erow])
                extra.append(t.to(x.dtype)[nrow].reshape(B, N, 1))
            xfn = torch.cat([x] + extra, dim=2)
        cols += [torch.zeros_like(cols[0])] * (ops.EDGE_SCALARS - len(cols))
        return torch.stack(cols, dim=2), xfn

    def _packed(self) -> "ops.PackedMPLayer":
        """Persistent weight images of this layer for the current mode (dropout scale) -- rebuilt when a
        parameter changes (``PackedMPLayer.ensure``) or on ``refresh_packed()``."""
        dscale = ops.drop_params(self.fe.dropout_p)[1] if self.training else 1.0
        params = tuple(l.weight for l in (*self.fe.net, *self.fn.net))
        # keyed by the DEVICE of the weights as well: ``nn.DataParallel`` (the reference's multi-GPU mode,
        # setup_training.py:1418-1421) replicates a module by shallow-copying its __dict__, so all replicas see this one
        # dict, each from its own device's thread -- every replica gets (and replaces) only its own device's entry
        key = (dscale, ops.FWD_F16, params[0].device.index if params[0].is_cuda else -1)
        cache = self.__dict__.setdefault("_pack_cache", {})
        pk = cache.get(key)
        if pk is None or any(a is not b for a, b in zip(pk.params, params)):
            plist = tuple(q for l in (*self.fe.net, *self.fn.net) for q in (l.weight, l.bias))
            pk = cache[key] = ops.PackedMPLayer(params, self.input_node_size, self.output_node_size, key[0], key[1], plist=plist)
        return pk

    def packed_sets(self):
        """The weight-image sets built so far (``train.TrainStep`` rebuilds those of a whole network together)."""
        return list(self.__dict__.get("_pack_cache", {}).values())

    def refresh_packed(self):
        """Re-pack after an update torch cannot see (a kernel writing the parameters' storage directly)."""
        for pk in self.packed_sets():
            pk.refresh()

    def __repr__(self):
        return f"{self.__class__.__name__}(fe = {self.fe}, \n fn = {self.fn})"


class MPNet(nn.Module):
    """``mp_iters`` MPLayers with generator-/discriminator-specific hooks around them."""

    def __init__(self, num_particles: int, input_node_size: int, mp_iters: int = 2,
                 fe_layers: list = [96, 160, 192], fn_layers: list = [256, 256], fe1_layers: list = None,
                 fn1_layers: list = None, hidden_node_size: int = 32, output_node_size: int = 0,
                 final_activation: str = "", linear_args: dict = {}, mp_args: dict = {},
                 mp_args_first_layer: dict = {}, mask_args: dict = {}):
        super().__init__()
        self.num_particles = num_particles
        self.input_node_size = input_node_size
        self.hidden_node_size = hidden_node_size
        self.output_node_size = output_node_size if output_node_size > 0 else hidden_node_size
        self.mp_iters = mp_iters
        self.final_activation = final_activation
        self.linear_args = linear_args
        self.mask_args = mask_args
        first = {**mp_args, **mp_args_first_layer}
        self._init_mask(**mask_args)
        sizes = [input_node_size] + [hidden_node_size] * (mp_iters - 1) + [self.output_node_size]
        self.mp_layers = nn.ModuleList()
        for k in range(mp_iters):
            fe_k = (fe1_layers or fe_layers) if k == 0 else fe_layers
            fn_k = (fn1_layers or fn_layers) if k == 0 else fn_layers
            self.mp_layers.append(MPLayer(sizes[k], fe_k, fn_k, sizes[k + 1], **(first if k == 0 else mp_args),
                                          **linear_args))

    def _run_layers(self, x, use_mask, mask, labels, njp):
        """The loop over ``mp_layers`` (mpgan/model.py:511-512).  Each layer is told which layer takes its output (a transient
        attribute, set per call: ``nn.DataParallel`` replicas then see their own device's layers)."""
        n = len(self.mp_layers)
        for k, layer in enumerate(self.mp_layers):
            layer.__dict__["_next_layer"] = self.mp_layers[k + 1] if k + 1 < n else None
            x = layer(x, use_mask, mask, labels, njp)
        return x

    def forward(self, x: Tensor, labels: Tensor = None) -> Tensor:
        x = self._pre_mp(x, labels)
        x, use_mask, mask, njp = self._get_mask(x, labels, **self.mask_args)
        x = self._run_layers(x, use_mask, mask, labels, njp)
        x = self._post_mp(x, labels, use_mask, mask, njp)
        return self._finish(x, mask)

    def _finish(self, x, mask):
        """Final activation (reference ``_final_activation``, :533-538) and ``_final_mask``."""
        if self.final_activation == "tanh":
            x = torch.tanh(x)
        elif self.final_activation == "sigmoid":
            x = torch.sigmoid(x)
        return self._final_mask(x, mask, **self.mask_args)

    # hooks
    def _init_mask(self, **mask_args):
        pass

    def _pre_mp(self, x, labels):
        return x

    def _get_mask(self, x, labels, **mask_args):
        return x, False, None, None

    def _post_mp(self, x, labels, use_mask, mask, num_jet_particles):
        return x

    def _final_mask(self, x, mask, **mask_args):
        return x

    def __repr__(self):
        return f"MPLayers = {self.mp_layers})"


def _rank_mask(first_feature: Tensor, labels: Tensor, num_particles: int) -> Tensor:
    """mask_c: the n = int(label * N) lowest-noise particles are real (reference :689-699: rank by
    ``argsort().argsort()``).  The rank of particle i is the number of particles that sort before it -- counted
    directly (one comparison + one sum over an [B, N, N] boolean) instead of two device sorts; equal values are
    ordered by index."""
    n_minus_1 = (labels[:, -1] * num_particles).int() - 1
    xi, xj = first_feature.unsqueeze(2), first_feature.unsqueeze(1)
    idx = torch.arange(first_feature.shape[1], device=first_feature.device)
    before = (xj < xi) | ((xj == xi) & (idx.unsqueeze(0) < idx.unsqueeze(1)).unsqueeze(0))
    rank = before.sum(2)
    return (rank <= n_minus_1.unsqueeze(1)).unsqueeze(2).float()


class MPGenerator(MPNet):
    def __init__(self, lfc: bool = False, lfc_latent_size: int = 128, **mpnet_args):
        super().__init__(**mpnet_args)
        self.lfc = lfc
        if lfc:
            self.lfc_layer = nn.Linear(lfc_latent_size, self.num_particles * self.input_node_size)

    def _init_mask(self, mask_learn: bool = False, mask_learn_sep: bool = False, fmg: list = [64], **mask_args):
        _unsupported(mask_learn=mask_learn, mask_learn_sep=mask_learn_sep)

    def _pre_mp(self, x, labels):
        if self.lfc:
            x = ops.FusedLinearFn.apply(x, self.lfc_layer.weight, self.lfc_layer.bias, False, 0.2, 0.0, False)
            x = x.reshape(x.shape[0], self.num_particles, self.input_node_size)
        return x

    def _get_mask(self, x, labels=None, mask_learn=False, mask_learn_bin=True, mask_learn_sep=False, mask_c=True,
                  mask_fne_np=False, **mask_args):
        if not mask_c:
            return x, False, None, None
        if x.is_cuda:  # one launch (ops.rank_mask) instead of the comparison / sum / compare chain below
            return x, True, ops.rank_mask(x[:, :, 0], labels, self.num_particles).unsqueeze(2), None
        return x, True, _rank_mask(x[:, :, 0], labels, self.num_particles), None

    def _finish(self, x, mask):
        if x.is_cuda and self.final_activation in ops.ACT_CODES and not self.mask_args.get("mask_feat_bin", False):
            # tanh + the (mask - 0.5) column in one launch each way
            return ops.GenTailFn.apply(x, mask, ops.ACT_CODES[self.final_activation])
        return super()._finish(x, mask)

    def generate_into(self, x: Tensor, labels: Tensor, out: Tensor) -> Tensor:
        """``forward`` for callers that own the output rows (``out`` [B, N, F+1], e.g. the second half of a
        discriminator batch) and need no gradient: ``train.TrainStep``'s D step and bulk generation."""
        assert not torch.is_grad_enabled() and x.is_cuda
        x = self._pre_mp(x, labels)
        x, use_mask, mask, njp = self._get_mask(x, labels, **self.mask_args)
        x = self._run_layers(x, use_mask, mask, labels, njp)
        return ops.gen_tail_into(x, mask, ops.ACT_CODES[self.final_activation], out)

    def noise_mask_ok(self) -> bool:
        """The mask is a function of the input noise alone (mask_c on the noise's own first feature: no latent layer in
        front): a caller may draw both in one launch (``ops.normal_noise_masked``) and hand the mask in as ``premask``."""
        return bool(self.mask_args.get("mask_c", True)) and not self.mask_args.get("mask_feat_bin", False) and not self.lfc

    def generate_parts(self, x: Tensor, labels: Tensor, feat_out: Tensor = None, mask_out: Tensor = None, ign_out: Tensor = None,
                       premask=None):
        """``forward`` without gluing the mask column on: (particle features [B, N, F] after the final activation, mask
        [B, N, 1], None).  A discriminator's ``features_parts`` takes them as they are -- no mask column to write, to split off
        again and to pad a gradient for (``train.TrainStep``; the modules' ``forward`` keeps the reference's [B, N, F+1]
        tensors).  ``feat_out`` / ``mask_out`` (without gradients): the caller's rows to write into."""
        assert x.is_cuda and self.mask_args.get("mask_c", True) and not self.mask_args.get("mask_feat_bin", False)
        x = self._pre_mp(x, labels)
        if premask is not None:       # (mask [B, N], 1 - mask) drawn with the noise
            mask2d = premask[0]
        else:
            mask2d = ops.rank_mask(x[:, :, 0], labels, self.num_particles, out=None if mask_out is None else mask_out.view(x.shape[0], -1))
        mask = mask2d.unsqueeze(2)
        x = self._run_layers(x, True, mask, labels, None)
        act = ops.ACT_CODES[self.final_activation]
        if feat_out is not None:
            assert not torch.is_grad_enabled()
            return ops.gen_tail_into(x, None, act, feat_out), mask, None
        return ops.GenTailFn.apply(x, None, act), mask, None

    def _final_mask(self, x, mask, mask_feat_bin: bool = False, **mask_args):
        _unsupported(mask_feat_bin=mask_feat_bin)
        return torch.cat((x, mask - 0.5), dim=2) if mask is not None else x

    def __repr__(self):
        lfc_str = f"LFC = {self.lfc_layer},\n" if self.lfc else ""
        return f"{self.__class__.__name__}({lfc_str}MPLayers = {self.mp_layers})"


class MPDiscriminator(MPNet):
    def __init__(self, dea: bool = True, dea_sum: bool = True, fnd: list = [], mask_fnd_np: bool = False,
                 **mpnet_args):
        super().__init__(output_node_size=1 if not dea else 0, **mpnet_args)
        self.dea, self.dea_sum, self.mask_fnd_np = dea, dea_sum, bool(mask_fnd_np)
        if dea:   # (mask_fnd_np: the fraction of real particles is one more input of the final network, :803)
            self.fnd_layer = LinearNet(fnd, input_size=self.hidden_node_size + int(self.mask_fnd_np), output_size=1,
                                       final_linear=True, **self.linear_args)

    def _get_mask(self, x, labels, mask_manual=False, mask_learn=False, mask_learn_sep=False, mask_c=True,
                  mask_fne_np=False, mask_fnd_np=False, **mask_args):
        use_mask = bool(mask_manual or mask_learn or mask_c or mask_learn_sep)
        mask = x[:, :, -1:] + 0.5 if (use_mask or mask_fnd_np) else None   # (:881-883)
        if use_mask:
            x = x[:, :, :-1]
        njp = torch.mean(mask, dim=1) if mask_fne_np else None           # fraction of real particles, [B, 1] (:888)
        if njp is not None and all(l.fused for l in self.mp_layers):
            # the fused layers take the mask as data (its factor on the senders has no gradient there; nothing upstream of a
            # mask column is ever trained: the generator's comes out of a ranking) -- consistently so: the particle count
            # derived from it carries none either, instead of the partial gradient its columns alone would give
            njp = njp.detach()
        return x, use_mask, mask, njp

    def fused_head(self):
        """(weight [1, F], bias, mean-pooling?, sigmoid?, dropout p) when pooling + ``fnd_layer`` + final activation are
        the single-launch head of ``ops.DiscHeadFn`` (``dea`` with an empty ``fnd`` list: the reference default), else None."""
        if (not self.dea or self.mask_fnd_np or len(self.fnd_layer.net) != 1 or not self.fnd_layer.plain
                or self.final_activation not in ("", "sigmoid")):
            return None
        lin = self.fnd_layer.net[0]
        return lin.weight, lin.bias, not self.dea_sum, self.final_activation == "sigmoid", self.fnd_layer.dropout_p

    def features(self, x: Tensor, labels: Tensor = None):
        """The message-passing part of ``forward``: (last layer's node features [B, N, F], mask [B, N, 1] or None)."""
        x = self._pre_mp(x, labels)
        x, use_mask, mask, njp = self._get_mask(x, labels, **self.mask_args)
        x = self._run_layers(x, use_mask, mask, labels, njp)
        return x, (mask if use_mask else None)

    def parts_ok(self) -> bool:
        """Whether ``features_parts`` is ``features`` (the mask is used as a mask only: the reference's default)."""
        a = self.mask_args
        return (bool(a.get("mask_c", True)) and not a.get("mask_fne_np", False) and not a.get("mask_fnd_np", False)
                and not self.mask_fnd_np and not any(l.clabels or l.mask_fne_np for l in self.mp_layers))

    def features_parts(self, x3: Tensor, mask: Tensor, labels: Tensor = None, ignore: Tensor = None):
        """``features`` for callers that hold the particle features [B, N, F] and the mask [B, N, 1] apart (``parts_ok``)."""
        x = self._pre_mp(x3, labels)
        x = self._run_layers(x, True, mask, labels, None)
        return x, mask

    def forward(self, x: Tensor, labels: Tensor = None) -> Tensor:
        head = self.fused_head() if (x.is_cuda and not ops.double_backward_on(x.device)) else None
        if head is None:
            return super().forward(x, labels)
        y, mask = self.features(x, labels)
        w, b, mean, sigmoid, p = head
        return ops.DiscHeadFn.apply(y, mask, w, b, mean, sigmoid, p, self.training).unsqueeze(1)

    def _post_mp(self, x, labels, use_mask, mask, num_jet_particles):
        mean = not (self.dea and self.dea_sum)
        if use_mask:
            x = (x * mask).sum(1)
            if mean:
                x = x / (mask.sum(1) + 1e-12)
        else:
            x = x.mean(1) if mean else x.sum(1)
        if not self.dea:
            return x
        if self.mask_fnd_np:
            x = torch.cat((num_jet_particles, x), dim=1)   # (:826-827: the reference needs mask_fne_np on as well)
        return self.fnd_layer(x.contiguous())

    def __repr__(self):
        dea_str = f",\nFND = {self.fnd_layer}" if self.dea else ""
        return f"{self.__class__.__name__}(MPLayers = {self.mp_layers}{dea_str})"

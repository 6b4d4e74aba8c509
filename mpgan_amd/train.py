"""One G+D training iteration on the fused path, host side.

Mirrors the reference's ``train_D`` / ``train_G`` (train.py:398-523) for the default recipe:
LSGAN loss (``--loss ls``, train.py:368-370, :471-472), RMSprop (setup_training.py:1511-1513),
``num_critic = num_gen = 1``, generator noise ~ N(0, sd=0.2) sampled on the device every step
(train.py:100-141), D in train mode (dropout on) in both sub-steps, G in eval mode in the D step.

Two pieces of work the reference does and throws away are not done (results-neutral, SURVEY.md
section 3.1): the D step does not back-propagate into G (its gradients are zeroed before use,
train.py:495), and the G step does not form D's weight gradients (zeroed at train.py:420).

MI355X specifics: parameters and gradients of each network live in ONE flat buffer (a single
fused RMSprop launch, a single RCCL all-reduce per network per step), and the whole iteration is
captured into hipGraphs (three segments, split at the two gradient all-reduces) and replayed.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
from torch import nn

from . import _lib, ops, dist as mdist
from .mpgan import MPGenerator, MPDiscriminator

LR = {  # setup_training.py:848-872 (lr_disc, lr_gen) per jet type for model = mpgan
    "g": (3e-5, 1e-5), "t": (6e-5, 2e-5), "q": (1.5e-5, 0.5e-5),
}


def default_mpgan(num_particles: int = 30, disc_dropout: float = 0.5, gen_dropout: float = 0.0, device="cuda"):
    """MPGenerator / MPDiscriminator exactly as ``setup_training.setup_mpgan`` builds them from the
    reference's default arguments (setup_training.py:1195-1293, defaults :415-548)."""
    def lin(p):
        return {"leaky_relu_alpha": 0.2, "dropout_p": p, "batch_norm": False, "spectral_norm": False}
    mp_args = {"pos_diffs": False, "all_ef": False, "coords": "polarrel", "delta_coords": False, "delta_r": False,
               "int_diffs": False, "clabels": 0, "mask_fne_np": False, "fully_connected": True, "num_knn": 10,
               "self_loops": True, "sum": True}
    common = {"num_particles": num_particles, "hidden_node_size": 32, "fe_layers": [96, 160, 192],
              "fn_layers": [256, 256], "fn1_layers": None}
    mask_args = {"mask_feat": False, "mask_feat_bin": False, "mask_weights": False, "mask_manual": False,
                 "mask_exp": False, "mask_real_only": False, "mask_learn": False, "mask_learn_bin": True,
                 "mask_learn_sep": False, "fmg": [64], "mask_disc_sep": False, "mask_fnd_np": False,
                 "mask_c": True, "mask_fne_np": False}
    G = MPGenerator(mp_iters=2, fe1_layers=None, final_activation="tanh", output_node_size=3, input_node_size=32,
                    lfc=False, lfc_latent_size=128, **common, mp_args=dict(mp_args),
                    mp_args_first_layer={"clabels": 0}, linear_args=lin(gen_dropout), mask_args=dict(mask_args))
    D = MPDiscriminator(mp_iters=2, fe1_layers=None, final_activation="sigmoid", input_node_size=3, dea=True,
                        dea_sum=True, fnd=[], mask_fnd_np=False, **common, mp_args=dict(mp_args),
                        mp_args_first_layer={"clabels": 0, "all_ef": False}, linear_args=lin(disc_dropout),
                        mask_args=dict(mask_args))
    return G.to(device), D.to(device)


def default_gapt(num_particles: int = 30, disc_dropout: float = 0.5, gen_dropout: float = 0.0, device="cuda",
                 use_isab: bool = False):
    """GAPT_G / GAPT_D as ``setup_training.setup_gapt`` builds them from the reference's defaults
    (setup_training.py:1296-1347; 4 / 2 SAB layers, 4 heads, embed 64: :552-581)."""
    from .gapt import GAPT_G, GAPT_D

    def lin(p):
        return {"leaky_relu_alpha": 0.2, "dropout_p": p, "batch_norm": False, "spectral_norm": False}
    common = {"num_particles": num_particles, "num_heads": 4, "embed_dim": 64, "sab_fc_layers": [],
              "use_mask": True, "use_isab": use_isab, "num_isab_nodes": 10}
    G = GAPT_G(sab_layers=4, output_feat_size=3, final_fc_layers=[], dropout_p=gen_dropout, layer_norm=False,
               **common, linear_args=lin(gen_dropout))
    D = GAPT_D(sab_layers=2, input_feat_size=3, final_fc_layers=[], dropout_p=disc_dropout, layer_norm=False,
               **common, linear_args=lin(disc_dropout))
    return G.to(device), D.to(device)


LR_GAPT = (1.5e-4, 0.5e-4)  # setup_training.py:856-857, :869-870


class FlatParams:
    """All parameters of a module re-pointed into one flat fp32 buffer, with a flat gradient buffer
    whose views are pre-installed as ``p.grad`` (autograd accumulates into them in place) and a
    flat RMSprop state.  state_dict() keys/shapes of the module are untouched."""

    def __init__(self, module: nn.Module):
        ps = [p for p in module.parameters()]
        self.module = module
        n = sum(p.numel() for p in ps)
        dev = ps[0].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.sq = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in ps:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k
        self.n = n

    def zero_grad(self):
        self.grad.zero_()

    def rmsprop(self, lr: float, alpha: float = 0.99, eps: float = 1e-8, gscale: float = 1.0):
        _lib.check(_lib.lib().mpg_rmsprop(C.c_void_p(self.flat.data_ptr()), C.c_void_p(self.grad.data_ptr()),
                                          C.c_void_p(self.sq.data_ptr()), self.n, lr, alpha, eps, gscale,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mpg_rmsprop")


def _set_requires_grad(module: nn.Module, flag: bool):
    for p in module.parameters():
        p.requires_grad_(flag)


class TrainStep:
    """G+D iteration (train.py:829-878 body) with static buffers, optional hipGraph replay and an
    optional process group for data-parallel gradient averaging (RCCL over xGMI)."""

    def __init__(self, G: nn.Module, D: nn.Module, batch_size: int, num_particles: int, latent: int = 32,
                 lr_disc: float = 3e-5, lr_gen: float = 1e-5, noise_std: float = 0.2, use_graphs: bool = True,
                 process_group=None, world_size: int = 1, batch_real_fake: bool = True):
        self.G, self.D = G, D
        # train_D evaluates D on the real and on the generated batch (train.py:432-447).  D has no cross-sample
        # coupling (no batch norm), so one pass over the concatenated 2B jets gives the same outputs and the same
        # summed gradients as the reference's two passes -- with half the launches and twice the workgroups per
        # launch (jets have different multiplicities; more workgroups than CUs evens that out).
        self.batch_real_fake = batch_real_fake
        self.B, self.N, self.latent = batch_size, num_particles, latent
        self.lr_disc, self.lr_gen, self.noise_std = lr_disc, lr_gen, noise_std
        self.pg, self.world = process_group, world_size
        dev = next(G.parameters()).device
        self.dev = dev
        self.fG, self.fD = FlatParams(G), FlatParams(D)
        self.data = torch.zeros(batch_size, num_particles, 4, device=dev)
        self.labels = torch.zeros(batch_size, 1, device=dev)
        self._target = torch.cat([torch.ones(batch_size, device=dev), torch.zeros(batch_size, device=dev)])
        self.D_loss = torch.zeros((), device=dev)
        self.G_loss = torch.zeros((), device=dev)
        self.use_graphs = use_graphs
        self._graphs = None
        self.fixed_noise = None  # tests: (noise_D, noise_G) used instead of fresh samples

    # -- the three segments between collectives ------------------------------------------------
    def _noise(self, which: int = 0):
        if self.fixed_noise is not None:
            return self.fixed_noise[which]
        return torch.empty(self.B, self.N, self.latent, device=self.dev).normal_(0.0, self.noise_std)

    def _seg_D(self):  # train_D up to and including backward (train.py:419-460)
        # parameter gradients are added straight into the flat buffers (no AccumulateGrad kernel per parameter)
        ops.OPTIONS["grad_into_param"] = True
        ops.bump_seed(self.dev)
        self.D.train(); self.G.eval()
        self.fD.zero_grad()
        _set_requires_grad(self.D, True)
        with torch.no_grad():
            fake = self.G(self._noise(0), self.labels)
        if self.batch_real_fake:
            out = self.D(torch.cat([self.data, fake], 0), torch.cat([self.labels, self.labels], 0))
            # mean over the real half of (out - 1)^2 + mean over the generated half of out^2, without slicing
            loss = ((out - self._target.reshape(out.shape)) ** 2).sum() / self.B
        else:
            out_r = self.D(self.data.clone(), self.labels)
            out_f = self.D(fake, self.labels)
            loss = ((out_r - 1.0) ** 2).mean() + (out_f ** 2).mean()
        self._backward(loss)
        self.D_loss.copy_(loss.detach())

    @staticmethod
    def _backward(loss):
        """loss.backward() with the stand-alone Linear layers' weight gradients collected and issued as grouped
        launches that add straight into the flat gradient buffers."""
        ops.DEFERRED_WGRAD = ops.WgradBatch()
        try:
            loss.backward()
            ops.DEFERRED_WGRAD.flush()
        finally:
            ops.DEFERRED_WGRAD = None

    @staticmethod
    def _refresh_packed(module: nn.Module):
        # mpg_rmsprop wrote the parameters behind autograd's back: rebuild the layers' cached weight images
        for m in module.modules():
            if hasattr(m, "refresh_packed"):
                m.refresh_packed()

    def _seg_G(self):  # D_optimizer.step() (train.py:461) + train_G up to backward (:494-520)
        self.fD.rmsprop(self.lr_disc, gscale=1.0 / self.world)
        self._refresh_packed(self.D)
        self.G.train()
        self.fG.zero_grad()
        _set_requires_grad(self.D, False)
        fake = self.G(self._noise(1), self.labels)
        out = self.D(fake, self.labels)
        loss = ((out - 1.0) ** 2).mean()
        self._backward(loss)
        _set_requires_grad(self.D, True)
        self.G_loss.copy_(loss.detach())

    def _seg_end(self):  # G_optimizer.step() (train.py:521)
        self.fG.rmsprop(self.lr_gen, gscale=1.0 / self.world)
        self._refresh_packed(self.G)
        ops.OPTIONS["grad_into_param"] = False

    def _allreduce(self, flat: FlatParams):
        mdist.allreduce_sum_(flat.grad, self.pg, self.world)  # sum; 1/world is folded into rmsprop

    def _eager(self):
        self._seg_D(); self._allreduce(self.fD)
        self._seg_G(); self._allreduce(self.fG)
        self._seg_end()

    def capture(self, warmup: int = 3):
        """Warm up eagerly on a side stream, then capture the three segments into hipGraphs."""
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        graphs = []
        pool = None
        # one graph per segment between collectives; without a process group the whole iteration is one graph
        groups = [(self._seg_D,), (self._seg_G,), (self._seg_end,)] if (self.world > 1 or os.environ.get("MPG_SPLIT_GRAPHS")) else \
                 [(self._seg_D, self._seg_G, self._seg_end)]
        for segs in groups:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                for seg in segs:
                    seg()
            pool = g.pool()
            graphs.append(g)
        self._graphs = graphs

    def set_batch(self, data: torch.Tensor, labels: torch.Tensor):
        self.data.copy_(data, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)

    def step(self):
        if self.use_graphs and self._graphs is None:
            self.capture()
        if not self.use_graphs:
            self._eager()
            return
        if len(self._graphs) == 1:
            self._graphs[0].replay()
            return
        gD, gG, gE = self._graphs
        gD.replay(); self._allreduce(self.fD)
        gG.replay(); self._allreduce(self.fG)
        gE.replay()

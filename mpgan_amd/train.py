"""One G+D training iteration on the fused path, host side.

Mirrors the reference's ``train_D`` / ``train_G`` (train.py:398-523): losses ``ls`` (default, train.py:368-370,
:471-472), ``og``, ``w``, ``hinge`` (``calc_D_loss`` :331-395, ``calc_G_loss`` :465-476), optimizers RMSprop
(default), Adam, Adadelta as ``setup_training.optimizers`` builds them (setup_training.py:1511-1523),
``num_critic = num_gen = 1``, generator noise ~ N(0, sd=0.2) sampled on the device every step
(train.py:100-141), D in train mode (dropout on) in both sub-steps, G in eval mode in the D step.
The gradient penalty (train.py:286-324, ``--gp``) needs a second derivative through D.  The fused ops are first-order
only (``once_differentiable``), so the penalty's own pass D(interpolated) takes the double-backward route
(``ops.double_backward_route``: every product an ``ops.MatMulFn`` on the HIP GEMM, the rest ATen) while D(real) and
D(generated) stay on the fused kernels; both discriminators (``MPDiscriminator``, ``GAPT_D``) have that route.

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


def default_mpgan(num_particles: int = 30, disc_dropout: float = 0.5, gen_dropout: float = 0.0, device="cuda",
                  loss: str = "ls", spectral_norm_gen: bool = False, spectral_norm_disc: bool = False,
                  batch_norm_gen: bool = False, batch_norm_disc: bool = False):
    """MPGenerator / MPDiscriminator exactly as ``setup_training.setup_mpgan`` builds them from the
    reference's default arguments (setup_training.py:1195-1293, defaults :415-548); ``loss`` picks D's final
    activation as :1250 does (none for ``w`` / ``hinge``, sigmoid otherwise); the four normalisation switches are
    ``--spectral-norm-gen/-disc`` and ``--batch-norm-gen/-disc`` (:254-262 -> linear_args, :1207-1223)."""
    def lin(p, bn=False, sn=False):
        return {"leaky_relu_alpha": 0.2, "dropout_p": p, "batch_norm": bn, "spectral_norm": sn}
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
                    mp_args_first_layer={"clabels": 0}, linear_args=lin(gen_dropout, batch_norm_gen, spectral_norm_gen),
                    mask_args=dict(mask_args))
    D = MPDiscriminator(mp_iters=2, fe1_layers=None, final_activation="" if loss in ("w", "hinge") else "sigmoid",
                        input_node_size=3, dea=True,
                        dea_sum=True, fnd=[], mask_fnd_np=False, **common, mp_args=dict(mp_args),
                        mp_args_first_layer={"clabels": 0, "all_ef": False},
                        linear_args=lin(disc_dropout, batch_norm_disc, spectral_norm_disc), mask_args=dict(mask_args))
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


OPTIMIZERS = ("rmsprop", "adam", "adadelta")


class FlatParams:
    """All parameters of a module re-pointed into one flat fp32 buffer, with a flat gradient buffer
    whose views are pre-installed as ``p.grad`` (autograd accumulates into them in place) and the flat
    optimiser state.  state_dict() keys/shapes of the module are untouched.

    ``optimizer``: "rmsprop" (torch.optim.RMSprop defaults), "adam" (weight_decay 5e-4, betas as given) or
    "adadelta" -- the three choices of ``setup_training.optimizers`` (setup_training.py:1511-1523), each ONE fused
    launch over the flat buffer.  ``state_dict()`` / ``load_state_dict()`` speak the matching ``torch.optim``
    class's own format, so the reference's ``*_optim_<epoch>.pt`` files (train.py:534-535, reloaded at
    setup_training.py:1525-1535) are read and written as they are."""

    def __init__(self, module: nn.Module, optimizer: str = "rmsprop", betas=(0.9, 0.999), weight_decay: float = 5e-4):
        if optimizer not in OPTIMIZERS:
            raise ValueError(f"optimizer must be one of {OPTIMIZERS}, got {optimizer!r}")
        # Only what is TRAINED goes into the flat buffers: frozen parameters (spectral norm's power-iteration vectors,
        # requires_grad = False) are never stepped by the reference's optimizers either -- filtered out by requires_grad
        # or left with grad None (setup_training.py:1500-1523) -- and Adam's weight decay would otherwise move them.
        all_ps = list(module.parameters())
        self._trained_idx = [i for i, p in enumerate(all_ps) if p.requires_grad]   # positions in module.parameters()
        self._n_all = len(all_ps)
        ps = [all_ps[i] for i in self._trained_idx]
        if not ps:
            raise ValueError("FlatParams: the module has no trainable parameter")
        self.params = ps
        self.module = module
        self.optimizer, self.betas, self.weight_decay = optimizer, tuple(betas), float(weight_decay)
        n = sum(p.numel() for p in ps)
        dev = ps[0].device
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.sq = torch.zeros(n, device=dev, dtype=torch.float32)       # square_avg / exp_avg_sq
        self.aux = torch.zeros(n, device=dev, dtype=torch.float32) if optimizer != "rmsprop" else None  # exp_avg / acc_delta
        # optimiser steps taken.  Adam needs the count in its arithmetic, so it lives in device memory (a replayed
        # hipGraph must see it advance); the others only report it in state_dict(): host counter.
        self.step_count = torch.zeros((), device=dev, dtype=torch.float32)
        self._host_steps = 0
        self._spans = []
        off = 0
        for p in ps:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            self._spans.append((off, k, tuple(p.shape)))
            off += k
        self.n = n
        self.lr = None  # last learning rate used (for state_dict's param_groups)
        # TrainStep: the device whose dropout / noise seed travels with this optimizer's state dict (``SEED_KEY`` in its
        # parameter group: torch.optim's own load_state_dict carries unknown group keys along untouched)
        self.seed_device = None
        self.seed_rank = 0     # this process's rank in the data-parallel group (TrainStep sets it): saved beside the seed

    def zero_grad(self):
        self.grad.zero_()

    # -- the fused step -----------------------------------------------------------------------------
    def step(self, lr: float, gscale: float = 1.0, zero_grad: bool = False, advance_seed: torch.Tensor = None):
        """One optimiser step over the flat buffer; ``gscale`` multiplies the gradient first; ``zero_grad``: the gradient
        buffer is cleared by the same launch, behind its last use; ``advance_seed``: the device's dropout / noise seed
        (``ops.seed_tensor``) is moved on by ``ops.SEED_STEP`` in the same launch -- ``ops.bump_seed`` of the next iteration
        without a kernel of its own."""
        self.lr = lr
        vp = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L = _lib.lib()
        z = int(zero_grad)
        cnt, add = (None, 0) if advance_seed is None else (vp(advance_seed), ops.SEED_STEP)
        if self.optimizer == "rmsprop":
            _lib.check(L.mpg_rmsprop(vp(self.flat), vp(self.grad), vp(self.sq), self.n, lr, 0.99, 1e-8, gscale, z, cnt, add, st),
                       "mpg_rmsprop")
        elif self.optimizer == "adam":
            _lib.check(L.mpg_adam(vp(self.flat), vp(self.grad), vp(self.aux), vp(self.sq), vp(self.step_count), self.n,
                                  lr, self.betas[0], self.betas[1], 1e-8, self.weight_decay, gscale, z, cnt, add, st), "mpg_adam")
        else:
            _lib.check(L.mpg_adadelta(vp(self.flat), vp(self.grad), vp(self.sq), vp(self.aux), self.n, lr, 0.9, 1e-6,
                                      gscale, z, cnt, add, st), "mpg_adadelta")
        if not (self.flat.is_cuda and torch.cuda.is_current_stream_capturing()):  # a capture records the launch, it does not run it
            self._host_steps += 1

    def note_step(self):
        """Count one step that ran inside a replayed hipGraph (``TrainStep.step`` calls this)."""
        self._host_steps += 1

    @property
    def steps(self) -> float:
        return float(self.step_count) if self.optimizer == "adam" else float(self._host_steps)

    def rmsprop(self, lr: float, alpha: float = 0.99, eps: float = 1e-8, gscale: float = 1.0):
        assert self.optimizer == "rmsprop" and alpha == 0.99 and eps == 1e-8
        self.step(lr, gscale)

    # -- torch.optim-compatible state ---------------------------------------------------------------
    _STATE_KEYS = {"rmsprop": ("square_avg", None), "adam": ("exp_avg_sq", "exp_avg"), "adadelta": ("square_avg", "acc_delta")}
    SEED_KEY = "mpgan_amd_seed"
    SEED_RANK_KEY = "mpgan_amd_seed_rank"

    def _torch_optimizer(self, lr):
        """A torch.optim instance over detached CPU stand-ins of the parameters: the source of truth for the
        ``param_groups`` defaults of this torch version."""
        stand_ins = [torch.zeros(shape) for _, _, shape in self._spans]
        if self.optimizer == "rmsprop":
            return torch.optim.RMSprop(stand_ins, lr=lr)
        if self.optimizer == "adam":
            return torch.optim.Adam(stand_ins, lr=lr, weight_decay=self.weight_decay, betas=self.betas)
        return torch.optim.Adadelta(stand_ins, lr=lr)

    def state_dict(self, lr: float = None, filtered: bool = True) -> dict:
        """What a ``torch.optim.<Optimizer>`` would hold after the same steps.  ``filtered`` (default): built over the
        trainable parameters only, as ``setup_training.optimizers`` does under ``--spectral-norm-gen``
        (``filter(lambda p: p.requires_grad, ...)``, setup_training.py:1500-1509) -- and identical to the unfiltered form for
        a module without frozen parameters.  ``filtered=False``: built over ALL of ``module.parameters()``, frozen ones
        included without state (what the reference's other branch gives a spectral-norm discriminator)."""
        lr = self.lr if lr is None else lr
        sd = self._torch_optimizer(1e-2 if lr is None else lr).state_dict()
        k_sq, k_aux = self._STATE_KEYS[self.optimizer]
        steps = torch.tensor(self.steps, dtype=torch.float32)
        if not filtered:
            sd["param_groups"][0]["params"] = list(range(self._n_all))
        if float(steps) > 0:
            for i, (off, k, shape) in enumerate(self._spans):
                ent = {"step": steps.clone(), k_sq: self.sq[off:off + k].view(shape).clone()}
                if k_aux is not None:
                    ent[k_aux] = self.aux[off:off + k].view(shape).clone()
                sd["state"][i if filtered else self._trained_idx[i]] = ent
        if self.seed_device is not None:
            sd["param_groups"][0][self.SEED_KEY] = ops.get_seed(self.seed_device)
            sd["param_groups"][0][self.SEED_RANK_KEY] = int(self.seed_rank)
        return sd

    def load_state_dict(self, sd: dict):
        """Take over the per-parameter state of a ``torch.optim`` state dict of the matching optimizer class.  Its indices
        are positions either in the list of trainable parameters or in all of ``module.parameters()`` (see
        ``state_dict``): told apart by the length of its parameter group; entries may be missing (parameters that were
        never stepped).  Returns the learning rate recorded in it."""
        k_sq, k_aux = self._STATE_KEYS[self.optimizer]
        state = sd["state"]
        groups = sd.get("param_groups") or [{}]
        n_listed = sum(len(g.get("params", ())) for g in groups)
        if n_listed == len(self._spans) or (n_listed == 0 and len(state) <= len(self._spans)):
            index = list(range(len(self._spans)))
        elif n_listed == self._n_all:
            index = self._trained_idx
        else:
            raise ValueError(f"optimizer state lists {n_listed} parameters; the module has {len(self._spans)} trainable "
                             f"of {self._n_all}")
        known = set(index)
        stray = [k for k in state if int(k) not in known]
        if stray:
            raise ValueError(f"optimizer state has entries for parameters {stray} that are not trained here")
        self.sq.zero_()
        if self.aux is not None:
            self.aux.zero_()
        steps = 0.0
        for (off, k, shape), i in zip(self._spans, index):
            ent = state.get(i, state.get(str(i)))
            if ent is None:
                continue
            if k_sq not in ent:
                raise ValueError(f"state of parameter {i} has no {k_sq!r}: not a torch.optim {self.optimizer} state dict")
            if tuple(ent[k_sq].shape) != shape:
                raise ValueError(f"state of parameter {i} has shape {tuple(ent[k_sq].shape)}, expected {shape}")
            self.sq[off:off + k].copy_(ent[k_sq].reshape(-1))
            if k_aux is not None:
                self.aux[off:off + k].copy_(ent[k_aux].reshape(-1))
            steps = max(steps, float(ent.get("step", 0.0)))
        self.step_count.fill_(steps)
        self._host_steps = int(steps)
        self.lr = groups[0].get("lr", self.lr)
        if self.seed_device is not None and self.SEED_KEY in groups[0]:
            # a resumed run goes on with the noise / dropout stream where the saved one stopped, not from its start -- on the rank
            # that wrote the file.  The reference's checkpoint is ONE file per epoch (train.py:534-535): every other rank of a
            # resumed data-parallel run moves the saved value by its distance in rank (ops.rerank_seed), so that no two ranks
            # share noise or masks behind a resume either.
            ops.set_seed(ops.rerank_seed(int(groups[0][self.SEED_KEY]), int(groups[0].get(self.SEED_RANK_KEY, 0)), self.seed_rank),
                         self.seed_device)
        return self.lr

    def versions(self) -> int:
        """Sum of the autograd version counters of the parameters: changes when anything but the fused optimiser (which
        works on the flat buffer, behind autograd's back) writes them -- ``load_state_dict``, an in-place edit."""
        return sum(p._version for p in self.params)


def _set_requires_grad(flat: "FlatParams", flag: bool):
    for p in flat.params:    # (the trained ones: frozen parameters stay frozen)
        p.requires_grad_(flag)


def _forward_writes_no_state(module: nn.Module) -> bool:
    """No batch norm (running_mean / running_var / num_batches_tracked) and no spectral norm (weight_u / weight_v) anywhere
    in ``module``: its forward reads parameters and buffers only."""
    from .mpgan.model import LinearNet, SpectralNorm
    for m in module.modules():
        if isinstance(m, (SpectralNorm, nn.modules.batchnorm._BatchNorm)) or (isinstance(m, LinearNet) and not m.plain):
            return False
    return True


LOSSES = ("ls", "og", "w", "hinge")


def d_loss(loss: str, out: torch.Tensor, B: int, real: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``calc_D_loss`` (train.py:331-395, without label smoothing / noise) on D's outputs for the CONCATENATED
    batch ``out[:B]`` = real, ``out[B:]`` = generated: D_real_loss + D_fake_loss, each a mean over its B jets.
    Written on the whole vector with a 0/1 target (no slicing: a slice costs a zero-fill and a copy in autograd)."""
    out = out.reshape(-1)
    if real is None:
        real = (torch.arange(2 * B, device=out.device) < B).to(out.dtype)   # 1 for the real half
    if loss == "ls":
        return ((out - real) ** 2).sum() / B
    if loss == "og":  # nn.BCELoss (clamps its logarithms at -100)
        return torch.nn.functional.binary_cross_entropy(out, real, reduction="sum") / B
    sign = 1.0 - 2.0 * real                                             # -1 real, +1 generated
    if loss == "w":
        return (sign * out).sum() / B
    if loss == "hinge":
        return torch.relu(1.0 + sign * out).sum() / B
    raise ValueError(f"loss must be one of {LOSSES}, got {loss!r}")


def g_loss(loss: str, out: torch.Tensor) -> torch.Tensor:
    """``calc_G_loss`` (train.py:465-476) on D's outputs for generated jets."""
    out = out.reshape(-1)
    if loss == "ls":
        return ((out - 1.0) ** 2).mean()
    if loss == "og":   # nn.BCELoss against ones: logarithm clamped at -100, its backward's denominator at 1e-12
        return torch.nn.functional.binary_cross_entropy(out, torch.ones_like(out))
    if loss in ("w", "hinge"):
        return -out.mean()
    raise ValueError(f"loss must be one of {LOSSES}, got {loss!r}")


class TrainStep:
    """G+D iteration (train.py:829-878 body) with static buffers, optional hipGraph replay and an
    optional process group for data-parallel gradient averaging (RCCL over xGMI)."""

    def __init__(self, G: nn.Module, D: nn.Module, batch_size: int, num_particles: int, latent: int = 32,
                 lr_disc: float = 3e-5, lr_gen: float = 1e-5, noise_std: float = 0.2, use_graphs: bool = True,
                 process_group=None, world_size: int = 1, batch_real_fake: bool = True, loss: str = "ls",
                 optimizer: str = "rmsprop", betas=(0.9, 0.999), gp_lambda: float = 0.0,
                 graph_collectives: Optional[bool] = None):
        if loss not in LOSSES:
            raise ValueError(f"loss must be one of {LOSSES}, got {loss!r}")
        self.gp_lambda = float(gp_lambda)
        self.GP = torch.zeros((), device=next(G.parameters()).device)   # last penalty value (the reference's losses["gp"])
        self.fixed_alpha = None   # tests: the interpolation weights [B, 1, 1] instead of fresh uniform samples
        self.G, self.D = G, D
        self.loss = loss
        # train_D evaluates D on the real and on the generated batch (train.py:432-447).  D has no cross-sample
        # coupling (no batch norm), so one pass over the concatenated 2B jets gives the same outputs and the same
        # summed gradients as the reference's two passes -- with half the launches and twice the workgroups per
        # launch (jets have different multiplicities; more workgroups than CUs evens that out).
        # (with batch norm in D the two passes normalise over their own B jets each: kept as the reference's two passes)
        self.batch_real_fake = batch_real_fake and not any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in D.modules())
        self.B, self.N, self.latent = batch_size, num_particles, latent
        self.lr_disc, self.lr_gen, self.noise_std = lr_disc, lr_gen, noise_std
        self.pg, self.world = process_group, world_size
        # RCCL collectives can be captured into a hipGraph like kernels: the two gradient all-reduces then sit INSIDE
        # one graph and a multi-rank iteration is a single replay.  Opt-in (argument, or MPG_GRAPH_COLLECTIVES=1):
        # the default keeps the all-reduces as ordinary calls between three graph segments.
        if graph_collectives is None:
            graph_collectives = os.environ.get("MPG_GRAPH_COLLECTIVES") == "1"
        self.graph_collectives = bool(graph_collectives) and process_group is not None
        dev = next(G.parameters()).device
        self.dev = dev
        self.state = ops.dev_state(dev)
        self.fG = FlatParams(G, optimizer, betas)
        self.fD = FlatParams(D, optimizer, betas)
        if dev.type == "cuda":
            # The generator's noise and every dropout mask come from counter-based streams keyed by the DEVICE seed, which
            # torch.manual_seed does not reach.  Unless the caller has set it (ops.set_seed), derive it here from torch's
            # seed -- the reference's contract: torch.manual_seed(seed), setup_training.py:184 -- and this process's rank,
            # so that data-parallel ranks never share noise or masks; it is saved / restored with G's optimizer state.
            # A second TrainStep on the same device under the same (torch seed, rank) -- a bench's secondary workload, a step
            # re-created for another batch size -- leaves the stream where the first one has brought it; a new torch.manual_seed
            # in between starts it afresh.
            rank = 0
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                rank = torch.distributed.get_rank(process_group)
            key = (torch.initial_seed(), rank)
            if self.state.seed_is_default and self.state.auto_seed_key != key:
                ops.set_seed(ops.derived_seed(*key), dev, _auto=True)
                self.state.auto_seed_key = key
            self.fG.seed_device = dev
            self.fG.seed_rank = rank
            # arrival counters of the sender-chunked launches (ops._tickets): allocated and zeroed HERE, on the default stream, for
            # the step's largest launch (the D step's 2B jets) -- never inside a capture, never first on a side stream
            ops.reserve_tickets(dev, 2 * batch_size * ((num_particles + 31) // 32))
        self.data = torch.zeros(batch_size, num_particles, 4, device=dev)
        self.labels = torch.zeros(batch_size, 1, device=dev)
        self._real = torch.cat([torch.ones(batch_size, device=dev), torch.zeros(batch_size, device=dev)])
        # the discriminator's real + generated batch of the D step: real jets in the first half (set_batch), the
        # generator's output rows land in the second half; labels likewise
        self._dcat = torch.zeros(2 * batch_size, num_particles, 4, device=dev)
        self._labels2 = torch.zeros(2 * batch_size, 1, device=dev)
        # ... and, where both networks take it (``generate_parts`` / ``features_parts``), held APART: particle features, mask and
        # 1 - mask of the 2B jets.  The reference glues the mask on as a fourth column, D splits it off again and autograd pads
        # the gradient back to four columns: three elementwise launches per pass that carry no information
        self._x3 = torch.zeros(2 * batch_size, num_particles, 3, device=dev)
        self._mask2 = torch.zeros(2 * batch_size, num_particles, 1, device=dev)
        self._ign2 = torch.zeros(2 * batch_size, num_particles, device=dev)
        self.parts = (dev.type == "cuda" and hasattr(G, "generate_parts") and hasattr(D, "features_parts") and D.parts_ok()
                      and getattr(G, "use_mask", True) and not getattr(G, "lfc", False)
                      and getattr(G, "mask_args", {}).get("mask_c", True) and os.environ.get("MPG_PARTS", "1") != "0")
        self.D_loss = torch.zeros((), device=dev)
        self.G_loss = torch.zeros((), device=dev)
        self.use_graphs = use_graphs and dev.type == "cuda"
        self._graphs = None
        # The G step's generator forward depends on nothing the D step writes (G's weights, fresh noise, the labels): it is
        # launched at the top of the D step on a second stream and runs BESIDE it -- the edge launches of a batch of 256 jets
        # are one workgroup per CU and as long as their fullest jet, so a fifth of the chip idles at the end of each; work
        # from an independent stream starts on those CUs (measured on one box: 104.3k -> 105.9k jets/s).  Message-passing
        # generators only: the attention blocks are one-wave-per-jet latency chains with no idle CUs to fill, and a second
        # stream beside them cost 4.7 % (650.6k -> 619.8k).  MPG_GEN_AHEAD=0 switches it off.
        # Only for a generator whose forward writes NO module state: with batch norm (running statistics) or spectral norm
        # (power-iteration vectors) the forked train-mode forward would update what the D step's eval-mode call reads on
        # the main stream at the same time -- a race, and the reference's order the other way round (train.py:432-447 before
        # :500-511).
        self.gen_ahead = (dev.type == "cuda" and isinstance(G, MPGenerator) and _forward_writes_no_state(G)
                          and os.environ.get("MPG_GEN_AHEAD", "1") != "0")
        self._side = None
        self._fake_ahead = None
        # where the generator-ahead branch joins: at the end of the D segment when that segment is a hipGraph of its own (a
        # capture must end with every forked stream joined), otherwise only where the G step picks its jets up -- so that the D
        # all-reduce (graph_collectives: inside the one graph; eagerly: between the segments) waits for D's backward and its
        # weight-gradient stream alone, not for the generator's forward beside them
        self._defer_join = False
        self._join_pending = False
        self._fork_late = False
        self.gen_join = None       # ("seg_D" | "seg_G": where the last iteration / capture joined the branch; tests read it)
        # The launches that only produce weight gradients (mpg_edge_dw + reduction, the grouped node-network weight
        # gradients: about a quarter of the step) feed nothing before the optimizer: they run on a second side stream, forked
        # per layer behind mpg_edge_bwd and joined at the end of the backward (before the all-reduce / optimizer step), so
        # that they start on CUs the one-round data-gradient launches leave idle and their launch boundaries stop
        # serialising with the data path.  MPG_WGRAD_SIDE=0 switches it off.
        self.wgrad_side = dev.type == "cuda" and os.environ.get("MPG_WGRAD_SIDE", "1") != "0"
        self._wside = None
        self.bridge = dev.type == "cuda" and os.environ.get("MPG_BRIDGE", "1") != "0"
        # the generator's noise and its jets' masks drawn by one launch (MPG_NOISE_MASK=0: mpg_normal, then mpg_rank_mask)
        self.noise_mask = dev.type == "cuda" and os.environ.get("MPG_NOISE_MASK", "1") != "0"
        self.fixed_noise = None  # tests: (noise_D, noise_G) used instead of fresh samples
        self._seen_versions = (self.fD.versions(), self.fG.versions())
        # the flat gradient buffers start as zeros and every optimizer launch of an iteration leaves them cleared again
        # (FlatParams.step(zero_grad=True)): no memset of its own at the top of train_D / train_G.  A caller that accumulates
        # into the networks' .grad between iterations calls ``mark_grads_dirty()``.
        self._clean = {"D": True, "G": True}

    # -- the three segments between collectives ------------------------------------------------
    def _noise(self, which: int = 0):
        if self.fixed_noise is not None:
            return self.fixed_noise[which]
        if self.dev.type == "cuda":   # counter-based, keyed by the device seed (bumped once per iteration) and the draw's site
            return ops.normal_noise((self.B, self.N, self.latent), self.noise_std, site=which, device=self.dev)
        return torch.empty(self.B, self.N, self.latent, device=self.dev).normal_(0.0, self.noise_std)

    def _noise_masked(self, which: int, mask_out=None, ign_out=None):
        """(noise, premask): the generator's input and -- when its mask depends on nothing else (``noise_mask_ok``) -- the
        jets' masks from the same launch, written into the caller's rows when given; premask None otherwise."""
        if (self.fixed_noise is None and self.noise_mask and self.dev.type == "cuda"
                and getattr(self.G, "noise_mask_ok", lambda: False)() and (self.N * self.latent) % 2 == 0):
            z, m, ig = ops.normal_noise_masked((self.B, self.N, self.latent), self.noise_std, self.labels, site=which, device=self.dev,
                                               mask_out=None if mask_out is None else mask_out.view(self.B, -1),
                                               ignore_out=None if ign_out is None else ign_out.view(self.B, -1))
            return z, (m, ig)
        return self._noise(which), None

    def _fused_ends(self) -> bool:
        """Generator able to write its jets into a caller-owned batch and discriminator whose pooling / last Linear /
        final activation / loss are the single fused head (``ops.disc_head_loss``): the default MPGAN and GAPT
        configurations.  Then an iteration has no autograd node and no elementwise ATen kernel between the last
        message-passing / attention block and the loss, in either direction."""
        return (self.dev.type == "cuda" and hasattr(self.G, "generate_into") and hasattr(self.D, "features")
                and getattr(self.D, "fused_head", lambda: None)() is not None and self.batch_real_fake
                and not self.gp_lambda)

    def _bridge(self) -> bool:
        """GAPT: gen's ``final_fc`` + tanh and disc's ``input_embedding`` as one launch each way (``ops.GenDiscBridgeFn``;
        MPG_BRIDGE=0: the three launches)."""
        if not self.bridge or not hasattr(self.G, "bridge_head") or not hasattr(self.D, "bridge_tail"):
            return False
        h, t = self.G.bridge_head(), self.D.bridge_tail()
        # (final_fc's weight is a view into the flat parameter buffer: the launch reads its rows as float4 -- a generator composed
        # so that the view starts off a 16-byte boundary takes the three launches instead)
        return h is not None and t is not None and ops.bridge_fusable(h[0].shape[1], h[0].shape[0], t[0].shape[0]) \
            and t[0].shape[1] == h[0].shape[0] and h[0].data_ptr() % 16 == 0

    def _head_loss(self, y, mask, gen_step: bool, n_jets: int, loss_out, wgrad: bool):
        w, b, mean, sigmoid, p = self.D.fused_head()
        grads = (w.grad, None if b is None else b.grad) if wgrad else None
        _, dy = ops.disc_head_loss(y, mask, w, b, mean=mean, sigmoid=sigmoid, p_drop=p, training=self.D.training,
                                   loss=self.loss, n_real=self.B, gen_step=gen_step, count=self.B, loss_out=loss_out,
                                   want_dy=True, wgrad=grads)
        return dy

    def _seg_D(self):  # train_D up to and including backward (train.py:419-460)
        # parameter gradients are added straight into the flat buffers (no AccumulateGrad kernel per parameter)
        self.state.grad_into_param = True
        self.state.order_cache = None   # (ops.jet_order: the masks of this iteration live where last iteration's did)
        # (the dropout / noise seed of this iteration was set by the last launch of the iteration before: _seg_end)
        self.D.train()
        # the generator-ahead branch: forked at the top of the segment (MPG_GEN_AHEAD_LATE=0: beside the D step's own generator
        # call), or -- default -- behind the D step's last data-gradient launch, beside the weight-gradient tail (_backward): the
        # lower layer's mpg_edge_dw, its reduction, the grouped weight gradients, the optimizer and the packing are small or
        # latency-bound launches that leave most CUs idle, and two full-chip forwards fill them
        self._fork_late = self.gen_ahead and self.wgrad_side and os.environ.get("MPG_GEN_AHEAD_LATE", "1") != "0"
        if self.gen_ahead and not self._fork_late:
            self._fork_generator()
        self.G.eval()
        if not self._clean["D"]:     # (cleared by the optimizer launch of the iteration before: see _seg_G)
            self.fD.zero_grad()
        self._clean["D"] = False
        _set_requires_grad(self.fD, True)
        try:
            self._seg_D_body()
        finally:
            if self.gen_ahead:
                if self._defer_join:
                    self._join_pending = True
                elif self._side is not None:   # join: everything of this segment is ordered before whatever follows it
                    torch.cuda.current_stream(self.dev).wait_stream(self._side)
                    self.gen_join = "seg_D"

    def _fork_generator(self):
        """train_G's ``gen_data = gen(...)`` (train.py:500-511) on the side stream, in training mode."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        # G's weight images are shared by this forward and the D step's own generator call: built (on first use, or behind an
        # outside write) on THIS stream, before the fork -- built inside the forward they would be written on the side stream
        # while the other call reads them
        self.G.train()
        for m in self.G.modules():
            if hasattr(m, "_packed") and getattr(m, "fused", False):
                m._packed().ensure()
        self._side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self._side):
            if self.parts and self._fused_ends():
                z, pm = self._noise_masked(1)
                self._fake_ahead = self.G.generate_parts(z, self.labels, premask=pm)
            else:
                self._fake_ahead = self.G(self._noise(1), self.labels)
        for t in (self._fake_ahead if isinstance(self._fake_ahead, tuple) else (self._fake_ahead,)):
            if t is not None:
                t.record_stream(torch.cuda.current_stream(self.dev))   # (its consumer, the G step, runs on this stream)

    def _seg_D_body(self):
        if self._fused_ends():
            # real jets sit in the first half of the static batch; the generator writes the second half itself
            B = self.B
            if self.parts and self._bridge():
                with torch.no_grad():
                    z, pm = self._noise_masked(0, self._mask2[B:], self._ign2[B:])
                    pre, _, _ = self.G.generate_rows(z, self.labels, mask_out=self._mask2[B:], ign_out=self._ign2[B:], premask=pm)
                    W1, b1, act1 = self.G.bridge_head()
                # (the generator takes no gradient in train_D: its final_fc enters the launch as plain data)
                head = (W1.detach(), None if b1 is None else b1.detach(), act1)
                y, mask = self.D.features_rows(pre, head, self._x3, self._mask2, self._labels2, ignore=self._ign2)
            elif self.parts:
                with torch.no_grad():
                    z, pm = self._noise_masked(0, self._mask2[B:], self._ign2[B:])
                    self.G.generate_parts(z, self.labels, feat_out=self._x3[B:], mask_out=self._mask2[B:], ign_out=self._ign2[B:], premask=pm)
                y, mask = self.D.features_parts(self._x3, self._mask2, self._labels2, ignore=self._ign2)
            else:
                with torch.no_grad():
                    self.G.generate_into(self._noise(0), self.labels, self._dcat[B:])
                y, mask = self.D.features(self._dcat, self._labels2)
            dy = self._head_loss(y, mask, False, 2 * self.B, self.D_loss, True)
            self._backward(y, dy)
            return
        with torch.no_grad():
            fake = self.G(self._noise(0), self.labels)
        if self.batch_real_fake:
            out = self.D(torch.cat([self.data, fake], 0), torch.cat([self.labels, self.labels], 0))
        else:
            out = torch.cat([self.D(self.data.clone(), self.labels).reshape(-1), self.D(fake, self.labels).reshape(-1)])
        loss = d_loss(self.loss, out, self.B, self._real)
        self.D_loss.copy_(loss.detach())   # (D_real_loss + D_fake_loss: the reference's losses["D"] leaves the penalty out)
        if self.gp_lambda:
            gp = self.gradient_penalty(self.data, fake)
            self.GP.copy_(gp.detach())
            loss = loss + gp
        self._backward(loss)

    def gradient_penalty(self, real: torch.Tensor, fake: torch.Tensor) -> torch.Tensor:
        """``gradient_penalty`` of train.py:286-324:  gp_lambda * mean_b (|| dD(x_b)/dx_b ||_2 - 1)^2  at
        x = a real + (1 - a) generated, a ~ U[0, 1) per jet; D is called without labels, the norm runs over all
        particles and features of a jet (mask column included) with 1e-12 under the root.  D(x) runs on the
        double-backward route, so that the penalty can be back-propagated into D's parameters (a discriminator that is
        plain torch is twice differentiable as it is; the fused kernels decline a second derivative loudly)."""
        B = real.shape[0]
        alpha = self.fixed_alpha if self.fixed_alpha is not None else torch.rand(B, 1, 1, device=real.device)
        x = (alpha * real + (1 - alpha) * fake.detach()).requires_grad_(True)
        with ops.double_backward_route(self.dev):
            prob = self.D(x)
            grads = torch.autograd.grad(prob, x, torch.ones_like(prob), create_graph=True, retain_graph=True)[0]
        norm = torch.sqrt((grads.reshape(B, -1) ** 2).sum(1) + 1e-12)
        return self.gp_lambda * ((norm - 1) ** 2).mean()

    def _backward(self, root, grad=None):
        """root.backward(grad) with the stand-alone Linear layers' weight gradients collected and issued as grouped
        launches that add straight into the flat gradient buffers."""
        self.state.deferred_wgrad = ops.WgradBatch()
        if self.wgrad_side:
            if self._wside is None:
                self._wside = torch.cuda.Stream(device=self.dev)
            self.state.wgrad_stream = self._wside
        try:
            torch.autograd.backward([root], None if grad is None else [grad])
            self.state.deferred_wgrad.flush()
            if self._fork_late:   # (D step only: _seg_G clears the flag before its own backward; never behind a failed backward)
                self._fork_late = False
                self._fork_generator()
        finally:
            self.state.deferred_wgrad = None
            self._fork_late = False
            if self.wgrad_side:
                # join: the weight gradients are complete before whatever follows the backward (all-reduce, optimizer step)
                self.state.wgrad_stream = None
                torch.cuda.current_stream(self.dev).wait_stream(self._wside)
                self.state.wgrad_keep.clear()

    @staticmethod
    def _refresh_packed(module: nn.Module):
        # the fused optimiser wrote the parameters behind autograd's back: rebuild the layers' cached weight images,
        # all layers of the network in as few launches as the pack-job limit allows
        packs = []
        for m in module.modules():
            if hasattr(m, "packed_sets"):
                packs += m.packed_sets()
            elif hasattr(m, "refresh_packed"):
                m.refresh_packed()
        if packs:
            ops.refresh_many(packs)

    def _seg_G(self):  # D_optimizer.step() (train.py:461) + train_G up to backward (:494-520)
        self._fork_late = False
        # optimizer.zero_grad() of the next train_D (train.py:419) rides in this launch: the buffer is cleared behind its last use
        self.fD.step(self.lr_disc, gscale=1.0 / self.world, zero_grad=True)
        self._clean["D"] = True
        self._refresh_packed(self.D)
        self.G.train()
        if not self._clean["G"]:
            self.fG.zero_grad()
        self._clean["G"] = False
        _set_requires_grad(self.fD, False)
        if self._join_pending:       # (the deferred join of the generator-ahead branch: its jets are used from here on)
            self._join_pending = False
            if self._side is not None:
                torch.cuda.current_stream(self.dev).wait_stream(self._side)
                self.gen_join = "seg_G"
        fake, self._fake_ahead = self._fake_ahead, None
        parts = self.parts and self._fused_ends()
        bridge = parts and fake is None and self._bridge()
        if bridge:
            z, pm = self._noise_masked(1)
            fake = self.G.generate_rows(z, self.labels, premask=pm)
        elif fake is None:
            if parts:
                z, pm = self._noise_masked(1)
                fake = self.G.generate_parts(z, self.labels, premask=pm)
            else:
                fake = self.G(self._noise(1), self.labels)
        if bridge:
            y, mask = self.D.features_rows(fake[0], self.G.bridge_head(), None, fake[1], self.labels, ignore=fake[2])
            dy = self._head_loss(y, mask, True, self.B, self.G_loss, False)
            self._backward(y, dy)
        elif parts:
            y, mask = self.D.features_parts(fake[0], fake[1], self.labels, ignore=fake[2])
            dy = self._head_loss(y, mask, True, self.B, self.G_loss, False)
            self._backward(y, dy)
        elif self._fused_ends():
            y, mask = self.D.features(fake, self.labels)
            dy = self._head_loss(y, mask, True, self.B, self.G_loss, False)
            self._backward(y, dy)
        else:
            out = self.D(fake, self.labels)
            loss = g_loss(self.loss, out)
            self._backward(loss)
            self.G_loss.copy_(loss.detach())
        _set_requires_grad(self.fD, True)

    def _seg_end(self):  # G_optimizer.step() (train.py:521)
        # ... and so does the next iteration's seed (what ops.bump_seed at the head of train_D would do in a launch of its own)
        self.fG.step(self.lr_gen, gscale=1.0 / self.world, zero_grad=True, advance_seed=ops.seed_tensor(self.dev))
        self._clean["G"] = True
        self._refresh_packed(self.G)
        self.state.grad_into_param = False

    def mark_grads_dirty(self):
        """Tell the step that something outside it wrote the networks' .grad buffers (they are views of the flat gradient
        buffers): both are cleared HERE, eagerly, on the current stream.  The iteration itself has no memset -- every optimizer
        launch leaves its buffer cleared behind its last use, so ``.grad`` reads as zeros after ``step()`` (log gradient norms
        between ``_seg_D`` / ``_seg_G`` and the following segment) -- and a captured hipGraph contains none either: a host
        flag read at capture time could not reach a replay."""
        self.fD.zero_grad()
        self.fG.zero_grad()
        self._clean = {"D": True, "G": True}

    def _allreduce(self, flat: FlatParams):
        mdist.allreduce_sum_(flat.grad, self.pg, self.world)  # sum; 1/world is folded into the optimiser step

    def _eager(self):
        self._defer_join = True
        try:
            self._seg_D(); self._allreduce(self.fD)
            self._seg_G(); self._allreduce(self.fG)
            self._seg_end()
        finally:
            self._defer_join = False

    def _training_state(self):
        """Everything an iteration changes: parameters, optimiser moments and step counters, the dropout seed, the losses --
        and what a FORWARD changes: the modules' buffers (batch norm's running statistics and batch counter) and frozen
        parameters (spectral norm's power-iteration vectors, written in place by ``SpectralNorm.weight``)."""
        ts = [self.D_loss, self.G_loss, self.GP, ops.seed_tensor(self.dev)]
        for f in (self.fD, self.fG):
            ts += [f.flat, f.sq, f.step_count] + ([f.aux] if f.aux is not None else [])
            ts += list(f.module.buffers()) + [p for p in f.module.parameters() if not p.requires_grad]
        return ts

    def capture(self, warmup: int = 3):
        """Warm up eagerly on a side stream, then capture the three segments into hipGraphs.  The warm-up iterations are
        real ones (kernels get loaded, their LDS limits set, the allocator's pools filled -- none of which may happen
        during a capture): the training state is saved before and put back after them, so capturing -- which ``step``
        does on its first call -- leaves parameters, optimiser state, step counters and the dropout seed where they
        were.  (A run resumed from a checkpoint continues from exactly the state it loaded.)"""
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            if warmup:
                saved = [t.clone() for t in self._training_state()]
                host = (self.fD._host_steps, self.fG._host_steps)
                # torch's generator on this device (the noise, the gradient penalty's interpolation weights and the dropout
                # of the double-backward route draw from it): the warm-up's draws are taken back as well
                rng = torch.cuda.get_rng_state(self.dev) if self.dev.type == "cuda" else None
            for _ in range(warmup):
                self._eager()
            if warmup:
                for t, v in zip(self._training_state(), saved):
                    t.copy_(v)
                self.fD._host_steps, self.fG._host_steps = host
                if rng is not None:
                    torch.cuda.synchronize(self.dev)
                    torch.cuda.set_rng_state(rng, self.dev)
                self._refresh_packed(self.D)
                self._refresh_packed(self.G)
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        graphs = []
        pool = None
        # one graph per segment between collectives; without a process group the whole iteration is one graph
        split = (self.world > 1 or self.pg is not None or os.environ.get("MPG_SPLIT_GRAPHS")) and not self.graph_collectives
        if self.graph_collectives:
            if torch.distributed.get_backend(self.pg) != "nccl":
                raise RuntimeError("graph_collectives needs an nccl (RCCL) process group: only its collectives are stream operations")
            rD, rG = (lambda: self._allreduce(self.fD)), (lambda: self._allreduce(self.fG))
            groups = [(self._seg_D, rD, self._seg_G, rG, self._seg_end)]
        elif split:
            groups = [(self._seg_D,), (self._seg_G,), (self._seg_end,)]
        else:
            groups = [(self._seg_D, self._seg_G, self._seg_end)]
        for segs in groups:
            g = torch.cuda.CUDAGraph()
            self._defer_join = len(segs) > 1     # (D and G segments in ONE graph: the branch may stay open across them)
            try:
                with torch.cuda.graph(g, pool=pool):
                    for seg in segs:
                        seg()
            finally:
                self._defer_join = False
            pool = g.pool()
            graphs.append(g)
        self._graphs = graphs

    def set_batch(self, data: torch.Tensor, labels: torch.Tensor):
        self.data.copy_(data, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self._dcat[:self.B].copy_(self.data)
        self._x3[:self.B].copy_(self.data[..., :3])
        self._mask2[:self.B].copy_(self.data[..., 3:] + 0.5)
        self._ign2[:self.B].copy_(0.5 - self.data[..., 3])
        self._labels2[:self.B].copy_(self.labels)
        self._labels2[self.B:].copy_(self.labels)

    def sync_external_writes(self):
        """Parameters written from outside since the last step -- ``load_state_dict`` of a checkpoint (resume,
        setup_training.py:1406-1416), an in-place edit -- went into the flat buffer (the parameters are views of it), but
        the layers' packed weight images are only rebuilt behind an optimiser step, inside the captured segments: rebuild
        them now, eagerly.  Called by ``step``; cheap when nothing changed (one pass over the version counters)."""
        seen = (self.fD.versions(), self.fG.versions())
        if seen != self._seen_versions:
            if self.dev.type == "cuda":
                self._refresh_packed(self.D)
                self._refresh_packed(self.G)
            self._seen_versions = seen

    def check_range(self):
        """Raise ``FloatingPointError`` if the run has left the numeric range of the fused path (fp16 operands
        with power-of-two scales): a weight image overflowed or a weight / loss is no longer finite.  Synchronises -- call it
        once per epoch, next to the reference's loss logging (train.py:526-540), not per iteration."""
        import math
        st = ops.range_status(self.dev)
        losses = (float(self.D_loss), float(self.G_loss))
        if st or not all(math.isfinite(v) for v in losses):
            raise FloatingPointError(
                f"mpgan_amd: outside the fused path's numeric range (guard word {st}: 1 = |weight x operand scale| > 65504 in an "
                f"fp16 image, 2 = non-finite weight; losses D {losses[0]}, G {losses[1]})")

    def step(self):
        self.sync_external_writes()
        if self.use_graphs and self._graphs is None:
            self.capture()
        if not self.use_graphs:
            self._eager()
            return
        if len(self._graphs) == 1:
            self._graphs[0].replay()
        else:
            gD, gG, gE = self._graphs
            gD.replay(); self._allreduce(self.fD)
            gG.replay(); self._allreduce(self.fG)
            gE.replay()
        self.fD.note_step(); self.fG.note_step()

    # -- optimiser state in the reference's checkpoint format (train.py:534-535, setup_training.py:1525-1535) -----
    def optimizer_state_dicts(self):
        """(D_optimizer.state_dict(), G_optimizer.state_dict()) as the reference's ``save_models`` stores them."""
        return self.fD.state_dict(self.lr_disc), self.fG.state_dict(self.lr_gen)

    def load_optimizer_state_dicts(self, sd_D: dict, sd_G: dict):
        self.fD.load_state_dict(sd_D)
        self.fG.load_state_dict(sd_G)

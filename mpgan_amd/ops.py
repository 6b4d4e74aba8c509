"""torch.autograd front-ends over the C ABI (include/mpgan_amd.h).

Everything here hands raw device pointers + sizes to libmpgan_amd.so on torch's current HIP
stream; torch is used for memory, autograd bookkeeping and a few tiny reductions only.  There is
no CPU path: tensors must live on a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import sys
import threading
from typing import Optional

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import MpgGemm, MpgEdgeFwd, MpgEdgeBwd, MpgEdgeDw, MpgPackJob, MpgChain, MpgReduceJob, check

H1, H2, H3 = 96, 160, 192
# Exact power-of-two operand scales of the forward products (csrc/edge_common.h): fp16 hi/lo pairs keep their 22 bits
# only for |x| >= 2^-3, so the forward images hold SC * W and the activations are split as ascale * x.
SC_W2, SC_W3 = 16.0, 64.0          # = SC_W2 / SC_W3 of csrc/edge_common.h
SC_WN, SC_ACT = 64.0, 8.0          # node-network (mpg_chain) images / activations
TAG_E0, TAG_E1, TAG_E2, TAG_N0, TAG_N1, TAG_N2, TAG_GENERIC = 1, 2, 3, 4, 5, 6, 7

# ------------------------------------------------------------------------------------- state
# Read-only configuration of the arithmetic (set before building models; not step state).
OPTIONS = {
    "skip_masked": True,      # edge forward: skip zero-masked senders (they contribute exactly 0)
    # launches with more workgroups than CUs (the discriminator's real + generated batch) hand their jets out heaviest first
    # (mpg_jet_order): no effect on results, 136 -> 113 us on the 2B launches of the headline configuration
    "lpt_order": True,
    # the node network fn as the epilogue of the fused edge forward (mpg_edge_fwd_fn: one launch instead of two, same bits);
    # False = mpg_edge_fwd followed by mpg_chain
    "fn_epilogue": os.environ.get("MPG_FN_EPILOGUE", "1") != "0",
    # ... and, in the backward, the layer's dx chain and the layer-below's node-network input-gradient chain as the epilogue of the
    # data-gradient kernel (mpg_edge_bwd_fn: one launch instead of three); False = mpg_edge_bwd followed by the mpg_chain calls
    "bwd_epilogue": os.environ.get("MPG_BWD_EPILOGUE", "1") != "0",
    # sender-chunked launches (N > 32 on few jets: SC > 1) keep the node network as the edge forward's epilogue too: the workgroup of
    # a (jet, receiver block) that arrives LAST adds up the chunks' partial sums (mpg_edge_fwd_fn with MpgEdgeFwd.tickets); False =
    # mpg_edge_fwd, a sum over the chunk axis and the mpg_chain calls
    "fn_chunks": os.environ.get("MPG_FN_CHUNKS", "1") != "0",
    # the per-workgroup reduction of mpg_edge_dw inside the layer's grouped split-K reduction launch (mpg_splitk_reduce_group_dw)
    "dw_reduce_grouped": os.environ.get("MPG_DW_REDUCE_GROUPED", "1") != "0",
    # product form of the fused edge forward (MpgEdgeFwd.two_term): 0 = three 16-bit terms in every product; 1 = fe.net.2 on two terms
    # (its input E2 as the one fp16 value that is parked for the backward anyway): -15 % per launch, pre-activations of fe.net.2 to
    # ~1e-4 of their scale instead of ~5e-7.  MPG_FWD_TWO_TERM.
    "fwd_two_term": int(os.environ.get("MPG_FWD_TWO_TERM", "0")),
}
NUM_CUS = 256
# Forward products (they decide LeakyReLU signs) are split as fp16 hi/lo with the operand scales below (~2^-21 per
# product); the fused edge backward works in fp16 as well (csrc/edge_bwd2_impl.h).  Range: |e2| < 1023, node activations
# < 8188, |W * dscale| < 1023 -- see INTEGRATION.md.
FWD_F16 = True


class DeviceState:
    """Everything mutable the fused ops keep between calls, ONE INSTANCE PER DEVICE (SURVEY.md section 8b:
    the reference's ``nn.DataParallel`` drives each device from its own thread, and autograd runs a device's
    backward nodes on that device's worker thread -- so the key is the device, not the calling thread).

    seed            64-bit dropout seed in device memory (a captured hipGraph sees a new value on every replay)
    tags / last_tag dropout-site tag counter of fused-op invocations on this device
    grad_into_param ``FusedMPLayerFn.backward`` adds parameter gradients straight into ``param.grad`` and returns
                    None for them (``train.TrainStep`` switches it on around its backward; leave it off when
                    ``torch.autograd.grad``, hooks or anything else needs the gradients as autograd values)
    deferred_wgrad  a ``WgradBatch`` that collects the stand-alone Linear layers' weight gradients of the
                    backward in flight (``TrainStep`` flushes it as grouped launches), or None
    wgrad_stream    a second stream for the launches that only produce WEIGHT gradients (``mpg_edge_dw`` + its reduction, the
                    grouped node-network weight gradients): nothing reads them before the optimizer, so ``FusedMPLayerFn.backward``
                    forks them off behind ``mpg_edge_bwd`` and goes on with the data gradients; ``TrainStep`` sets it around a
                    backward and joins before the optimizer step / all-reduce.  ``wgrad_keep`` holds every tensor those
                    launches touch until the join (the allocator must not hand their memory to main-stream work meanwhile)
    double_backward modules built while it is set (``double_backward_route``) take the route whose backward is itself
                    differentiable: the gradient penalty's ``torch.autograd.grad(..., create_graph=True)`` (train.py:304-311)
    """

    def __init__(self, index: int):
        self.index = index
        self._seed = None
        self.seed_is_default = True   # nobody has called set_seed on this device yet (TrainStep then seeds it itself)
        self.auto_seed_key = None     # (torch seed, rank) TrainStep last seeded this device from: a second TrainStep under the same pair leaves the stream where it is
        self.tags = itertools.count(1)
        self.last_tag = 0
        self.tag_log = None
        self.sign_tap = None   # tests: a list that collects, per fused MPLayer call with a backward, the tensors its kink decisions can be read from
        self.grad_into_param = False
        self.deferred_wgrad = None
        self.wgrad_stream = None
        self.wgrad_keep = []
        self.double_backward = False
        self._status = None

    @property
    def status(self) -> torch.Tensor:
        """Range guard of the packed weight images: a device word ``mpg_pack_many`` ORs into (1: an fp16 image element beyond
        65504 after scaling, 2: a non-finite weight).  ``range_status`` reads it."""
        if self._status is None:
            dev = torch.device("cuda", self.index) if self.index >= 0 else torch.device("cpu")
            self._status = torch.zeros((1,), dtype=torch.int32, device=dev)
        return self._status

    @property
    def seed(self) -> torch.Tensor:
        if self._seed is None:  # created on first use: the host-side state exists without touching the device
            dev = torch.device("cuda", self.index) if self.index >= 0 else torch.device("cpu")
            self._seed = torch.full((1,), 0x243F6A8885A308D3, dtype=torch.int64, device=dev)
        return self._seed


_states = {}
_states_lock = threading.Lock()


def _dev_index(device) -> int:
    """Device ordinal of ``device`` (a torch.device, a string or an int); a bare "cuda" means the CURRENT device,
    not device 0.  CPU (host-logic tests of ``train.TrainStep`` with toy modules) maps to -1."""
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.type == "cpu":
        return -1
    if d.index is not None:
        return d.index
    return torch.cuda.current_device() if torch.cuda.is_available() else 0


def dev_state(device) -> DeviceState:
    idx = _dev_index(device)
    st = _states.get(idx)
    if st is None:
        with _states_lock:
            st = _states.get(idx)
            if st is None:
                st = _states[idx] = DeviceState(idx)
    return st


def seed_tensor(device) -> torch.Tensor:
    """Per-device 64-bit dropout seed living in device memory.  ``bump_seed`` advances it; call once per
    training iteration."""
    return dev_state(device).seed


def range_status(device="cuda", clear: bool = False) -> int:
    """What the range guard has seen on ``device`` since it was last cleared (0 = nothing; synchronises): bit 0 = a weight times
    its operand scale left fp16's range while the images were packed (the fused forward then multiplies with inf), bit 1 =
    a weight was not finite."""
    st = dev_state(device).status
    v = int(st.item())
    if clear:
        st.zero_()
    return v


def set_seed(value: int, device="cuda", _auto: bool = False):
    """Set the device-resident 64-bit seed that keys every counter-based stream of the fused path on ``device``: the dropout
    masks and -- inside ``train.TrainStep`` -- the generator's input noise.  ``torch.manual_seed`` does NOT reach these
    streams.  ``TrainStep`` seeds a device nobody has seeded from ``torch.initial_seed()`` and its rank (so that
    ``torch.manual_seed(seed)`` keeps the reference's meaning, setup_training.py:184, and ranks draw different noise and
    masks); call this after constructing it to choose the value yourself, with a different value on every rank
    (``dist.rank_seed``)."""
    v = int(value) & 0xFFFFFFFFFFFFFFFF          # the 64-bit pattern as the kernels read it (an int64 tensor holds it signed)
    seed_tensor(device).fill_(v - (1 << 64) if v >= (1 << 63) else v)
    st = dev_state(device)
    st.seed_is_default = _auto   # (TrainStep's own choice does not count as the caller's)
    if not _auto:
        st.auto_seed_key = None


def derived_seed(torch_seed: int, rank: int) -> int:
    """The device seed ``TrainStep`` derives from torch's seed and the data-parallel rank: distinct per (seed, rank)."""
    return (torch_seed * 0x9E3779B97F4A7C15 + (rank + 1) * 0xD1B54A32D192ED03 + 0x243F6A8885A308D3) & 0x7FFFFFFFFFFFFFFF


RANK_TERM = 0xD1B54A32D192ED03   # what one step in rank adds to ``derived_seed``


def rerank_seed(saved: int, saved_rank: int, rank: int) -> int:
    """The seed a checkpoint written by rank ``saved_rank`` hands to rank ``rank``: the saved value moved by the rank term of
    ``derived_seed`` once per rank of distance (mod 2^64).  The reference keeps ONE ``G_optim_<epoch>.pt`` per epoch
    (train.py:534-535), written by one process: loaded as it is, every rank of a resumed data-parallel run would draw rank 0's
    noise and dropout masks from there on.  The saving rank itself gets the saved value back bit for bit (resume == uninterrupted
    run); the others get streams of their own, distinct for every rank."""
    return (int(saved) + (int(rank) - int(saved_rank)) * RANK_TERM) & 0xFFFFFFFFFFFFFFFF


def get_seed(device="cuda") -> int:
    """The seed's current value as an unsigned 64-bit number (synchronises; checkpoints)."""
    return int(seed_tensor(device).item()) & 0xFFFFFFFFFFFFFFFF


SEED_STEP = 0x1E3779B97F4A7C15   # what an iteration adds to the seed


def bump_seed(device="cuda"):
    seed_tensor(device).add_(SEED_STEP)


NOISE_TAG = 0x4E000000   # site tags of ``normal_noise`` (dropout sites stay below 2^27: next_tag)


def normal_noise(shape, std: float, site: int = 0, device="cuda", mean: float = 0.0) -> torch.Tensor:
    """A fresh [*shape] tensor of N(mean, std^2) samples from the counter-based stream of ``mpg_normal``: keyed by the
    device's seed (``bump_seed`` once per iteration) and ``site`` (which draw of the iteration).  The generator's input noise
    (train.py:100-141) inside a captured iteration: no torch generator state to carry through the graph."""
    out = torch.empty(shape, device=device, dtype=torch.float32)
    check(_lib.lib().mpg_normal(_p(out), out.numel(), _p(seed_tensor(out.device)), NOISE_TAG + int(site), mean, std, _stream()),
          "mpg_normal")
    return out


def next_tag(device="cuda", kind: str = "", thr: int = 0) -> int:
    """A fresh dropout-site tag base (8 sites per call) for one fused-op invocation on ``device``.  ``kind`` / ``thr`` name the
    invocation in ``DeviceState.tag_log`` (tests: a list there collects (kind, tag base, thr) of every invocation, in host
    order, so that a whole iteration's keep masks can be dumped site by site with ``dropout_mask``)."""
    st = dev_state(device)
    st.last_tag = (next(st.tags) % (1 << 24)) * 8
    if st.tag_log is not None:
        st.tag_log.append((kind, st.last_tag, int(thr)))
    return st.last_tag


def last_tag(device="cuda") -> int:
    """Tag base of the most recent fused-op invocation on ``device`` (tests: dump that call's dropout masks)."""
    return dev_state(device).last_tag


def drop_params(p: float):
    """thr = round(256 p) (keep <=> byte >= thr); scale = 256/(256-thr)."""
    thr = int(round(256.0 * p))
    if thr <= 0:
        return 0, 1.0
    if thr >= 256:
        raise ValueError("dropout p too close to 1")
    return thr, 256.0 / (256.0 - thr)


TICKET_SLOTS = 32


def _tickets(device, n: int) -> torch.Tensor:
    """Arrival counters for ONE launch whose workgroups hand a reduction to the last arriver: ``n`` zeros that the launch leaves
    zero.  Every call gets the next of ``TICKET_SLOTS`` regions of a per-device buffer: launches that run side by side on two
    streams (the generator-ahead branch beside the D step's own generator call) must not count on the same words, and a
    captured launch keeps the region it was given.  A buffer that has been handed out is NEVER freed (a replayed hipGraph keeps
    raw pointers into it: a larger request gets a new buffer, the old one stays on ``DeviceState``), and a new buffer is zeroed
    and the device synchronised before its first region goes out, whichever stream asks first (``TrainStep`` reserves it for its
    largest launch when it is built)."""
    buf = reserve_tickets(device, n)
    st = dev_state(device)
    st._ticket_i = (st._ticket_i + 1) % TICKET_SLOTS
    return buf[st._ticket_i]


def reserve_tickets(device, n: int) -> torch.Tensor:
    """Make sure the device's ticket buffer has regions of at least ``n`` words (``TrainStep`` calls this when it is built, for its
    largest launch: a capture with ``warmup=0`` then finds the buffer ready)."""
    st = dev_state(device)
    buf = getattr(st, "_tickets", None)
    per = max(4096, n)
    if buf is None or buf.shape[1] < per:
        if buf is not None:
            st._tickets_retired = getattr(st, "_tickets_retired", []) + [buf]
        buf = torch.zeros((TICKET_SLOTS, per), dtype=torch.int32, device=device)
        # the zeros are there for every stream (the launches that use them run on several).  Inside a capture -- a module captured
        # without a TrainStep in front of it -- the fill is a node of that graph, ordered before the captured launches.
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize(device)
        st._tickets, st._ticket_i = buf, 0
    return buf


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor], offset_elems: int = 0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + 4 * offset_elems)


def _chk(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"{name}: expected a float32 tensor on the GPU (got {t.dtype} on {t.device}); "
                           "mpgan_amd has no CPU path")


# ------------------------------------------------------------------------------------- GEMM
def gemm(A, lda, B, ldb, Cout, ldc, M, N, K, *, ak=True, bk=True, a_off=0, b_off=0, c_off=0,
         A2=None, lda2=0, K1=0, bias=None, out_scale=1.0, act=False, alpha=0.2,
         drop=None, gate=None, resid=None, ldr=0, accumulate=False, splitk=1, split_stride=0, f16=False,
         ones_col=False):
    """Thin wrapper of mpg_gemm.  drop = (seed_t, tag, thr, scale) applies forward dropout to C;
    gate = (H, ldh, gate_act, seed_t, tag, thr, scale) multiplies C by d(dropout o act)/dz."""
    g = MpgGemm()
    g.A, g.A2, g.lda, g.lda2, g.K1 = _p(A, a_off), _p(A2), lda, lda2, K1
    g.B, g.ldb = _p(B, b_off), ldb
    g.C, g.ldc = _p(Cout, c_off), ldc
    g.M, g.N, g.K = M, N, K
    g.split_stride = split_stride
    g.bias, g.out_scale, g.act, g.alpha = _p(bias), out_scale, int(act), alpha
    g.seed = None
    if drop is not None and drop[2]:
        g.seed = _p(drop[0]); g.drop_tag, g.drop_thr, g.drop_scale = drop[1], drop[2], drop[3]
    if gate is not None:
        H, ldh, gact, seed_t, tag, thr, scale = gate
        g.gateH, g.ldh, g.gate_act = _p(H), ldh, int(gact)
        if thr:
            g.seed = _p(seed_t); g.gate_tag, g.gate_thr, g.gate_scale = tag, thr, scale
    g.resid, g.ldr = _p(resid), ldr
    g.accumulate = int(accumulate)
    g.f16 = int(f16)
    g.ones_col = int(ones_col)
    check(_lib.lib().mpg_gemm(C.byref(g), int(ak), int(bk), splitk, _stream()), "mpg_gemm")


def linear_fwd(x, W, bias=None, *, act=False, alpha=0.2, drop=None, x2=None, w_col0=0, w_cols=None, resid=None, f16=None):
    """y = drop(act([x | x2] @ W[:, w_col0:w_col0+K]^T + bias)) (+ resid).  x [M,K1], W [N,ldw].  ``f16``: fp16 hi/lo
    operands (the forward's default: ~2^-21 per product, magnitudes below 65504) or bf16 hi/lo (~2^-17, any magnitude)."""
    M, K1 = x.shape
    K2 = 0 if x2 is None else x2.shape[1]
    K = K1 + K2 if w_cols is None else w_cols
    N = W.shape[0]
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    gemm(x, x.stride(0), W, W.stride(0), y, N, M, N, K, ak=True, bk=True, b_off=w_col0,
         A2=x2, lda2=0 if x2 is None else x2.stride(0), K1=K1, bias=bias, act=act, alpha=alpha, drop=drop,
         resid=resid, ldr=0 if resid is None else resid.stride(0), f16=FWD_F16 if f16 is None else f16)
    return y


def linear_bwd_data(dy, W, *, w_col0=0, w_cols=None, gate=None, out=None, accumulate=False, alpha=0.2):
    """dx = (dy @ W[:, w_col0:w_col0+K]) * gate.   dy [M,N], W [N,ldw] -> dx [M,K]."""
    M, N = dy.shape
    K = (W.shape[1] - w_col0) if w_cols is None else w_cols
    if out is None:
        out = torch.empty((M, K), device=dy.device, dtype=torch.float32)
    gemm(dy, dy.stride(0), W, W.stride(0), out, out.stride(0), M, K, N, ak=True, bk=False, b_off=w_col0,
         gate=gate, accumulate=accumulate, alpha=alpha)
    return out


def linear_bwd_weight(dy, x, *, out=None, out_col0=0, out_scale=1.0, bias_out=None):
    """dW[:, out_col0:out_col0+K] = out_scale * dy^T @ x  (dy [M,N], x [M,K]; split-K over M), and, when
    ``bias_out`` [N] is given, bias_out = column sums of dy (a virtual ones column of x, same launch)."""
    M, N = dy.shape
    K = x.shape[1]
    if out is None:
        out = torch.empty((N, K), device=dy.device, dtype=torch.float32)
    hb = int(bias_out is not None)
    tiles = ((N + 63) // 64) * ((K + hb + 63) // 64)
    splitk = max(1, min((M + 255) // 256, (1024 + tiles - 1) // tiles))
    part = torch.empty((splitk, N, K + hb), device=dy.device, dtype=torch.float32)
    gemm(dy, dy.stride(0), x, x.stride(0), part, K + hb, N, K + hb, M, ak=False, bk=False, out_scale=out_scale,
         splitk=splitk, split_stride=N * (K + hb), ones_col=bool(hb))
    check(_lib.lib().mpg_splitk_reduce(_p(part), splitk, N, K, hb, _p(out, out_col0), out.stride(0), _p(bias_out),
                                       _stream()), "mpg_splitk_reduce")
    return out


# (train.TrainStep sets DeviceState.deferred_wgrad around a backward: FusedLinearFn then queues its weight gradients
# there instead of launching one small split-K GEMM + reduction per layer; TrainStep flushes the queue -- grouped
# launches -- before the optimizer.)
# workgroups a single weight-gradient GEMM of a group aims for when choosing its split-K factor
WGRAD_TARGET_WGS = int(__import__("os").environ.get("MPG_WGRAD_TARGET", "512"))


GROUP_MAX = 16   # MPG_GROUP_MAX of include/mpgan_amd.h


class WgradBatch:
    """Weight gradients dW[:, col0:col0+K] = scale * dy^T @ x (+ bias = column sums of dy) collected and issued as
    ONE grouped split-K GEMM launch plus ONE grouped reduction (``linear_bwd_weight`` does one at a time)."""

    def __init__(self):
        self.jobs = []

    def add(self, dy, x, *, out, out_col0=0, out_scale=1.0, bias_out=None, accumulate=False):
        M, N = dy.shape
        K = x.shape[1]
        hb = int(bias_out is not None)
        # (a ones column that would sit alone in a tile column of its own -- K a multiple of 64 -- is folded into the
        # workgroups of tile column 0 by the kernel: csrc/gemm.hip)
        kcols = K if (hb and K % 64 == 0) else K + hb
        tiles = ((N + 63) // 64) * ((kcols + 63) // 64)
        splitk = max(1, min((M + 255) // 256, (WGRAD_TARGET_WGS + tiles - 1) // tiles))
        part = torch.empty((splitk, N, K + hb), device=dy.device, dtype=torch.float32)
        self.jobs.append((dy, x, out, out_col0, out_scale, bias_out, splitk, part, accumulate))

    def flush(self, dw=None):
        """``dw``: an ``MpgEdgeDw`` whose launch ran with ``defer_reduce`` -- its per-workgroup reduction rides in the first group's
        reduction launch (``mpg_splitk_reduce_group_dw``; with no job at all it is that launch alone)."""
        if dw is not None and not self.jobs:
            check(_lib.lib().mpg_splitk_reduce_group_dw(None, 0, C.byref(dw), _stream()), "mpg_splitk_reduce_group_dw")
        for i0 in range(0, len(self.jobs), GROUP_MAX):
            jobs = self.jobs[i0:i0 + GROUP_MAX]
            n = len(jobs)
            gs, sk, rj = (MpgGemm * n)(), (C.c_int * n)(), (MpgReduceJob * n)()
            for i, (dy, x, out, col0, scale, bias_out, splitk, part, acc) in enumerate(jobs):
                M, N = dy.shape
                K = x.shape[1]
                hb = int(bias_out is not None)
                g = gs[i]
                g.A, g.lda, g.B, g.ldb = _p(dy), dy.stride(0), _p(x), x.stride(0)
                g.C, g.ldc = _p(part), K + hb
                g.M, g.N, g.K = N, K + hb, M
                g.split_stride, g.out_scale, g.alpha, g.ones_col = N * (K + hb), scale, 0.2, hb
                sk[i] = splitk
                r = rj[i]
                r.part, r.S, r.N, r.K, r.has_bias = _p(part), splitk, N, K, hb
                r.out, r.ldo, r.bias, r.accumulate = _p(out, col0), out.stride(0), _p(bias_out), int(acc)
            check(_lib.lib().mpg_gemm_wgrad_group(gs, sk, n, _stream()), "mpg_gemm_wgrad_group")
            if dw is not None and i0 == 0:
                check(_lib.lib().mpg_splitk_reduce_group_dw(rj, n, C.byref(dw), _stream()), "mpg_splitk_reduce_group_dw")
            else:
                check(_lib.lib().mpg_splitk_reduce_group(rj, n, _stream()), "mpg_splitk_reduce_group")
        self.jobs = []


def gate(g, H, *, gate_act, alpha, seed_t=None, tag=0, thr=0, scale=1.0):
    M, N = g.shape
    out = torch.empty((M, N), device=g.device, dtype=torch.float32)
    check(_lib.lib().mpg_gate(_p(g), g.stride(0), _p(H), 0 if H is None else H.stride(0), _p(out), N, M, N,
                              int(gate_act), alpha, _p(seed_t), tag, thr, scale, _stream()), "mpg_gate")
    return out


def normal_noise_masked(shape, std: float, labels: torch.Tensor, site: int = 0, device="cuda", mean: float = 0.0,
                        mask_out: Optional[torch.Tensor] = None, ignore_out: Optional[torch.Tensor] = None):
    """``normal_noise`` of a generator's input [B, N, L] AND ``rank_mask`` of its first feature (the jets' masks, mask_c of
    mpgan/model.py:689-699) in one launch: (noise, mask [B, N], 1 - mask [B, N]).  Same values as the two calls."""
    B, N, L = shape
    out = torch.empty(shape, device=device, dtype=torch.float32)
    lab = labels[:, -1]
    if lab.dtype != torch.float32:
        lab = lab.float()
    mask = mask_out if mask_out is not None else torch.empty((B, N), device=device, dtype=torch.float32)
    ign = ignore_out if ignore_out is not None else torch.empty((B, N), device=device, dtype=torch.float32)
    assert mask.is_contiguous() and ign.is_contiguous() and mask.numel() == B * N and ign.numel() == B * N
    check(_lib.lib().mpg_normal_rank_mask(_p(out), B, N, L, _p(seed_tensor(out.device)), NOISE_TAG + int(site), mean, std,
                                          _p(lab), lab.stride(0), _p(mask), _p(ign), _stream()), "mpg_normal_rank_mask")
    return out, mask.view(B, N), ign.view(B, N)


def dropout_mask(rows: int, F: int, tag: int, thr: int, device="cuda"):
    """The {0,1} keep mask [rows, F] of dropout site ``tag`` under the current seed (tests)."""
    out = torch.empty((rows, F), device=device, dtype=torch.float32)
    check(_lib.lib().mpg_dropout_mask(_p(out), rows, F, _p(seed_tensor(device)), tag, thr, _stream()),
          "mpg_dropout_mask")
    return out


# ------------------------------------------------------------------------------------- edge
def pack_weights(W, rows, cols, *, col0=0, transpose=False, scale=1.0, f16=False):
    """bf16 hi/lo fragment image of W[:, col0:col0+cols] (or its transpose)."""
    r, c = (cols, rows) if transpose else (rows, cols)
    MT, QT = (r + 31) // 32, (c + 31) // 32
    img = torch.empty((2 * MT * QT * 2 * 512,), device=W.device, dtype=torch.bfloat16)
    check(_lib.lib().mpg_pack_weights(_p(W, col0), W.stride(0), r, c, int(transpose), scale, int(f16),
                                      C.c_void_p(img.data_ptr()), _stream()), "mpg_pack_weights")
    return img


def _img_elems(rows, cols):
    return 2 * ((rows + 31) // 32) * ((cols + 31) // 32) * 2 * 512   # 16-bit elements of a hi|lo image


PACK_MAX = 24    # MPG_PACK_MAX_JOBS of include/mpgan_amd.h


def refresh_many(packs):
    """Rebuild the weight images of several ``Packed*`` sets (all layers of a network after an optimizer step) with as
    few ``mpg_pack_many`` launches as the job limit allows."""
    todo = [j for pk in packs for j in pk.jobs()]
    for i0 in range(0, len(todo), PACK_MAX):
        chunk = todo[i0:i0 + PACK_MAX]
        jobs = (MpgPackJob * len(chunk))()
        status = C.c_void_p(dev_state(chunk[0][0].device).status.data_ptr())
        for j, (W, rows, cols, tr, scale, f16, rs, sc, img) in zip(jobs, chunk):
            j.W, j.ldw, j.rows, j.cols, j.transpose = _p(W), W.stride(0), rows, cols, tr
            j.scale, j.f16, j.img, j.row_split, j.split_cols = scale, int(f16), C.c_void_p(img.data_ptr()), rs, sc
            j.status = status
        check(_lib.lib().mpg_pack_many(jobs, len(chunk), _stream()), "mpg_pack_many")
    for pk in packs:
        pk._key = pk._current_key()


class PackedMPLayer:
    """All weight images one MPLayer call needs (edge network, node network, their transposes and the stacked
    a|c view of fe.net.0), in persistent buffers, rebuilt by ONE ``mpg_pack_many`` launch.

    ``ensure()`` rebuilds when a parameter's storage or autograd version changed (``optimizer.step()``,
    ``load_state_dict``).  Updates made behind torch's back -- ``train.TrainStep`` runs RMSprop on a flat buffer
    through ``mpg_rmsprop`` -- must call ``refresh()`` themselves (TrainStep does, inside its graph segments).
    """

    def __init__(self, params, F, out, dscale, f16, plist=None):
        W1, W2, W3, V1, V2, V3 = params
        self.plist = plist  # the twelve Parameters (W1, b1, ..., V3, c3) when built by MPLayer: .grad targets
        self.params, self.F, self.out, self.dscale, self.f16 = params, F, out, float(dscale), bool(f16)
        dev = W1.device
        KN = V1.shape[1]   # H3 + F (+ the conditioning columns appended to the node network's input)
        # name: (W, packed rows, packed cols, transpose, scale, f16, row_split, split_cols)
        spec = {
            "W2": (W2, H2, H1, 0, dscale * SC_W2, f16, 0, 0), "W3": (W3, H3, H2, 0, dscale * SC_W3, f16, 0, 0),
            "W3T": (W3, H2, H3, 1, dscale * SC_W3, True, 0, 0), "W2T": (W2, H1, H2, 1, dscale * SC_W2, True, 0, 0),
            "V1": (V1, V1.shape[0], KN, 0, SC_WN, f16, 0, 0), "V2": (V2, V2.shape[0], V2.shape[1], 0, SC_WN, f16, 0, 0),
            "V3": (V3, out, V3.shape[1], 0, SC_WN, f16, 0, 0),
            "V3T": (V3, V3.shape[1], out, 1, 1.0, False, 0, 0), "V2T": (V2, V2.shape[1], V2.shape[0], 1, 1.0, False, 0, 0),
            "V1T": (V1, KN, V1.shape[0], 1, 1.0, False, 0, 0),
            "W1S": (W1, 2 * H1, F, 0, SC_WN, f16, H1, F),        # [a-half ; c-half] of fe.net.0.weight
            "W1ST": (W1, F, 2 * H1, 1, 1.0, False, H1, F),
        }
        self.img = {k: torch.empty((_img_elems(v[1], v[2]),), device=dev, dtype=torch.bfloat16) for k, v in spec.items()}
        self._spec = spec
        self._key = None

    def _current_key(self):
        return tuple((q.data_ptr(), q._version) for q in self.params)

    def jobs(self):
        """(W, rows, cols, transpose, scale, f16, row_split, split_cols, image) per image."""
        return [v + (self.img[k],) for k, v in self._spec.items()]

    def refresh(self):
        refresh_many([self])

    def ensure(self):
        if self._key != self._current_key():
            self.refresh()
        return self

    def ptr(self, name):
        return C.c_void_p(self.img[name].data_ptr())


def chain(M, layers, **kw):
    """mpg_chain front-end.  ``layers``: dicts with img, K, N and optionally bias, nbias, act, drop=(tag,thr,scale),
    gate=(H, act, tag, thr, scale), out (tensor [M, >=N]), wscale (the image holds wscale * W)."""
    c = chain_struct(M, layers, **kw)
    check(_lib.lib().mpg_chain(C.byref(c), _stream()), "mpg_chain")


def chain_struct(M, layers, *, A, lda, K1, A2=None, lda2=0, a_slabs=1, a_slab_stride=0, in_gate=None, in_out=None,
                 alpha=0.2, seed_t=None, f16=False, ascale=1.0):
    """The ``MpgChain`` argument block of ``chain`` (also what ``mpg_edge_fwd_fn`` takes for its epilogue)."""
    c = MpgChain()
    c.A, c.lda, c.K1 = _p(A), lda, K1
    c.A2, c.lda2 = _p(A2), lda2
    c.a_slabs, c.a_slab_stride = a_slabs, a_slab_stride
    if in_gate is not None and in_gate[1]:
        c.in_tag, c.in_thr, c.in_scale = in_gate
    if in_out is not None:
        c.in_out, c.ld_in_out = _p(in_out), in_out.stride(0)
    c.M, c.nlayers, c.alpha, c.seed, c.f16 = M, len(layers), alpha, _p(seed_t), int(f16)
    c.ascale = ascale
    for i, d in enumerate(layers):
        L = c.L[i]
        L.Wimg, L.K, L.N = d["img"], d["K"], d["N"]
        L.wscale = d.get("wscale", 1.0)
        L.bias, L.nbias, L.act = _p(d.get("bias")), d.get("nbias", 0), int(d.get("act", False))
        if d.get("drop") is not None and d["drop"][1]:
            L.drop_tag, L.drop_thr, L.drop_scale = d["drop"]
        if d.get("gate") is not None:
            H, gact, tag, thr, scale = d["gate"]
            L.gateH, L.ldh, L.gate_act = _p(H), H.stride(0), int(gact)
            if thr:
                L.gate_tag, L.gate_thr, L.gate_scale = tag, thr, scale
        if d.get("resid") is not None:
            L.resid, L.ldr = _p(d["resid"]), d["resid"].stride(0)
        if d.get("out") is not None:
            L.out, L.ldo = _p(d["out"]), d["out"].stride(0)
    return c


EDGE_SCALARS = 2          # MPG_EDGE_SCALARS of include/mpgan_amd.h
MAX_CHUNK_SENDERS_ES = 116   # ... with edge scalars (their columns take part of the list's LDS)
PARK_BYTES_PER_BLOCK = 10240   # E2 / dZ2 of one (jet, receiver block, sender) block as fp16 fragments: 160 x 32 x 2 bytes
MAX_CHUNK_SENDERS = 160   # mpg_edge_bwd keeps the list of a chunk's unmasked senders in LDS (csrc/edge_bwd2.hip)


def jet_order(mask2d: torch.Tensor) -> torch.Tensor:
    """int32 [B]: the jets by decreasing number of unmasked particles (``mpg_jet_order``).  The last result per device is kept
    and handed out again for a mask with the same storage, shape and version -- the layers of one network pass share their
    mask.  (A stale order would only change in which order workgroups start, never a result: any permutation is valid.)"""
    st = dev_state(mask2d.device)
    key = (mask2d.data_ptr(), tuple(mask2d.shape), mask2d._version)
    hit = getattr(st, "order_cache", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    B, N = mask2d.shape
    order = torch.empty((B,), device=mask2d.device, dtype=torch.int32)
    check(_lib.lib().mpg_jet_order(_p(mask2d), B, N, C.c_void_p(order.data_ptr()), _stream()), "mpg_jet_order")
    st.order_cache = (key, order)
    return order


def _sender_chunks(B, N, max_chunk=None):
    """Sender chunks per (jet, receiver block).  A workgroup costs a 150 KiB LDS fill (about five senders' worth of
    time) plus its chunk's senders, and the 256 CUs take the workgroups of a launch in rounds: the chunk count that
    minimises  rounds * (fill + senders per chunk), with chunks of at least 8 and at most MAX_CHUNK_SENDERS senders
    (B = 256, N = 30: 1 -- every CU already has a workgroup; B = 16, N = 150: 3 -- 240 workgroups in one round, where
    doubling up to 320 workgroups would run two rounds at 62 % occupancy)."""
    RB = (N + 31) // 32
    wg = B * RB
    max_chunk = max_chunk or MAX_CHUNK_SENDERS
    forced = os.environ.get("MPG_FORCE_SC")   # experiments only (load balance of one-round launches)
    if forced:
        return max(int(forced), -(-N // max_chunk))
    best, best_cost = 1, None
    for sc in range(1, max(1, N // 8) + 1):
        per = -(-N // sc)
        if per > max_chunk:
            continue
        cost = -(-wg * sc // 256) * (5 + per)
        if best_cost is None or cost < best_cost:
            best, best_cost = sc, cost
    if best_cost is None:   # (N > 8 * MAX_CHUNK_SENDERS cannot happen below the kernels' own limits; be safe)
        best = -(-N // max_chunk)
    return best


def dw_workgroups(nblk, N):
    """Workgroups of an ``mpg_edge_dw`` launch: one per CU, more when a workgroup would get over 64 blocks.  The blocks are
    dealt in runs of R consecutive senders (csrc/edge_dw.hip), so the bound is R * ceil(nblk / R / nwg) <= 64."""
    R = max(d for d in range(1, 7) if N % d == 0)
    nruns = nblk // R
    return min(nruns, max(256, -(-nruns // (64 // R))))


def _fn_grad_chain(ctx, gy2):
    """Buffers and the ``MpgChain`` block of the node network's input-gradient chain of the layer behind ``ctx`` (the backward
    of mpgan/model.py:279) for the upstream gradient rows ``gy2`` [B*N, out]: (dz3, dz2, dz1, dh0, chain)."""
    x2, m1, ac, agg, h1, h2, W1, b2, b3, W2, W3, V1, V2, V3 = ctx.saved_tensors[:14]
    pk = ctx.packed
    B, N, F, agg_scale, alpha, thr, dscale, tag = ctx.cfg[:8]
    V, dev = B * N, gy2.device
    n1, n2, out_f = V1.shape[0], V2.shape[0], V3.shape[0]
    dz3 = torch.empty_like(gy2) if thr else gy2
    dz2 = torch.empty((V, n2), device=dev, dtype=torch.float32)
    dz1 = torch.empty((V, n1), device=dev, dtype=torch.float32)
    dh0 = torch.empty((V, V1.shape[1]), device=dev, dtype=torch.float32)  # [dagg | dx(node path) | (conditioning columns)]
    c = chain_struct(V, [dict(img=pk.ptr("V3T"), K=out_f, N=n2, gate=(h2, True, tag + TAG_N1, thr, dscale), out=dz2),
                         dict(img=pk.ptr("V2T"), K=n2, N=n1, gate=(h1, True, tag + TAG_N0, thr, dscale), out=dz1),
                         dict(img=pk.ptr("V1T"), K=n1, N=V1.shape[1], out=dh0)],
                     A=gy2, lda=gy2.stride(0), K1=out_f, in_gate=(tag + TAG_N2, thr, dscale), in_out=dz3 if thr else None,
                     alpha=alpha, seed_t=seed_tensor(dev), f16=False)
    return dz3, dz2, dz1, dh0, c


def _below_chain(prev, dx, x2, thr, alpha, V):
    """``_fn_grad_chain`` of the layer that produced this layer's input (``prev``: its backward context), fed with this layer's
    ``dx`` rows -- or None when that layer cannot take it: not a fused layer's direct output, another dropout mode or slope,
    rows that are not this layer's x, no backward pending there."""
    if prev is None or getattr(prev, "cfg", None) is None or getattr(prev, "packed", None) is None:
        return None
    try:
        saved = prev.saved_tensors
    except RuntimeError:   # (already released: its backward has run)
        return None
    pB, pN, pF, _, palpha, pthr, _, _ = prev.cfg[:8]
    h1, h2, V3 = saved[4], saved[5], saved[13]
    if h1 is None or h2 is None or pB * pN != V or V3.shape[0] != dx.shape[1] or pthr != thr or palpha != alpha:
        return None
    if not any(prev.needs_input_grad):
        return None
    return _fn_grad_chain(prev, dx)


class LayerHandoff:
    """What consecutive fused MPLayers of one network pass to each other around ``FusedMPLayerFn`` (``MPNet`` wires it):
    ``next`` = (PackedMPLayer, fe.net.0.bias) of the layer that will take this layer's output; ``ac_in`` / ``ac_out`` =
    (a | c [B*N, 192], the PackedMPLayer whose W1 image produced it, data pointer of the rows it was computed from, that
    set's parameter key)."""
    __slots__ = ("next", "ac_in", "ac_out", "prev_node")

    def __init__(self, next=None, ac_in=None, prev_node=None):
        # prev_node: the autograd node (a FusedMPLayerFn backward context) that produced this layer's x, or None -- this
        # layer's backward may then run that layer's node-network input-gradient chain in its own launch (mpg_edge_bwd_fn)
        self.next, self.ac_in, self.ac_out, self.prev_node = next, ac_in, None, prev_node


class FusedMPLayerFn(torch.autograd.Function):
    """MPLayer.forward (mpgan/model.py:206-282), default configuration: fully connected, no edge
    features, no conditioning labels; fe = 3 layers [96,160,192], fn = 2 hidden layers + linear."""

    @staticmethod
    def forward(ctx, x, mask, W1, b1, W2, b2, W3, b3, V1, c1, V2, c2, V3, c3, sum_agg, alpha, p_drop, training,
                packed=None, nbr=None, num_knn=0, es=None, nq=0, xfn=None, handoff=None, no_grad=False):
        """``nbr`` (from ``knn_sets``) restricts receiver i's senders to its ``num_knn`` nearest neighbours
        (``fully_connected=False``, mpgan/model.py:319-381); the mean then divides by ``num_knn`` (:267).

        ``handoff`` (a ``LayerHandoff`` or None) couples consecutive layers of a network: ``handoff.ac_in`` are this layer's
        layer-1 node terms a | c already computed by the launch that produced ``x`` (then no projection launch here), and
        ``handoff.next`` is the next layer's ``(PackedMPLayer, b1)``: where this call's edge launch runs the node network as
        its epilogue it appends that layer's projection and leaves the result in ``handoff.ac_out``.

        ``es`` [B, N senders, EDGE_SCALARS, N receivers] with ``nq`` live scalars: the edge features / row-tiled conditioning
        columns of the reference (mpgan/model.py:247-253, :297-313), one scalar per edge each; they multiply the columns
        ``W1[:, 2F : 2F + nq]`` (``Z1 = a_i + c_j + sum_q es_q w_q``) and get a gradient.  ``xfn`` [B, N, F + E]: the node
        network's view of the nodes when conditioning columns are appended to it (:270-276); ``V1`` then has E more columns."""
        _chk(x, "x")
        B, N, F = x.shape
        V = B * N
        dev = x.device
        thr, dscale = drop_params(p_drop) if training else (0, 1.0)
        seed_t = seed_tensor(dev)
        tag = next_tag(dev, "mplayer", thr)
        x2 = x.reshape(V, F)          # a view when x is a feature slice of a contiguous tensor (D's x[..., :-1]) ...
        if x2.stride(1) != 1:
            x2 = x2.contiguous()      # ... every consumer below takes the row stride, only unit column stride matters
        m1 = None if mask is None else mask.reshape(V).contiguous()
        f16 = FWD_F16
        out_f = V3.shape[0]
        if packed is None or packed.dscale != dscale or packed.f16 != f16:  # direct callers: pack for this call
            packed = PackedMPLayer((W1, W2, W3, V1, V2, V3), F, out_f, dscale, f16)
        pk = packed.ensure()

        def dr(site):
            return (tag + site, thr, dscale)

        # layer-1 node terms a | c = x [W1a ; W1c]^T (+ b1 on the a half): handed over by the launch that produced x, or one launch
        ac = None
        if handoff is not None and handoff.ac_in is not None:
            ac_pre, pk_pre, x_ptr, key_pre = handoff.ac_in   # (valid for these very rows and the weight images as they are now)
            if pk_pre is pk and key_pre == pk._key and x_ptr == x2.data_ptr() and tuple(ac_pre.shape) == (V, 2 * H1):
                ac = ac_pre
        if ac is None:
            ac = torch.empty((V, 2 * H1), device=dev, dtype=torch.float32)
            chain(V, [dict(img=pk.ptr("W1S"), K=F, N=2 * H1, bias=b1, nbias=H1, out=ac, wscale=SC_WN)], A=x2, lda=x2.stride(0), K1=F,
                  alpha=alpha, f16=f16, ascale=SC_ACT)
        SC = _sender_chunks(B, N, MAX_CHUNK_SENDERS_ES if es is not None else None)
        aggp = torch.empty((SC, V, H3), device=dev, dtype=torch.float32)
        e = MpgEdgeFwd()
        e.a, e.c, e.ld_ac, e.mask = _p(ac), _p(ac, H1), 2 * H1, _p(m1)
        e.W2img, e.W3img = pk.ptr("W2"), pk.ptr("W3")
        e.b2, e.b3, e.agg = _p(b2), _p(b3), _p(aggp)
        e.B, e.N, e.SC = B, N, SC
        agg_scale = 1.0 if sum_agg else 1.0 / (num_knn if nbr is not None else N)
        e.alpha, e.agg_scale = alpha, agg_scale
        e.nbr = None if nbr is None else C.c_void_p(nbr.data_ptr())
        e.seed, e.tag_base, e.thr, e.dscale = _p(seed_t), tag, thr, dscale
        e.skip_masked = int(OPTIONS["skip_masked"])
        e.f16 = int(f16)
        # product form of the edge layers (MpgEdgeFwd.two_term; edge scalars ride on the three-term kernels only)
        e.two_term = int(OPTIONS["fwd_two_term"]) if es is None else 0
        wq = None
        if es is not None:
            assert 0 < nq <= EDGE_SCALARS and tuple(es.shape) == (B, N, EDGE_SCALARS, N) and W1.shape[1] == 2 * F + nq
            es = es.detach().float().contiguous()
            wq = torch.zeros((EDGE_SCALARS, H1), device=dev, dtype=torch.float32)
            wq[:nq] = W1.detach()[:, 2 * F:2 * F + nq].t()
            e.es, e.wq = _p(es), _p(wq)
        RB = (N + 31) // 32
        order = None
        if m1 is not None and OPTIONS["lpt_order"] and B * RB * SC > NUM_CUS and B + N + 2 + (B + 63) // 64 * (N + 1) <= 16384:
            order = jet_order(m1.view(B, N))
            e.order = C.c_void_p(order.data_ptr())
        # ``no_grad``: the caller's torch.is_grad_enabled() was off (train_D's generator call): inside forward() grad mode is
        # always off and needs_input_grad still says what the PARAMETERS want, so without the flag such a call would write
        # everything a backward reads -- sign words, 10 KB of parked fragments per block, agg, h1, h2 -- for nothing
        need_grad = any(ctx.needs_input_grad) and not no_grad
        if need_grad and B * RB * N * PARK_BYTES_PER_BLOCK > 0x7fffffff:
            # (mpg_edge_fwd / mpg_edge_bwd return -7: the parked fragments are addressed with 32-bit offsets)
            raise RuntimeError(f"FusedMPLayerFn: {B} jets x {N} particles park {B * RB * N * PARK_BYTES_PER_BLOCK / 2**30:.1f} GiB of "
                               f"edge activations for the backward, beyond the kernels' 2 GiB per launch; split the batch "
                               f"(at most {0x7fffffff // (RB * N * PARK_BYTES_PER_BLOCK)} jets per call at this size)")
        sign3 = torch.empty((B * RB * N * 192,), device=dev, dtype=torch.int32) if need_grad else None
        e.sign3 = None if sign3 is None else C.c_void_p(sign3.data_ptr())
        # E2 (the second edge layer's output) parked as fp16 fragments for the backward, which takes LeakyReLU' from its
        # signs instead of recomputing the layer, and for the weight-gradient kernel
        stE2 = torch.empty((B * RB * N, H2, 32), device=dev, dtype=torch.float16) if need_grad else None
        e.stageE2 = None if stE2 is None else C.c_void_p(stE2.data_ptr())
        # node network fn: three chained layers -- as the epilogue of the edge launch where that form covers the call
        # (mpg_edge_fwd_fn: a whole jet per workgroup, the default widths), else one more launch
        xf2 = x2
        if xfn is not None:
            xf2 = xfn.detach().reshape(V, -1)
            if xf2.stride(1) != 1:
                xf2 = xf2.contiguous()
        assert V1.shape[1] == H3 + xf2.shape[1]
        n1, n2 = V1.shape[0], V2.shape[0]
        # (what only a backward reads -- agg for fn.net.0's weight gradient, the hidden activations -- is not written without one)
        h1 = torch.empty((V, n1), device=dev, dtype=torch.float32) if need_grad else None
        h2 = torch.empty((V, n2), device=dev, dtype=torch.float32) if need_grad else None
        y = torch.empty((V, out_f), device=dev, dtype=torch.float32)
        fn_layers = [dict(img=pk.ptr("V1"), K=V1.shape[1], N=n1, bias=c1, act=True, drop=dr(TAG_N0), out=h1, wscale=SC_WN),
                     dict(img=pk.ptr("V2"), K=n1, N=n2, bias=c2, act=True, drop=dr(TAG_N1), out=h2, wscale=SC_WN),
                     dict(img=pk.ptr("V3"), K=n2, N=out_f, bias=c3, act=False, drop=dr(TAG_N2), out=y, wscale=SC_WN)]
        fn_kw = dict(A2=xf2, lda2=xf2.stride(0), alpha=alpha, seed_t=seed_t, f16=f16, ascale=SC_ACT)
        rc = _lib.MPG_FN_NA
        if OPTIONS["fn_epilogue"] and es is None and (SC == 1 or OPTIONS["fn_chunks"]):
            if SC > 1:
                e.tickets = _p(_tickets(dev, B * RB))   # arrival counters of the (jet, receiver block)s: zero, and left zero
            if not need_grad and SC == 1:   # (with sender chunks the partial sums travel through the slabs of agg)
                e.agg = None
            cs = chain_struct(V, fn_layers, A=aggp, lda=H3, K1=H3, **fn_kw)
            cs2 = ac_next = None
            if handoff is not None and handoff.next is not None and out_f % 4 == 0 and out_f <= 32:
                # the next layer's a | c projection of the rows this launch produces, appended to the epilogue
                pk_n, b1_n = handoff.next
                if pk_n.F == out_f and pk_n.f16 == f16:
                    pk_n.ensure()
                    ac_next = torch.empty((V, 2 * H1), device=dev, dtype=torch.float32)
                    cs2 = chain_struct(V, [dict(img=pk_n.ptr("W1S"), K=out_f, N=2 * H1, bias=b1_n, nbias=H1, out=ac_next, wscale=SC_WN)],
                                       A=y, lda=out_f, K1=out_f, alpha=alpha, f16=f16, ascale=SC_ACT)
            rc = _lib.lib().mpg_edge_fwd_fn(C.byref(e), C.byref(cs), None if cs2 is None else C.byref(cs2), _stream())
            if rc != _lib.MPG_FN_NA:
                check(rc, "mpg_edge_fwd_fn")
                agg = aggp[0] if need_grad else None   # (SC > 1: the last workgroup to arrive left the chunks' total in slab 0)
                if cs2 is not None:
                    handoff.ac_out = (ac_next, pk_n, y.data_ptr(), pk_n._key)
        if rc == _lib.MPG_FN_NA:
            e.agg = _p(aggp)
            check(_lib.lib().mpg_edge_fwd(C.byref(e), _stream()), "mpg_edge_fwd")
            agg = aggp[0]
            for q in range(1, SC):   # (in chunk order, slab by slab: the order the epilogue form's last arriver takes -- the two routes
                agg = agg + aggp[q]  #  then agree bit for bit; torch.sum over the chunk axis adds in another order)
            chain(V, fn_layers, A=agg, lda=H3, K1=H3, **fn_kw)
        ctx.packed = pk
        ctx.prev_node = handoff.prev_node if (handoff is not None and need_grad) else None
        ctx.pre = None   # filled by the backward of the layer ABOVE when it has run this layer's input-gradient chain already

        if need_grad and dev_state(dev).sign_tap is not None:
            dev_state(dev).sign_tap.append(dict(B=B, N=N, ac=ac, stE2=stE2, sign3=sign3, h1=h1, h2=h2))
        ctx.save_for_backward(x2, m1, ac, agg, h1, h2, W1, b2, b3, W2, W3, V1, V2, V3, sign3, nbr, stE2, es, wq, xf2, order)
        ctx.cfg = (B, N, F, agg_scale, alpha, thr, dscale, tag, SC, f16, nq)
        return y.reshape(B, N, V3.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x2, m1, ac, agg, h1, h2, W1, b2, b3, W2, W3, V1, V2, V3, sign3, nbr, stE2, es, wq, xf2, order = ctx.saved_tensors
        pk = ctx.packed
        B, N, F, agg_scale, alpha, thr, dscale, tag, SC, f16, nq = ctx.cfg
        nbr_p = None if nbr is None else C.c_void_p(nbr.data_ptr())
        V = B * N
        dev = x2.device
        seed_t = seed_tensor(dev)
        gy2 = gy.reshape(V, -1).contiguous()

        def gt(H, site, act):
            return (H, H.stride(0), act, seed_t, tag + site, thr, dscale)

        need_w = any(ctx.needs_input_grad[2:14])   # False in the G step: D's weights get no update there
        need_x = ctx.needs_input_grad[0]

        # ---- node network fn (mpgan/model.py:279) backward: its input-gradient chain -- already run by the layer above as the
        #      epilogue of its data-gradient launch (ctx.pre, for exactly this upstream gradient), or launched below
        n1, n2, out_f = V1.shape[0], V2.shape[0], V3.shape[0]
        pre, fnb = ctx.pre, None
        ctx.pre = None
        if pre is not None and pre["gy_ptr"] == gy2.data_ptr() and tuple(gy2.shape) == (V, out_f):
            dz3, dz2, dz1, dh0 = pre["dz3"], pre["dz2"], pre["dz1"], pre["dh0"]
        else:
            dz3, dz2, dz1, dh0, fnb = _fn_grad_chain(ctx, gy2)
        dV1 = dV2 = dV3 = dc1 = dc2 = dc3 = None
        wb = WgradBatch()  # all six weight gradients of the layer go out as one grouped launch (below)
        # DeviceState.grad_into_param: add into the parameters' .grad buffers directly and return None for them
        direct = (need_w and dev_state(dev).grad_into_param and pk.plist is not None
                  and all(q.grad is not None and q.grad.is_contiguous() for q in pk.plist))
        if need_w:
            if direct:
                gW1, gb1, gW2, gb2, gW3, gb3, dV1, dc1, dV2, dc2, dV3, dc3 = (q.grad for q in pk.plist)
            else:
                dc3, dc2, dc1 = (torch.empty(t.shape[1], device=dev, dtype=torch.float32) for t in (dz3, dz2, dz1))
                dV3, dV2, dV1 = torch.empty_like(V3), torch.empty_like(V2), torch.empty_like(V1)
            wb.add(dz3, h2, out=dV3, bias_out=dc3, accumulate=direct)
            wb.add(dz2, h1, out=dV2, bias_out=dc2, accumulate=direct)
            wb.add(dz1, agg, out=dV1, out_col0=0, bias_out=dc1, accumulate=direct)
            wb.add(dz1, xf2, out=dV1, out_col0=H3, accumulate=direct)

        # ---- edge network backward: data path, then (if wanted) the weight-gradient pass
        RB = (N + 31) // 32
        nblk = B * RB * N
        dap = torch.empty((SC, V, H1), device=dev, dtype=torch.float32)
        dcp = torch.empty((RB, V, H1), device=dev, dtype=torch.float32)
        stZ2 = None
        if need_w:
            stZ2 = torch.empty((nblk, H2, 32), device=dev, dtype=torch.float16)   # fp16 fragments as the lanes hold them
            gexp = torch.empty((B * RB,), device=dev, dtype=torch.int32)           # gradient-unit exponent per (jet, receiver block)
        e = MpgEdgeBwd()
        e.a, e.c, e.ld_ac, e.mask = _p(ac), _p(ac, H1), 2 * H1, _p(m1)
        e.dagg, e.ld_dagg = _p(dh0), dh0.stride(0)
        e.sign3 = C.c_void_p(sign3.data_ptr())
        e.W2img = pk.ptr("W2")
        e.W3Timg, e.W2Timg = pk.ptr("W3T"), pk.ptr("W2T")
        e.b2 = _p(b2)
        e.da, e.dc = _p(dap), _p(dcp)
        e.stageE2 = C.c_void_p(stE2.data_ptr())
        e.stageZ2 = None if stZ2 is None else C.c_void_p(stZ2.data_ptr())
        e.gexp = None if stZ2 is None else C.c_void_p(gexp.data_ptr())
        e.B, e.N, e.SC = B, N, SC
        e.alpha, e.agg_scale, e.nbr = alpha, agg_scale, nbr_p
        e.seed, e.tag_base, e.thr, e.dscale = _p(seed_t), tag, thr, dscale
        e.f16 = int(f16)
        if order is not None:
            e.order = C.c_void_p(order.data_ptr())
        des = daq = None
        if es is not None:
            des = torch.zeros_like(es)   # (zero-masked senders' rows are not written)
            daq = torch.empty((SC, V, EDGE_SCALARS, H1), device=dev, dtype=torch.float32)
            e.es, e.wq, e.des, e.daq = _p(es), _p(wq), _p(des), _p(daq)
        if fnb is not None:
            check(_lib.lib().mpg_chain(C.byref(fnb), _stream()), "mpg_chain")
        # dx = dx(node path) + [da | dc] [W1a ; W1c]: one chained layer over the stacked transposed view -- as the epilogue of
        # the data-gradient launch where that form covers the call (a whole jet per workgroup), and behind it the node
        # network's input-gradient chain of the layer BELOW, which produced x (its backward then finds its work done)
        dx = cdx = None
        rc = _lib.MPG_FN_NA
        if need_x:
            dx = torch.empty((V, F), device=dev, dtype=torch.float32)
            if SC == 1 and RB == 1:
                cdx = chain_struct(V, [dict(img=pk.ptr("W1ST"), K=2 * H1, N=F, resid=dh0[:, H3:H3 + F], out=dx)],
                                   A=dap, lda=H1, K1=H1, A2=dcp, lda2=H1, alpha=alpha, f16=False)
            if cdx is not None and OPTIONS["bwd_epilogue"] and es is None:
                below = _below_chain(ctx.prev_node, dx, x2, thr, alpha, V)
                rc = _lib.lib().mpg_edge_bwd_fn(C.byref(e), C.byref(cdx), None if below is None else C.byref(below[4]), _stream())
                if rc == _lib.MPG_FN_NA and below is not None:   # (the pair is not covered: the layer alone may be)
                    below = None
                    rc = _lib.lib().mpg_edge_bwd_fn(C.byref(e), C.byref(cdx), None, _stream())
                if rc != _lib.MPG_FN_NA:
                    check(rc, "mpg_edge_bwd_fn")
                    if below is not None:
                        ctx.prev_node.pre = dict(gy_ptr=dx.data_ptr(), keep=dx, dz3=below[0], dz2=below[1], dz1=below[2], dh0=below[3])
        if rc == _lib.MPG_FN_NA:
            check(_lib.lib().mpg_edge_bwd(C.byref(e), _stream()), "mpg_edge_bwd")
        if SC == 1 and RB == 1:
            da, dc, dadc = dap[0], dcp[0], None
        else:
            # the chunks' partial da and the receiver blocks' partial dc, added slab by slab into [da | dc] rows: one launch
            dadc = torch.empty((V, 2 * H1), device=dev, dtype=torch.float32)
            check(_lib.lib().mpg_slab_sums(_p(dap), SC, V * H1, _p(dcp), RB, V * H1, _p(dadc), V, H1, _stream()), "mpg_slab_sums")
            da, dc = dadc[:, :H1], dadc[:, H1:]
        dW1 = db1 = dW2 = db2 = dW3 = db3 = None
        # The weight-gradient launches below feed nothing before the optimizer: with a side stream set (TrainStep) they are
        # forked off here, behind the data-gradient kernel that produced their inputs, and this stream goes straight on
        # to dx and the layers below.  Same launches, same order per parameter: results are bit-identical.
        st_dev = dev_state(dev)
        side = st_dev.wgrad_stream if (need_w and direct and es is None) else None
        main = torch.cuda.current_stream(dev) if side is not None else None
        if need_w:
            nwg = dw_workgroups(nblk, N)
            part = torch.empty((nwg, H3 * H2 + H2 * H1 + H3 + H2), device=dev, dtype=torch.float32)
            if direct:
                dW3, dW2, db3, db2 = gW3, gW2, gb3, gb2
            else:
                dW3, dW2 = torch.empty_like(W3), torch.empty_like(W2)
                db3, db2 = torch.empty_like(b3), torch.empty_like(b2)
            d = MpgEdgeDw()
            d.a, d.c, d.ld_ac, d.mask = _p(ac), _p(ac, H1), 2 * H1, _p(m1)
            d.dagg, d.ld_dagg = _p(dh0), dh0.stride(0)
            d.sign3 = C.c_void_p(sign3.data_ptr())
            d.stageE2, d.stageZ2 = C.c_void_p(stE2.data_ptr()), C.c_void_p(stZ2.data_ptr())
            d.gexp = C.c_void_p(gexp.data_ptr())
            d.part, d.nwg = _p(part), nwg
            d.dW3, d.dW2, d.db3, d.db2, d.accumulate = _p(dW3), _p(dW2), _p(db3), _p(db2), int(direct)
            d.B, d.N = B, N
            d.alpha, d.agg_scale, d.nbr = alpha, agg_scale, nbr_p
            d.seed, d.tag_base, d.thr, d.dscale = _p(seed_t), tag, thr, dscale
            d.f16 = int(f16)
            d.defer_reduce = int(OPTIONS["dw_reduce_grouped"])   # (its reduction rides in the grouped reduction launch below)
            if es is not None:
                d.es, d.wq = _p(es), _p(wq)
            # layer 1 (fe.net.0): a = W1[:, :F] x + b1, c = W1[:, F:] x
            if direct:
                dW1, db1 = gW1, gb1
            else:
                dW1 = torch.empty_like(W1)
                db1 = torch.empty(H1, device=dev, dtype=torch.float32)
            wb.add(da, x2, out=dW1, out_col0=0, bias_out=db1, accumulate=direct)
            wb.add(dc, x2, out=dW1, out_col0=F, accumulate=direct)
            if side is not None:
                # everything these launches read or write stays referenced until TrainStep joins the stream
                st_dev.wgrad_keep.append((ac, m1, dh0, sign3, stE2, stZ2, gexp, part, da, dc, dap, dcp, dadc, x2, xf2, agg, h1, h2,
                                          dz1, dz2, dz3, gy2, nbr, [j[7] for j in wb.jobs]))
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    check(_lib.lib().mpg_edge_dw(C.byref(d), _stream()), "mpg_edge_dw")
                    wb.flush(dw=d if d.defer_reduce else None)
            else:
                check(_lib.lib().mpg_edge_dw(C.byref(d), _stream()), "mpg_edge_dw")
                wb.flush(dw=d if d.defer_reduce else None)
            del stZ2
            if es is not None:   # the columns of the edge scalars: sum over receivers of daq
                dWq = daq.sum((0, 1))[:nq].t()
                if direct:
                    dW1[:, 2 * F:2 * F + nq] += dWq
                else:
                    dW1[:, 2 * F:2 * F + nq] = dWq
            if direct:  # already in .grad: autograd gets nothing to accumulate
                dW1 = db1 = dW2 = db2 = dW3 = db3 = dV1 = dc1 = dV2 = dc2 = dV3 = dc3 = None
        if need_x:
            if rc == _lib.MPG_FN_NA:   # (not done by the data-gradient launch: its own launch)
                if dadc is not None:
                    chain(V, [dict(img=pk.ptr("W1ST"), K=2 * H1, N=F, resid=dh0[:, H3:H3 + F], out=dx)],
                          A=dadc, lda=2 * H1, K1=2 * H1, alpha=alpha, f16=False)
                else:
                    chain(V, [dict(img=pk.ptr("W1ST"), K=2 * H1, N=F, resid=dh0[:, H3:H3 + F], out=dx)],
                          A=dap, lda=H1, K1=H1, A2=dc, lda2=dc.stride(0), alpha=alpha, f16=False)
            dx = dx.reshape(B, N, F)
        dxfn = None
        if len(ctx.needs_input_grad) > 23 and ctx.needs_input_grad[23] and dh0.shape[1] > H3 + F:
            # the conditioning columns appended to the node network's input (mpgan/model.py:270-276): their gradient is the
            # tail of dh0; the x columns of xfn are the same nodes as x, whose node-path gradient is already in dx above
            dxfn = torch.cat((torch.zeros((V, F), device=dev, dtype=torch.float32), dh0[:, H3 + F:]), dim=1).reshape(B, N, -1)
        return (dx, None, dW1, db1, dW2, db2, dW3, db3, dV1, dc1, dV2, dc2, dV3, dc3,
                None, None, None, None, None, None, None, des, None, dxfn, None, None)[:len(ctx.needs_input_grad)]


def _grad_target(t):
    """The .grad buffer a parameter gradient may be added into directly: the leaf's own, or -- for a contiguous
    view of a leaf (a row slice of nn.MultiheadAttention's in_proj_weight / in_proj_bias) -- the matching view of
    the leaf's .grad.  None when there is no such buffer."""
    if t is None:
        return None
    if t.is_leaf:
        return t.grad if (t.grad is not None and t.grad.is_contiguous()) else None
    base = t._base
    if base is None or not base.is_leaf or base.grad is None or not t.is_contiguous() or not base.grad.is_contiguous():
        return None
    return base.grad.as_strided(t.size(), t.stride(), t.storage_offset() - base.storage_offset() + base.grad.storage_offset())


class FusedLinearFn(torch.autograd.Function):
    """Linear -> [LeakyReLU] -> [Dropout] (one LinearNet layer; mpgan/model.py:77-83), optionally ``+ resid`` in the
    same launch (MAB's residual connections, gapt/model.py:131-137; only for a layer without activation)."""

    @staticmethod
    def forward(ctx, x, W, b, act, alpha, p_drop, training, resid=None):
        _chk(x, "x")
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if x2.stride(1) != 1 or x2.stride(0) < shp[-1]:   # (rows of any stride are fine -- a column slice of wider rows, D's x[..., :-1] -- only unit column stride matters)
            x2 = x2.contiguous()
        thr, dscale = drop_params(p_drop) if training else (0, 1.0)
        seed_t = seed_tensor(x.device)
        tag = next_tag(x.device, "linear", thr)
        if resid is not None and act:  # (the backward reads the activation's sign off the saved output)
            raise NotImplementedError("FusedLinearFn: a fused residual needs a layer without activation")
        r2 = None if resid is None else resid.reshape(-1, W.shape[0]).contiguous()
        y = linear_fwd(x2, W, b, act=act, alpha=alpha, drop=(seed_t, tag + TAG_GENERIC, thr, dscale) if thr else None,
                       resid=r2)
        ctx.has_resid = resid is not None
        ctx.save_for_backward(x2, W, y)
        ctx.wparam, ctx.bias = W, b  # (only their .grad buffers are touched, by the deferred weight-gradient path)
        ctx.cfg = (shp, act, alpha, thr, dscale, tag, b is not None)
        return y.reshape(*shp[:-1], W.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x2, W, y = ctx.saved_tensors
        shp, act, alpha, thr, dscale, tag, has_b = ctx.cfg
        g2 = gy.reshape(-1, W.shape[0]).contiguous()
        if act or thr:
            g2 = gate(g2, y, gate_act=act, alpha=alpha, seed_t=seed_tensor(g2.device), tag=tag + TAG_GENERIC,
                      thr=thr, scale=dscale)
        dW = db = None
        want_b = has_b and ctx.needs_input_grad[2]
        gW = gb = None
        st = dev_state(g2.device)
        if ctx.needs_input_grad[1] and st.grad_into_param and st.deferred_wgrad is not None:
            gW, gb = _grad_target(ctx.wparam), _grad_target(ctx.bias) if want_b else None
        if gW is not None and (not want_b or gb is not None):
            # TrainStep: queue dW (+ db) for the grouped launch at the end of the backward; it adds into .grad
            st.deferred_wgrad.add(g2, x2, out=gW, bias_out=gb, accumulate=True)
        elif ctx.needs_input_grad[1]:
            if want_b:
                db = torch.empty(W.shape[0], device=g2.device, dtype=torch.float32)
            dW = linear_bwd_weight(g2, x2, bias_out=db)
        elif want_b:
            db = g2.sum(0)
        dx = linear_bwd_data(g2, W).reshape(shp) if ctx.needs_input_grad[0] else None
        return dx, dW, db, None, None, None, None, (gy if ctx.has_resid else None)


class MatMulFn(torch.autograd.Function):
    """C = A B^T ("nt"), A B ("nn") or A^T B ("tn") on ``mpg_gemm`` -- with a backward that is written in terms of
    ``MatMulFn`` itself, so that it can be differentiated again, to any order.  The three forms are closed under
    differentiation:  nt: dA = G B (nn), dB = G^T A (tn);  nn: dA = G B^T (nt), dB = A^T G (tn);  tn: dA = B G^T (nt),
    dB = A G (nn).  This is the product the double-backward route is made of (``LinearNet._forward_dd``): the gradient
    penalty (train.py:286-324) differentiates D's input gradient once more, which the fused kernels
    (``once_differentiable``) decline."""

    @staticmethod
    def forward(ctx, A, B, form):
        _chk(A, "A"); _chk(B, "B")
        A, B = A.contiguous(), B.contiguous()
        ctx.form = form
        ctx.save_for_backward(A, B)
        if form == "nt":   # (bf16 hi/lo like the other two forms: gradients of any magnitude pass through all three)
            return linear_fwd(A, B, None, f16=False)
        if form == "nn":
            return linear_bwd_data(A, B)
        if form == "tn":
            return linear_bwd_weight(A, B)
        raise ValueError(f"MatMulFn: form must be nt / nn / tn, got {form!r}")

    @staticmethod
    def backward(ctx, G):
        A, B = ctx.saved_tensors
        nA, nB = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        mm = MatMulFn.apply
        if ctx.form == "nt":
            return (mm(G, B, "nn") if nA else None), (mm(G, A, "tn") if nB else None), None
        if ctx.form == "nn":
            return (mm(G, B, "nt") if nA else None), (mm(A, G, "tn") if nB else None), None
        return (mm(B, G, "nt") if nA else None), (mm(A, G, "nn") if nB else None), None


class double_backward_route:
    """``with ops.double_backward_route(device):`` -- modules called inside take the route that can be differentiated
    twice: every product a ``MatMulFn``, everything elementwise plain ATen (whose backward formulas are differentiable
    themselves), the edge matrix materialised as the reference builds it.  First-order-only pieces (the fused
    message-passing and attention kernels, batch norm) are not used there; a configuration that needs one raises."""

    def __init__(self, device="cuda"):
        self.state = dev_state(device)

    def __enter__(self):
        self.prev, self.state.double_backward = self.state.double_backward, True
        return self

    def __exit__(self, *exc):
        self.state.double_backward = self.prev
        return False


def double_backward_on(device) -> bool:
    return dev_state(device).double_backward


class FusedDropoutFn(torch.autograd.Function):
    """Stand-alone inverted dropout on the counter-based mask stream (MAB.dropout, gapt/model.py:132,137)."""

    @staticmethod
    def forward(ctx, x, p_drop, training):
        thr, scale = drop_params(p_drop) if training else (0, 1.0)
        if not thr:
            ctx.cfg = None
            return x
        _chk(x, "x")
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        tag = next_tag(x.device, "dropout", thr) + TAG_GENERIC
        ctx.cfg = (tag, thr, scale, x.shape)
        return gate(x2, None, gate_act=False, alpha=0.0, seed_t=seed_tensor(x.device), tag=tag, thr=thr,
                    scale=scale).reshape(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if ctx.cfg is None:
            return g, None, None
        tag, thr, scale, shp = ctx.cfg
        g2 = g.reshape(-1, shp[-1]).contiguous()
        return gate(g2, None, gate_act=False, alpha=0.0, seed_t=seed_tensor(g.device), tag=tag, thr=thr,
                    scale=scale).reshape(shp), None, None


def _attn_struct(q, k, v, ignore, o, P, B, L, S, H, d):
    a = _lib.MpgAttn()
    a.q, a.k, a.v = _p(q), _p(k), _p(v)
    a.ldq, a.ldk, a.ldv = q.stride(0), k.stride(0), v.stride(0)
    a.ignore = _p(ignore)
    a.o, a.ldo, a.P = _p(o), o.stride(0), _p(P)
    a.B, a.L, a.S, a.H, a.d = B, L, S, H, d
    return a


class FusedPackedAttnFn(torch.autograd.Function):
    """FusedAttnFn on PACKED projections, as MAB produces them: self-attention takes qkv [B*L, 3E] (kv = None),
    cross-attention q [B*L, E] and kv [B*S, 2E].  The backward writes dq / dk / dv straight into one gradient
    tensor per packed input -- slicing q, k, v out in autograd instead costs a zero-fill, a copy and an add per slice."""

    @staticmethod
    def forward(ctx, qx, kv, ignore, B, L, S, H):
        _chk(qx, "qkv")
        self_attn = kv is None
        E = qx.shape[1] // 3 if self_attn else qx.shape[1]
        d = E // H
        q = qx[:, :E]
        k, v = (qx[:, E:2 * E], qx[:, 2 * E:]) if self_attn else (kv[:, :E], kv[:, E:])
        o = torch.empty((B * L, E), device=qx.device, dtype=torch.float32)
        P = torch.empty((B, H, L, S), device=qx.device, dtype=torch.float32)
        a = _attn_struct(q, k, v, ignore, o, P, B, L, S, H, d)
        check(_lib.lib().mpg_attn_fwd(C.byref(a), _stream()), "mpg_attn_fwd")
        ctx.save_for_backward(qx, kv, P)
        ctx.ignore = ignore
        ctx.dims = (B, L, S, H, d, E, self_attn)
        return o

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        qx, kv, P = ctx.saved_tensors
        B, L, S, H, d, E, self_attn = ctx.dims
        go = go.contiguous()
        dqx = torch.empty_like(qx)
        dkv = None if self_attn else torch.empty_like(kv)
        q = qx[:, :E]
        k, v = (qx[:, E:2 * E], qx[:, 2 * E:]) if self_attn else (kv[:, :E], kv[:, E:])
        dq = dqx[:, :E]
        dk, dv = (dqx[:, E:2 * E], dqx[:, 2 * E:]) if self_attn else (dkv[:, :E], dkv[:, E:])
        a = _attn_struct(q, k, v, ctx.ignore, go, P, B, L, S, H, d)
        a.d_o = _p(go)
        a.dq, a.dk, a.dv = C.c_void_p(dq.data_ptr()), C.c_void_p(dk.data_ptr()), C.c_void_p(dv.data_ptr())
        a.lddq, a.lddk, a.lddv = dq.stride(0), dk.stride(0), dv.stride(0)
        check(_lib.lib().mpg_attn_bwd(C.byref(a), _stream()), "mpg_attn_bwd")
        return dqx, dkv, None, None, None, None, None


class FusedAttnFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d) + key mask) v per (jet, head): q [B*L, E], k, v [B*S, E] (row-strided
    views are fine), ignore [B*S] floats (1 = padded key) or None."""

    @staticmethod
    def forward(ctx, q, k, v, ignore, B, L, S, H):
        _chk(q, "q")
        E = q.shape[1]
        d = E // H
        o = torch.empty((B * L, E), device=q.device, dtype=torch.float32)
        P = torch.empty((B, H, L, S), device=q.device, dtype=torch.float32)
        a = _attn_struct(q, k, v, ignore, o, P, B, L, S, H, d)
        check(_lib.lib().mpg_attn_fwd(C.byref(a), _stream()), "mpg_attn_fwd")
        ctx.save_for_backward(q, k, v, P)
        ctx.ignore = ignore
        ctx.dims = (B, L, S, H, d)
        return o

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        q, k, v, P = ctx.saved_tensors
        B, L, S, H, d = ctx.dims
        go = go.contiguous()
        dq = torch.empty((B * L, H * d), device=q.device, dtype=torch.float32)
        dk = torch.empty((B * S, H * d), device=q.device, dtype=torch.float32)
        dv = torch.empty((B * S, H * d), device=q.device, dtype=torch.float32)
        a = _attn_struct(q, k, v, ctx.ignore, go, P, B, L, S, H, d)
        a.d_o = _p(go)
        a.dq, a.dk, a.dv = _p(dq), _p(dk), _p(dv)
        a.lddq, a.lddk, a.lddv = H * d, H * d, H * d
        check(_lib.lib().mpg_attn_bwd(C.byref(a), _stream()), "mpg_attn_bwd")
        return dq, dk, dv, None, None, None, None, None


# ------------------------------------------------------------------------------------- one launch per MAB
class PackedMAB:
    """Weight images of one MAB for ``mpg_mab_fwd`` / ``mpg_mab_bwd`` (fp16 images of the three weights for the forward
    products, bf16 images of their transposes for the gradient products), rebuilt by ONE ``mpg_pack_many`` launch;
    ``ensure`` / ``refresh`` as ``PackedMPLayer``."""

    def __init__(self, Win, Wo, Wf):
        E = Wo.shape[0]
        self.params = (Win, Wo, Wf)
        self._spec = {
            "Win": (Win, 3 * E, E, 0, SC_WN, True), "Wo": (Wo, E, E, 0, SC_WN, True), "Wf": (Wf, E, E, 0, SC_WN, True),
            "WinT": (Win, E, 3 * E, 1, 1.0, False), "WoT": (Wo, E, E, 1, 1.0, False), "WfT": (Wf, E, E, 1, 1.0, False),
        }
        self.img = {k: torch.empty((_img_elems(v[1], v[2]),), device=Wo.device, dtype=torch.bfloat16)
                    for k, v in self._spec.items()}
        self._key = None

    def _current_key(self):
        return tuple((q.data_ptr(), q._version) for q in self.params)

    def jobs(self):
        return [v + (0, 0, self.img[k]) for k, v in self._spec.items()]

    def refresh(self):
        refresh_many([self])

    def ensure(self):
        if self._key != self._current_key():
            self.refresh()
        return self

    def ptr(self, name):
        return C.c_void_p(self.img[name].data_ptr())


def mab_fusable(E: int, H: int, L: int, S: int) -> bool:
    """Shapes ``mpg_mab_fwd`` / ``mpg_mab_bwd`` take: sets of at most 160 tokens (up to 32: a wave or two per jet; beyond: a
    workgroup per jet, a wave per tile of 32 tokens), E = 32 or 64, heads of 16 features."""
    return E in (32, 64) and H * 16 == E and 1 <= L <= MAB_MAX_TOKENS and 1 <= S <= MAB_MAX_TOKENS


MAB_MAX_TOKENS = 160
MAB_CHAIN_TOKENS = 32      # (``mpg_mab_chain_fwd`` keeps a jet's rows in one wave's registers)


def _mab_struct(x2, y2, ignore, pk, bin_, bo, bf, B, L, S, E, H, alpha, ff_act, tag, thr_mab, sc_mab, thr_ff, sc_ff):
    m = _lib.MpgMab()
    m.x, m.ldx = _p(x2), x2.stride(0)
    m.y, m.ldy = (_p(x2), x2.stride(0)) if y2 is None else (_p(y2), y2.stride(0))
    m.ignore = _p(ignore)
    m.Win, m.bin, m.Wo, m.bo, m.Wf, m.bf = pk.ptr("Win"), _p(bin_), pk.ptr("Wo"), _p(bo), pk.ptr("Wf"), _p(bf)
    m.WinT, m.WoT, m.WfT = pk.ptr("WinT"), pk.ptr("WoT"), pk.ptr("WfT")
    m.B, m.L, m.S, m.E, m.H = B, L, S, E, H
    m.alpha, m.ff_act = alpha, int(ff_act)
    m.seed, m.tag = _p(seed_tensor(x2.device)), tag
    m.thr_mab, m.sc_mab, m.thr_ff, m.sc_ff = thr_mab, sc_mab, thr_ff, sc_ff
    m.wscale, m.ascale = SC_WN, SC_ACT
    return m


def _mab_set_ln(m, ln):
    if ln is not None:
        w1, b1, w2, b2, eps = ln
        m.ln1_w, m.ln1_b, m.ln2_w, m.ln2_b, m.ln_eps = _p(w1), _p(b1), _p(w2), _p(b2), float(eps)


def mab_forward(x2, y2, ignore, pk, bin_, bo, bf, B, L, S, H, *, alpha=0.2, ff_act=True, p_mab=0.0, p_ff=0.0,
                training=False, tag=None, save=False, ln=None):
    """``mpg_mab_fwd``: x2 [B*L, E] queries, y2 [B*S, E] keys/values or None (self-attention), ignore [B*S] floats or
    None.  Returns (out [B*L, E], o, z, tag) -- o and z only with ``save`` (what the backward needs).  ``ln``: (norm1.weight,
    norm1.bias, norm2.weight, norm2.bias, eps) of a block with ``layer_norm=True``; ``save`` then also keeps za (the input of
    norm1) and the return value is (out, o, z, tag, za)."""
    _chk(x2, "x")
    E = x2.shape[1]
    dev = x2.device
    thr_mab, sc_mab = drop_params(p_mab) if training else (0, 1.0)
    thr_ff, sc_ff = drop_params(p_ff) if training else (0, 1.0)
    if tag is None:
        tag = next_tag(dev, "mab", max(thr_mab, thr_ff))
    out = torch.empty((B * L, E), device=dev, dtype=torch.float32)
    o = torch.empty_like(out) if save else None
    z = torch.empty_like(out) if save else None
    m = _mab_struct(x2, y2, ignore, pk, bin_, bo, bf, B, L, S, E, H, alpha, ff_act, tag, thr_mab, sc_mab, thr_ff, sc_ff)
    m.out, m.ldo, m.save_o, m.save_z = _p(out), out.stride(0), _p(o), _p(z)
    za = None
    if ln is not None:
        _mab_set_ln(m, ln)
        za = torch.empty_like(out) if save else None
        m.save_za = _p(za)
    check(_lib.lib().mpg_mab_fwd(C.byref(m), _stream()), "mpg_mab_fwd")
    return (out, o, z, tag) if ln is None else (out, o, z, tag, za)


def _ln_param_grads(dn, gn, w, b, need_w, need_b):
    """(d weight, d bias) of a LayerNorm from the rows the block's backward left -- dn = gradient with respect to the norm's
    output, gn = dn times the normalised input: their column sums.  Inside a TrainStep backward they ride in the grouped
    weight-gradient launch as bias-sum jobs (and Nones are returned); otherwise summed here."""
    st = dev_state(dn.device)
    E = dn.shape[1]
    out = [None, None]
    for k, (rows, prm, need) in enumerate(((gn, w, need_w), (dn, b, need_b))):
        if not need:
            continue
        tgt = _grad_target(prm) if (st.grad_into_param and st.deferred_wgrad is not None) else None
        if tgt is not None:
            st.deferred_wgrad.add(rows, rows[:, :1], out=torch.empty((E, 1), device=dn.device, dtype=torch.float32),
                                  bias_out=tgt.reshape(-1), accumulate=True)
        else:
            out[k] = rows.sum(0)
    return out


def _mab_backward_block(x2, y2, ignore, o, z, params, pk, cfg, gout, need_x, need_y, need_w, ln=None, za=None, need_ln=False):
    """The backward of one attention block (``mpg_mab_bwd`` + its weight gradients: queued for the grouped launches inside a
    TrainStep backward, computed at once otherwise): (dx rows or None, dy rows or None, the six parameter gradients or Nones).
    ``ln`` / ``za``: the block's norms and the input of norm1 kept by the forward; with ``need_ln`` the rows for the norms'
    parameter gradients are produced and returned as a fourth value (dn1, gn1, dn2, gn2)."""
    B, L, S, E, H, alpha, ff_act, tag, p_mab, p_ff = cfg
    bin_, bo, bf = params[1], params[3], params[5]
    dev = x2.device
    cross = y2 is not None
    thr_mab, sc_mab = drop_params(p_mab)
    thr_ff, sc_ff = drop_params(p_ff)
    dout = gout.reshape(B * L, E).contiguous()
    m = _mab_struct(x2, y2, ignore, pk, bin_, bo, bf, B, L, S, E, H, alpha, ff_act, tag, thr_mab, sc_mab, thr_ff, sc_ff)
    m.save_o, m.save_z = _p(o), _p(z)
    m.dout, m.lddout = _p(dout), dout.stride(0)
    dx = torch.empty((B * L, E), device=dev, dtype=torch.float32) if need_x else None
    dy = torch.empty((B * S, E), device=dev, dtype=torch.float32) if (cross and need_y) else None
    m.dx, m.lddx, m.dy, m.lddy = _p(dx), E, _p(dy), E
    dqkv = dq = dkv = dza = du = None
    if need_w:
        if cross:
            dq = torch.empty((B * L, E), device=dev, dtype=torch.float32)
            dkv = torch.empty((B * S, 2 * E), device=dev, dtype=torch.float32)
            m.dq, m.lddq, m.dk, m.dv, m.lddkv = _p(dq), E, _p(dkv), _p(dkv, E), 2 * E
        else:
            dqkv = torch.empty((B * L, 3 * E), device=dev, dtype=torch.float32)
            m.dq, m.lddq, m.dk, m.dv, m.lddkv = _p(dqkv), 3 * E, _p(dqkv, E), _p(dqkv, 2 * E), 3 * E
        dza, du = torch.empty_like(dout), torch.empty_like(dout)
        m.dza, m.du = _p(dza), _p(du)
    elif L > 32 or S > 32:   # (large sets: the waves that own the key tiles read the rows of dza back)
        dza = torch.empty_like(dout)
        m.dza = _p(dza)
    lnrows = None
    if ln is not None:
        _mab_set_ln(m, ln)
        m.save_za = _p(za)
        if need_ln:
            lnrows = tuple(torch.empty((B * L, E), device=dev, dtype=torch.float32) for _ in range(4))
            m.dn1, m.gn1, m.dn2, m.gn2 = (_p(t) for t in lnrows)
    check(_lib.lib().mpg_mab_bwd(C.byref(m), _stream()), "mpg_mab_bwd")
    grads = [None] * 6
    if need_w:
        st = dev_state(dev)
        tg = None
        if st.grad_into_param and st.deferred_wgrad is not None:
            tg = [_grad_target(q) for q in params]
            if any(t is None for t in tg):
                tg = None
        if tg is not None:      # TrainStep: queued for the grouped launches, added into the flat gradient buffers
            gWin, gbin, gWo, gbo, gWf, gbf = tg
            wb = st.deferred_wgrad
            if cross:
                wb.add(dq, x2, out=gWin[:E], bias_out=gbin[:E], accumulate=True)
                wb.add(dkv, y2, out=gWin[E:], bias_out=gbin[E:], accumulate=True)
            else:
                wb.add(dqkv, x2, out=gWin, bias_out=gbin, accumulate=True)
            wb.add(dza, o, out=gWo, bias_out=gbo, accumulate=True)
            wb.add(du, z, out=gWf, bias_out=gbf, accumulate=True)
        else:
            def wgrad(dyv, xv):
                db = torch.empty(dyv.shape[1], device=dev, dtype=torch.float32)
                return linear_bwd_weight(dyv, xv, bias_out=db), db
            if cross:
                (wq, bq), (wkv, bkv) = wgrad(dq, x2), wgrad(dkv, y2)
                grads[0], grads[1] = torch.cat([wq, wkv], 0), torch.cat([bq, bkv], 0)
            else:
                grads[0], grads[1] = wgrad(dqkv, x2)
            grads[2], grads[3] = wgrad(dza, o)
            grads[4], grads[5] = wgrad(du, z)
    return (dx, dy, grads) if ln is None else (dx, dy, grads, lnrows)


def sab_chain_forward(x, ignore, H, alpha, ff_act, p_mab, p_ff, training, pks, params):
    """``FusedSABChainFn`` without a backward to prepare for: one launch, nothing kept but the last block's output."""
    B, L, E = x.shape
    dev = x.device
    x2 = x.reshape(B * L, E).contiguous()
    thr_mab, sc_mab = drop_params(p_mab) if training else (0, 1.0)
    thr_ff, sc_ff = drop_params(p_ff) if training else (0, 1.0)
    c = _lib.MpgMabChain()
    c.n = len(pks)
    inp, keep = x2, []
    for b in range(len(pks)):
        out = torch.empty((B * L, E), device=dev, dtype=torch.float32)
        m = _mab_struct(inp, None, ignore, pks[b], params[6 * b + 1], params[6 * b + 3], params[6 * b + 5], B, L, L, E, H, alpha, ff_act,
                        next_tag(dev, "mab", max(thr_mab, thr_ff)), thr_mab, sc_mab, thr_ff, sc_ff)
        m.out, m.ldo = _p(out), E
        c.blk[b] = m
        keep.append(out)
        inp = out
    check(_lib.lib().mpg_mab_chain_fwd(C.byref(c), _stream()), "mpg_mab_chain_fwd")
    return keep[-1].reshape(B, L, E)


class FusedSABChainFn(torch.autograd.Function):
    """Several self-attention blocks applied one after the other (the SABs of GAPT_G / GAPT_D, gapt/model.py:261-262, :341-342)
    with ONE forward launch (``mpg_mab_chain_fwd``: a wave keeps its jet's rows in registers from block to block); the backward
    runs block by block (``mpg_mab_bwd``).  ``params``: (in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, ff weight,
    ff bias) per block; ``pks``: the blocks' ``PackedMAB`` sets."""

    @staticmethod
    def forward(ctx, x, ignore, H, alpha, ff_act, p_mab, p_ff, training, pks, *params):
        B, L, E = x.shape
        n = len(pks)
        dev = x.device
        x2 = x.reshape(B * L, E).contiguous()
        thr_mab, sc_mab = drop_params(p_mab) if training else (0, 1.0)
        thr_ff, sc_ff = drop_params(p_ff) if training else (0, 1.0)
        c = _lib.MpgMabChain()
        c.n = n
        inp, outs, os_, zs_, tags = x2, [], [], [], []
        for b in range(n):
            bin_, bo, bf = params[6 * b + 1], params[6 * b + 3], params[6 * b + 5]
            tag = next_tag(dev, "mab", max(thr_mab, thr_ff))
            out, o, z = (torch.empty((B * L, E), device=dev, dtype=torch.float32) for _ in range(3))
            m = _mab_struct(inp, None, ignore, pks[b], bin_, bo, bf, B, L, L, E, H, alpha, ff_act, tag, thr_mab, sc_mab, thr_ff, sc_ff)
            m.out, m.ldo, m.save_o, m.save_z = _p(out), E, _p(o), _p(z)
            c.blk[b] = m
            outs.append(out); os_.append(o); zs_.append(z); tags.append(tag)
            inp = out
        check(_lib.lib().mpg_mab_chain_fwd(C.byref(c), _stream()), "mpg_mab_chain_fwd")
        ctx.save_for_backward(x2, ignore, *outs[:-1], *os_, *zs_)
        ctx.pks, ctx.params, ctx.n = pks, params, n
        ctx.cfgs = [(B, L, L, E, H, alpha, ff_act, tags[b], p_mab if training else 0.0, p_ff if training else 0.0) for b in range(n)]
        return outs[-1].reshape(B, L, E)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        n = ctx.n
        sv = ctx.saved_tensors
        x2, ignore = sv[0], sv[1]
        ins = [x2] + list(sv[2:2 + n - 1])            # the input rows of block b: x, then the outputs of the blocks before
        os_, zs_ = sv[2 + n - 1:2 + 2 * n - 1], sv[2 + 2 * n - 1:2 + 3 * n - 1]
        B, L, _, E = ctx.cfgs[0][:4]
        g = gout
        all_grads = [None] * (6 * n)
        for b in reversed(range(n)):
            prm = ctx.params[6 * b:6 * b + 6]
            need_w = any(ctx.needs_input_grad[9 + 6 * b:9 + 6 * b + 6])
            need_x = b > 0 or ctx.needs_input_grad[0]
            dx, _, grads = _mab_backward_block(ins[b], None, ignore, os_[b], zs_[b], prm, ctx.pks[b], ctx.cfgs[b], g, need_x, False, need_w)
            all_grads[6 * b:6 * b + 6] = grads
            g = dx
        return (None if g is None else g.reshape(B, L, E), None, None, None, None, None, None, None, None, *all_grads)


class FusedMABFn(torch.autograd.Function):
    """MAB.forward (gapt/model.py:124-139) as ONE launch each way (``mpg_mab_fwd`` / ``mpg_mab_bwd``).  x [B, L, E]
    queries, y [B, S, E] keys/values or None for self-attention.  The forward keeps only o (attention output) and z
    (input of the feed-forward layer); the backward recomputes the rest and hands the three pre-activation gradients to
    the grouped weight-gradient launches (``TrainStep``) or computes the weight gradients itself."""

    @staticmethod
    def forward(ctx, x, y, ignore, Win, bin_, Wo, bo, Wf, bf, H, alpha, ff_act, p_mab, p_ff, training, pk):
        """``x`` [1, 1, E] with ``y`` [B, S, E], B > 1: ONE query row shared by all jets (PMA's learned seed, gapt/model.py:170-174:
        ``S.repeat(B, 1, 1)``) -- read with row stride 0 instead of being copied B times, and its gradient summed over the jets by
        the grouped weight-gradient launch (as a bias sum) instead of a reduction launch of its own."""
        B, L, E = x.shape
        bcast = B == 1 and L == 1 and y is not None and y.shape[0] > 1
        if bcast:
            B = y.shape[0]
        S = L if y is None else y.shape[1]
        x2 = x.reshape(1, E).contiguous().expand(B, E) if bcast else x.reshape(B * L, E).contiguous()   # (row stride 0)
        y2 = None if y is None else y.reshape(B * S, E).contiguous()
        ctx.bcast = bcast
        out, o, z, tag = mab_forward(x2, y2, ignore, pk, bin_, bo, bf, B, L, S, H, alpha=alpha, ff_act=ff_act,
                                     p_mab=p_mab, p_ff=p_ff, training=training, save=True)
        ctx.save_for_backward(x2, y2, ignore, o, z, bin_, bo, bf)
        ctx.pk, ctx.params = pk, (Win, bin_, Wo, bo, Wf, bf)
        ctx.seed = x if (bcast and x.is_leaf) else None     # (the parameter itself: its .grad takes the summed gradient directly)
        ctx.cfg = (B, L, S, E, H, alpha, ff_act, tag, p_mab if training else 0.0, p_ff if training else 0.0)
        return out.reshape(B, L, E)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x2, y2, ignore, o, z, bin_, bo, bf = ctx.saved_tensors
        B, L, S, E, H, alpha, ff_act, tag, p_mab, p_ff = ctx.cfg
        dev = x2.device
        dx, dy, grads = _mab_backward_block(x2, y2, ignore, o, z, ctx.params, ctx.pk, ctx.cfg, gout,
                                            ctx.needs_input_grad[0], ctx.needs_input_grad[1], any(ctx.needs_input_grad[3:9]))
        if ctx.bcast and dx is not None:
            # the shared query row's gradient: the sum over the jets -- a bias-sum job of the grouped launch when there is one
            st = dev_state(dev)
            gS = _grad_target(ctx.seed) if (st.grad_into_param and st.deferred_wgrad is not None and ctx.seed is not None) else None
            if gS is not None:
                st.deferred_wgrad.add(dx, dx[:, :1], out=torch.empty((E, 1), device=dev, dtype=torch.float32), bias_out=gS.reshape(-1),
                                      accumulate=True)
                dx = None
            else:
                dx = dx.sum(0).reshape(1, 1, E)
            return (dx, None if dy is None else dy.reshape(B, S, E), None, *grads, None, None, None, None, None, None, None)
        return (None if dx is None else dx.reshape(B, L, E), None if dy is None else dy.reshape(B, S, E), None,
                *grads, None, None, None, None, None, None, None)


class FusedMABLayerNormFn(torch.autograd.Function):
    """``FusedMABFn`` for a block with ``layer_norm=True`` (gapt/model.py:118-120, :131-136): ``nn.LayerNorm`` behind each of
    the two residuals, inside the same launch each way (``mpg_mab_fwd`` / ``mpg_mab_bwd`` with the norms' parameters set: one wave
    per jet).  The forward also keeps za, the input of norm1; the backward leaves, per row, the gradients with respect to the
    norms' outputs and those times the normalised inputs -- their column sums, the norms' parameter gradients, ride in the grouped
    weight-gradient launch.  x [B, L, E], y [B, S, E] or None."""

    @staticmethod
    def forward(ctx, x, y, ignore, Win, bin_, Wo, bo, Wf, bf, n1w, n1b, n2w, n2b, eps, H, alpha, ff_act, p_mab, p_ff, training, pk):
        B, L, E = x.shape
        S = L if y is None else y.shape[1]
        x2 = x.reshape(B * L, E).contiguous()
        y2 = None if y is None else y.reshape(B * S, E).contiguous()
        ln = (n1w, n1b, n2w, n2b, eps)
        out, o, z, tag, za = mab_forward(x2, y2, ignore, pk, bin_, bo, bf, B, L, S, H, alpha=alpha, ff_act=ff_act,
                                         p_mab=p_mab, p_ff=p_ff, training=training, save=True, ln=ln)
        ctx.save_for_backward(x2, y2, ignore, o, z, za, n1w, n1b, n2w, n2b)
        ctx.pk, ctx.params, ctx.eps = pk, (Win, bin_, Wo, bo, Wf, bf), eps
        ctx.cfg = (B, L, S, E, H, alpha, ff_act, tag, p_mab if training else 0.0, p_ff if training else 0.0)
        return out.reshape(B, L, E)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x2, y2, ignore, o, z, za, n1w, n1b, n2w, n2b = ctx.saved_tensors
        B, L, S, E = ctx.cfg[:4]
        need = ctx.needs_input_grad
        need_ln = any(need[9:13])
        dx, dy, grads, rows = _mab_backward_block(x2, y2, ignore, o, z, ctx.params, ctx.pk, ctx.cfg, gout, need[0], need[1],
                                                  any(need[3:9]), ln=(n1w, n1b, n2w, n2b, ctx.eps), za=za, need_ln=need_ln)
        g1 = g2 = (None, None)
        if need_ln:
            g1 = _ln_param_grads(rows[0], rows[1], n1w, n1b, need[9], need[10])
            g2 = _ln_param_grads(rows[2], rows[3], n2w, n2b, need[11], need[12])
        return (None if dx is None else dx.reshape(B, L, E), None if dy is None else dy.reshape(B, S, E), None, *grads,
                g1[0], g1[1], g2[0], g2[1], None, None, None, None, None, None, None, None)


# ------------------------------------------------------------------------------------- per-jet pieces around the layers
def rank_mask(first_feature: torch.Tensor, labels: torch.Tensor, num_particles: int, out: Optional[torch.Tensor] = None,
              with_ignore: bool = False, ignore_out: Optional[torch.Tensor] = None):
    """mask_c (mpgan/model.py:689-699): [B, N] floats, 1 for the n = int(label * N) particles of each jet with the
    smallest first feature.  ``first_feature`` [B, N] may be a strided view (x[:, :, 0]); one launch.  ``with_ignore``: returns
    (mask, 1 - mask), the second written by the same launch (the key mask of GAPT's attention blocks)."""
    _chk(first_feature, "first_feature")
    B, N = first_feature.shape
    lab = labels[:, -1]
    if lab.dtype != torch.float32:
        lab = lab.float()
    if out is None:
        out = torch.empty((B, N), device=first_feature.device, dtype=torch.float32)
    ign = ignore_out if ignore_out is not None else (torch.empty((B, N), device=first_feature.device, dtype=torch.float32) if with_ignore else None)
    assert out.is_contiguous() and (ign is None or ign.is_contiguous())
    check(_lib.lib().mpg_rank_mask(_p(first_feature), first_feature.stride(0), first_feature.stride(1), _p(lab), lab.stride(0),
                                   B, N, _p(out), _p(ign), _stream()), "mpg_rank_mask")
    return (out, ign) if (with_ignore or ignore_out is not None) else out


def knn_sets(x: torch.Tensor, mask: Optional[torch.Tensor], num_knn: int, self_loops: bool = True):
    """The k-nearest-neighbour graph of MPLayer._getA_knn (mpgan/model.py:319-381) as bit masks [B * N, ceil(N / 32)]
    (int32): bit j of row (b, i) is set when sender j is among the ``num_knn`` nearest neighbours of receiver i
    (zero-masked senders pushed 1e4 times further away, :333-335).  One launch; no gradient (the selection is discrete)."""
    _chk(x, "x")
    B, N, F = x.shape
    x2 = x.reshape(B * N, F)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    m1 = None if mask is None else mask.reshape(B * N).contiguous()
    nbr = torch.empty((B * N, (N + 31) // 32), device=x.device, dtype=torch.int32)
    check(_lib.lib().mpg_knn_sets(_p(x2), x2.stride(0), _p(m1), B, N, F, num_knn, int(self_loops),
                                  C.c_void_p(nbr.data_ptr()), _stream()), "mpg_knn_sets")
    return nbr


ACT_CODES = {"": 0, "tanh": 1, "sigmoid": 2}


def gen_tail_into(y, mask, act: int, out):
    """out[..., :F] = act(y), out[..., F] = mask - 0.5, written into the caller's [B, N, F+1] buffer (a view with unit
    feature stride is fine); no autograd -- the D step's generator pass runs under ``no_grad`` and lands its jets
    directly in the second half of the discriminator's real+generated batch."""
    B, N, F = y.shape
    y2 = y.reshape(B * N, F)
    if y2.stride(1) != 1:
        y2 = y2.contiguous()
    assert out.shape == (B, N, F + (mask is not None)) and out.stride(2) == 1 and out.stride(0) == N * out.stride(1)
    m1 = None if mask is None else mask.reshape(B * N)
    check(_lib.lib().mpg_gen_tail_fwd(_p(y2), y2.stride(0), _p(m1), _p(out), out.stride(1), B * N, F, act, _stream()),
          "mpg_gen_tail_fwd")
    return out


class GenTailFn(torch.autograd.Function):
    """Final activation + mask column of a generator (MPNet._final_activation / MPGenerator._final_mask,
    mpgan/model.py:533-538, :741-757): out = [act(y) | mask - 0.5], one launch each way."""

    @staticmethod
    def forward(ctx, y, mask, act):
        _chk(y, "y")
        B, N, F = y.shape
        out = torch.empty((B, N, F + (mask is not None)), device=y.device, dtype=torch.float32)
        gen_tail_into(y, mask, act, out)
        ctx.save_for_backward(out)
        ctx.cfg = (B, N, F, act)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        B, N, F, act = ctx.cfg
        if dout.stride(2) != 1 or dout.stride(0) != N * dout.stride(1):
            dout = dout.contiguous()
        dy = torch.empty((B * N, F), device=out.device, dtype=torch.float32)
        check(_lib.lib().mpg_gen_tail_bwd(_p(dout), dout.stride(1), _p(out), out.stride(1), _p(dy), F, B * N, F, act, _stream()),
              "mpg_gen_tail_bwd")
        return dy.reshape(B, N, F), None, None


def bridge_fusable(K: int, F: int, E: int) -> bool:
    """Shapes ``GenDiscBridgeFn`` takes (csrc/bridge.hip: sixteen lanes to a row of 64)."""
    return K == 64 and E == 64 and (1 <= F <= 4 or F == 8)


class GenDiscBridgeFn(torch.autograd.Function):
    """The rows between a GAPT generator's last block and the discriminator's first one as ONE launch each way
    (``mpg_bridge_fwd`` / ``mpg_bridge_bwd``):  feat = act1(pre W1' + b1)  (gen's ``final_fc`` + final activation,
    gapt/model.py:263-265), then  e = dropout(LeakyReLU(feat W2' + b2))  (disc's ``input_embedding``, :336-339).

    ``pre`` [Bg, N, K]: the generated jets' rows; ``feat_buf``: None, or a [B, N, F] batch whose first B - Bg jets are real
    ones -- the generated features are written behind them, in place, and every row is embedded (the D step's real +
    generated batch).  Returns (feat [Bg, N, F] -- None with ``feat_buf`` --, e [B, N, E]).  Weight gradients go through the grouped launches like
    ``FusedLinearFn``'s."""

    @staticmethod
    def forward(ctx, pre, W1, b1, feat_buf, W2, b2, act1, act2, alpha, p_drop, training):
        _chk(pre, "pre")
        Bg, N, K = pre.shape
        F, E = W1.shape[0], W2.shape[0]
        pre2 = pre.reshape(Bg * N, K)
        if pre2.stride(1) != 1 or pre2.stride(0) % 4:
            pre2 = pre2.contiguous()
        if feat_buf is None:
            feat = torch.empty((Bg, N, F), device=pre.device, dtype=torch.float32)
        else:
            feat = feat_buf
            assert feat.is_contiguous() and feat.shape[1:] == (N, F) and feat.shape[0] >= Bg
        B = feat.shape[0]
        e = torch.empty((B, N, E), device=pre.device, dtype=torch.float32)
        thr, dscale = drop_params(p_drop) if training else (0, 1.0)
        tag = next_tag(pre.device, "bridge", thr) + TAG_GENERIC
        q = _lib.MpgBridge()
        q.x, q.ldx, q.W1, q.b1, q.act1 = _p(pre2), pre2.stride(0), _p(W1), _p(b1), int(act1)
        q.feat, q.ldf = _p(feat), F
        q.M, q.row0, q.K, q.F, q.E = B * N, (B - Bg) * N, K, F, E
        q.W2, q.b2, q.act2, q.alpha = _p(W2), _p(b2), int(act2), alpha
        q.seed, q.tag, q.thr, q.dscale = _p(seed_tensor(pre.device)), tag, thr, dscale
        q.e, q.lde = _p(e), E
        check(_lib.lib().mpg_bridge_fwd(q, _stream()), "mpg_bridge_fwd")
        ctx.save_for_backward(pre2, W1, W2, feat, e)
        ctx.set_materialize_grads(False)   # (feat usually goes nowhere else: no zeros filled in for its gradient)
        ctx.params = (W1, b1, W2, b2)
        ctx.cfg = (Bg, B, N, K, F, E, int(act1), int(act2), alpha, thr, dscale, tag)
        return (feat if feat_buf is None else None), e   # (a caller's own batch is written in place: nothing new to hand back)

    @staticmethod
    @once_differentiable
    def backward(ctx, gfeat, ge):
        pre2, W1s, W2s, feat, e = ctx.saved_tensors
        W1, b1, W2, b2 = ctx.params
        Bg, B, N, K, F, E, act1, act2, alpha, thr, dscale, tag = ctx.cfg
        need = ctx.needs_input_grad
        if ge is None:   # (only feat was used downstream)
            ge = torch.zeros((B, N, E), device=pre2.device, dtype=torch.float32)
        want1 = need[1] or (b1 is not None and need[2])
        want2 = need[4] or (b2 is not None and need[5])
        M, row0 = B * N, (B - Bg) * N
        dev = pre2.device
        ge2 = ge.reshape(M, E)
        if ge2.stride(1) != 1 or ge2.stride(0) % 4:
            ge2 = ge2.contiguous()
        g2 = torch.empty((M, E), device=dev, dtype=torch.float32) if want2 else None
        g1 = torch.empty((Bg * N, F), device=dev, dtype=torch.float32) if want1 else None
        dx = torch.empty((Bg * N, K), device=dev, dtype=torch.float32) if need[0] else None
        gf = None
        if gfeat is not None and (want1 or need[0]):
            gf = gfeat.reshape(-1, F)[row0:] if gfeat.shape[0] == B else gfeat.reshape(-1, F)
            gf = gf.contiguous()
        q = _lib.MpgBridgeBwd()
        q.ge, q.ldge, q.e, q.lde, q.feat, q.ldf = _p(ge2), ge2.stride(0), _p(e), E, _p(feat), F
        q.gfeat, q.ldgf, q.W1, q.W2 = _p(gf), F, _p(W1s), _p(W2s)
        q.M, q.row0, q.K, q.F, q.E, q.act1, q.act2, q.alpha = M, row0, K, F, E, act1, act2, alpha
        q.seed, q.tag, q.thr, q.dscale = _p(seed_tensor(dev)), tag, thr, dscale
        q.g2, q.ldg2, q.g1, q.ldg1, q.dx, q.lddx = _p(g2), E, _p(g1), F, _p(dx), K
        if want1 or want2 or need[0]:
            check(_lib.lib().mpg_bridge_bwd(q, _stream()), "mpg_bridge_bwd")
        st = dev_state(dev)
        outs = {}
        for key, g, xin, W, b, nW, nb in (("1", g1, pre2, W1, b1, need[1], b1 is not None and need[2]),
                                          ("2", g2, feat.reshape(M, F), W2, b2, need[4], b2 is not None and need[5])):
            dW = db = None
            if g is not None:
                gW = gb = None
                if nW and st.grad_into_param and st.deferred_wgrad is not None:
                    gW, gb = _grad_target(W), _grad_target(b) if nb else None
                if gW is not None and (not nb or gb is not None):
                    st.deferred_wgrad.add(g, xin, out=gW, bias_out=gb, accumulate=True)
                elif nW:
                    if nb:
                        db = torch.empty(W.shape[0], device=dev, dtype=torch.float32)
                    dW = linear_bwd_weight(g, xin, bias_out=db)
                elif nb:
                    db = g.sum(0)
            outs[key] = (dW, db)
        return (None if dx is None else dx.reshape(Bg, N, K), outs["1"][0], outs["1"][1], None, outs["2"][0], outs["2"][1],
                None, None, None, None, None)


LOSS_CODES = {"ls": 0, "og": 1, "w": 2, "hinge": 3}


def _head_struct(y, mask, w, b, mean, sigmoid, p_drop, training, tag, out, pooled, aux):
    B, N, F = y.shape
    h = _lib.MpgDiscHead()
    h.y, h.ldy, h.mask = _p(y), y.stride(1), _p(mask)
    h.w, h.bias = _p(w), _p(b)
    h.B, h.N, h.F, h.mean, h.sigmoid = B, N, F, int(mean), int(sigmoid)
    thr, dscale = drop_params(p_drop) if training else (0, 1.0)
    h.seed, h.tag, h.thr, h.dscale = _p(seed_tensor(y.device)), tag, thr, dscale
    h.out, h.pooled, h.aux = _p(out), _p(pooled), _p(aux)
    h.loss = -1
    return h


def _head_inputs(y, mask):
    _chk(y, "y")
    B, N, F = y.shape
    if y.stride(2) != 1 or y.stride(0) != N * y.stride(1):
        y = y.contiguous()
    m = None if mask is None else mask.reshape(B, N)
    if m is not None and not m.is_contiguous():
        m = m.contiguous()
    return y, m


class DiscHeadFn(torch.autograd.Function):
    """Discriminator head (mpgan/model.py:812-829 + fnd_layer + :537): out[b] = act(drop(w . pool_i(mask * y) + bias)).
    One launch forward; backward: one launch for dy plus one small reduction for dw / db."""

    @staticmethod
    def forward(ctx, y, mask, w, b, mean, sigmoid, p_drop, training):
        y, m = _head_inputs(y, mask)
        B, N, F = y.shape
        dev = y.device
        out = torch.empty((B,), device=dev, dtype=torch.float32)
        pooled = torch.empty((B, F), device=dev, dtype=torch.float32)
        aux = torch.empty((2 * B,), device=dev, dtype=torch.float32)
        tag = next_tag(dev, "head", drop_params(p_drop)[0] if training else 0) + TAG_GENERIC
        wv = w.reshape(-1)
        h = _head_struct(y, m, wv, b, mean, sigmoid, p_drop, training, tag, out, pooled, aux)
        check(_lib.lib().mpg_disc_head_fwd(C.byref(h), _stream()), "mpg_disc_head_fwd")
        ctx.save_for_backward(y, m, w, b, out, pooled, aux)
        ctx.cfg = (mean, sigmoid, p_drop, training, tag)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        y, m, w, b, out, pooled, aux = ctx.saved_tensors
        mean, sigmoid, p_drop, training, tag = ctx.cfg
        B, N, F = y.shape
        dev = y.device
        h = _head_struct(y, m, w.reshape(-1), b, mean, sigmoid, p_drop, training, tag, out, pooled, aux)
        gout = gout.contiguous()
        h.gout = _p(gout)
        dy = torch.empty((B, N, F), device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        h.dy, h.ld_dy = _p(dy), F
        dw = db = None
        if ctx.needs_input_grad[2]:
            st = dev_state(dev)
            gw = _grad_target(w) if st.grad_into_param else None
            gb = _grad_target(b) if (st.grad_into_param and b is not None) else None
            if gw is not None and (b is None or gb is not None):
                h.dw, h.db, h.accumulate = _p(gw), _p(gb), 1
            else:
                dw = torch.empty_like(w)
                db = None if b is None else torch.empty_like(b)
                h.dw, h.db, h.accumulate = _p(dw), _p(db), 0
        check(_lib.lib().mpg_disc_head_bwd(C.byref(h), _stream()), "mpg_disc_head_bwd")
        return dy, None, dw, db, None, None, None, None


def disc_head_loss(y, mask, w, b, *, mean, sigmoid, p_drop, training, loss, n_real, gen_step, count, loss_out,
                   want_dy=True, wgrad=None):
    """Head forward, the named loss and its gradient in two launches (+ one small reduction), no autograd node:
    returns (out [B], dy [B, N, F] or None).  ``loss_out`` (0-dim tensor) receives the loss value; ``wgrad`` =
    (dw, db) buffers the head's own parameter gradients are ADDED to (the flat .grad views), or None."""
    y, m = _head_inputs(y, mask)
    B, N, F = y.shape
    dev = y.device
    out = torch.empty((B,), device=dev, dtype=torch.float32)
    pooled = torch.empty((B, F), device=dev, dtype=torch.float32)
    aux = torch.empty((2 * B,), device=dev, dtype=torch.float32)
    terms = torch.empty((B,), device=dev, dtype=torch.float32)
    tag = next_tag(dev, "head", drop_params(p_drop)[0] if training else 0) + TAG_GENERIC
    h = _head_struct(y, m, w.reshape(-1), b, mean, sigmoid, p_drop, training, tag, out, pooled, aux)
    h.loss, h.gen_step, h.n_real, h.inv_count = LOSS_CODES[loss], int(gen_step), n_real, 1.0 / count
    h.terms, h.loss_out = _p(terms), _p(loss_out)
    dy = torch.empty((B, N, F), device=dev, dtype=torch.float32) if want_dy else None
    h.dy, h.ld_dy = _p(dy), F
    if wgrad is not None:
        h.dw, h.db, h.accumulate = _p(wgrad[0]), _p(wgrad[1]), 1
    check(_lib.lib().mpg_disc_head_loss(C.byref(h), _stream()), "mpg_disc_head_loss")
    return out, dy


class BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm1d in training mode over the rows of x [M, F] (LinearNet's optional normalisation, mpgan/model.py:58-60,
    :80-81) on ``mpg_batchnorm_*``; returns (y, batch mean, biased batch variance) -- the caller keeps the running
    statistics (``LinearNet._bn``)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        _chk(x, "x")
        x2 = x.contiguous()
        M, F = x2.shape
        nchunk = max(1, min(256, M // 64))
        part = torch.empty((nchunk, F), device=x.device, dtype=torch.float32)
        mean, var = (torch.empty(F, device=x.device, dtype=torch.float32) for _ in range(2))
        check(_lib.lib().mpg_batchnorm_stats(_p(x2), x2.stride(0), M, F, _p(part), nchunk, _p(mean), _p(var), _stream()),
              "mpg_batchnorm_stats")
        y = torch.empty_like(x2)
        check(_lib.lib().mpg_batchnorm_apply(_p(x2), x2.stride(0), _p(mean), _p(var), _p(w), _p(b), eps, _p(y), y.stride(0), M, F,
                                             _stream()), "mpg_batchnorm_apply")
        ctx.save_for_backward(x2, w, mean, var)
        ctx.eps, ctx.params = eps, (w, b)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gm, _gv):
        x2, w, mean, var = ctx.saved_tensors
        M, F = x2.shape
        g2 = g.contiguous()
        nchunk = max(1, min(256, M // 64))
        part = torch.empty((2 * nchunk, F), device=g.device, dtype=torch.float32)
        sums = torch.empty(2 * F, device=g.device, dtype=torch.float32)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        wp, bp = ctx.params
        st = dev_state(g.device)
        # (a frozen norm -- the discriminator's inside train_G -- keeps its .grad buffer as the optimizer launch left it: cleared)
        wanted = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gw = _grad_target(wp) if (st.grad_into_param and wanted) else None
        gb = _grad_target(bp) if (st.grad_into_param and wanted) else None
        direct = gw is not None and gb is not None
        dw = gw if direct else torch.empty(F, device=g.device, dtype=torch.float32)
        db = gb if direct else torch.empty(F, device=g.device, dtype=torch.float32)
        check(_lib.lib().mpg_batchnorm_bwd(_p(g2), g2.stride(0), _p(x2), x2.stride(0), _p(mean), _p(var), _p(w), ctx.eps, _p(part), nchunk,
                                           _p(sums), _p(dx), F, _p(dw), _p(db), int(direct), M, F, _stream()), "mpg_batchnorm_bwd")
        return dx, (None if direct else dw), (None if direct else db), None


def batchnorm_eval(x, w, b, mean, var, eps):
    """BatchNorm1d with given (running) statistics: one elementwise launch, no gradient bookkeeping of its own."""
    x2 = x.contiguous()
    M, F = x2.shape
    y = torch.empty_like(x2)
    check(_lib.lib().mpg_batchnorm_apply(_p(x2), x2.stride(0), _p(mean), _p(var), _p(w), _p(b), eps, _p(y), y.stride(0), M, F, _stream()),
          "mpg_batchnorm_apply")
    return y


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension (GAPT's MAB.norm1 / norm2, gapt/model.py:118-120, :131-136): one launch
    forward, one launch + a fixed-order reduction of the weight / bias gradients backward."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        _chk(x, "x")
        shp = x.shape
        E = shp[-1]
        x2 = x.reshape(-1, E)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        M = x2.shape[0]
        y = torch.empty((M, E), device=x.device, dtype=torch.float32)
        stats = torch.empty((M, 2), device=x.device, dtype=torch.float32)
        check(_lib.lib().mpg_layernorm_fwd(_p(x2), x2.stride(0), _p(w), _p(b), _p(y), E, _p(stats), M, E, eps, _stream()),
              "mpg_layernorm_fwd")
        ctx.save_for_backward(x2, w, stats)
        ctx.params = (w, b)
        ctx.shp = shp
        return y.reshape(shp)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x2, w, stats = ctx.saved_tensors
        M, E = x2.shape
        g2 = g.reshape(M, E)
        if g2.stride(1) != 1:
            g2 = g2.contiguous()
        dx = torch.empty((M, E), device=g.device, dtype=torch.float32)
        nwaves = 4 * min(256, (M + 3) // 4)
        part = torch.empty((nwaves, 2, E), device=g.device, dtype=torch.float32)
        st = dev_state(g.device)
        wp, bp = ctx.params
        dw = db = None
        if not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            # frozen norm (the discriminator's inside train_G): no parameter gradient is formed at all -- its .grad buffer was
            # cleared by the optimizer launch before and must stay clean for the next train_D (TrainStep has no zero_grad there)
            tw = tb = None
            acc = 0
        else:
            gw = _grad_target(wp) if st.grad_into_param else None
            gb = _grad_target(bp) if st.grad_into_param else None
            if gw is not None and gb is not None:
                tw, tb, acc = gw, gb, 1
            else:
                dw, db = torch.empty_like(w), torch.empty_like(w)
                tw, tb, acc = dw, db, 0
        check(_lib.lib().mpg_layernorm_bwd(_p(g2), g2.stride(0), _p(x2), x2.stride(0), _p(w), _p(stats), _p(dx), E, _p(part),
                                           nwaves, _p(tw), _p(tb), acc, M, E, _stream()), "mpg_layernorm_bwd")
        return dx.reshape(ctx.shp), dw, db, None

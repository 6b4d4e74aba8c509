#!/usr/bin/env python3
"""Headline benchmark: jets/sec of one full G+D training iteration, MPGAN gluon-30, B = 256 per GPU.

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without torchrun's environment starts N ranks itself (one per GPU, RCCL,
127.0.0.1 rendezvous) BEFORE anything here touches a GPU and waits for them; under
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` it is one of the ranks.

One "step" = train_D + train_G (reference train.py:398-523) on one synthetic JetNet-30-like batch
resident in HBM, fresh generator noise and dropout masks every step, optimizer updates included.
Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      -- the dominant kernel's algorithmic FLOP (or byte) rate vs the MI355X peak, from HIP-event timings
                   taken in this process on the launch stream
  cpu_baseline  -- the CPU oracle (own port of the reference step) timed on this box's host cores
                   on a bounded sample (rank 0, N = 1 only)
  secondary     -- (default run, --gpus 1) the two other single-GPU workloads of BASELINE.json, 20 warm-up + 100 timed
                   iterations each in the same process: GAPT B = 512 (config 4) and MPGAN N = 150, B = 16 (config 5's
                   per-GPU shard), each with its own value / ms_per_step / roofline
Other workloads (``--model gapt``, ``--particles 150 --batch 16``) can also be run as the line itself: same fields under
their own metric name.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_16BIT = 2.5e15  # dense bf16/f16 MFMA, MI355X (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM = 8.0e12         # HBM3E bytes/s, same guide

H1, H2, H3 = 96, 160, 192
SPLIT = 3                 # MFMA MACs issued per algorithmic MAC by a three-term product (hi/lo operand split, csrc/common.h)


# --------------------------------------------------------------------------- work models
def edge_flops_per_jet(N, F):
    """Algorithmic FLOPs of fe on one jet, dense reference formulation."""
    return N * N * 2 * (2 * F * 96 + 96 * 160 + 160 * 192)


def node_flops_per_jet(N, F, out):
    return N * 2 * ((192 + F) * 256 + 256 * 256 + 256 * out)


def iteration_flops_per_jet(N):
    g = edge_flops_per_jet(N, 32) * 2 + node_flops_per_jet(N, 32, 32) + node_flops_per_jet(N, 32, 3)
    d = edge_flops_per_jet(N, 3) + edge_flops_per_jet(N, 32) + node_flops_per_jet(N, 3, 32) + node_flops_per_jet(N, 32, 32)
    return 3 * (3 * d + 2 * g)  # forward (3 D + 2 G) + backward counted as 2x forward


def executed_flops_per_jet(N, valid_frac):
    """MFMA FLOPs the fused path actually ISSUES per jet and G+D iteration: the logical MACs of every launch in the
    step (x2) times the number of 16-bit terms the product runs in -- 3 (hi/lo x hi/lo without lo*lo) for every forward
    product and the node network; 2 for the edge backward's data-gradient
    products (weight hi + lo times the gradient rounded to fp16); 1 for the edge weight gradients.  Differences to the
    algorithmic figure: layer 1 of fe is the factorised a_i + c_j (two node-level products instead of one per edge);
    masked senders are skipped (``valid_frac`` = mean multiplicity / N of the batch); train_D does not back-propagate into G and train_G forms no weight gradients of D (both results-neutral,
    train.py:420, :495).  Tile padding (30 receivers on 32 lanes, K rounded up to 32) is not counted."""
    E = N * N * valid_frac                      # edges that are computed per jet-layer

    def layer(F, out, fwd, bwd_x, bwd_w, first):
        f = 0.0
        node_fwd = 2 * ((H3 + F) * 256 + 256 * 256 + 256 * out) * N
        ac = 2 * F * 2 * H1 * N
        f += fwd * 3 * (ac + E * 2 * (H1 * H2 + H2 * H3) + node_fwd)
        if bwd_x or bwd_w:
            f += 3 * node_fwd                                                   # fn input-gradient chain
            f += E * 2 * 2 * (H2 * H3 + H2 * H1)                                # dE2, dE1 (2 terms; nothing is recomputed)
            if bwd_w:
                f += E * 2 * (H3 * H2 + H2 * H1) + 3 * (node_fwd + ac)          # dW3, dW2 (1 term); fn dW; dW1
            if not first:
                f += 3 * ac                                                     # dx through [W1a ; W1c]
        return f
    G = lambda fwd, bx, bw: layer(32, 32, fwd, bx, bw, True) + layer(32, 3, fwd, bx, bw, False)
    D = lambda fwd, bx, bw: layer(3, 32, fwd, bx, bw, True) + layer(32, 32, fwd, bx, bw, False)
    step_D = G(1, 0, 0) + 2 * D(1, 1, 1)        # G forward only; D forward + full backward on real + generated
    step_G = G(1, 1, 1) + D(1, 1, 0)            # D: data gradient only
    return step_D + step_G


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU leg: the cores this process may run on, capped at the GPU box's CPU
    share for one GPU (16) -- os.cpu_count() reports the whole host and oversubscribes."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("MPGAN_BENCH_CPU_THREADS", "16"))))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # >= 100 timed, >= 20 warm-up iterations
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="mpgan", choices=["mpgan", "gapt"],
                    help="mpgan = the headline config; gapt = BASELINE config 4 (not the bench line)")
    ap.add_argument("--batch", type=int, default=0, help="jets per GPU (weak scaling); default 256 (mpgan) / 512 (gapt)")
    ap.add_argument("--particles", type=int, default=30)
    ap.add_argument("--jets", default="g", choices=["g", "t", "q"],
                    help="jet type (reference --jets): picks the learning rates of setup_training.py:848-872 and the synthetic "
                         "multiplicity law; g = the headline, t = BASELINE config 3 (top jets)")
    ap.add_argument("--dist", default=None, choices=["gluon", "top", "quark", "uniform"],
                    help="particle-multiplicity law of the synthetic jets (default: the --jets type's own)")
    ap.add_argument("--gp", type=float, default=0.0,
                    help="gradient-penalty weight (reference --gp; train.py:286-324): the D step takes the double-backward route; "
                         "implies --loss w unless given")
    ap.add_argument("--loss", default=None, choices=["ls", "og", "w", "hinge"])
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the GAPT B=512 and MPGAN N=150 B=16 legs that follow the headline run at --gpus 1")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 outside torchrun: become the launcher.  No HIP call has been made in this process (importing the
    package imports torch, which initialises nothing on the device), so starting children is safe; the library is
    built first because ranks must not compile."""
    from mpgan_amd import _lib
    _lib.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("launching", " ".join(cmd))
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts this runs on only support dmabuf IPC; RCCL's intra-node transport shares
    # device buffers between the ranks' processes, and with the legacy mode hipIpcGetMemHandle fails ("invalid
    # argument").  The image exports it already -- this keeps it when the caller's environment was scrubbed, and
    # never overrides a value the caller set.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    from mpgan_amd import dist as mdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would misreport n_gpus")
    # Rehearsal on a one-GPU box (tests/test_gpu_dist.py): MPGAN_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and the
    # exchange on gloo (RCCL refuses two ranks on one device).  Never set in a real run.
    share = os.environ.get("MPGAN_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit(f"rank with LOCAL_RANK={local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, world, pg = mdist.init_from_env("gloo" if share else "nccl", dev)
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    B, N = args.batch or (256 if args.model == "mpgan" else 512), args.particles
    mult = args.dist or JET_LAW[args.jets]
    loss = args.loss or ("w" if args.gp else "ls")
    out, ts = run_workload(torch, dist, args.model, B, N, args.steps, args.warmup, dev, rank, world, pg, mult,
                           not args.no_graphs, share, jets=args.jets, loss=loss, gp=args.gp)
    headline = args.model == "mpgan" and N == 30 and B == 256 and args.jets == "g" and not args.gp and loss == "ls"
    if world > 1 or pg is not None:
        # what the exchange really was: ranks in the RCCL communicator, its version, and whether the two all-reduces sat
        # between three hipGraph segments (default) or inside one graph (MPG_GRAPH_COLLECTIVES=1)
        out["config"]["rccl_ranks"] = dist.get_world_size() if dist.is_initialized() else 1
        out["config"]["collective_backend"] = dist.get_backend(pg) if dist.is_initialized() else None
        try:
            out["config"]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001 -- informational only
            out["config"]["rccl_version"] = None
    out["config"]["graph_collectives"] = bool(ts.graph_collectives)
    out["config"]["graphs_per_step"] = len(ts._graphs) if ts._graphs else 0

    # ------------------------------------------------------------------ roofline of the dominant kernel
    # rank 0 only, after the timed region, with the collectives switched off (the other ranks are not taking part)
    if rank == 0 and not args.no_roofline:
        ts.world, ts.pg = 1, None
        out["roofline"], out["kernels"] = roofline(torch, ts, args.model, dev, measured_traffic=headline)
        log("roofline leg done", out["roofline"]["kernel"], out["roofline"]["frac"])
    del ts

    # ------------------------------------------------------------------ the other single-GPU workloads of BASELINE.json
    # (config 4: GAPT B = 512; config 5's per-GPU shard: MPGAN N = 150, B = 16), same process, after the headline legs:
    # their own value / ms_per_step / roofline under one extra key -- metric, value and config above stay the headline's
    if rank == 0 and world == 1 and headline and not args.no_secondary:
        out["secondary"] = {}
        for key, (model2, B2, N2) in {"gapt_n30_b512": ("gapt", 512, 30), "mpgan_n150_b16": ("mpgan", 16, 150)}.items():
            torch.cuda.empty_cache()
            o2, ts2 = run_workload(torch, dist, model2, B2, N2, 100, 20, dev, 0, 1, None, mult, not args.no_graphs, False)
            sec = {"metric": o2["metric"], "value": o2["value"], "unit": o2["unit"], "ms_per_step": o2["ms_per_step"],
                   "steps": 100, "warmup": 20, "config": o2["config"], "losses": o2["losses"]}
            if not args.no_roofline:
                sec["roofline"], sec["kernels"] = roofline(torch, ts2, model2, dev, measured_traffic=False, workload=key)
            del ts2
            if not args.no_cpu_baseline:   # the oracle's iteration of THIS workload on the host cores, a short bounded sample
                sec["cpu_baseline"] = cpu_baseline(torch, model2, N2, B2, short=True)
            out["secondary"][key] = sec
            log("secondary", key, f"{o2['value']:.0f} jets/s")
        # the headline workload with the edge forward's opt-in product form (MpgEdgeFwd.two_term = 1: fe.net.2 on two 16-bit terms,
        # MPG_FWD_TWO_TERM=1).  NOT the line's value: the default keeps three terms -- with two, pre-activations are known to ~1e-4
        # instead of ~5e-7 of their scale and ~100x more LeakyReLU branches differ from fp32's
        from mpgan_amd import ops as _ops
        torch.cuda.empty_cache()
        prev_tt, _ops.OPTIONS["fwd_two_term"] = _ops.OPTIONS["fwd_two_term"], 1
        try:
            o4, ts4 = run_workload(torch, dist, "mpgan", 256, 30, 100, 20, dev, 0, 1, None, mult, not args.no_graphs, False)
            sec = {k: o4[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "losses")}
            sec["metric"] = "jets/sec (G+D step) mpgan N=30 bs=256, fe.net.2 on two 16-bit terms (opt-in)"
            if not args.no_roofline:
                sec["roofline"], sec["kernels"] = roofline(torch, ts4, "mpgan", dev, measured_traffic=False)
            del ts4
        finally:
            _ops.OPTIONS["fwd_two_term"] = prev_tt
        out["secondary"]["mpgan_n30_b256_two_term"] = sec
        log("secondary", "mpgan_n30_b256_two_term", f"{o4['value']:.0f} jets/s")
        # the gradient-penalty route (reference --gp 10 --loss w, train.py:286-324): D(interpolated) on the double-backward
        # route (mpg_gemm + ATen, the N x N edge tensor in memory), everything else on the fused kernels; eager -- its
        # autograd.grad(create_graph=True) is host-driven -- a few iterations, its own timing only
        torch.cuda.empty_cache()
        o3, ts3 = run_workload(torch, dist, "mpgan", 256, 30, 10, 3, dev, 0, 1, None, mult, False, False, loss="w", gp=10.0)
        del ts3
        out["secondary"]["mpgan_n30_b256_gp"] = {k: o3[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config", "losses")}
        log("secondary", "mpgan_n30_b256_gp", f"{o3['value']:.0f} jets/s")

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(torch, args.model, N, B)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # rank 0 arrives after its extra legs; nobody tears the group down under it
        dist.destroy_process_group()


LOSS_NAME = {"ls": "LSGAN", "og": "BCE GAN", "w": "Wasserstein", "hinge": "hinge"}
JET_LAW = {"g": "gluon", "t": "top", "q": "quark"}
JET_NAME = {"g": "gluon", "t": "top", "q": "quark"}


def run_workload(torch, dist, model, B, N, steps, warmup, dev, rank, world, pg, mult, graphs, share, jets="g", loss="ls", gp=0.0):
    """``warmup`` untimed + ``steps`` timed G+D iterations of ``model`` at B jets per GPU, N particles; returns the JSON
    line's fields for it (value = whole-job jets/s over the MAX time over ranks) and the TrainStep."""
    from mpgan_amd import train, ops, dist as mdist
    from mpgan_amd.data import synthetic_jets
    torch.manual_seed(4 + rank)  # setup_training.py:184 (+ rank: every rank draws its own noise)
    if model == "mpgan":
        G, D = train.default_mpgan(N, disc_dropout=0.5, device=dev, loss=loss)
        latent, (lr_d, lr_g) = 32, train.LR[jets]      # setup_training.py:848-872: per jet type
    else:
        G, D = train.default_gapt(N, disc_dropout=0.5, device=dev)
        latent, (lr_d, lr_g) = 64, train.LR_GAPT
    if world > 1:  # one-time parameter broadcast from rank 0
        mdist.broadcast_module(G, 0, pg)
        mdist.broadcast_module(D, 0, pg)
    ts = train.TrainStep(G, D, B, N, latent=latent, lr_disc=lr_d, lr_gen=lr_g, use_graphs=graphs,
                         process_group=pg, world_size=world, loss=loss, gp_lambda=gp)
    data, labels = synthetic_jets(B, N, seed=4 + rank, dist=mult)
    ts.set_batch(data.to(dev), labels.to(dev))
    ops.set_seed(mdist.rank_seed(0x5EED, rank), dev)
    valid_frac = float((data[..., 3] > 0).float().mean())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    log(f"rank {rank}/{world}: {model} built, B={B} N={N}; warm-up ({warmup} steps, hipGraph capture)")
    for _ in range(warmup):
        ts.step()
    barrier()
    log("warm-up done; timing")
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    jets_per_s = world * B * steps / dt
    log(f"timed {steps} steps in {dt:.3f} s -> {jets_per_s:.0f} jets/s")
    d_loss, g_loss = float(ts.D_loss), float(ts.G_loss)

    two_term = bool(ops.OPTIONS["fwd_two_term"]) and model == "mpgan"   # (MPG_FWD_TWO_TERM=1: never the headline's arithmetic)
    headline = model == "mpgan" and N == 30 and B == 256 and jets == "g" and not gp and loss == "ls" and not two_term
    variant = (("" if jets == "g" else f" {JET_NAME[jets]} jets") + (f" loss={loss}" if loss != "ls" else "") + (f" gp={gp:g}" if gp else "")
               + (" fe.net.2 on two 16-bit terms" if two_term else ""))
    out = {
        "metric": "jets/sec (G+D step) MPGAN gluon N=30 bs=256 @1/2/4/8 MI355X" if headline
                  else f"jets/sec (G+D step) {model} N={N} bs={B}{variant}",
        "value": jets_per_s, "unit": "jets/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("split-16-bit MFMA, fp32 accumulate, fp32 in/out: f16 hi/lo x3 terms (forward; fe.net.2 x2: MPG_FWD_TWO_TERM), f16 x2 / x1 "
                  if two_term else "split-16-bit MFMA, fp32 accumulate, fp32 in/out: f16 hi/lo x3 terms (forward), f16 x2 / x1 ") +
                 "terms in per-sender dithered units (edge backward: data / weight gradients), bf16 hi/lo x3 (node network "
                 "gradients)" if model == "mpgan" else
                 "split-16-bit MFMA, fp32 accumulate, fp32 in/out: f16 hi/lo x3 terms (forward), bf16 hi/lo x3 (gradients)",
        "data": "synthetic",
        "config": {"workload": f"{model.upper()} {JET_NAME[jets]}-like jets (--jets {jets}: lr_disc {lr_d:g}, lr_gen {lr_g:g}), N={N} particles, "
                               f"B={B} per GPU, one train_D+train_G iteration ({LOSS_NAME[loss]}"
                               + (f" + gradient penalty {gp:g} on the double-backward route" if gp else "") + ", RMSprop, D dropout 0.5)",
                   "jets": jets, "lr_disc": lr_d, "lr_gen": lr_g, "loss": loss, "gp_lambda": gp, "fwd_two_term": two_term,
                   "global_batch": world * B, "particles": N, "multiplicity": mult,
                   "mean_multiplicity": valid_frac * N, "parallelism": f"dp{world}", "hip_graphs": graphs,
                   **({"rehearsal": "all ranks share cuda:0, gloo exchange (MPGAN_BENCH_SHARE_GPU)"} if share else {})},
        "losses": {"D": d_loss, "G": g_loss},
    }
    if model == "mpgan":
        out["algorithmic_gflop_per_jet"] = iteration_flops_per_jet(N) / 1e9
        out["executed_gflop_per_jet"] = executed_flops_per_jet(N, valid_frac) / 1e9
        out["whole_step_mfma_frac"] = jets_per_s / world * iteration_flops_per_jet(N) / PEAK_MFMA_16BIT
        out["whole_step_executed_mfma_frac"] = jets_per_s / world * executed_flops_per_jet(N, valid_frac) / PEAK_MFMA_16BIT
    return out, ts


# --------------------------------------------------------------------------- roofline leg
def _work(name, a):
    """(algorithmic FLOPs, algorithmic HBM bytes) of one launch of entry point ``name`` with arguments ``a``."""
    import ctypes
    o = a[0]._obj if hasattr(a[0], "_obj") else (a[0] if isinstance(a[0], ctypes.Structure) else None)   # (byref(struct) or the struct itself)
    if name == "mpg_edge_fwd_fn":   # the edge forward's two dense layers + the node network's three, one launch
        c = a[1]._obj
        edges, rows = o.B * o.N * o.N, o.B * o.N
        fn = sum(2 * rows * c.L[l].K * c.L[l].N for l in range(3))
        return edges * 2 * (H1 * H2 + H2 * H3) + fn, rows * (2 * H1 + (c.L[0].K - H3) + c.L[2].N) * 4
    if name == "mpg_edge_bwd_fn":   # the data-gradient products + the node network's three transposed layers, one launch
        # (a[1] = the layer's dx chain, cdx.nlayers layers; a[2] = the lower layer's node-network input-gradient chain or NULL)
        edges, rows = o.B * o.N * o.N, o.B * o.N
        fn, io = 0, 2 * H1 + 2 * H1
        for arg in a[1:3]:
            c = getattr(arg, "_obj", None)
            if c is None or c.nlayers <= 0:
                continue
            fn += sum(2 * rows * c.L[l].K * c.L[l].N for l in range(c.nlayers))
            io += c.L[c.nlayers - 1].N
        return edges * 2 * (H1 * H2 + H2 * H3) + fn, rows * (io + H3) * 4
    if name in ("mpg_edge_fwd", "mpg_edge_bwd", "mpg_edge_dw"):
        edges = o.B * o.N * o.N
        # the two dense layers the kernel fuses per edge: forward e2 = W2 e1, e3 = W3 e2; backward dE2 = W3^T dZ3,
        # dE1 = W2^T dZ2; weight gradients dW3 = dZ3 E2^T, dW2 = dZ2 E1^T -- (96*160 + 160*192) MAC = 92,160 FLOP per
        # edge each.  The backward's recomputation of layer 2 and the number of 16-bit terms per product are execution cost.
        io = {"mpg_edge_fwd": 2 * H1 + H3, "mpg_edge_bwd": 2 * H1 + H3 + 2 * H1, "mpg_edge_dw": 2 * H1 + H3}[name]
        return edges * 2 * (H1 * H2 + H2 * H3), o.B * o.N * io * 4
    if name in ("mpg_attn_fwd", "mpg_attn_bwd"):
        E = o.H * o.d
        fl = 4 * o.B * o.H * o.L * o.S * o.d
        by = 4 * (o.B * (o.L + 2 * o.S) * E + o.B * o.L * E + o.B * o.H * o.L * o.S)
        return (fl, by) if name == "mpg_attn_fwd" else (2 * fl, 2 * by)
    if name in ("mpg_mab_fwd", "mpg_mab_bwd"):
        # one attention block per jet: in-projection (3E x E on L | S tokens), scores and weighted sum per head,
        # out-projection and feed-forward layer (E x E each); bytes = x, y in and out (+ o, z kept for the backward) one
        # way, x, y, z, dout in and dx, dy, dq|dk|dv, dza, du out the other, plus the weights once
        E, L, S, cross = o.E, o.L, o.S, int(o.y != o.x)
        fl = o.B * (2 * E * E * (L + 2 * S) + 4 * L * S * E + 4 * L * E * E)
        wts = 4 * 5 * E * E
        if name == "mpg_mab_fwd":
            return fl, 4 * o.B * E * (L + cross * S + L + (2 * L if o.save_z else 0)) + wts
        rows_out = (L if o.dx else 0) + (S if o.dy else 0) + ((L + 2 * S + 2 * L) if o.dza else 0)
        return 2 * fl, 4 * o.B * E * (L + cross * S + 2 * L + rows_out) + 2 * wts
    if name == "mpg_mab_chain_fwd":   # the blocks of mpg_mab_fwd back to back: a block's input rows are the registers of the one before
        fl = by = 0
        for b in range(o.n):
            q = o.blk[b]
            fl += q.B * (2 * q.E * q.E * 3 * q.L + 4 * q.L * q.L * q.E + 4 * q.L * q.E * q.E)
            by += 4 * q.B * q.E * (q.L + (2 * q.L if q.save_z else 0)) + 4 * 5 * q.E * q.E
        return fl, by + 4 * o.blk[0].B * o.blk[0].E * o.blk[0].L
    if name == "mpg_bridge_fwd":      # rows in (K) -> features (F) -> rows out (E)
        gen = o.M - o.row0
        return 2 * gen * o.K * o.F + 2 * o.M * o.F * o.E, 4 * (gen * o.K + o.M * o.F + o.M * o.E)
    if name == "mpg_bridge_bwd":
        gen = o.M - o.row0
        return 2 * gen * o.K * o.F + 2 * o.M * o.F * o.E, 4 * (2 * o.M * o.E + (o.M * o.E if o.g2 else 0) + (gen * o.K if o.dx else 0))
    if name == "mpg_gemm":
        return 2 * o.M * o.N * o.K, 4 * (o.M * o.K + o.N * o.K + o.M * o.N * (2 if o.resid else 1))
    if name == "mpg_chain":
        fl = by = 0
        for l in range(o.nlayers):
            L = o.L[l]
            fl += 2 * o.M * L.K * L.N
            by += 4 * o.M * L.N * (1 if L.out else 0) + 4 * L.K * L.N
        return fl, by + 4 * o.M * o.L[0].K
    return 0, 0


def roofline(torch, ts, model, dev, measured_traffic=True, workload=None):
    """Time every launch of every C-ABI entry point with HIP events on the launch stream during a few eager
    (un-captured) iterations of the same step, and price the one that takes the most time."""
    from mpgan_amd import _lib
    lib = _lib.lib()
    rec = {}
    ITER = 4

    class Proxy:
        def __getattr__(self, k):
            fn = getattr(lib, k)
            if not k.startswith("mpg_"):
                return fn

            def timed(*a):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn(*a)
                e1.record()
                rec.setdefault(k, []).append((e0, e1) + _work(k, a))
                return r
            return timed

    saved = _lib._lib
    _lib._lib = Proxy()
    try:
        for _ in range(ITER):
            ts._eager()
        torch.cuda.synchronize(dev)
    finally:
        _lib._lib = saved
    kern, tot = {}, {}
    for name, evs in rec.items():
        per = len(evs) // ITER
        evs = evs[per:]  # drop the first iteration
        ms = [a.elapsed_time(b) for a, b, _, _ in evs]
        kern[name] = {"launches_per_step": per, "avg_ms": sum(ms) / len(ms), "max_ms": max(ms),
                      "ms_per_step": sum(ms) / (ITER - 1)}
        tot[name] = (sum(ms), sum(e[2] for e in evs), sum(e[3] for e in evs), len(evs))
    name = max(tot, key=lambda k: tot[k][0])
    ms_sum, flop_sum, byte_sum, nl = tot[name]
    kname = name.replace("mpg_", "") + "_kernel"
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    # PMC measurement of this kernel (tools/final_profiles.sh, FETCH_SIZE x2 + WRITE_SIZE), bytes per launch; it belongs to the
    # headline workload the counters were collected on and is left null for any other
    # (``workload``: the secondary workloads' own entries, profiles/hbm_traffic.json: {"secondary": {workload: {kernel: ...}}}).
    # The counters are not collected in this run (rocprofv3 --pmc passes of their own): ``traffic_source`` says which committed
    # profile the figure comes from and whether the kernel sources it was measured on are the ones this library was built from
    # -- if not, the figure is DROPPED (null), not carried along stale.
    if os.path.isfile(tpath):
        tj = json.load(open(tpath))
        if workload is not None:
            traffic = tj.get("secondary", {}).get(workload, {}).get(kname, {}).get("bytes_per_launch")
        elif measured_traffic:
            traffic = tj.get(kname, {}).get("bytes_per_launch")
        src = tj.get("source") or {}
        now = _lib.source_digest()
        traffic_source = {"file": "profiles/hbm_traffic.json", "profile": src.get("profile"),
                          "measured_on_source_digest": src.get("source_digest"), "this_source_digest": now,
                          "current": src.get("source_digest") == now}
        if traffic is not None and not traffic_source["current"]:
            traffic_source["dropped_stale_bytes_per_launch"] = traffic
            traffic = None
    # MPGAN's fused edge kernels are MFMA-bound by >100x; GAPT's launches are HBM / latency bound
    mfma = name.startswith("mpg_edge") or (flop_sum / max(byte_sum, 1) > PEAK_MFMA_16BIT / PEAK_HBM)
    if mfma:
        ach = flop_sum / (ms_sum * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": PEAK_MFMA_16BIT / 1e12, "unit": "TFLOP/s",
                "frac": ach * 1e12 / PEAK_MFMA_16BIT, "traffic": traffic, "traffic_source": traffic_source,
                "flop_per_launch": flop_sum / nl, "avg_launch_ms": ms_sum / nl,
                "note": "algorithmic FLOPs of the two fused dense layers (one MAC = 2 FLOP, all B*N*N edges) over the "
                        "HIP-event time of all launches of the kernel in a step; the forward issues 3 MFMA MACs per "
                        "algorithmic MAC (hi/lo split: 1/3 is the ceiling of its frac), the data-gradient kernel 2 per MAC "
                        "(ceiling 1/2), and tools/ubench/mfma_power.hip measures 1.45 "
                        "PFLOP/s (not 2.5) as the dense f16 MFMA rate this chip sustains on random operands"}
    else:
        ach = byte_sum / (ms_sum * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                "frac": ach * 1e9 / PEAK_HBM, "traffic": traffic, "traffic_source": traffic_source,
                "bytes_per_launch": byte_sum / nl, "avg_launch_ms": ms_sum / nl,
                "launches_per_step_all_kernels": sum(v["launches_per_step"] for v in kern.values()),
                "note": "algorithmic bytes (operands read once + results written once, fp32) over the HIP-event time of "
                        "all launches of the kernel in a step; at this size every launch is latency-bound (a few us of "
                        "work), so the launch count per step is the figure to lower"}
    return roof, kern


# --------------------------------------------------------------------------- CPU leg
def cpu_baseline(torch, model, N, B_gpu, short=False):
    """The oracle's restatement of the same iteration on this box's host cores, bounded samples: BASELINE config 1
    (B = 32; >= 10 timed iterations after a warm-up) and the GPU leg's own batch size (as many as fit ~15 s), fp32,
    D dropout 0.5 (Bernoulli masks)."""
    from oracle import train_ref as T
    from oracle.mpgan_ref import RandKeeps
    cores = host_threads()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads")
    if model == "mpgan":
        shapes, lat, lrs = (T.mpgan_param_shapes(True), T.mpgan_param_shapes(False)), 32, (3e-5, 1e-5)
    else:
        shapes, lat, lrs = (T.gapt_param_shapes(True), T.gapt_param_shapes(False)), 64, (1.5e-4, 0.5e-4)

    def sample(B, min_iters, budget_s):
        sdG, sdD = T.init_state_dict(shapes[0], 1), T.init_state_dict(shapes[1], 2)
        stD, stG = {}, {}
        data, labels = T.synthetic_batch(B, N, seed=4)
        keeps = (RandKeeps(), RandKeeps(), RandKeeps())
        times = []
        t_start = time.perf_counter()
        while len(times) < 1 + min_iters or (time.perf_counter() - t_start < budget_s and len(times) < 1 + 30):
            nD, nG = torch.randn(B, N, lat) * 0.2, torch.randn(B, N, lat) * 0.2
            t0 = time.perf_counter()
            T.train_iteration(model, sdD, sdG, stD, stG, data, labels, nD, nG, *lrs, p_disc=0.5, keeps=keeps)
            times.append(time.perf_counter() - t0)
            if len(times) > 1 and time.perf_counter() - t_start > 4 * budget_s:
                break
        timed = times[1:]
        dt = sum(timed) / len(timed)
        log(f"cpu B={B}: {len(timed)} timed iterations, {dt:.3f} s each")
        return {"value": B / dt, "ms_per_step": 1e3 * dt, "iterations": len(timed), "batch": B}

    # BASELINE config 1's batch (32) where an iteration takes a fraction of a second (N = 30); at N = 150 one
    # iteration of 32 jets is ~20 s of CPU: the sample there is the GPU leg's own batch alone
    # (``short``: the secondary workloads' legs -- fewer iterations, so that the default run stays within a few minutes)
    if short:
        small = sample(32, 4, 4.0) if N <= 40 else sample(B_gpu, 2, 8.0)
    else:
        small = sample(32, 10, 8.0) if N <= 40 else sample(B_gpu, 3, 15.0)
    res = {"value": small["value"], "unit": "jets/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{small['iterations']} timed G+D iterations (after 1 warm-up) at B={small['batch']}, N={N}"
                     f"{' (BASELINE config 1 batch)' if small['batch'] == 32 else ''}, {model}, fp32, torch {torch.__version__} "
                     "CPU ops, D dropout 0.5; the port skips the reference's results-neutral wasted work (G backward in "
                     "train_D, D weight gradients in train_G)",
           "ms_per_step": small["ms_per_step"], "iterations": small["iterations"]}
    if B_gpu != small["batch"]:
        big = sample(B_gpu, 2, 8.0) if short else sample(B_gpu, 3, 15.0)
        res["at_gpu_batch"] = {"value": big["value"], "unit": "jets/s", "batch": B_gpu, "ms_per_step": big["ms_per_step"],
                               "iterations": big["iterations"]}
    return res


if __name__ == "__main__":
    main()

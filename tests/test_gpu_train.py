"""GPU: the full G+D iteration on the fused path vs goldens captured from the reference modules."""
import numpy as np
import pytest
import torch

from conftest import load_golden, summarize, rel_err

pytestmark = pytest.mark.gpu


def _setup(B, N, disc_dropout=0.0, use_graphs=False, seedG=41, seedD=42):
    from oracle import train_ref as T
    from mpgan_amd import train
    G, D = train.default_mpgan(N, disc_dropout=disc_dropout)
    G.load_state_dict(T.init_state_dict(T.mpgan_param_shapes(True), seedG, torch.float32))
    D.load_state_dict(T.init_state_dict(T.mpgan_param_shapes(False), seedD, torch.float32))
    return G, D


def test_state_dict_manifest_matches_reference():
    import json, os
    from conftest import GOLDEN
    G, D = _setup(4, 30)
    with open(os.path.join(GOLDEN, "manifests.json")) as f:
        m = json.load(f)
    assert {k: list(v.shape) for k, v in G.state_dict().items()} == m["mpgan_G"]
    assert {k: list(v.shape) for k, v in D.state_dict().items()} == m["mpgan_D"]


def test_nets_forward_vs_reference_golden():
    from oracle import train_ref as T
    g = load_golden("mpgan_nets_f32.npz")
    G, D = _setup(6, 30, seedG=11, seedD=12)
    G.eval(); D.eval()
    dev = "cuda"
    gout = G(torch.from_numpy(g["noise"]).to(dev), torch.from_numpy(g["labels"]).to(dev))
    dout = D(torch.from_numpy(g["data"]).to(dev), torch.from_numpy(g["labels"]).to(dev))
    assert rel_err(gout.detach().cpu().numpy(), g["gout"]) < 1e-4
    assert rel_err(dout.detach().cpu().numpy(), g["dout"]) < 1e-4


def test_train_step_vs_reference_golden():
    """Two iterations, dropout 0, the reference's learning rates: losses, first-iteration gradients
    and parameter UPDATES match what the reference modules + torch RMSprop produced (fp64 golden)."""
    from mpgan_amd import train
    g = load_golden("train_step_mpgan.npz")
    B, N = g["data"].shape[:2]
    G, D = _setup(B, N)
    init = {("G", k): v.detach().clone() for k, v in G.state_dict().items()}
    init.update({("D", k): v.detach().clone() for k, v in D.state_dict().items()})
    ts = train.TrainStep(G, D, B, N, lr_disc=float(g["lr_d"]), lr_gen=float(g["lr_g"]), use_graphs=False)
    ts.set_batch(torch.from_numpy(g["data"]).float().cuda(), torch.from_numpy(g["labels"]).float().cuda())
    ts.fixed_noise = (torch.from_numpy(g["noise_D"]).float().cuda(), torch.from_numpy(g["noise_G"]).float().cuda())
    for it in range(2):
        ts._seg_D()
        if it == 0:
            for k, p in D.named_parameters():
                assert rel_err(summarize(k, p.grad), g["gradD__" + k]) < 2e-3, k
        ts._seg_G()
        if it == 0:
            for k, p in G.named_parameters():
                assert rel_err(summarize(k, p.grad), g["gradG__" + k]) < 2e-3, k
        ts._seg_end()
        assert abs(float(ts.D_loss) - float(g[f"D_loss{it}"])) < 1e-4 * abs(float(g[f"D_loss{it}"]))
        assert abs(float(ts.G_loss) - float(g[f"G_loss{it}"])) < 1e-4 * abs(float(g[f"G_loss{it}"]))
    # parameter values after both iterations (summaries: sum, l2, 64 samples)
    for net, mod in (("D", D), ("G", G)):
        for k, p in mod.named_parameters():
            assert rel_err(summarize(k, p.data), g[f"post{net}__" + k]) < 1e-4, (net, k)


def test_graph_replay_equals_eager():
    """Dropout off: kernels and RMSprop are deterministic, so three hipGraph replays must give
    bit-identical parameters to three eager iterations.  Dropout on: replays stay finite and
    every replay draws new masks (the seed lives in device memory)."""
    from mpgan_amd import train, ops
    from oracle.train_ref import synthetic_batch
    B, N = 16, 30
    data, labels = synthetic_batch(B, N, seed=3)
    res = []
    for use_graphs in (False, True):
        G, D = _setup(B, N, disc_dropout=0.0)
        ts = train.TrainStep(G, D, B, N, use_graphs=use_graphs)
        ts.set_batch(data.cuda(), labels.cuda())
        gen = torch.Generator(device="cuda").manual_seed(5)
        ts.fixed_noise = (torch.randn(B, N, 32, device="cuda", generator=gen) * 0.2,
                          torch.randn(B, N, 32, device="cuda", generator=gen) * 0.2)
        if use_graphs:
            ts.capture(warmup=0)
        for _ in range(3):
            ts.step()
        torch.cuda.synchronize()
        res.append((ts.fD.flat.clone(), ts.fG.flat.clone(), float(ts.D_loss), float(ts.G_loss)))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]

    G, D = _setup(B, N, disc_dropout=0.5)
    ts = train.TrainStep(G, D, B, N, use_graphs=True)
    ts.set_batch(data.cuda(), labels.cuda())
    losses = []
    for _ in range(4):
        ts.step()
        losses.append(float(ts.D_loss))
    assert all(np.isfinite(l) for l in losses)
    assert len(set(losses)) == 4  # fresh noise and fresh dropout masks on every replay

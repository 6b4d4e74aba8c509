"""Host-side pieces around the hot path that need no GPU: synthetic data, (un)normalisation, checkpoint / loss-history
formats of a reference run, optimiser state dicts in torch.optim layout, per-device op state."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_synthetic_jets_match_the_oracle_generator():
    from mpgan_amd.data import synthetic_jets
    from oracle.train_ref import synthetic_batch
    for B, N, d in ((64, 30, "gluon"), (64, 30, "uniform"), (16, 150, "gluon")):
        a, b = synthetic_jets(B, N, 5, d), synthetic_batch(B, N, 5, d)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        data, labels = a
        n = (labels[:, 0] * N).int()                       # the round trip the generator's mask_c relies on
        assert torch.equal(n, (data[..., 3] > 0).sum(1).int())
    with pytest.raises(ValueError):
        synthetic_jets(4, 30, dist="nope")


def test_normalise_unnormalise_round_trip_and_dataset():
    from mpgan_amd.data import JetArrayDataset, unnormalise_jets, FEATURE_MAXES
    rs = np.random.RandomState(0)
    raw = rs.uniform(0, 1, size=(20, 30, 4)).astype(np.float32) * np.array(FEATURE_MAXES["t"], dtype=np.float32)
    raw[..., 3] = (rs.uniform(size=(20, 30)) > 0.3)
    raw[..., :3] *= raw[..., 3:]
    ds = JetArrayDataset(raw, jet_type="t", split="all")
    x, lab = ds[3]
    assert x.shape == (30, 4) and abs(float(lab) - raw[3, :, 3].sum() / 30) < 1e-6
    assert float(ds.particle_data[..., 2].min()) >= -0.5 - 1e-6 and float(ds.particle_data[..., 2].max()) <= 0.5 + 1e-6
    back = unnormalise_jets(ds.particle_data, "t")
    assert torch.allclose(back, torch.from_numpy(raw[..., :3]), atol=1e-6)
    tr, va = JetArrayDataset(raw, split="train"), JetArrayDataset(raw, split="valid")
    assert len(tr) == 14 and len(va) == 6


def test_checkpoint_formats_round_trip(tmp_path):
    from mpgan_amd import checkpoint as ck
    D, G = torch.nn.Linear(3, 2), torch.nn.Linear(4, 3)
    oD, oG = torch.optim.RMSprop(D.parameters(), lr=3e-5), torch.optim.RMSprop(G.parameters(), lr=1e-5)
    D(torch.randn(5, 3)).sum().backward(); oD.step()
    G(torch.randn(5, 4)).sum().backward(); oG.step()
    mp = str(tmp_path / "models")
    assert ck.latest_epoch(mp) == 0
    ck.save_models(D, G, oD, oG, mp, 5)
    ck.save_models(D, G, oD, oG, mp, 10)
    torch.save(D.state_dict(), os.path.join(mp, "D_15.pt"))      # G_15 missing: epoch 15 does not count
    assert sorted(os.listdir(mp))[:4] == ["D_10.pt", "D_15.pt", "D_5.pt", "D_optim_10.pt"]
    assert ck.latest_epoch(mp) == 10
    D2, G2 = torch.nn.Linear(3, 2), torch.nn.Linear(4, 3)
    ck.load_models(D2, G2, mp, 10)
    assert torch.equal(D2.weight, D.weight) and torch.equal(G2.bias, G.bias)
    o2D, o2G = torch.optim.RMSprop(D2.parameters(), lr=1.0), torch.optim.RMSprop(G2.parameters(), lr=1.0)
    ck.load_optimizers(o2D, o2G, mp, 10)
    assert o2D.state_dict()["param_groups"][0]["lr"] == 3e-5
    assert torch.equal(o2G.state_dict()["state"][0]["square_avg"], oG.state_dict()["state"][0]["square_avg"])
    # loss histories
    keys, eval_keys = ck.loss_keys(gp=False, fpnd=False, fpd=True, efp=False)
    assert keys == ["D", "Dr", "Df", "G", "w1p", "w1m", "fpd"]
    losses = {"D": [0.5, 0.4, 0.3], "Dr": [0.2, 0.2, 0.1], "Df": [0.3, 0.2, 0.2], "G": [0.9, 0.8, 0.7],
              "w1p": [[1e-3, 1e-4]], "w1m": [[2e-3, 2e-4]], "fpd": [[0.5, 0.01]]}
    lp = str(tmp_path / "losses")
    ck.save_losses(losses, lp)
    back = ck.load_losses(lp, keys, eval_keys, start_epoch=1, save_epochs=5)
    assert back["D"] == [0.5, 0.4] and back["w1p"] == [[1e-3, 1e-4]] and back["fpd"] == [[0.5, 0.01]]
    assert ck.load_losses(lp, ["nope"])["nope"] == []


@pytest.mark.parametrize("opt", ["rmsprop", "adam", "adadelta"])
def test_flat_params_speak_torch_optim_state_dicts(opt):
    """FlatParams.state_dict() loads into the matching torch.optim class and vice versa (the reference's
    *_optim_<epoch>.pt files, train.py:534-535 / setup_training.py:1525-1535)."""
    from mpgan_amd.train import FlatParams
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 2))
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 2))
    ref.load_state_dict(net.state_dict())
    cls = {"rmsprop": torch.optim.RMSprop, "adam": torch.optim.Adam, "adadelta": torch.optim.Adadelta}[opt]
    kw = {"weight_decay": 5e-4, "betas": (0.5, 0.9)} if opt == "adam" else {}
    to = cls(ref.parameters(), lr=1e-3, **kw)
    for _ in range(3):
        to.zero_grad()
        ref(torch.randn(4, 6)).pow(2).sum().backward()
        to.step()
    fp = FlatParams(net, opt, betas=(0.5, 0.9))
    assert fp.state_dict()["state"] == {}                      # no step taken yet: empty, like torch
    lr = fp.load_state_dict(to.state_dict())
    assert lr == 1e-3 and fp.steps == 3
    sd = fp.state_dict()
    tsd = to.state_dict()
    assert sd["param_groups"][0].keys() == tsd["param_groups"][0].keys()
    assert sd["param_groups"][0]["params"] == tsd["param_groups"][0]["params"]
    for i, ent in tsd["state"].items():
        assert sd["state"][i].keys() == ent.keys(), (sd["state"][i].keys(), ent.keys())
        for k, v in ent.items():
            assert torch.allclose(sd["state"][i][k].float().cpu(), torch.as_tensor(v).float()), (i, k)
    fresh = cls(ref.parameters(), lr=5.0, **kw)
    fresh.load_state_dict(sd)                                  # and torch accepts what we write
    assert fresh.state_dict()["param_groups"][0]["lr"] == 1e-3
    with pytest.raises(ValueError):
        fp.load_state_dict({"state": {0: {"nope": torch.zeros(5, 6)}, 1: {}, 2: {}, 3: {}}, "param_groups": [{}]})


def test_device_state_is_per_device_not_global():
    from mpgan_amd import ops
    a, b = ops.dev_state(0), ops.dev_state(1)
    assert a is not b and a is ops.dev_state("cuda:0") and ops.dev_state("cpu").index == -1
    a.grad_into_param = True
    assert b.grad_into_param is False
    a.grad_into_param = False
    t0 = ops.next_tag(0); t1 = ops.next_tag(1); t0b = ops.next_tag(0)
    assert t0b - t0 == 8 and ops.last_tag(1) == t1 and ops.last_tag(0) == t0b
    for name in ("DEFERRED_WGRAD", "LAST_TAG", "_seed", "_tag_counter"):
        assert not hasattr(ops, name), name                    # no step state at module level
    assert "grad_into_param" not in ops.OPTIONS


def test_fused_backwards_decline_double_backward():
    """Every fused autograd Function is marked once_differentiable: a create_graph=True pass through it raises
    instead of silently dropping second-order terms (reference train.py:304-311 would do exactly that)."""
    import inspect
    from mpgan_amd import ops
    for fn in (ops.FusedMPLayerFn, ops.FusedLinearFn, ops.FusedDropoutFn, ops.FusedPackedAttnFn, ops.FusedAttnFn):
        src = inspect.getsource(fn)
        assert "@once_differentiable\n    def backward" in src, fn


def test_jetnet_file_reader(tmp_path):
    """JetNet's on-disk layout (particle_features [n, N, 4], jet_features [n, 4] = pt, eta, mass, num_particles) read
    from a file the test writes itself, normalised as train.py:41-67 configures JetNet: x / max + shift per particle
    feature, num_particles * (1 / num_hits) as the label, 70 / 30 train / valid split."""
    from mpgan_amd.data import JetArrayDataset, read_jetnet_file, FEATURE_MAXES, FEATURE_SHIFTS
    rs = np.random.RandomState(1)
    n, N = 40, 30
    mult = rs.randint(1, N + 1, size=n)
    pf = rs.uniform(-1, 1, size=(n, N, 4)).astype(np.float32)
    pf[..., 3] = np.arange(N)[None] < mult[:, None]
    pf[..., :3] *= pf[..., 3:]
    jf = np.stack([rs.uniform(800, 1600, n), rs.normal(0, 1, n), rs.uniform(0, 200, n), mult], axis=1).astype(np.float32)
    np.savez(tmp_path / "g.npz", particle_features=pf, jet_features=jf)
    a, b = read_jetnet_file(str(tmp_path / "g.npz"))
    assert np.array_equal(a, pf) and np.array_equal(b, jf)
    tr = JetArrayDataset.from_jetnet_file(str(tmp_path), "g", 30, split="train")
    va = JetArrayDataset.from_jetnet_file(str(tmp_path), "g", 30, split="valid")
    assert len(tr) == 28 and len(va) == 12
    x, lab = va[2]
    want = pf[30] / np.array(FEATURE_MAXES["g"], dtype=np.float32) + np.array(FEATURE_SHIFTS, dtype=np.float32)
    assert np.allclose(x.numpy(), want, atol=1e-7) and x.dtype == torch.float32
    assert float(lab) == float(np.float32(mult[30]) * np.float32(1.0 / 30))     # the product with the reciprocal
    assert int(float(lab) * 30) == mult[30]
    # a shorter particle axis takes the first (pT-ordered) particles and counts the multiplicity again
    cut = JetArrayDataset.from_jetnet_file(str(tmp_path), "g", 10, split="all")
    assert cut.particle_data.shape == (n, 10, 4)
    assert np.allclose(cut.jet_features[:, 0].numpy() * 10, np.minimum(mult, 10))
    # the 150-particle file name, a missing file, a malformed array
    np.savez(tmp_path / "t150.npz", particle_features=np.zeros((3, 150, 4), np.float32))
    assert JetArrayDataset.from_jetnet_file(str(tmp_path), "t", 150, split="all").particle_data.shape == (3, 150, 4)
    with pytest.raises(FileNotFoundError):
        JetArrayDataset.from_jetnet_file(str(tmp_path), "q", 30)
    np.savez(tmp_path / "q.npz", particle_features=np.zeros((3, 30, 3), np.float32))
    with pytest.raises(ValueError):
        JetArrayDataset.from_jetnet_file(str(tmp_path), "q", 30)


def test_flat_params_leave_frozen_parameters_alone():
    """Spectral norm's power-iteration vectors (requires_grad = False) are not part of the flat buffers, never stepped,
    and optimiser state dicts come and go in both index conventions of the reference's optimizers
    (setup_training.py:1500-1523: filtered by requires_grad, or all of module.parameters())."""
    from mpgan_amd.mpgan import LinearNet
    from mpgan_amd.train import FlatParams
    net = LinearNet([8, 6], input_size=5, output_size=2, final_linear=True, spectral_norm=True)
    names = [k for k, _ in net.named_parameters()]
    frozen = [i for i, (k, p) in enumerate(net.named_parameters()) if not p.requires_grad]
    assert frozen and all(("weight_u" in names[i]) or ("weight_v" in names[i]) for i in frozen)
    u_before = [p.detach().clone() for p in net.parameters() if not p.requires_grad]
    fp = FlatParams(net, "adam", betas=(0.5, 0.9))
    assert fp.n == sum(p.numel() for p in net.parameters() if p.requires_grad)
    assert all(p.grad is None for p in net.parameters() if not p.requires_grad)
    assert all(torch.equal(a, b) for a, b in zip(u_before, [p for p in net.parameters() if not p.requires_grad]))
    trained = [p for p in net.parameters() if p.requires_grad]
    for ref in (torch.optim.Adam(trained, lr=1e-3, weight_decay=5e-4, betas=(0.5, 0.9)),
                torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-4, betas=(0.5, 0.9))):
        for p in trained:
            p.grad.fill_(0.25)
        ref.step()
        sd = ref.state_dict()
        fp.load_state_dict(sd)                                           # the reference's file, either convention
        assert fp.steps == 1.0
        filtered = len(sd["param_groups"][0]["params"]) == len(trained)
        ours = fp.state_dict(1e-3, filtered=filtered)
        assert list(ours["state"].keys()) == list(sd["state"].keys())
        for i in sd["state"]:
            assert torch.equal(ours["state"][i]["exp_avg_sq"].cpu(), sd["state"][i]["exp_avg_sq"])
        ref.load_state_dict(ours)                                        # ... and torch.optim reads ours
    with pytest.raises(ValueError):
        fp.load_state_dict({"state": {}, "param_groups": [{"params": list(range(fp._n_all + 3))}]})


def test_spectral_norm_two_forwards_before_one_backward():
    """train_D runs D(real) and D(fake) and backpropagates once (train.py:432-460): the power iteration of the second
    forward must not invalidate what autograd saved for the first."""
    from mpgan_amd.mpgan.model import SpectralNorm
    torch.manual_seed(0)
    sn = SpectralNorm(torch.nn.Linear(6, 4))
    u0 = sn.module.weight_u.detach().clone()
    w1 = sn.weight()
    w2 = sn.weight()
    (w1.sum() + (w2 * w2).sum()).backward()
    assert sn.module.weight_bar.grad is not None and bool(torch.isfinite(sn.module.weight_bar.grad).all())
    assert not torch.equal(sn.module.weight_u, u0) and not sn.module.weight_u.requires_grad
    # one power iteration as the reference does it (spectral_normalization.py:29-39)
    ref = torch.nn.Linear(6, 4)
    torch.manual_seed(0)
    sn2 = SpectralNorm(torch.nn.Linear(6, 4))
    w, u = sn2.module.weight_bar.detach(), sn2.module.weight_u.detach().clone()
    v = torch.mv(w.t(), u); v = v / (v.norm() + 1e-12)
    u = torch.mv(w, v); u = u / (u.norm() + 1e-12)
    assert torch.allclose(sn2.weight(), w / (u.dot(w.mv(v)) + 1e-12), atol=1e-7)


def test_synthetic_jet_laws_and_learning_rates_per_jet_type():
    """``bench.py --jets {g,t,q}``: the learning rates of setup_training.py:848-872 and the synthetic multiplicity laws standing in for
    the three JetNet jet types (top jets nearly fill their 30 slots, quark jets are lighter than gluon jets)."""
    import bench
    from mpgan_amd import train
    from mpgan_amd.data import synthetic_jets
    assert train.LR == {"g": (3e-5, 1e-5), "t": (6e-5, 2e-5), "q": (1.5e-5, 0.5e-5)}
    assert bench.JET_LAW == {"g": "gluon", "t": "top", "q": "quark"}
    mean = {}
    for law in ("gluon", "top", "quark", "uniform"):
        data, labels = synthetic_jets(512, 30, seed=3, dist=law)
        n = (data[..., 3] > 0).sum(1)
        assert int(n.min()) >= 1 and int(n.max()) <= 30
        assert torch.equal((labels[:, 0] * 30).round().long(), n)          # labels = multiplicity / N
        assert bool(((data[..., 3] > 0)[:, :-1] >= (data[..., 3] > 0)[:, 1:]).all())   # real particles first
        mean[law] = float(n.float().mean())
    assert mean["top"] > 27 > mean["gluon"] > mean["quark"] > mean["uniform"]

"""Drop-in boundary, checked against the reference where it can be imported
(the build container; skipped on the GPU box, where /root/reference does not exist):

* ``install_as_reference_packages()`` makes the reference's ``setup_training`` build OUR classes, with its own
  default arguments, and the import line of the reference's ``train.py:7`` resolves;
* the published ``trained_models/mp_g`` generator checkpoint loads (all keys matched);
* ``augment`` / ``mask_manual`` reproduce the reference functions under the same RNG state.
"""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")


@pytest.fixture
def as_reference_packages():
    import mpgan_amd
    saved = {k: sys.modules.get(k) for k in ("mpgan", "gapt", "setup_training")}
    mpgan_amd.install_as_reference_packages()
    sys.path.insert(0, REF)
    sys.modules.pop("setup_training", None)
    try:
        yield
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _default_args(setup_training, extra=()):
    argv = sys.argv
    sys.argv = ["train.py", "--name", "t", "--no-save-zero"] + list(extra)
    try:
        args = setup_training.parse_args()
    finally:
        sys.argv = argv
    args.device = "cpu"
    return setup_training.process_args(args) if hasattr(setup_training, "process_args") else args


def test_train_py_import_line_resolves():
    """``from mpgan import augment, mask_manual`` (reference train.py:7) with our package installed."""
    import mpgan_amd
    saved = {k: sys.modules.get(k) for k in ("mpgan", "gapt")}
    try:
        mpgan_amd.install_as_reference_packages()
        ns = {}
        exec("from mpgan import augment, mask_manual\nfrom mpgan import MPGenerator, MPDiscriminator, MPNet, MPLayer, LinearNet\n"
             "from gapt import GAPT_G, GAPT_D, MAB, SAB, ISAB, PMA, LinearNet as L2", ns)
        assert isinstance(ns["augment"], types.ModuleType) and callable(ns["augment"].augment)
        assert callable(ns["mask_manual"])
        assert ns["MPGenerator"].__module__.startswith("mpgan_amd")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@needs_ref
def test_reference_setup_builds_our_classes_and_loads_published_weights(as_reference_packages):
    import setup_training
    import mpgan_amd
    try:
        args = _default_args(setup_training)
    except SystemExit:
        pytest.skip("reference argument parser needs options this environment cannot supply")
    G = setup_training.setup_mpgan(args, gen=True)
    D = setup_training.setup_mpgan(args, gen=False)
    assert type(G) is mpgan_amd.MPGenerator and type(D) is mpgan_amd.MPDiscriminator
    Gg = setup_training.setup_gapt(args, gen=True)
    Dg = setup_training.setup_gapt(args, gen=False)
    assert type(Gg) is mpgan_amd.GAPT_G and type(Dg) is mpgan_amd.GAPT_D
    ck = os.path.join(REF, "trained_models", "mp_g", "G_best_epoch.pt")
    if not os.path.isfile(ck):
        cands = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(REF, "trained_models", "mp_g")) for f in fs
                 if f.endswith(".pt")]
        assert cands, "no published generator checkpoint found"
        ck = cands[0]
    res = G.load_state_dict(torch.load(ck, map_location="cpu"))
    assert not res.missing_keys and not res.unexpected_keys


class _Args:
    device = "cpu"
    num_hits = 30
    aug_r90 = aug_f = aug_t = aug_s = True
    translate_ratio = 0.125
    translate_pn_ratio = 0.05
    scale_sd = 0.125
    mask_real_only = False
    mask_exp = False


@needs_ref
def test_augment_and_mask_manual_match_reference():
    import importlib.util
    from mpgan_amd.mpgan import augment as ours, mask_manual as ours_mask

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ref_aug = load("_ref_augment", os.path.join(REF, "mpgan", "augment.py"))
    ref_mask = load("_ref_mask_utils", os.path.join(REF, "mpgan", "mask_utils.py"))
    args = _Args()
    X = torch.randn(16, 30, 3) * 0.3
    for p in (0.3, 0.8, 1):
        torch.manual_seed(11)
        a = ours.augment(args, X.clone(), p)
        torch.manual_seed(11)
        b = ref_aug.augment(args, X.clone(), p)
        assert torch.allclose(a, b, atol=1e-6), p
    for fn in ("rand_flip", "rand_90_rotation", "rand_translate", "rand_translate_per_node", "rand_scale"):
        torch.manual_seed(5)
        a = getattr(ours, fn)(args, X.clone())
        torch.manual_seed(5)
        b = getattr(ref_aug, fn)(args, X.clone())
        assert torch.allclose(a, b, atol=1e-6), fn
    for flags in ({}, {"mask_exp": True}, {"mask_real_only": True}):
        a2 = _Args()
        for k, v in flags.items():
            setattr(a2, k, v)
        assert torch.equal(ours_mask(a2, X, -0.2), ref_mask.mask_manual(a2, X, -0.2)), flags


def test_augment_without_reference():
    """Shape / invariants on their own (runs everywhere)."""
    from mpgan_amd.mpgan import augment as A, mask_manual
    args = _Args()
    torch.manual_seed(0)
    X = torch.randn(8, 30, 4)
    Y = A.augment(args, X, 0.5)
    assert Y.shape == X.shape and torch.equal(Y[..., 2:], X[..., 2:])      # pT and mask untouched
    assert torch.equal(A.augment(args, X, 1), X)                           # the reference's p == 1 shortcut
    r = A.rand_90_rotation(args, X)
    assert torch.allclose((r[..., :2] ** 2).sum(-1), (X[..., :2] ** 2).sum(-1), atol=1e-5)
    m = mask_manual(args, X[..., :3], 0.0)
    assert m.shape == (8, 30, 4) and set(m[..., 3].unique().tolist()) <= {-0.5, 0.5}

"""CPU: the C-ABI library builds/loads here, exports every symbol include/mpgan_amd.h declares, its
struct layouts match the ctypes mirrors, and the host-side logic behaves (no compute calls)."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mpgan_amd.h")


def _declared():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"^int\s+(mpg_\w+)\s*\(", txt, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from mpgan_amd import _lib
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names  # ctypes table and header list the same entry points


def test_struct_layouts_match_header():
    """sizeof of every struct as gcc sees the header == ctypes.sizeof of the Python mirror."""
    from mpgan_amd import _lib
    structs = ["MpgGemm", "MpgEdgeFwd", "MpgEdgeBwd", "MpgEdgeDw", "MpgAttn", "MpgPackJob", "MpgChainLayer", "MpgChain", "MpgReduceJob",
               "MpgDiscHead", "MpgMab", "MpgMabChain", "MpgBridge", "MpgBridgeBwd"]
    src = '#include <stdio.h>\n#include "mpgan_amd.h"\nint main(){' + "".join(
        f'printf("{s} %zu\\n", sizeof({s}));' for s in structs) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.dirname(HEADER), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for s in structs:
        assert int(sizes[s]) == ctypes.sizeof(getattr(_lib, s)), s


def test_header_constants_match_their_python_mirrors():
    """The limits the header defines and the kernels' LDS plans imply are the ones ops.py plans its launches with."""
    from mpgan_amd import ops
    txt = open(HEADER).read()
    defs = {k: int(v) for k, v in re.findall(r"^#define\s+(MPG_\w+)\s+(\d+)\s*$", txt, flags=re.M)}
    assert defs["MPG_GROUP_MAX"] == ops.GROUP_MAX and defs["MPG_PACK_MAX_JOBS"] == ops.PACK_MAX
    assert defs["MPG_EDGE_SCALARS"] == ops.EDGE_SCALARS
    from mpgan_amd import _lib
    assert defs["MPG_MAB_CHAIN_MAX"] == _lib.MAB_CHAIN_MAX and defs["MPG_FN_NA"] if "MPG_FN_NA" in defs else True
    csrc = os.path.join(ROOT, "mpgan_amd", "csrc")
    bwd, fwd = open(os.path.join(csrc, "edge_bwd2_impl.h")).read(), open(os.path.join(csrc, "edge_fwd2_impl.h")).read()
    lim = lambda src, name: int(re.search(r"constexpr int " + name + r" = (\d+);", src).group(1))
    assert lim(bwd, "B2_LIST_MAX") == ops.MAX_CHUNK_SENDERS <= lim(fwd, "F2_LIST_MAX")
    assert lim(bwd, "B2_LIST_MAX_Q") == ops.MAX_CHUNK_SENDERS_ES


def test_host_helpers():
    from mpgan_amd import ops
    assert ops.drop_params(0.0) == (0, 1.0)
    assert ops.drop_params(0.5) == (128, 2.0)
    thr, sc = ops.drop_params(0.3)
    assert thr == 77 and abs(sc - 256.0 / 179.0) < 1e-12
    with pytest.raises(ValueError):
        ops.drop_params(0.9999)
    assert ops._sender_chunks(256, 30) == 1        # every CU already has a workgroup
    assert ops._sender_chunks(512, 30) == 1
    assert ops._sender_chunks(16, 150) == 3        # 240 workgroups: one round of the 256 CUs (4 chunks would run two)
    assert ops._sender_chunks(1, 1) == 1
    for B, N in ((1, 150), (3, 40), (16, 150), (2, 1000)):   # chunks never exceed what mpg_edge_bwd's sender list holds
        sc = ops._sender_chunks(B, N)
        assert -(-N // sc) <= ops.MAX_CHUNK_SENDERS
    assert ops.mab_fusable(64, 4, 30, 30) and ops.mab_fusable(32, 2, 1, 30) and ops.mab_fusable(64, 4, 10, 30)
    assert ops.mab_fusable(64, 4, 150, 150) and ops.mab_fusable(64, 4, 1, 160) and not ops.mab_fusable(64, 4, 161, 161)   # (large-set kernels: 5 tiles of 32)
    assert not ops.mab_fusable(64, 8, 30, 30) and not ops.mab_fusable(128, 8, 30, 30)
    a, b = ops.next_tag(), ops.next_tag()
    assert b - a == 8


def test_modules_build_on_cpu_with_reference_state_dict_keys():
    import json
    from mpgan_amd import train
    from oracle import train_ref as T
    G, D = train.default_mpgan(30, device="cpu")
    Gg, Dg = train.default_gapt(30, device="cpu")
    m = json.load(open(os.path.join(ROOT, "tests", "golden", "manifests.json")))
    assert {k: list(v.shape) for k, v in G.state_dict().items()} == m["mpgan_G"]
    assert {k: list(v.shape) for k, v in D.state_dict().items()} == m["mpgan_D"]
    assert {k: list(v.shape) for k, v in Gg.state_dict().items()} == m["gapt_G"]
    assert {k: list(v.shape) for k, v in Dg.state_dict().items()} == m["gapt_D"]
    assert list(Dg.state_dict().keys()) == list(m["gapt_D"].keys()) or sorted(Dg.state_dict()) == sorted(m["gapt_D"])
    # reference-format state dicts load (strict)
    G.load_state_dict(T.init_state_dict(T.mpgan_param_shapes(True), 0))
    Dg.load_state_dict(T.init_state_dict(T.gapt_param_shapes(False), 0))


def test_product_has_no_cpu_path_and_says_so():
    from mpgan_amd import train
    G, _ = train.default_mpgan(30, device="cpu")
    x = torch.randn(2, 30, 32)
    labels = torch.full((2, 1), 0.5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        G(x, labels)


def test_unsupported_options_raise():
    from mpgan_amd.mpgan import MPLayer, LinearNet
    from mpgan_amd.gapt import MAB
    with pytest.raises(NotImplementedError):   # an input column the reference's own forward never fills
        MPLayer(32, [96, 160, 192], [256, 256], 32, int_diffs=True)
    bn_sn = LinearNet([8, 8], input_size=4, batch_norm=True, spectral_norm=True)   # both normalisations exist (un-fused)
    assert not bn_sn.plain and "bn.0.running_mean" in bn_sn.state_dict() and "net.0.module.weight_bar" in bn_sn.state_dict()
    assert not MPLayer(32, [96, 160, 192], [256, 256], 32, batch_norm=True).fused
    # the default configuration (and its k-NN form) is on the fused kernels ...
    assert MPLayer(32, [96, 160, 192], [256, 256], 32).fused
    assert MPLayer(32, [96, 160, 192], [256, 256], 32, fully_connected=False, num_knn=10).fused
    # ... and so are the edge-feature / conditioning options that are a scalar per edge each (at most ops.EDGE_SCALARS) or
    # fold into the node terms; every other combination builds the reference's layer shapes and takes the un-fused route
    for kw, fe_in, fn_in, fused in ((dict(pos_diffs=True), 65, 224, True),
                                    (dict(pos_diffs=True, all_ef=False, delta_coords=True), 67, 224, True),
                                    (dict(clabels=1, mask_fne_np=True), 66, 226, True),
                                    (dict(pos_diffs=True, clabels=1, mask_fne_np=True), 67, 226, False),        # three scalars
                                    (dict(pos_diffs=True, fully_connected=False, num_knn=5), 65, 224, True),    # k-NN distances
                                    (dict(clabels=1, fully_connected=False, num_knn=5), 65, 225, False)):       # tiled over rank rows
        m = MPLayer(32, [96, 160, 192], [256, 256], 32, **kw)
        assert m.fused == fused and m.fe.net[0].in_features == fe_in and m.fn.net[0].in_features == fn_in, kw
    m = MPLayer(16, [64, 48], [40], 8)
    assert not m.fused and [l.out_features for l in m.fe.net] == [64, 48] and [l.out_features for l in m.fn.net] == [40, 8]
    m = MAB(64, 4, layer_norm=True)            # LayerNorm is on the HIP path (ops.LayerNormFn)
    assert [k for k in m.state_dict() if "norm" in k] == ["norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"]


def test_discriminator_with_conditioning_options_has_the_reference_manifest():
    """MPDiscriminator built with the keywords setup_training.setup_mpgan passes when the conditioning options are on:
    same state-dict keys and shapes as the reference module had when the golden was made."""
    import numpy as np
    from conftest import load_golden
    from gen_golden import D_OPT
    from mpgan_amd.mpgan import MPDiscriminator
    g = load_golden("mpdisc_opt_f64.npz")
    D = MPDiscriminator(**D_OPT)
    assert [k for k in D.state_dict()] == list(g["keys"])
    assert [str(tuple(v.shape)) for v in D.state_dict().values()] == list(g["shapes"])
    assert not any(l.fused for l in D.mp_layers) and D.fused_head() is None
    # the same with two scalars per edge (distance + clabels): every layer on the fused kernels
    from gen_golden import D_OPT2
    g2 = load_golden("mpdisc_opt2_f64.npz")
    D2 = MPDiscriminator(**D_OPT2)
    assert [k for k in D2.state_dict()] == list(g2["keys"])
    assert [str(tuple(v.shape)) for v in D2.state_dict().values()] == list(g2["shapes"])
    assert all(l.fused and l.n_es == 2 for l in D2.mp_layers)


def test_product_does_not_import_the_oracle():
    import sys
    import importlib
    for mod in [m for m in sys.modules if m.startswith("mpgan_amd")]:
        src = getattr(sys.modules[mod], "__file__", None)
        if src and src.endswith(".py"):
            assert "import oracle" not in open(src).read() and "from oracle" not in open(src).read(), mod

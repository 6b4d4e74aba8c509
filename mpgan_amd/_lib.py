"""ctypes binding of libmpgan_amd.so (the C ABI declared in include/mpgan_amd.h).

The library is built in-tree (``mpgan_amd/lib/libmpgan_amd.so``) by ``build()`` with
``hipcc --offload-arch=gfx950``; there is no CPU fallback: importing the ops without the
library raises, and every entry point's non-zero return raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# (MPG_LIBDIR: another directory for the built library and its object cache -- two builds of the kernels, e.g. under different
# MPG_HIPCC_FLAGS, side by side for same-box A/B runs; the product loads mpgan_amd/lib)
LIBDIR = os.environ.get("MPG_LIBDIR") or os.path.join(_HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libmpgan_amd.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

_u64p = C.POINTER(C.c_uint64)
_fp = C.c_void_p  # device pointers travel as integers


class MpgGemm(C.Structure):
    _fields_ = [
        ("A", _fp), ("A2", _fp), ("lda", C.c_int), ("lda2", C.c_int), ("K1", C.c_int),
        ("B", _fp), ("ldb", C.c_int),
        ("C", _fp), ("ldc", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("split_stride", C.c_longlong),
        ("bias", _fp), ("out_scale", C.c_float), ("act", C.c_int), ("alpha", C.c_float),
        ("seed", _fp),
        ("drop_tag", C.c_uint32), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float),
        ("gateH", _fp), ("ldh", C.c_int), ("gate_act", C.c_int),
        ("gate_tag", C.c_uint32), ("gate_thr", C.c_uint32), ("gate_scale", C.c_float),
        ("resid", _fp), ("ldr", C.c_int),
        ("accumulate", C.c_int), ("f16", C.c_int), ("ones_col", C.c_int),
    ]


class MpgEdgeFwd(C.Structure):
    _fields_ = [
        ("a", _fp), ("c", _fp), ("ld_ac", C.c_int), ("mask", _fp),
        ("W2img", _fp), ("W3img", _fp), ("b2", _fp), ("b3", _fp),
        ("agg", _fp),
        ("B", C.c_int), ("N", C.c_int), ("SC", C.c_int),
        ("alpha", C.c_float), ("agg_scale", C.c_float),
        ("seed", _fp), ("tag_base", C.c_uint32), ("thr", C.c_uint32), ("dscale", C.c_float),
        ("skip_masked", C.c_int), ("f16", C.c_int),
        ("sign3", _fp), ("nbr", _fp), ("stageE2", _fp),
        ("es", _fp), ("wq", _fp), ("order", _fp), ("tickets", _fp),
        ("two_term", C.c_int),
    ]


class MpgEdgeBwd(C.Structure):
    _fields_ = [
        ("a", _fp), ("c", _fp), ("ld_ac", C.c_int), ("mask", _fp),
        ("dagg", _fp), ("ld_dagg", C.c_int),
        ("sign3", _fp),
        ("W2img", _fp), ("W3Timg", _fp), ("W2Timg", _fp),
        ("b2", _fp),
        ("da", _fp), ("dc", _fp),
        ("stageE2", _fp), ("stageZ2", _fp),
        ("B", C.c_int), ("N", C.c_int), ("SC", C.c_int),
        ("alpha", C.c_float), ("agg_scale", C.c_float),
        ("seed", _fp), ("tag_base", C.c_uint32), ("thr", C.c_uint32), ("dscale", C.c_float),
        ("f16", C.c_int), ("nbr", _fp), ("gexp", _fp),
        ("es", _fp), ("wq", _fp), ("des", _fp), ("daq", _fp), ("order", _fp),
    ]


class MpgEdgeDw(C.Structure):
    _fields_ = [
        ("a", _fp), ("c", _fp), ("ld_ac", C.c_int), ("mask", _fp),
        ("dagg", _fp), ("ld_dagg", C.c_int),
        ("sign3", _fp), ("stageE2", _fp), ("stageZ2", _fp),
        ("part", _fp), ("nwg", C.c_int),
        ("dW3", _fp), ("dW2", _fp), ("db3", _fp), ("db2", _fp), ("accumulate", C.c_int),
        ("B", C.c_int), ("N", C.c_int),
        ("alpha", C.c_float), ("agg_scale", C.c_float),
        ("seed", _fp), ("tag_base", C.c_uint32), ("thr", C.c_uint32), ("dscale", C.c_float),
        ("f16", C.c_int), ("nbr", _fp), ("gexp", _fp),
        ("es", _fp), ("wq", _fp), ("defer_reduce", C.c_int),
    ]


class MpgReduceJob(C.Structure):
    _fields_ = [
        ("part", _fp), ("S", C.c_int), ("N", C.c_int), ("K", C.c_int), ("has_bias", C.c_int),
        ("out", _fp), ("ldo", C.c_int), ("bias", _fp), ("accumulate", C.c_int),
    ]


class MpgPackJob(C.Structure):
    _fields_ = [
        ("W", _fp), ("ldw", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("transpose", C.c_int),
        ("scale", C.c_float), ("f16", C.c_int), ("img", _fp), ("row_split", C.c_int), ("split_cols", C.c_int),
        ("status", _fp),
    ]


class MpgChainLayer(C.Structure):
    _fields_ = [
        ("Wimg", _fp), ("bias", _fp), ("nbias", C.c_int),
        ("K", C.c_int), ("N", C.c_int), ("act", C.c_int),
        ("drop_tag", C.c_uint32), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float),
        ("gateH", _fp), ("ldh", C.c_int), ("gate_act", C.c_int),
        ("gate_tag", C.c_uint32), ("gate_thr", C.c_uint32), ("gate_scale", C.c_float),
        ("resid", _fp), ("ldr", C.c_int),
        ("out", _fp), ("ldo", C.c_int),
        ("wscale", C.c_float),
    ]


class MpgChain(C.Structure):
    _fields_ = [
        ("A", _fp), ("lda", C.c_int), ("K1", C.c_int),
        ("A2", _fp), ("lda2", C.c_int),
        ("a_slabs", C.c_int), ("a_slab_stride", C.c_uint64),
        ("in_tag", C.c_uint32), ("in_thr", C.c_uint32), ("in_scale", C.c_float),
        ("in_out", _fp), ("ld_in_out", C.c_int),
        ("M", C.c_int), ("nlayers", C.c_int), ("alpha", C.c_float), ("seed", _fp), ("f16", C.c_int),
        ("ascale", C.c_float),
        ("L", MpgChainLayer * 3),
    ]


class MpgAttn(C.Structure):
    _fields_ = [
        ("q", _fp), ("k", _fp), ("v", _fp), ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int),
        ("ignore", _fp), ("o", _fp), ("ldo", C.c_int), ("P", _fp), ("d_o", _fp),
        ("dq", _fp), ("dk", _fp), ("dv", _fp), ("lddq", C.c_int), ("lddk", C.c_int), ("lddv", C.c_int),
        ("B", C.c_int), ("L", C.c_int), ("S", C.c_int), ("H", C.c_int), ("d", C.c_int),
    ]


class MpgDiscHead(C.Structure):
    _fields_ = [
        ("y", _fp), ("ldy", C.c_int), ("mask", _fp), ("w", _fp), ("bias", _fp),
        ("B", C.c_int), ("N", C.c_int), ("F", C.c_int), ("mean", C.c_int), ("sigmoid", C.c_int),
        ("seed", _fp), ("tag", C.c_uint32), ("thr", C.c_uint32), ("dscale", C.c_float),
        ("out", _fp), ("pooled", _fp), ("aux", _fp),
        ("loss", C.c_int), ("gen_step", C.c_int), ("n_real", C.c_int), ("inv_count", C.c_float),
        ("gout", _fp), ("terms", _fp), ("loss_out", _fp),
        ("dy", _fp), ("ld_dy", C.c_int), ("dw", _fp), ("db", _fp), ("accumulate", C.c_int),
    ]


class MpgBridge(C.Structure):
    _fields_ = [
        ("x", _fp), ("ldx", C.c_int), ("W1", _fp), ("b1", _fp), ("act1", C.c_int),
        ("feat", _fp), ("ldf", C.c_int),
        ("M", C.c_int), ("row0", C.c_int), ("K", C.c_int), ("F", C.c_int), ("E", C.c_int),
        ("W2", _fp), ("b2", _fp), ("act2", C.c_int), ("alpha", C.c_float),
        ("seed", _fp), ("tag", C.c_uint32), ("thr", C.c_uint32), ("dscale", C.c_float),
        ("e", _fp), ("lde", C.c_int),
    ]


class MpgBridgeBwd(C.Structure):
    _fields_ = [
        ("ge", _fp), ("ldge", C.c_int), ("e", _fp), ("lde", C.c_int), ("feat", _fp), ("ldf", C.c_int),
        ("gfeat", _fp), ("ldgf", C.c_int), ("W1", _fp), ("W2", _fp),
        ("M", C.c_int), ("row0", C.c_int), ("K", C.c_int), ("F", C.c_int), ("E", C.c_int), ("act1", C.c_int), ("act2", C.c_int),
        ("alpha", C.c_float),
        ("seed", _fp), ("tag", C.c_uint32), ("thr", C.c_uint32), ("dscale", C.c_float),
        ("g2", _fp), ("ldg2", C.c_int), ("g1", _fp), ("ldg1", C.c_int), ("dx", _fp), ("lddx", C.c_int),
    ]


class MpgMab(C.Structure):
    _fields_ = [
        ("x", _fp), ("ldx", C.c_int), ("y", _fp), ("ldy", C.c_int), ("ignore", _fp),
        ("Win", _fp), ("bin", _fp), ("Wo", _fp), ("bo", _fp), ("Wf", _fp), ("bf", _fp),
        ("B", C.c_int), ("L", C.c_int), ("S", C.c_int), ("E", C.c_int), ("H", C.c_int),
        ("alpha", C.c_float), ("ff_act", C.c_int),
        ("seed", _fp), ("tag", C.c_uint32), ("thr_mab", C.c_uint32), ("sc_mab", C.c_float),
        ("thr_ff", C.c_uint32), ("sc_ff", C.c_float), ("wscale", C.c_float), ("ascale", C.c_float),
        ("out", _fp), ("ldo", C.c_int), ("save_o", _fp), ("save_z", _fp),
        ("WinT", _fp), ("WoT", _fp), ("WfT", _fp),
        ("dout", _fp), ("lddout", C.c_int), ("dx", _fp), ("lddx", C.c_int), ("dy", _fp), ("lddy", C.c_int),
        ("dq", _fp), ("lddq", C.c_int), ("dk", _fp), ("dv", _fp), ("lddkv", C.c_int),
        ("dza", _fp), ("du", _fp),
        ("ln1_w", _fp), ("ln1_b", _fp), ("ln2_w", _fp), ("ln2_b", _fp), ("ln_eps", C.c_float),
        ("save_za", _fp), ("dn1", _fp), ("gn1", _fp), ("dn2", _fp), ("gn2", _fp),
    ]


MPG_FN_NA = -100   # mpg_edge_fwd_fn: "not one of my shapes" (include/mpgan_amd.h)

MAB_CHAIN_MAX = 4   # MPG_MAB_CHAIN_MAX of include/mpgan_amd.h


class MpgMabChain(C.Structure):
    _fields_ = [("blk", MpgMab * MAB_CHAIN_MAX), ("n", C.c_int)]


# name -> (restype, argtypes); kept in step with include/mpgan_amd.h (tests check the symbol list)
SIGNATURES = {
    "mpg_gemm": (C.c_int, [C.POINTER(MpgGemm), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mpg_splitk_reduce": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_void_p]),
    "mpg_gemm_wgrad_group": (C.c_int, [C.POINTER(MpgGemm), C.POINTER(C.c_int), C.c_int, C.c_void_p]),
    "mpg_splitk_reduce_group": (C.c_int, [C.POINTER(MpgReduceJob), C.c_int, C.c_void_p]),
    "mpg_splitk_reduce_group_dw": (C.c_int, [C.POINTER(MpgReduceJob), C.c_int, C.POINTER(MpgEdgeDw), C.c_void_p]),
    "mpg_slab_sums": (C.c_int, [_fp, C.c_int, C.c_uint64, _fp, C.c_int, C.c_uint64, _fp, C.c_int, C.c_int, C.c_void_p]),
    "mpg_gate": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                           _fp, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]),
    "mpg_dropout_mask": (C.c_int, [_fp, C.c_uint64, C.c_int, _fp, C.c_uint32, C.c_uint32, C.c_void_p]),
    "mpg_pack_weights": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _fp, C.c_void_p]),
    "mpg_pack_many": (C.c_int, [C.POINTER(MpgPackJob), C.c_int, C.c_void_p]),
    "mpg_chain": (C.c_int, [C.POINTER(MpgChain), C.c_void_p]),
    "mpg_edge_fwd": (C.c_int, [C.POINTER(MpgEdgeFwd), C.c_void_p]),
    "mpg_edge_fwd_fn": (C.c_int, [C.POINTER(MpgEdgeFwd), C.POINTER(MpgChain), C.POINTER(MpgChain), C.c_void_p]),
    "mpg_edge_bwd": (C.c_int, [C.POINTER(MpgEdgeBwd), C.c_void_p]),
    "mpg_edge_bwd_fn": (C.c_int, [C.POINTER(MpgEdgeBwd), C.POINTER(MpgChain), C.POINTER(MpgChain), C.c_void_p]),
    "mpg_edge_dw": (C.c_int, [C.POINTER(MpgEdgeDw), C.c_void_p]),
    "mpg_attn_fwd": (C.c_int, [C.POINTER(MpgAttn), C.c_void_p]),
    "mpg_attn_bwd": (C.c_int, [C.POINTER(MpgAttn), C.c_void_p]),
    "mpg_mab_fwd": (C.c_int, [C.POINTER(MpgMab), C.c_void_p]),
    "mpg_mab_bwd": (C.c_int, [C.POINTER(MpgMab), C.c_void_p]),
    "mpg_mab_chain_fwd": (C.c_int, [C.POINTER(MpgMabChain), C.c_void_p]),
    "mpg_knn_sets": (C.c_int, [_fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_void_p]),
    "mpg_rank_mask": (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_void_p]),
    "mpg_jet_order": (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_void_p]),
    "mpg_gen_tail_fwd": (C.c_int, [_fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mpg_gen_tail_bwd": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mpg_disc_head_fwd": (C.c_int, [C.POINTER(MpgDiscHead), C.c_void_p]),
    "mpg_disc_head_bwd": (C.c_int, [C.POINTER(MpgDiscHead), C.c_void_p]),
    "mpg_disc_head_loss": (C.c_int, [C.POINTER(MpgDiscHead), C.c_void_p]),
    "mpg_layernorm_fwd": (C.c_int, [_fp, C.c_int, _fp, _fp, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "mpg_layernorm_bwd": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p]),
    "mpg_batchnorm_stats": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, C.c_void_p]),
    "mpg_batchnorm_apply": (C.c_int, [_fp, C.c_int, _fp, _fp, _fp, _fp, C.c_float, _fp, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mpg_batchnorm_bwd": (C.c_int, [_fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, C.c_float, _fp, C.c_int, _fp, _fp, C.c_int, _fp, _fp,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mpg_rmsprop": (C.c_int, [_fp, _fp, _fp, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _fp, C.c_uint64, C.c_void_p]),
    "mpg_adam": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                           C.c_float, C.c_int, _fp, C.c_uint64, C.c_void_p]),
    "mpg_adadelta": (C.c_int, [_fp, _fp, _fp, _fp, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _fp, C.c_uint64,
                               C.c_void_p]),
    "mpg_bridge_fwd": (C.c_int, [C.POINTER(MpgBridge), C.c_void_p]),
    "mpg_bridge_bwd": (C.c_int, [C.POINTER(MpgBridgeBwd), C.c_void_p]),
    "mpg_normal": (C.c_int, [_fp, C.c_uint64, _fp, C.c_uint32, C.c_float, C.c_float, C.c_void_p]),
    "mpg_normal_rank_mask": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_uint32, C.c_float, C.c_float, _fp, C.c_int, _fp, _fp,
                                       C.c_void_p]),
}


# -fno-slp-vectorize: packed fp32 VALU ops (v_pk_add/mul/fma_f32) that the SLP vectoriser forms are slower than the
# scalar ops beside MFMAs on gfx950 (MI355X_MICROARCH.md)
EXTRA_FLAGS = os.environ.get("MPG_HIPCC_FLAGS", "-fno-slp-vectorize").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


DIGEST_FILE = os.path.join(LIBDIR, "libmpgan_amd.source_digest")


def _stale():
    """Is the library older than what it is compiled from?  By CONTENT where the build left its sources' digest beside the library
    (a checkout or a copy that resets modification times must not make a fresh library look stale inside a multi-rank job, where
    nothing may be compiled); by modification time otherwise."""
    if not os.path.isfile(LIBPATH):
        return True
    if os.path.isfile(DIGEST_FILE):
        try:
            with open(DIGEST_FILE) as f:
                return f.read().strip() != source_digest()
        except OSError:
            pass
    t = os.path.getmtime(LIBPATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _includes(path, seen):
    """Transitive closure of the quoted includes of ``path`` (resolved beside the including file, then in INCLUDE)."""
    import re
    if path in seen:
        return
    seen.add(path)
    with open(path, "r", errors="replace") as f:
        text = f.read()
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        for root in (os.path.dirname(path), INCLUDE):
            cand = os.path.normpath(os.path.join(root, inc))
            if os.path.isfile(cand):
                _includes(cand, seen)
                break


_compiler_id_cache = {}


def _compiler_id(hipcc: str) -> str:
    """What identifies the compiler behind ``hipcc`` for the object cache: its resolved path and its own ``--version`` text (ROCm
    / clang version, install directory); if that cannot be run, the binary's size and mtime.  A ROCm upgrade or another
    HIPCC must not link objects of the old compiler into the library."""
    if hipcc not in _compiler_id_cache:
        real = os.path.realpath(hipcc)
        try:
            ver = subprocess.run([hipcc, "--version"], check=True, capture_output=True, text=True, env=_compiler_env(),
                                 timeout=60).stdout
        except Exception:   # noqa: BLE001 -- the fallback still tells two installs apart
            st = os.stat(real) if os.path.exists(real) else None
            ver = "unversioned %s %s" % ((st.st_size, int(st.st_mtime)) if st else ("?", "?"))
        _compiler_id_cache[hipcc] = real + "\n" + ver
    return _compiler_id_cache[hipcc]


def _digest(src, flags, hipcc=None) -> str:
    import hashlib
    deps = set()
    _includes(os.path.abspath(src), deps)
    h = hashlib.sha1(" ".join(flags).encode())
    if hipcc is not None:
        h.update(b"\0" + _compiler_id(hipcc).encode() + b"\0")
    for d in sorted(deps):
        with open(d, "rb") as f:
            h.update(os.path.relpath(d, CSRC).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def source_digest() -> str:
    """Digest of everything the library is compiled from (csrc/*.hip, csrc/*.h, include/*.h): what a measurement taken on the
    built library belongs to.  ``profiles/hbm_traffic.json`` carries the digest of the tree its counters were collected on;
    ``bench.py`` reports a measured ``roofline.traffic`` only while it still equals this one."""
    import hashlib
    h = hashlib.sha1()
    for d in sorted(sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))):
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def _compiler_env():
    """Environment for the hipcc children: without a profiler's preload.  ``rocprofv3`` injects a library that
    initialises the GPU in every process it is loaded into; hipcc then execs clang -- a GPU-initialised exec, which
    this pool forbids (it takes the machine down)."""
    env = dict(os.environ)
    for k in list(env):
        if k == "LD_PRELOAD" or k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTRACER")):
            env.pop(k)
    return env


def _under_profiler() -> bool:
    pre = os.environ.get("LD_PRELOAD", "").lower()
    return any(t in pre for t in ("rocprof", "roctracer", "rocp_")) or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def _note_digest():
    """Leave the sources' digest beside a library that is fresh by modification time and has none yet (a library built before
    the digest existed): from then on staleness is judged by content."""
    if not os.path.isfile(DIGEST_FILE):
        try:
            tmp = DIGEST_FILE + ".%d.tmp" % os.getpid()
            with open(tmp, "w") as f:
                f.write(source_digest() + "\n")
            os.replace(tmp, DIGEST_FILE)
        except OSError:   # (a read-only tree: the modification-time rule stays)
            pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into mpgan_amd/lib/libmpgan_amd.so (in-tree).

    Safe against concurrent callers (ranks of one torchrun job, parallel test workers): an exclusive file lock is
    held for the whole build, objects and the library are written under temporary names and moved into place."""
    if not force and not _stale():
        _note_digest()
        return LIBPATH
    import fcntl
    import tempfile
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():  # another process built it while this one waited
            return LIBPATH
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        env = _compiler_env()
        tmpdir = tempfile.mkdtemp(prefix=".build.", dir=LIBDIR)
        # Object cache (LIBDIR/obj, git-ignored with the library): an object is reused when the digest of its source, of
        # every header it includes (transitively, quoted includes), of the compiler flags and of the compiler's identity
        # (_compiler_id) is unchanged -- editing one kernel
        # recompiles its translation units only.  ``force`` (the driver's "does it build" check) ignores the cache.
        objdir = os.path.join(LIBDIR, "obj")
        os.makedirs(objdir, exist_ok=True)
        try:
            jobs, objs = [], []
            flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + EXTRA_FLAGS
            for src in sources():
                base = os.path.basename(src)[:-4]
                obj = os.path.join(objdir, "%s.%s.o" % (base, _digest(src, flags, hipcc)))
                objs.append(obj)
                if force or not os.path.isfile(obj):
                    tmpobj = os.path.join(tmpdir, base + ".o")
                    jobs.append((tmpobj, [hipcc] + flags + ["-I", INCLUDE, "-c", src, "-o", tmpobj], obj, base))
            # the translation units are independent (one takes over a minute): compile them side by side
            from concurrent.futures import ThreadPoolExecutor

            def run(job):
                if verbose:
                    print(" ".join(job[1]))
                subprocess.run(job[1], check=True, env=env)
                for old in glob.glob(os.path.join(objdir, job[3] + ".*.o")):   # superseded objects of this unit
                    os.remove(old)
                os.replace(job[0], job[2])
            if jobs:
                with ThreadPoolExecutor(max_workers=max(1, min(os.cpu_count() or 4, 8, len(jobs)))) as ex:
                    list(ex.map(run, jobs))
            tmplib = os.path.join(tmpdir, "libmpgan_amd.so")
            subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmplib] + objs, check=True, env=env)
            os.replace(tmplib, LIBPATH)
            with open(os.path.join(tmpdir, "digest"), "w") as f:
                f.write(source_digest() + "\n")
            os.replace(os.path.join(tmpdir, "digest"), DIGEST_FILE)
        finally:
            import shutil
            shutil.rmtree(tmpdir, ignore_errors=True)
    return LIBPATH


_lib = None


def lib():
    """The loaded library.  Raises if it is absent.  A stale or missing library is rebuilt on first use ONLY in a
    plain single process: under a multi-rank launch (WORLD_SIZE > 1) or a profiler preload nothing is compiled
    here -- run ``python __graft_entry__.py`` (build only) first."""
    global _lib
    if _lib is None:
        # (MPG_LIB_STALE_OK=1 beside MPG_LIBDIR: load that directory's library as it is -- an A/B against a build of OLDER sources)
        if _stale() and not (os.environ.get("MPG_LIB_STALE_OK") == "1" and os.environ.get("MPG_LIBDIR") and os.path.isfile(LIBPATH)):
            can_build = os.path.isfile(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) and os.path.isdir(CSRC)
            guarded = int(os.environ.get("WORLD_SIZE", "1")) > 1 or _under_profiler()
            if can_build and not guarded:
                build()
            elif not os.path.isfile(LIBPATH):
                raise RuntimeError("libmpgan_amd.so is missing%s: run `python __graft_entry__.py` (build) first"
                                   % (" and this process must not compile (multi-rank launch or profiler preload)"
                                      if can_build else " and hipcc is not available"))
            elif can_build:
                raise RuntimeError("libmpgan_amd.so is older than its sources and this process must not compile "
                                   "(multi-rank launch or profiler preload): run `python __graft_entry__.py` first")
        _lib = C.CDLL(LIBPATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}")

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) where no GPU is visible
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def summarize(name, t):
    """Same summary as tests/gen_golden.py: sum, l2, 64 sampled entries."""
    import zlib
    flat = t.detach().double().reshape(-1).cpu()
    rs = np.random.RandomState(zlib.crc32(name.encode()) % (2**31))
    idx = rs.randint(0, flat.numel(), size=64)
    return np.concatenate([[flat.sum().item(), flat.norm().item()], flat[idx].numpy()])


def rel_err(a, b):
    """max|a-b| / max|b|  -- the parity metric used throughout (per tensor)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))

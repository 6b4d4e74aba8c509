"""GPU parity of the attention core (mpg_attn_fwd / mpg_attn_bwd) on its own, against an fp64 restatement of what
nn.MultiheadAttention computes between its projections as MAB uses it (gapt/model.py:107-129): per (jet, head)
softmax(q k^T / sqrt(d) + key-padding mask) v.  Covers the one-wave-per-(jet, head) fast path (S <= 32, L <= 64,
d in {8, 16, 32}) and the generic kernels behind it.  Tolerance 1e-3 relative (north star), asserted at 1e-4."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
TIGHT = 1e-4


def _ref(q, k, v, ignore, B, L, S, H):
    E = q.shape[1]
    d = E // H
    qh = q.double().reshape(B, L, H, d).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, S, H, d).permute(0, 2, 1, 3)
    vh = v.double().reshape(B, S, H, d).permute(0, 2, 1, 3)
    sc = qh @ kh.transpose(2, 3) / d ** 0.5
    if ignore is not None:
        sc = sc.masked_fill(ignore.reshape(B, 1, 1, S) != 0, float("-inf"))
    o = torch.softmax(sc, -1) @ vh
    return o.permute(0, 2, 1, 3).reshape(B * L, E)


@pytest.mark.parametrize("B,L,S,H,d,masked", [
    (3, 30, 30, 4, 16, True), (5, 10, 30, 4, 16, True), (5, 30, 10, 4, 16, False), (7, 1, 30, 4, 16, True),
    (2, 64, 32, 2, 8, False), (2, 17, 5, 3, 32, True),          # fast path, other head sizes
    (2, 150, 150, 4, 16, True), (2, 40, 33, 4, 16, False), (2, 30, 30, 4, 12, False),   # generic kernels
])
def test_attention_core(B, L, S, H, d, masked):
    from mpgan_amd import ops
    rs = np.random.RandomState(B + L + S + H + d)
    dev = torch.device("cuda:0")
    E = H * d
    mk = lambda n: torch.from_numpy(rs.normal(size=(n, E))).float().to(dev)
    q, k, v, go = mk(B * L), mk(B * S), mk(B * S), mk(B * L)
    ignore = None
    if masked:
        ig = (rs.uniform(size=(B, S)) < 0.3).astype(np.float32)
        ig[:, 0] = 0  # at least one real key per jet
        ignore = torch.from_numpy(ig.reshape(-1)).to(dev)
    qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))
    o = ops.FusedAttnFn.apply(qg, kg, vg, ignore, B, L, S, H)
    o.backward(go)
    q64, k64, v64 = (t.double().clone().requires_grad_(True) for t in (q, k, v))
    ro = _ref(q64, k64, v64, ignore, B, L, S, H)
    ro.backward(go.double())
    assert rel_err(o.detach().cpu().numpy(), ro.detach().cpu().numpy()) < TIGHT
    for got, ref in ((qg.grad, q64.grad), (kg.grad, k64.grad), (vg.grad, v64.grad)):
        assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < TIGHT


def test_attention_strided_views():
    """q, k, v as column slices of one [rows, 3E] projection output (how MAB feeds them), fast path."""
    from mpgan_amd import ops
    rs = np.random.RandomState(0)
    dev = torch.device("cuda:0")
    B, L, H, d = 4, 30, 4, 16
    E = H * d
    qkv = torch.from_numpy(rs.normal(size=(B * L, 3 * E))).float().to(dev)
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    o = ops.FusedAttnFn.apply(q, k, v, None, B, L, L, H)
    ro = _ref(q, k, v, None, B, L, L, H)
    assert rel_err(o.cpu().numpy(), ro.cpu().numpy()) < TIGHT

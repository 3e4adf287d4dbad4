"""CPU: pins oracle/torch_ref.py::attention_bf16 (the restatement with the kernels' two bf16 rounding points, used by the
GPU parity tests at 1e-3) to oracle/torch_ref.py::attention (the fp32 restatement of the reference's softmax attention, itself
pinned to the real reference modules through the golden fixtures): same function up to the roundings of P and dS."""

import pytest
import torch

from oracle import torch_ref as R


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def _inputs(B, H, Lq, Lk, seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: R.bf16_round(torch.randn(*s, generator=g))
    return mk(B, H, Lq, 64), mk(B, H, Lk, 64), mk(B, H, Lk, 64), mk(B, H, Lq, 64)


@pytest.mark.parametrize("masked,p", [(False, 0.0), (True, 0.0), (True, 0.1)])
def test_faithful_oracle_is_the_fp32_oracle_up_to_two_roundings(masked, p):
    q, k, v, do = _inputs(2, 3, 70, 70, 1)
    mask = torch.full((70, 70), -float("inf")).triu(1) if masked else None
    drop = (p, (11, 22)) if p > 0 else None
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref = R.attention(qr, kr, vr, mask=mask, drop=drop)
    o_ref.backward(do)
    o, lse, dq, dk, dv = R.attention_bf16(q, k, v, mask=mask, drop=drop, dout=do)
    # bf16 P: ~2e-3 rel-L2 on random data (what tests/gpu_checks.py used to allow the kernels: 3e-3 / 4e-3)
    assert _rel(o, o_ref.detach()) < 3e-3
    assert _rel(dq, qr.grad) < 4e-3 and _rel(dk, kr.grad) < 4e-3 and _rel(dv, vr.grad) < 4e-3
    # ... and it is NOT the fp32 result merely rounded at the end: the roundings are really there
    assert _rel(o, R.bf16_round(o_ref.detach())) > 3e-4
    # lse is the fp32 log-sum-exp
    s = (q @ k.transpose(-1, -2)) / 8.0
    if mask is not None:
        s = s + mask
    assert torch.allclose(lse, torch.logsumexp(s, dim=-1), atol=2e-5, rtol=1e-5)


def test_faithful_forward_does_not_depend_on_the_key_order():
    """the integer running maximum makes the rounded P independent of intermediate maxima: permuting the keys (= visiting
    the key tiles in another order) changes nothing but the fp32 summation order"""
    q, k, v, _ = _inputs(1, 2, 40, 96, 3)
    perm = torch.randperm(96, generator=torch.Generator().manual_seed(0))
    o1, l1 = R.attention_bf16(q, k, v)
    o2, l2 = R.attention_bf16(q, k[:, :, perm], v[:, :, perm])
    assert _rel(o2, o1) < 1e-4 and torch.allclose(l1, l2, atol=1e-5)
    # chunked evaluation with a running INTEGER maximum (what the kernel does) == one-shot evaluation, P bit for bit
    s2 = (q @ k.transpose(-1, -2)) * (0.125 * R.LOG2E)
    m_run = torch.full(s2.shape[:-1] + (1,), -float("inf"))
    acc = torch.zeros(1, 2, 40, 64)
    for c in range(0, 96, 32):
        blk = s2[..., c:c + 32]
        m_new = torch.maximum(m_run, torch.ceil(blk.amax(-1, keepdim=True)))
        acc = acc * torch.exp2(m_run - m_new)
        acc = acc + R.bf16_round(torch.exp2(blk - m_new)) @ v[:, :, c:c + 32]
        m_run = m_new
    m_fin = torch.ceil(s2.amax(-1, keepdim=True))
    one_shot = R.bf16_round(torch.exp2(s2 - m_fin)) @ v
    assert torch.equal(m_run, m_fin) and _rel(acc, one_shot) < 1e-6


def test_fully_masked_rows_and_ragged_keys():
    q, k, v, do = _inputs(1, 1, 5, 9, 4)
    mask = torch.zeros(5, 9)
    mask[2] = -float("inf")                  # a row that sees nothing
    o, lse, dq, dk, dv = R.attention_bf16(q, k, v, mask=mask, dout=do)
    assert torch.isfinite(dq).all() and torch.isfinite(dk).all() and torch.isfinite(dv).all()
    assert float(dq[0, 0, 2].abs().max()) == 0.0


def test_attention_dropout_generator_statistics():
    """oracle/torch_ref.py::attn_drop_keep_mask (= csrc/common.h drop_tilekey / drop_rot / drop_elem, round 4: one hash per
    (score row, 32-key tile) and one 24-bit multiply-add per element instead of one hash per element): the keep rate is 1 - p per
    column and per row, and the 32 elements that share a tile key are pairwise uncorrelated (the multipliers are fixed odd 24-bit
    constants, half of the elements use the key rotated by 12 bits)."""
    import torch
    from oracle import torch_ref as R
    n_rows, n_cols = 20000, 384
    rows = torch.arange(n_rows, dtype=torch.int64)[:, None]
    cols = torch.arange(n_cols, dtype=torch.int64)[None, :]
    for p in (0.1, 0.5):
        keep = R.attn_drop_keep_mask((77, 4242), rows, cols, p)
        d = (~keep).double()
        assert abs(float(d.mean()) - p) < 2e-3
        assert float((d.mean(0) - p).abs().max()) < 5.0 * (p * (1 - p) / n_rows) ** 0.5          # every column (= element slot)
        assert abs(float(d.mean(1).std()) - (p * (1 - p) / n_cols) ** 0.5) < 1e-3               # rows scatter binomially
        t = (d - d.mean(0, keepdim=True)).view(n_rows, n_cols // 32, 32)
        cm = torch.einsum("ntj,ntk->jk", t, t) / (n_rows * (n_cols // 32)) / (p * (1 - p))
        off = cm - torch.diag(torch.diag(cm))
        assert float(off.abs().max()) < 5.0 / (n_rows * (n_cols // 32)) ** 0.5, float(off.abs().max())
        # triples inside one accumulator quad (consecutive multipliers of the same key)
        a, b = d[:, 0::32] * d[:, 1::32], d[:, 2::32]
        assert abs(float((a * b).sum() / a.sum()) - p) < (0.03 if p < 0.5 else 0.01)
    # a different seed / row block gives a different mask
    k1 = R.attn_drop_keep_mask((77, 4242), rows[:64], cols, 0.1)
    k2 = R.attn_drop_keep_mask((78, 4242), rows[:64], cols, 0.1)
    assert float((k1 != k2).double().mean()) > 0.1

"""CPU checks of the host-side wiring around kernels that only run on the GPU: the product path refuses CPU tensors
(no fallback), and the autograd plumbing of ops.assemble_tokens routes gradients as cat + broadcast would (the kernel
launch itself is stubbed out here; its output is checked bit for bit on the GPU in tests/gpu_checks.py)."""
import pytest
import torch

from dreamvla_amd import losses, ops
from dreamvla_amd._lib import DvlaError


def test_fused_losses_refuse_cpu_tensors():
    n, S = 2, 1
    out = (torch.rand(()), torch.rand(()), torch.randn(n, 2, 1, 196, 768), None, None, None, None, None, None, None)
    batch = {"image_primary": torch.randn(n, 4, 3, 224, 224), "image_wrist": torch.randn(n, 4, 3, 224, 224)}
    with pytest.raises(TypeError):
        losses.calvin_losses(out, batch, sequence_length=S, use_dit_head=True, fused=True)
    total, parts = losses.calvin_losses(out, batch, sequence_length=S, use_dit_head=True)      # fused=None: ATen checker path
    assert torch.isfinite(total) and float(parts["image"]) > 0


def test_assemble_tokens_refuses_cpu_tensors():
    with pytest.raises((DvlaError, TypeError)):
        ops.assemble_tokens([torch.zeros(1, 2, 3, 8, dtype=torch.bfloat16)], None)


def test_assemble_tokens_gradient_routing(monkeypatch):
    class FakeLib:
        @staticmethod
        def dvla_assemble_tokens(*a):
            return 0
    monkeypatch.setattr(ops._lib, "load", lambda: FakeLib)
    monkeypatch.setattr(ops, "_req", lambda t, name, dtype=None: t)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    B, S, H = 2, 3, 8
    BF = torch.bfloat16
    text = torch.randn(B, 1, 1, H).to(BF).requires_grad_(True)
    img = torch.randn(B, S, 4, H).to(BF).requires_grad_(True)
    tok = torch.randn(1, 1, 5, H).to(BF).requires_grad_(True)
    pos = torch.randn(1, S, 1, H).to(BF).requires_grad_(True)
    y = ops.assemble_tokens([text.expand(B, S, 1, H), img, tok.expand(B, S, -1, -1)], pos)
    assert y.shape == (B, S, 10, H)
    g = torch.randn(B, S, 10, H).to(BF)
    y.backward(g)
    gf = g.float()
    assert torch.allclose(text.grad.float(), gf[:, :, 0:1].sum(1, keepdim=True), atol=0.05)
    assert torch.equal(img.grad, g[:, :, 1:5])
    assert torch.allclose(tok.grad.float(), gf[:, :, 5:10].sum((0, 1), keepdim=True), atol=0.1)
    assert torch.allclose(pos.grad.float(), gf.sum((0, 2)).reshape(1, S, 1, H), atol=0.1)


def test_auto_split_k_picks_the_measured_minima():
    """ops.auto_split_k against the round-6 sweep on an MI355X (profiles/r06_small_dw_sweep.txt and the large-shape sweep of the same
    probe): for the step's weight gradients the heuristic's slice count is the measured minimum (or within 3 % of it), and the
    invariants it is built on hold -- no split for outputs that fill the chip, slices of at least ~1.3 k of K, at most 16."""
    from dreamvla_amd.ops import auto_split_k
    measured_min = {(1024, 3072, 20832): 5, (1024, 4096, 20832): 4, (4096, 1024, 20832): 4, (1024, 1024, 20832): 16,
                    (3072, 1024, 91840): 5, (1024, 4096, 91840): 4, (1024, 1024, 91840): 16, (2304, 768, 10752): 8}
    for (M, N, K), sk in measured_min.items():
        assert auto_split_k(M, N, K) == sk, (M, N, K, auto_split_k(M, N, K))
    # within 3 % of the minimum once the XCD map covers ragged grids (252 workgroups): 61.3 vs 60.6 us, 61.6 vs 60.5 us
    assert auto_split_k(768, 3072, 10752) == 7 and auto_split_k(3072, 768, 10752) == 7
    for (M, N, K) in [(20832, 4096, 1024), (88256, 3072, 768), (4096, 4096, 20832)]:
        assert auto_split_k(M, N, K) == 1                      # the output alone fills (three quarters of) the chip
    for (M, N, K) in [(768, 768, 10752), (768, 1536, 94976), (1024, 768, 7168), (512, 512, 4096), (256, 256, 1 << 20)]:
        sk = auto_split_k(M, N, K)
        assert 1 <= sk <= 16 and K // sk >= 1280
    assert auto_split_k(1024, 1024, 2048) == 1                 # short contractions are never split


def test_layernorm_row_group_map_covers_the_buffer_exactly_once():
    """The row map of dvla_layernorm_*_rows (csrc/layernorm.hip::ln_buf_row, restated): logical row r -> buffer row
    (r / grp) * gstride + goff + r % grp.  For every (grp, gstride, goff) the model uses, the mapped rows plus the rows the backward
    kernel zero-fills (the wave that owns a group's first row writes the sequence's rows outside [goff, goff + grp)) are a partition
    of the buffer: every row of the whole-stream gradient is written exactly once (tests/gpu_checks.py::check_layernorm_last_tokens
    checks the values on the GPU)."""
    def buf_row(r, grp, gstride, goff):
        return (r // grp) * gstride + goff + r % grp
    for (n, grp, gstride, goff) in [(5, 196, 205, 9), (3, 256, 265, 9), (7, 21, 21, 0), (4, 1, 40, 39), (6, 196, 212, 0), (6, 16, 212, 196)]:
        rows = n * grp
        mapped = [buf_row(r, grp, gstride, goff) for r in range(rows)]
        assert len(set(mapped)) == rows and all(0 <= m < n * gstride for m in mapped)
        assert all(goff <= m % gstride < goff + grp for m in mapped)
        zeroed = [(r // grp) * gstride + j for r in range(0, rows, grp) for j in range(gstride) if not (goff <= j < goff + grp)]
        assert sorted(mapped + zeroed) == list(range(n * gstride))
    # the two LayerNorms of the resampler (map_output) tile their buffer: media rows [0, 196) and latent rows [196, 212) of every sequence
    a = {buf_row(r, 196, 212, 0) for r in range(6 * 196)}
    b = {buf_row(r, 16, 212, 196) for r in range(6 * 16)}
    assert not (a & b) and a | b == set(range(6 * 212))

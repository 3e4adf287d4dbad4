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

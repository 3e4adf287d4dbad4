"""Pins oracle/torch_ref.py::clip_adamw_step against the reference's optimizer step itself -- torch's
clip_grad_norm_ + AdamW with bf16 parameters (importable here and on the GPU box) -- on CPU."""
import pytest

torch = pytest.importorskip("torch")
from oracle import torch_ref as R  # noqa: E402


def test_oracle_matches_torch_clip_adamw():
    g = torch.Generator().manual_seed(5)
    shapes = [(37, 19), (128,), (5, 7, 3)]
    params = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.3).to(torch.bfloat16)) for s in shapes]
    mine = [p.detach().clone() for p in params]
    m = [torch.zeros_like(p) for p in mine]
    v = [torch.zeros_like(p) for p in mine]
    try:
        opt = torch.optim.AdamW(params, lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, fused=True)
    except (RuntimeError, TypeError):
        pytest.skip("fused AdamW not available on CPU in this torch build")
    for step in range(1, 5):
        grads = [(torch.randn(p.shape, generator=g) * (3.0 if step % 2 else 0.01)).to(torch.bfloat16) for p in params]
        for p, gr in zip(params, grads):
            p.grad = gr.clone()
        norm_t = torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()
        norm_o = R.clip_adamw_step(mine, grads, m, v, step, 1e-2, (0.9, 0.95), 1e-8, 0.05, max_norm=0.1)
        # torch returns the norm of bf16 gradients as a bf16 tensor (per-tensor norms and the total are rounded to bf16);
        # the oracle / kernel keep it in fp32 -- AdamW is scale-invariant in the gradient up to eps, so the half-ulp in the
        # clip coefficient does not show in the parameters below
        assert abs(float(norm_t) - float(norm_o)) <= 2 ** -7 * float(norm_t)
        for a, b in zip(params, mine):
            d = a.detach().float() - b.float()
            # same fp32 math and bf16 storage; the bf16-vs-fp32 clip coefficient moves a few percent of the elements by
            # one rounding step of the gradient / moments: exact after step 1, <= 3e-3 rel-L2 afterwards
            assert float(d.norm() / b.float().norm()) <= (0.0 if step == 1 else 3e-3)
            assert float((d != 0).float().mean()) <= 0.15

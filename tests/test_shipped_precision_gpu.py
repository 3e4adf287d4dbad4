"""GPU: the module driven the way the SHIPPED scripts drive it (scripts/CALVIN_ABC_D/DreamVLA/finetune.sh:13,22:
`--precision fp32 --bf16_module vision_encoder`), step for step as train.py:122-174 and utils/train_utils.py:98-608 do:
model.float(); model.vision_encoder.bfloat16().requires_grad_(False); clip frozen; .to(device); _init_model_type(); (DDP wrap
omitted: one process) torch.optim.AdamW over the requires_grad parameters; autocast context = suppress (fp32); forward in
train mode; the reference loss block; backward; clip_grad_norm_; optimizer.step().

Round 1 raised TypeError on fp32 trainable parameters.  Now the fp32 parameters are masters and the kernels run on bf16
shadows (dreamvla_amd.ops.shadow): parameters, gradients and optimizer state stay fp32 exactly as train.py builds them."""
from contextlib import suppress

import pytest
import torch

BF = torch.bfloat16


def _build(cfg):
    from dreamvla_amd.dreamvla_model import DreamVLA
    from oracle import weights
    m = DreamVLA(clip_device="cpu", vit_checkpoint_path=None, **cfg)
    m.load_state_dict(weights.fill_state_dict(m.state_dict()), strict=True)
    return m


@pytest.mark.gpu
def test_train_py_flow_fp32_masters_bf16_vision_encoder():
    from dreamvla_amd import losses
    from tests import model_checks as C
    fx = C.load("dreamvla_B.pt")          # obs head + DiT head, 2 layers, S = 2
    cfg = dict(fx["cfg"])
    # ---- train.py:122-174 ----
    model = _build(cfg)
    model = model.float()                                   # --precision fp32
    model.vision_encoder.bfloat16()                         # --bf16_module vision_encoder
    model.vision_encoder.requires_grad_(False)
    model.clip_model.requires_grad_(False)
    model = model.to("cuda")
    model._init_model_type()
    params = [p for p in model.parameters() if p.requires_grad]
    assert all(p.dtype == torch.float32 for p in params)
    optimizer = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-4)
    autocast = suppress                                     # get_autocast("fp32")
    # ---- one window of the fixture's synthetic batch, labels as utils/train_utils.py:98-157 build them ----
    from oracle import weights
    S = fx["S"]
    b = weights.synthetic_batch(2, S, window=fx["window"], seed=fx["seed"], heads=())
    b["actions"][..., 6:] = (b["actions"][..., 6:] > 0.5).float()
    batch = {k: v.to("cuda") for k, v in b.items()}
    lab = losses.label_actions(batch["actions"], S, 3)
    model.train()
    losses_seen = []
    for step in range(3):
        optimizer.zero_grad()
        with autocast():
            out = model(batch["image_primary"][:, :S], batch["image_wrist"][:, :S], batch["state"][:, :S],
                        batch["text_token"][:, :S], action=batch["actions"][:, :S], action_label=lab, mode="train")
            total, _ = losses.calvin_losses(out, batch, sequence_length=S, use_dit_head=cfg["use_dit_head"], label_action=lab)
        total.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
        optimizer.step()
        losses_seen.append(float(total))
    assert all(l == l and abs(l) < 1e4 for l in losses_seen), losses_seen
    grads = [p.grad for p in params if p.grad is not None]
    assert len(grads) > 50 and all(g.dtype == torch.float32 for g in grads)
    assert all(bool(torch.isfinite(g).all()) for g in grads)
    st = optimizer.state[params[0]]
    assert st["exp_avg"].dtype == torch.float32                      # fp32 optimizer state, as under the reference
    assert model.state_dict()["transformer_backbone.h.0.attn.c_attn.weight"].dtype == torch.float32
    assert model.state_dict()["vision_encoder.blocks.0.attn.qkv.weight"].dtype == BF


@pytest.mark.gpu
def test_fp32_master_forward_matches_bf16_module():
    """same weights (bf16-representable), eval mode: fp32 masters on bf16 shadows vs the module cast to bf16.  The GEMM
    weights are identical (the shadows ARE the bf16 parameters); biases, LayerNorm affine parameters and learned tokens are
    read as fp32 in one run and as bf16 in the other (same values, different kernel paths / summation orders), so the
    outputs agree to bf16 noise, well inside the fixture's per-output tolerance -- not bit for bit."""
    from tests import model_checks as C
    fx = C.load("dreamvla_A.pt")
    inp = {k: v.to("cuda") for k, v in C.golden_inputs(fx).items()}
    outs = []
    for mode in ("bf16", "fp32_master"):
        m = C.build_hip_model(fx["cfg"])
        m = m.to(BF) if mode == "bf16" else m.float()
        if mode == "fp32_master":
            m.vision_encoder.bfloat16()
        m = m.to("cuda")
        m._init_model_type()
        m.eval()
        from dreamvla_amd.ops import GemmTuner
        GemmTuner.enabled = False            # same kernel configuration in both runs
        try:
            with torch.no_grad():
                outs.append(m(inp["image_primary"].to(BF), inp["image_wrist"].to(BF), inp["state"].to(BF), inp["text_token"],
                              action_label=fx["action_label"].to("cuda", BF), mode="train"))
        finally:
            GemmTuner.enabled = True
    for a, b in zip(*outs):
        if a is None:
            assert b is None
            continue
        r = C.rel_l2(a, b)
        assert r <= 1.5e-2, r      # two bf16 runs that round at different points: ~2x the fixture's bf16 deviation (6e-3)
    for outs_ in outs:             # and each of them sits inside the fixture's per-output tolerance against the fp32 reference
        for res in C.compare_outputs(outs_, fx["train"], 3e-2, "precision", fx=fx):
            assert res["ok"], res


@pytest.mark.gpu
def test_shadow_gradient_is_fp32_and_matches_bf16_path():
    from dreamvla_amd import ops
    torch.manual_seed(0)
    x = torch.randn(300, 256, device="cuda").to(BF)
    w32 = (torch.randn(512, 256, device="cuda") * 0.05).to(BF).float().requires_grad_(True)
    b32 = torch.randn(512, device="cuda").requires_grad_(True)
    w16 = w32.detach().to(BF).requires_grad_(True)
    b16 = b32.detach().clone().requires_grad_(True)
    dy = torch.randn(300, 512, device="cuda").to(BF)
    ops.GemmTuner.enabled = False
    try:
        ops.linear(x, w32, b32, act="gelu_erf").backward(dy)
        ops.linear(x, w16, b16, act="gelu_erf").backward(dy)
    finally:
        ops.GemmTuner.enabled = True
    assert w32.grad.dtype == torch.float32 and b32.grad.dtype == torch.float32
    assert torch.equal(w32.grad, w16.grad.float()) and torch.equal(b32.grad, b16.grad)
    # the shadow is refreshed when (and only when) the master changes
    s1 = ops.shadow(w32)
    assert ops.shadow(w32).data_ptr() == s1.data_ptr()
    with torch.no_grad():
        w32.add_(1.0)
    s2 = ops.shadow(w32)
    assert not torch.equal(s1, s2) and torch.equal(s2, w32.detach().to(BF))

"""Eager PyTorch-ROCm comparator on the GPU box (not a test, not part of bench.py's timed path).

The reference itself cannot travel to the GPU box, so this runs the oracle's functional restatement of the reference
forward (oracle/model_ref.py: the same ATen op sequence -- F.linear / F.layer_norm / SDPA-style softmax attention /
GELU -- minus dropout and minus the (B,1,L,L) mask copy, both of which would only make eager slower) in bf16 on the
GPU with autograd, the same loss block, clip_grad_norm_ and fused AdamW: the "reference run eagerly on PyTorch-ROCm
with --precision bf16 casting" of SURVEY.md section 8d, at the same B = 32, S = 7, head set C.
usage: python tests/gpu_eager_baseline.py [--heads C] [--batch 32] [--steps 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from dreamvla_amd import losses  # noqa: E402
from dreamvla_amd.dreamvla_model import DreamVLA  # noqa: E402
from dreamvla_amd.synthetic import synthetic_batch  # noqa: E402
from oracle import model_ref as M  # noqa: E402
from oracle import torch_ref as R  # noqa: E402


def sdpa_attention(q, k, v, scale=None, mask=None, drop=None, drop_cols=None):
    m = None if mask is None else mask.to(device=q.device, dtype=q.dtype)
    return torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=m, scale=scale)


def run(heads="C", B=32, steps=3, S=7):
    R.attention = sdpa_attention           # what timm 0.9.16 / GPT2SdpaAttention dispatch to
    dev, BF = "cuda", torch.bfloat16
    cfg = bench.model_cfg(heads, S)
    m = DreamVLA(clip_device="cpu", vit_checkpoint_path=None, **cfg)
    sd = {k: (v.to(dev, BF) if torch.is_floating_point(v) else v.to(dev)) for k, v in m.state_dict().items()}
    del m
    sd["attention_mask"] = sd["attention_mask"].float()
    params = []
    for k, v in sd.items():
        if torch.is_floating_point(v) and not k.startswith(("clip_model.", "vision_encoder.")) and k != "attention_mask" \
                and "decoder_position_embedding" not in k:
            v.requires_grad_(True)
            params.append(v)
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-4, fused=True)
    b = synthetic_batch(B, S, window=S + 3, seed=1234, heads=bench.label_heads(heads))
    b["actions"][..., 6:] = (b["actions"][..., 6:] > 0.5).float()
    batch = {k: (v.to(dev, BF) if torch.is_floating_point(v) else v.to(dev)) for k, v in b.items()}
    lab = losses.label_actions(batch["actions"], S, 3)
    g = torch.Generator(device=dev).manual_seed(1)

    def step():
        opt.zero_grad(set_to_none=True)
        noise = torch.randn(8 * B * S, 3, 7, device=dev, generator=g).to(BF)
        tstep = torch.randint(0, 100, (8 * B * S,), device=dev, generator=g)
        with torch.no_grad():
            tf = M.clip_text(sd, "clip_model", batch["text_token"][:, :S].flatten(0, 1))
        out = M.dreamvla_forward(sd, cfg, batch["image_primary"][:, :S], batch["image_wrist"][:, :S], batch["state"][:, :S],
                                 batch["text_token"][:, :S], action_label=lab, mode="train", dit_noise=noise,
                                 dit_timestep=tstep, text_feature=tf)
        total, _ = losses.calvin_losses(out, batch, sequence_length=S, use_dit_head=cfg["use_dit_head"], label_action=lab)
        total.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()
        return total

    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": B / dt, "unit": "samples/s", "ms_per_step": dt * 1e3, "steps": steps, "warmup": 2, "dtype": "bf16",
            "sample": f"eager PyTorch-ROCm restatement of the reference step (hipBLASLt GEMMs + SDPA + ATen elementwise, "
                      f"clip_grad_norm_ + fused AdamW), B={B}, S={S}, head set {heads}, same box; no dropout, no dense mask copy"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heads", default="C")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    print(json.dumps(run(args.heads, args.batch, args.steps)))


if __name__ == "__main__":
    main()

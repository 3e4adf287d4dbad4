"""ORACLE -- test infrastructure.  Deterministic weight recipe shared by the real reference (golden generation,
build container), the oracle and the HIP model (GPU box): every tensor is a pure function of its state_dict key and
shape, so ~400 M parameters never have to be committed.  Values are rounded to bf16 so that the bf16 HIP model and
the fp32 oracle / reference see IDENTICAL weights."""
import zlib

import torch

KEEP = ("attention_mask", "pos_embed", "position_embedding", "decoder_pos_embed")  # constructed tables: keep as built


def recipe_tensor(key, shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    t = torch.randn(tuple(shape), generator=g)
    last = key.rsplit(".", 1)[-1]
    if len(shape) >= 2 and ("weight" in last or "latents" in last or "text_projection" in last or "in_proj" in last):
        if "c_attn" in key or "c_fc" in key or ("c_proj" in key and "transformer_backbone" in key):
            fan_in = shape[0]                       # HF Conv1D (in, out)
        elif "token_embedding" in key:
            fan_in = 2500.0                         # std 0.02
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
        t = t / (fan_in ** 0.5)
    elif last == "weight":                          # LayerNorm gains
        t = 1.0 + 0.1 * t
    elif last in ("bias", "in_proj_bias"):
        t = 0.02 * t
    elif last == "logit_scale":
        t = torch.tensor(2.6592)
    else:                                           # learned tokens, mask tokens, cls tokens, position embeddings ...
        t = 0.02 * t
    return t.to(torch.bfloat16).to(dtype)


def fill_state_dict(sd, skip_prefixes=()):
    """Return a new state_dict: every floating tensor replaced by the recipe (except constructed tables)."""
    out = {}
    for k, v in sd.items():
        keep = any(s in k for s in KEEP) and "transformer_backbone_position_embedding" not in k \
            and not k.endswith("positional_embedding")
        if keep or not torch.is_floating_point(v) or any(k.startswith(p) for p in skip_prefixes):
            out[k] = v.clone()
        else:
            out[k] = recipe_tensor(k, v.shape, v.dtype)
    return out


def synthetic_batch(B, S, window=None, seed=1234, heads=()):
    """Seeded synthetic CALVIN-like batch; values are bf16-representable."""
    W = window or S
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    u = lambda *s: torch.rand(*s, generator=g).to(torch.bfloat16).float()
    batch = {
        "image_primary": r(B, W, 3, 224, 224),
        "image_wrist": r(B, W, 3, 224, 224),
        "state": torch.cat([u(B, W, 6), (torch.rand(B, W, 1, generator=g) > 0.5).float()], dim=-1),
        "text_token": torch.randint(1, 49000, (B, 77), generator=g).unsqueeze(1).repeat(1, W, 1),
        "actions": torch.cat([u(B, W, 6) * 2 - 1, (torch.rand(B, W, 1, generator=g) > 0.5).float()], dim=-1),
    }
    # EOT token (largest id) at a random position, as clip.tokenize produces
    eot = torch.randint(5, 77, (B,), generator=g)
    for b in range(B):
        batch["text_token"][b, :, eot[b]] = 49407
        batch["text_token"][b, :, eot[b] + 1:] = 0
    if "depth" in heads:
        batch["depth_primary"] = u(B, W, 1, 224, 224) * 10 + 0.01
        batch["depth_wrist"] = u(B, W, 1, 224, 224) * 10 + 0.01
    if "dino" in heads:
        batch["dino_primary"], batch["dino_wrist"] = r(B, W, 256, 768), r(B, W, 256, 768)
    if "sam" in heads:
        batch["sam_primary"], batch["sam_wrist"] = r(B, W, 256, 256), r(B, W, 256, 256)
    if "traj" in heads:
        batch["tracks"], batch["tracks_gripper"] = r(B, W, 784, 2) * 2, r(B, W, 784, 2) * 2
    return batch

"""CLIP ViT-B/32 text tower on the HIP kernels (frozen; `DreamVLA.clip_model`, dreamvla_model.py:511-514,643-650).

Restates openai/CLIP `clip/model.py` (CLIP.encode_text, Transformer, ResidualAttentionBlock, QuickGELU,
build_attention_mask) for the text half only, with the original parameter names so a real CLIP state_dict
(`token_embedding.weight`, `positional_embedding`, `transformer.resblocks.*`, `ln_final.*`, `text_projection`)
loads with strict=False.  The pretrained checkpoint is not available offline: weights are CLIP's own random
init unless `checkpoints/clip/ViT-B-32.pt` exists ("parity unpinned" for the text tower).
"""
import os

import numpy as np
import torch
from torch import nn

from . import ops
from .nn import LayerNorm, Linear


class _MHAParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names (in_proj_weight (3E,E) = [q;k;v], in_proj_bias, out_proj)."""

    def __init__(self, d_model, n_head):
        super().__init__()
        self.num_heads = n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = Linear(d_model, d_model)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head):
        super().__init__()
        assert d_model // n_head == 64
        self.attn = _MHAParams(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", Linear(d_model, d_model * 4))
        self.mlp.add_module("gelu", nn.Identity())   # QuickGELU is fused into c_fc's epilogue
        self.mlp.add_module("c_proj", Linear(d_model * 4, d_model))
        self.ln_2 = LayerNorm(d_model)

    def forward(self, x, mask_tables):
        qkv = ops.linear(self.ln_1(x), self.attn.in_proj_weight, self.attn.in_proj_bias)
        o = ops.self_attention(qkv, self.attn.num_heads, mask_tables=mask_tables)
        x = self.attn.out_proj(o, residual=x)
        x = ops.mlp(self.ln_2(x), self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight,
                    self.mlp.c_proj.bias, act="quick_gelu", residual=x)
        return x


class Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class CLIPTextEncoder(nn.Module):
    def __init__(self, embed_dim=512, context_length=77, vocab_size=49408, width=512, heads=8, layers=12):
        super().__init__()
        self.context_length = context_length
        self.vocab_size = vocab_size
        self.transformer = Transformer(width, layers, heads)
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.ln_final = LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.register_buffer("causal_mask", torch.full((context_length, context_length), float("-inf")).triu_(1),
                             persistent=False)
        self.initialize_parameters()

    def initialize_parameters(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        proj_std = (self.transformer.width ** -0.5) * ((2 * self.transformer.layers) ** -0.5)
        attn_std = self.transformer.width ** -0.5
        fc_std = (2 * self.transformer.width) ** -0.5
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)

    @property
    def dtype(self):
        return self.token_embedding.weight.dtype

    def encode_text(self, text):
        """text: int64 (N, 77) BPE tokens -> (N, embed_dim) features taken at the EOT (arg-max id) position."""
        x = self.token_embedding(text) + self.positional_embedding           # gather + add (index plumbing)
        mt = ops.mask_tables_for(self.causal_mask)
        for blk in self.transformer.resblocks:
            x = blk(x, mt)
        x = self.ln_final(x)
        x = x[torch.arange(x.shape[0], device=x.device), text.argmax(dim=-1)]
        return ops.linear(x, self.text_projection, None, conv1d=True)          # x @ text_projection


def load(name="ViT-B/32", device="cpu", jit=False, download_root=None):
    """Drop-in for `clip.load(name, device)` -> (model, preprocess) restricted to what DreamVLA uses (the text tower and the
    image transform).  `name` = a readable TorchScript / state-dict CLIP checkpoint: its text-tower tensors are loaded and
    every text-tower key must be present.  There is no download here (the reference fetches "ViT-B/32" from the network when
    the file is missing): without a checkpoint the frozen tower is RANDOMLY initialised and the language conditioning is
    meaningless -- that is said loudly (a warning; DVLA_REQUIRE_CLIP=1 turns it into an error), never silently."""
    import warnings
    from .preprocess import clip_image_preprocess
    model = CLIPTextEncoder()
    if isinstance(name, str) and os.path.isfile(name):
        try:
            sd = torch.jit.load(name, map_location="cpu").state_dict()
        except RuntimeError:       # not a TorchScript archive: a plain state_dict checkpoint
            sd = torch.load(name, map_location="cpu")
            sd = sd.get("state_dict", sd)
        want = set(model.state_dict())
        keep = {k: v for k, v in sd.items() if k in want}
        missing = sorted(want - set(keep))
        if missing:
            raise RuntimeError(f"clip_text.load({name!r}): the checkpoint lacks text-tower tensors {missing[:6]}"
                               f"{' ...' if len(missing) > 6 else ''} ({len(missing)} of {len(want)})")
        model.load_state_dict(keep, strict=True)
    else:
        msg = (f"clip_text.load({name!r}): no CLIP checkpoint file -- the frozen text tower is RANDOMLY initialised "
               f"(the reference would download ViT-B/32; there is no network here).  Place the checkpoint at the path the "
               f"caller passes (DreamVLA: checkpoints/clip/ViT-B-32.pt) for meaningful language conditioning.")
        if os.environ.get("DVLA_REQUIRE_CLIP") == "1":
            raise FileNotFoundError(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    model = model.to(device).eval()
    return model, clip_image_preprocess

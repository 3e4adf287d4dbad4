"""Flamingo-style PerceiverResampler on the HIP kernels -- host-side mirror of
/root/reference/models/perceiver_resampler.py (same constructor arguments, parameter names and shapes:
latents, layers.{i}.0.{norm_media,norm_latents,to_q,to_kv,to_out}, layers.{i}.1.{0,1,3}, norm).

Per layer (perceiver_resampler.py:35-61,124-128):
    x_n = LN_media(x); l_n = LN_latents(latents)
    q = to_q(l_n); [k | v] = to_kv(cat(x_n, l_n)); latents += to_out(softmax(q k^T / 8) v)   (8 heads x 64)
    latents += W2 gelu(W1 LN(latents))                                                        (no biases)
`sim - sim.amax()` before the softmax (line 57) is the usual max subtraction the fused kernel does anyway; the
`q * scale` pre-scaling (line 53) is folded into the kernel's score scale.
"""
import os

import torch
from torch import nn

from . import ops
from .nn import LayerNorm, Linear


def exists(val):
    return val is not None


def FeedForward(dim, mult=4):
    inner_dim = int(dim * mult)
    # nn.Sequential only to keep the reference's state_dict keys (1.0.*, 1.1.weight, 1.3.weight); the resampler
    # calls the fused MLP kernel path directly instead of Sequential.forward.
    return nn.Sequential(LayerNorm(dim), Linear(dim, inner_dim, bias=False), nn.GELU(), Linear(inner_dim, dim, bias=False))


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        assert dim_head == 64, "HIP attention kernels are specialised for head_dim 64"
        self.scale = dim_head ** -0.5
        self.heads = heads
        inner_dim = dim_head * heads
        self.norm_media = LayerNorm(dim)
        self.norm_latents = LayerNorm(dim)
        self.to_q = Linear(dim, inner_dim, bias=False)
        self.to_kv = Linear(dim, inner_dim * 2, bias=False)
        self.to_out = Linear(inner_dim, dim, bias=False)

    def forward(self, x, latents):
        """x: (n, n1, D) media tokens, latents: (n, n2, D).  Returns latents + attention (the residual add is fused
        into the to_out GEMM epilogue)."""
        if os.environ.get("DVLA_LN_CONCAT") == "0":      # (same-box A/B of the fused concatenation: the ATen form)
            xn = self.norm_media(x)
            ln = self.norm_latents(latents)
            q = self.to_q(ln)
            kv = self.to_kv(torch.cat((xn, ln), dim=-2))
        else:
            # the two LayerNorms write the two row ranges of ONE (n, n1 + n2, D) buffer: no torch.cat, and in backward no copies of
            # the cat's strided gradient slices (ops._LayerNormConcat).  The queries take their own LayerNorm of the 16 latents
            # (a second, tiny launch): slicing them out of the buffer would send a zero-filled (n, n1 + n2, D) gradient back into it.
            both = ops.layer_norm_concat(x, self.norm_media.weight, self.norm_media.bias, self.norm_media.eps,
                                         latents, self.norm_latents.weight, self.norm_latents.bias, self.norm_latents.eps)
            q = self.to_q(self.norm_latents(latents))
            kv = self.to_kv(both)
        o = ops.cross_attention(q, kv, self.heads, scale=self.scale)
        return self.to_out(o, residual=latents)


class PerceiverResampler(nn.Module):
    def __init__(self, *, dim, depth=6, dim_head=64, heads=8, num_latents=64, max_num_media=None,
                 max_num_frames=None, ff_mult=4):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.frame_embs = nn.Parameter(torch.randn(max_num_frames, dim)) if exists(max_num_frames) else None
        self.media_time_embs = nn.Parameter(torch.randn(max_num_media, 1, dim)) if exists(max_num_media) else None
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                              FeedForward(dim=dim, mult=ff_mult)]))
        self.norm = LayerNorm(dim)

    def forward(self, x):
        """x: (b, T, F, v, D) -> (b, T, num_latents, D)"""
        b, T, F, v = x.shape[:4]
        if exists(self.frame_embs):
            x = x + self.frame_embs[:F].view(1, 1, F, 1, -1)
        x = x.reshape(b, T, F * v, x.shape[-1])
        if exists(self.media_time_embs):
            x = x + self.media_time_embs[:T]
        D = x.shape[-1]
        xm = x.reshape(b * T, F * v, D)
        latents = self.latents.to(x.dtype).unsqueeze(0).expand(b * T, -1, -1).contiguous()
        for attn, ff in self.layers:
            latents = attn(xm, latents)
            h = ff[0](latents)
            latents = ops.mlp(h, ff[1].weight, None, ff[3].weight, None, act="gelu_erf", residual=latents)
        return self.norm(latents).view(b, T, -1, D)

"""MAE ViT-B/16 encoder on the HIP kernels -- host-side mirror of /root/reference/models/vit_mae.py.

Same constructor, parameter names and shapes as the reference `MaskedAutoencoderViT` (including the MAE
decoder half, which DreamVLA never calls but whose tensors are part of the checkpoint / state_dict surface,
vit_mae.py:82-97).  Only `forward_encoder` is on the hot path (dreamvla_model.py:672-673).

Deviation (documented in DESIGN.md): the reference's `random_masking(x, 0.0)` (vit_mae.py:157-182,194) keeps
all 196 patch tokens but in a random per-sample order and DreamVLA drops `ids_restore`.  Attention without
positional terms after the embedding is permutation-equivariant, so the un-shuffled result is identical up
to fp rounding (<= 6e-6 fp32).  This implementation keeps patch order (ids_restore =
identity) and consumes no RNG.
"""
from functools import partial

import numpy as np
import torch
from torch import nn

from . import ops
from .nn import Block, LayerNorm, Linear, PatchEmbed


# ---- fixed 2-D sin-cos position tables (numpy float32 -> identical bits to vit_mae.py:8-53) -----------
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.
    omega = 1. / 10000 ** omega
    pos = pos.reshape(-1)
    out = np.einsum('m,d->md', pos, omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed_from_grid(embed_dim, grid):
    assert embed_dim % 2 == 0
    emb_h = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0])
    emb_w = get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.meshgrid(grid_w, grid_h)  # w goes first (as in the reference)
    grid = np.stack(grid, axis=0).reshape([2, 1, grid_size, grid_size])
    pos_embed = get_2d_sincos_pos_embed_from_grid(embed_dim, grid)
    if cls_token:
        pos_embed = np.concatenate([np.zeros([1, embed_dim]), pos_embed], axis=0)
    return pos_embed


class MaskedAutoencoderViT(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.,
                 norm_layer=LayerNorm, norm_pix_loss=False):
        super().__init__()
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
                                     for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        # MAE decoder half: parameters only (never executed by DreamVLA)
        self.decoder_embed = Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([
            Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True, norm_layer=norm_layer)
            if decoder_embed_dim // decoder_num_heads == 64 else _ParamOnlyBlock(decoder_embed_dim, mlp_ratio)
            for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        self.initialize_weights()

    def initialize_weights(self):
        gs = int(self.patch_embed.num_patches ** .5)
        pos_embed = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], gs, cls_token=True)
        self.pos_embed.data.copy_(torch.from_numpy(pos_embed).float().unsqueeze(0))
        dpe = get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], gs, cls_token=True)
        self.decoder_pos_embed.data.copy_(torch.from_numpy(dpe).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.cls_token, std=.02)
        torch.nn.init.normal_(self.mask_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward_encoder(self, x, mask_ratio):
        """x: (n, 3, H, W) -> (n, 1 + num_patches, embed_dim), mask (n, L) zeros, ids_restore identity."""
        if mask_ratio != 0.0:
            raise NotImplementedError("the DreamVLA hot path only calls forward_encoder(mask_ratio=0.0) "
                                      "(models/dreamvla_model.py:672-673)")
        n = x.shape[0]
        L = self.patch_embed.num_patches
        # patch-embed GEMM + bias; the fixed pos-embed add rides in the residual slot of the epilogue
        x = self.patch_embed(x, pos=self.pos_embed[0, 1:, :])
        cls_token = (self.cls_token + self.pos_embed[:, :1, :]).to(x.dtype)
        if x.dtype == torch.bfloat16 and not torch.is_grad_enabled():
            D = x.shape[-1]       # the frozen encoder of DreamVLA: one gather-write pass instead of torch.cat
            x = ops.assemble_tokens([cls_token.reshape(1, 1, 1, D).expand(n, 1, 1, D), x.reshape(n, 1, L, D)]).view(n, L + 1, D)
        else:
            x = torch.cat((cls_token.expand(n, -1, -1), x), dim=1)
        for blk in self.blocks:
            x = blk(x)
        x = self.norm(x)
        mask = torch.zeros(n, L, device=x.device)
        ids_restore = torch.arange(L, device=x.device).unsqueeze(0).expand(n, -1)
        return x, mask, ids_restore


class _ParamOnlyBlock(nn.Module):
    """Holds timm-Block-shaped parameters for head_dim != 64 blocks that are never executed (MAE decoder:
    512 wide / 16 heads = 32)."""

    def __init__(self, dim, mlp_ratio):
        super().__init__()
        self.norm1 = LayerNorm(dim)
        self.attn = nn.Module()
        self.attn.qkv = Linear(dim, dim * 3, bias=True)
        self.attn.proj = Linear(dim, dim)
        self.norm2 = LayerNorm(dim)
        self.mlp = nn.Module()
        self.mlp.fc1 = Linear(dim, int(dim * mlp_ratio))
        self.mlp.fc2 = Linear(int(dim * mlp_ratio), dim)

    def forward(self, x):
        raise NotImplementedError("MAE decoder blocks are not on the DreamVLA hot path")

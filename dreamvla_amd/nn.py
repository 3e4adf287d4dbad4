"""nn.Module building blocks that run on the HIP kernels.

Parameter names / shapes mirror the third-party modules the reference instantiates, so state_dict keys are
identical:
  * Linear, LayerNorm      : torch.nn.Linear / LayerNorm parameter layout (weight (out,in), bias)
  * Conv1D                 : transformers.pytorch_utils.Conv1D (weight (in,out), bias (out))   [models/gpt2.py:53]
  * Mlp, Attention, Block  : timm==0.9.16 vision_transformer (norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2)
  * PatchEmbed             : timm PatchEmbed (proj = Conv2d(k=s=16) parameters), computed as an im2col GEMM
"""
import torch
from torch import nn

from . import ops


class Linear(nn.Linear):
    """nn.Linear whose forward is the fused HIP GEMM (+bias, optional activation / residual / dropout)."""

    def forward(self, x, act="none", residual=None, dropout_p=0.0, ln_eps=None):
        """ln_eps: x is layer-normalised (no affine parameters) inside the GEMM -- Linear(LayerNorm(x)) in one launch
        (ops.linear_ln: evaluation-time shapes only, see ops.ln_fusable)"""
        if ln_eps is not None:
            return ops.linear_ln(x, self.weight, self.bias, eps=ln_eps, act=act, residual=residual)
        return ops.linear(x, self.weight, self.bias, act=act, residual=residual, dropout_p=dropout_p)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)

    def fork(self, x):
        """(x, LayerNorm(x)) for pre-LN residual blocks: backward adds the residual-stream gradient inside the
        LayerNorm-backward kernel (ops._LayerNormFork)."""
        return ops.layer_norm_fork(x, self.weight, self.bias, self.eps)

    def last_tokens(self, x, keep):
        """LayerNorm(x[:, -keep:, :]) as (n * keep, D) without copying the strided slice (ops._LayerNormRows)."""
        return ops.layer_norm_last_tokens(x, self.weight, self.bias, self.eps, keep)


class Conv1D(nn.Module):
    """HF GPT-2 Conv1D: y = x @ W + b with W of shape (nx, nf)."""

    def __init__(self, nf, nx):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.empty(nx, nf))
        self.bias = nn.Parameter(torch.zeros(nf))
        nn.init.normal_(self.weight, std=0.02)

    def forward(self, x, act="none", residual=None, dropout_p=0.0):
        return ops.linear(x, self.weight, self.bias, act=act, conv1d=True, residual=residual, dropout_p=dropout_p)


class Mlp(nn.Module):
    """timm Mlp: fc1 -> act -> fc2 (dropouts are 0 everywhere the reference uses it)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act="gelu_erf", bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Linear(in_features, hidden_features, bias=bias)
        self.fc2 = Linear(hidden_features, out_features, bias=bias)
        self.act_name = act

    def forward(self, x, residual=None, ln_eps=None):
        if ln_eps is not None:          # fc1(LayerNorm(x)) in one launch (evaluation-time shapes)
            return self.fc2(self.fc1(x, act=self.act_name, ln_eps=ln_eps), residual=residual)
        return ops.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, act=self.act_name,
                       residual=residual)


class Attention(nn.Module):
    """timm Attention: fused qkv Linear -> (B,N,3,h,d) -> SDPA (scale d^-0.5) -> proj.  head_dim 64 runs on the MFMA kernels
    (csrc/attention.hip); any other head_dim <= 128 (DiT-S: 96) on the short-sequence kernel (csrc/attention_small.hip,
    sequences of at most 64 tokens, no mask) -- ops.self_attention raises for anything else."""

    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        assert dim % num_heads == 0, "dim must be a multiple of num_heads"
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = Linear(dim, dim)

    def forward(self, x, residual=None, ln_eps=None):
        qkv = self.qkv(x, ln_eps=ln_eps)
        o = ops.self_attention(qkv, self.num_heads, scale=self.scale, head_dim=self.head_dim)
        return self.proj(o, residual=residual)


class Block(nn.Module):
    """timm Block (pre-LN): x + attn(norm1(x)); x + mlp(norm2(x)).  norm_layer is a callable dim -> LayerNorm;
    `affine=False` gives the DiT block (LayerNorm without parameters, action_model/models.py:129-131)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, norm_layer=None, act="gelu_erf"):
        super().__init__()
        norm_layer = norm_layer or LayerNorm
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act=act)

    def forward(self, x):
        if isinstance(self.norm1, LayerNorm) and isinstance(self.norm2, LayerNorm):
            r, n = self.norm1.fork(x)
            x = self.attn(n, residual=r)
            r, n = self.norm2.fork(x)
            return self.mlp(n, residual=r)
        x = self.attn(self.norm1(x), residual=x)
        x = self.mlp(self.norm2(x), residual=x)
        return x

    def forward_shared_suffix(self, prefix, suffix, n_seq):
        """This block on a batch of sequences [prefix_b ; suffix] whose trailing tokens are IDENTICAL in every sequence --
        the dream-head decoders feed 9 per-sample query tokens followed by 196 / 256 copies of `mask_token + position`
        (models/dreamvla_model.py:800-809).  norm1 and the qkv projection are row-wise, so for the shared rows they are
        computed ONCE (n_suffix rows instead of n_seq * n_suffix: 96 % of this block's qkv GEMM, its input-gradient GEMM and
        its weight-gradient contraction) and broadcast into the per-sequence qkv buffer; from the attention on everything
        is per sequence.  Same function values as forward(cat(prefix, suffix.expand)); autograd sums the shared rows'
        gradient over the batch before the (tiny) backward GEMMs.   prefix (n_seq, n_q, D), suffix (n_suffix, D)."""
        x = ops.concat_shared_suffix(prefix, suffix)
        qkv = ops.concat_shared_suffix(self.attn.qkv(self.norm1(prefix)), self.attn.qkv(self.norm1(suffix)))
        o = ops.self_attention(qkv, self.attn.num_heads, scale=self.attn.scale)
        x = self.attn.proj(o, residual=x)
        r, n = self.norm2.fork(x)
        return self.mlp(n, residual=r)


class PatchEmbed(nn.Module):
    """timm PatchEmbed: Conv2d(in_chans, embed_dim, k = s = patch) -> flatten(2).transpose(1,2).
    A stride = kernel convolution is a GEMM over non-overlapping patches: (n*gh*gw, C*p*p) x (embed, C*p*p)^T."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x, pos=None):
        n, c, hh, ww = x.shape
        p = self.patch_size[0]
        gh, gw = hh // p, ww // p
        # im2col (pure data movement): (n, c, gh, p, gw, p) -> (n*gh*gw, c*p*p) in the conv weight's (c, kh, kw) order
        cols = x.view(n, c, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(n * gh * gw, c * p * p)
        w = self.proj.weight.view(self.proj.weight.shape[0], -1)
        # pos: optional (gh*gw, embed) table added per patch position (row m of the GEMM is patch m % (gh*gw))
        y = ops.linear(cols, w, self.proj.bias, residual=pos, res_rows=(gh * gw if pos is not None else 0))
        return y.view(n, gh * gw, -1)

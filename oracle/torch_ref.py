"""ORACLE -- test infrastructure, NOT product code.

Plain-PyTorch fp32 CPU restatement of the DreamVLA hot path's operators, used only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker for the HIP path.  Each function
cites the reference lines it follows.  Pinning: tests/test_oracle_vs_reference.py checks these functions
and oracle/model_ref.py against the REAL reference modules imported from /root/reference (build
container only) and against the golden fixtures under tests/golden/ (everywhere).

Nothing under dreamvla_amd/ or models/ imports this module.
"""
import math

import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------------
# activations
# ---------------------------------------------------------------------------------------------------


def act(x, name):
    if name in ("none", None):
        return x
    if name in ("gelu", "gelu_erf"):      # timm Mlp act_layer=nn.GELU (erf)  -- models/vit_mae.py:73 via timm Block
        return F.gelu(x)
    if name in ("gelu_tanh", "gelu_new"):  # HF ACT2FN["gelu_new"] -- models/gpt2.py:294; DiT approx_gelu models.py:133
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))
    if name == "relu":
        return F.relu(x)
    if name == "silu":
        return F.silu(x)
    if name == "quick_gelu":              # openai/CLIP model.py QuickGELU
        return x * torch.sigmoid(1.702 * x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    raise ValueError(name)


# ---------------------------------------------------------------------------------------------------
# stateless dropout RNG -- bit-for-bit restatement of dreamvla_amd/csrc/common.h (hash32 / drop_rowkey /
# drop_hash_rk).  The reference uses ATen's Philox stream (models/gpt2.py:56-57,435), which cannot be
# reproduced; parity with the reference is checked at p = 0 and the dropout path is checked against THIS
# restatement of our own generator.
# ---------------------------------------------------------------------------------------------------
_M32 = 0xFFFFFFFF


def _hash32(x):
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def drop_keep_mask(seed, idx_hi, idx_lo, p):
    """keep <=> hash >= floor(p * 2^32).  idx_hi / idx_lo: int64 tensors (broadcastable)."""
    seed_lo, seed_hi = int(seed[0]) & _M32, int(seed[1]) & _M32
    rowkey = (_hash32((idx_hi & _M32) ^ seed_hi) + seed_lo) & _M32
    h = _hash32((rowkey + ((idx_lo & _M32) * 0x9E3779B9 & _M32)) & _M32)
    thr = min(int(p * 4294967296.0), 4294967295)
    return h >= thr


_DROP_C = (0xb60881, 0x554da5, 0x6dada9, 0x9e0fff, 0xd1517f, 0x966d65, 0x764223, 0xb59e1d,
           0x80cd71, 0x769d3b, 0xba1a8f, 0xf85869, 0xf94c5b, 0x905af7, 0xec5577, 0xaa3cd1)


def attn_drop_keep_mask(seed, idx_hi, idx_lo, p):
    """the attention kernels' dropout generator (csrc/common.h, round 4): one hash per (score row, 32-key tile), one 24-bit
    multiply-add per element.  idx_hi = score row id, idx_lo = (compacted) key index; int64 tensors (broadcastable)."""
    seed_lo, seed_hi = int(seed[0]) & _M32, int(seed[1]) & _M32
    rowkey = (_hash32((idx_hi & _M32) ^ seed_hi) + seed_lo) & _M32
    tile, j = idx_lo >> 5, idx_lo & 31
    tk = _hash32((rowkey + ((tile & _M32) * 0x9E3779B9 & _M32)) & _M32)
    half = (j >> 2) & 1
    x = torch.where(half == 1, ((tk >> 12) | (tk << 20)) & _M32, tk)
    c = torch.tensor(_DROP_C, dtype=torch.int64, device=idx_lo.device)[(j & 3) + 4 * (j >> 3)]
    v = ((x & 0xFFFFFF) * c + tk) & _M32
    thr = min(int(p * 4294967296.0), 4294967295)
    return v >= thr


def dropout_elementwise(x2, p, seed):
    """(rows, cols) tensor: idx_hi = row, idx_lo = col."""
    rows, cols = x2.shape
    keep = drop_keep_mask(seed, torch.arange(rows, dtype=torch.int64)[:, None], torch.arange(cols, dtype=torch.int64)[None, :], p)
    return torch.where(keep, x2 / (1.0 - p), torch.zeros_like(x2))


# ---------------------------------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------------------------------
def layer_norm(x, weight, bias, eps):
    """nn.LayerNorm over the last dim (vit_mae.py:77 eps 1e-6; gpt2.py:312-315 eps 1e-5; DiT no affine)."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def linear(x, w, b=None, conv1d=False):
    """nn.Linear (w: (out,in)) or HF Conv1D (w: (in,out); transformers pytorch_utils.Conv1D, gpt2.py:53)."""
    y = x @ (w if conv1d else w.t())
    return y if b is None else y + b


def _drop_rows(B, H, Lq, batch_index):
    """dropout row ids (b*H + h)*Lq + i of a batch (or of the rows `batch_index` of a larger batch)"""
    b = torch.arange(B, dtype=torch.int64) if batch_index is None else torch.as_tensor(batch_index, dtype=torch.int64)
    return ((b.view(B, 1, 1) * H + torch.arange(H, dtype=torch.int64).view(1, H, 1)) * Lq
            + torch.arange(Lq, dtype=torch.int64).view(1, 1, Lq)).unsqueeze(-1)


def attention(q, k, v, scale=None, mask=None, drop=None, drop_cols=None, batch_index=None):
    """softmax(q k^T * scale + mask) v over (B, H, L, d) tensors.
    timm Attention -> F.scaled_dot_product_attention (vit_mae.py:202-203 via Block);
    GPT2Attention._attn (gpt2.py:61-84): scores / sqrt(d) + additive mask, softmax, dropout, @ v;
    PerceiverAttention (perceiver_resampler.py:53-60).  drop = (p, seed) applies our hash mask to the probabilities
    with idx_hi = (b*H + h)*Lq + i, idx_lo = drop_cols[j] (default j; with a compacted key axis the kernel hashes the
    compact key index).  `batch_index`: the true batch indices of the given rows when they are a subset of a larger batch."""
    B, H, Lq, d = q.shape
    Lk = k.shape[2]
    scale = d ** -0.5 if scale is None else scale
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    if drop is not None and drop[0] > 0:
        pd, seed = drop
        rows = _drop_rows(B, H, Lq, batch_index)
        cols = (torch.arange(Lk, dtype=torch.int64) if drop_cols is None else drop_cols.to(torch.int64)).view(1, 1, 1, Lk)
        keep = attn_drop_keep_mask(seed, rows, cols, pd)
        p = torch.where(keep, p / (1.0 - pd), torch.zeros_like(p))
    return torch.matmul(p, v)


LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453


def attention_bf16(q, k, v, scale=None, mask=None, drop=None, drop_cols=None, dout=None, batch_index=None):
    """The SAME attention as `attention` above (same reference lines), restated with the two bf16 rounding points every
    bf16 flash attention has -- the probabilities that enter the P.V matrix product and the score gradients that enter
    the dQ / dK products -- placed exactly where dreamvla_amd/csrc/attention.hip places them, so that the HIP kernels can
    be held to 1e-3 instead of a tolerance that absorbs those roundings (round-2 VERDICT).  fp32 everywhere else.

      forward   s2 = (q k^T) * (scale * log2 e) + mask;  M = ceil(rowmax s2)  (an INTEGER: every online-softmax rescale
                in the kernel is an exact power of two, so rounding P commutes with it and the result does not depend on the
                tile order);  p = 2^(s2 - M);  l = sum_j p  (fp32, unrounded p);  P~ = bf16(keep ? p : 0);
                o = bf16((P~ v) / (keep_rate l));  lse = (M + log2 l) ln 2
      backward  P = 2^(s2 - lse log2 e)  (from the saved lse, unrounded);  dP' = keep ? dout v^T : 0;
                delta = rowsum(dout * o)  (o as stored: bf16);  dS~ = bf16(P (c1 dP' - scale delta)),  c1 = scale / keep_rate;
                dq = bf16(dS~ k),  dk = bf16(dS~^T q),  dv = bf16((bf16(keep ? P : 0)^T dout) / keep_rate)
    (round 6: the dropout's 1 / keep_rate is applied to the accumulated output and to the accumulated dV instead of to every
    probability before its bf16 rounding, and rides on the fma that forms dS -- the same attention-probability dropout
    (gpt2.py:82), one multiply per output element instead of one per score)

    Pinned by tests/test_attention_oracle.py: equals `attention` (and its autograd) up to those two roundings -- 3e-3 /
    4e-3 rel-L2 on random data -- and is exactly invariant to the order in which key tiles are visited.
    (B, H, L, d) tensors; returns (o, lse) or (o, lse, dq, dk, dv) when `dout` is given."""
    B, H, Lq, d = q.shape
    Lk = k.shape[2]
    scale = d ** -0.5 if scale is None else scale
    q, k, v = q.float(), k.float(), v.float()
    s = torch.matmul(q, k.transpose(-1, -2))
    s2 = s * (scale * LOG2E)
    vis = None
    if mask is not None:
        vis = (mask == 0).expand(B, H, Lq, Lk) if mask.dim() < 4 else (mask == 0)
        s2 = torch.where(vis, s2, torch.full_like(s2, float("-inf")))
    m = torch.ceil(s2.amax(dim=-1, keepdim=True))
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = torch.exp2(s2 - m)
    l = p.sum(dim=-1, keepdim=True)
    keep = None
    inv_keep = 1.0
    if drop is not None and drop[0] > 0:
        pd, seed = drop
        rows = _drop_rows(B, H, Lq, batch_index)
        cols = (torch.arange(Lk, dtype=torch.int64) if drop_cols is None else drop_cols.to(torch.int64)).view(1, 1, 1, Lk)
        keep = attn_drop_keep_mask(seed, rows, cols, pd)
        inv_keep = 1.0 / (1.0 - pd)
    pdrop = p if keep is None else torch.where(keep, p, torch.zeros_like(p))
    some = l > 0                    # a row that sees no key: o = 0, lse = +inf (the kernel's convention; the reference never builds one)
    o = bf16_round(torch.where(some, torch.matmul(bf16_round(pdrop), v) * (inv_keep / torch.where(some, l, torch.ones_like(l))),
                               torch.zeros(1)))
    lse = torch.where(some, (m + torch.log2(torch.where(some, l, torch.ones_like(l)))) * LN2,
                      torch.full_like(l, float("inf"))).squeeze(-1)
    if dout is None:
        return o, lse
    dout = dout.float()
    P = torch.exp2(s2 - (lse * LOG2E).unsqueeze(-1))
    if vis is not None:
        P = torch.where(vis, P, torch.zeros_like(P))
    dP = torch.matmul(dout, v.transpose(-1, -2))
    if keep is not None:
        dP = torch.where(keep, dP, torch.zeros_like(dP))
    delta = (dout * o).sum(dim=-1, keepdim=True)
    dS = bf16_round(P * (dP * (scale * inv_keep) - delta * scale))
    Pd = P if keep is None else torch.where(keep, P, torch.zeros_like(P))
    dq = bf16_round(torch.matmul(dS, k))
    dk = bf16_round(torch.matmul(dS.transpose(-1, -2), q))
    dv = bf16_round(torch.matmul(bf16_round(Pd).transpose(-1, -2), dout) * inv_keep)
    return o, lse, dq, dk, dv


def split_qkv(qkv, H):
    """(B, L, 3*H*d) -> q, k, v as (B, H, L, d): timm `reshape(B,N,3,h,d).permute(2,0,3,1,4)`; GPT-2
    `split(H, dim=2)` + `_split_heads` (gpt2.py:136-142,160-164)."""
    B, L, W = qkv.shape
    d = W // (3 * H)
    t = qkv.view(B, L, 3, H, d).permute(2, 0, 3, 1, 4)
    return t[0], t[1], t[2]


def merge_heads(o):
    B, H, L, d = o.shape
    return o.permute(0, 2, 1, 3).reshape(B, L, H * d)


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


def clip_adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay, max_norm=None):
    """One `clip_grad_norm_(max_norm)` + AdamW step on lists of bf16 tensors (updated in place), restating torch's
    semantics with bf16 parameters (train.py optimizer step of the reference; torch/optim/adamw.py fused path:
    fp32 math per element, parameters and both moments stored in bf16; clip_grad_norm_: total L2 norm in fp32,
    coef = min(1, max_norm / (norm + 1e-6)), gradients scaled in place i.e. rounded to bf16).  Returns the norm."""
    norm = None
    coef = 1.0
    if max_norm is not None:
        norm = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
        coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    b1, b2 = betas
    bc1 = 1.0 - b1 ** step
    bc2_sqrt = (1.0 - b2 ** step) ** 0.5
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        gf = (g.float() * coef).to(torch.bfloat16).float()
        pf = p.float() * (1.0 - lr * weight_decay)
        mf = m.float() + (gf - m.float()) * (1.0 - b1)
        vf = b2 * v.float() + (1.0 - b2) * gf * gf
        denom = vf.sqrt() / bc2_sqrt + eps
        pf = pf - (lr / bc1) * (mf / denom)
        p.copy_(pf.to(torch.bfloat16)); m.copy_(mf.to(torch.bfloat16)); v.copy_(vf.to(torch.bfloat16))
    return norm

"""GPT-2 trunk on the HIP kernels -- host-side mirror of /root/reference/models/gpt2.py.

Same module tree / parameter names as the reference's trimmed HF GPT-2 (h.{i}.ln_1, attn.c_attn, attn.c_proj,
ln_2, mlp.c_fc, mlp.c_proj, ln_f; Conv1D weights are (in, out)).  Pre-LN blocks, gelu_new MLP, dropout 0.1 on
embeddings / attention probabilities / both residual branches in training (HF GPT2Config defaults, gpt2.py:56-57,
296,435).  The attention mask is the additive block mask of generate_attention_mask; the HF causal `bias` buffer
is never applied by the reference (gpt2.py:61-84) and is not materialised here.

Per block the work is 6 kernel launches forward: LN -> c_attn GEMM -> fused attention (mask bit tables, tile
skipping, in-kernel dropout) -> c_proj GEMM (+bias, dropout, residual in the epilogue) -> LN -> fused MLP
(c_fc GEMM + bias + gelu_new, c_proj GEMM + bias + dropout + residual).
"""
import math

import torch
from torch import nn

from . import ops
from .nn import Conv1D, LayerNorm


class GPT2Config:
    """The subset of transformers.GPT2Config the reference reads (dreamvla_model.py:301-308), same defaults."""

    def __init__(self, hidden_size=768, n_layer=12, n_head=12, vocab_size=50257, n_inner=None,
                 activation_function="gelu_new", resid_pdrop=0.1, embd_pdrop=0.1, attn_pdrop=0.1,
                 layer_norm_epsilon=1e-5, initializer_range=0.02, max_position_embeddings=1024,
                 attn_implementation="sdpa"):
        self.hidden_size, self.n_layer, self.n_head, self.vocab_size = hidden_size, n_layer, n_head, vocab_size
        self.n_inner, self.activation_function = n_inner, activation_function
        self.resid_pdrop, self.embd_pdrop, self.attn_pdrop = resid_pdrop, embd_pdrop, attn_pdrop
        self.layer_norm_epsilon, self.initializer_range = layer_norm_epsilon, initializer_range
        self.max_position_embeddings = max_position_embeddings
        self.attn_implementation = attn_implementation

    @property
    def num_hidden_layers(self):
        return self.n_layer

    @property
    def num_attention_heads(self):
        return self.n_head


class GPT2Attention(nn.Module):
    def __init__(self, config, layer_idx=None):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.embed_dim // self.num_heads
        if self.head_dim != 64:
            raise ValueError("HIP attention kernels are specialised for head_dim 64 "
                             f"(hidden {self.embed_dim} / heads {self.num_heads} = {self.head_dim})")
        self.layer_idx = layer_idx
        self.c_attn = Conv1D(3 * self.embed_dim, self.embed_dim)
        self.c_proj = Conv1D(self.embed_dim, self.embed_dim)
        self.attn_pdrop, self.resid_pdrop = config.attn_pdrop, config.resid_pdrop

    def forward(self, hidden_states, mask_tables=None, residual=None):
        qkv = self.c_attn(hidden_states)
        o = ops.self_attention(qkv, self.num_heads, scale=1.0 / math.sqrt(self.head_dim), mask_tables=mask_tables,
                               dropout_p=self.attn_pdrop if self.training else 0.0)
        return self.c_proj(o, residual=residual, dropout_p=self.resid_pdrop if self.training else 0.0)


GPT2SdpaAttention = GPT2Attention  # same math; the eager/sdpa switch of the reference only picks an ATen path


class GPT2MLP(nn.Module):
    def __init__(self, intermediate_size, config):
        super().__init__()
        self.c_fc = Conv1D(intermediate_size, config.hidden_size)
        self.c_proj = Conv1D(config.hidden_size, intermediate_size)
        self.act_name = config.activation_function
        self.resid_pdrop = config.resid_pdrop

    def forward(self, hidden_states, residual=None):
        return ops.mlp(hidden_states, self.c_fc.weight, self.c_fc.bias, self.c_proj.weight, self.c_proj.bias,
                       act=self.act_name, conv1d=True, residual=residual,
                       dropout_p=self.resid_pdrop if self.training else 0.0)


class GPT2Block(nn.Module):
    def __init__(self, config, layer_idx=None):
        super().__init__()
        hidden_size = config.hidden_size
        inner_dim = config.n_inner if config.n_inner is not None else 4 * hidden_size
        self.ln_1 = LayerNorm(hidden_size, eps=config.layer_norm_epsilon)
        self.attn = GPT2Attention(config, layer_idx=layer_idx)
        self.ln_2 = LayerNorm(hidden_size, eps=config.layer_norm_epsilon)
        self.mlp = GPT2MLP(inner_dim, config)

    def forward(self, hidden_states, mask_tables=None):
        res, normed = self.ln_1.fork(hidden_states)     # x and LN(x): backward adds dL/dx of both paths in one kernel
        hidden_states = self.attn(normed, mask_tables=mask_tables, residual=res)
        res, normed = self.ln_2.fork(hidden_states)
        return self.mlp(normed, residual=res)


class GPT2Model(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_dim = config.hidden_size
        self.embd_pdrop = config.embd_pdrop
        self.h = nn.ModuleList([GPT2Block(config, layer_idx=i) for i in range(config.num_hidden_layers)])
        self.ln_f = LayerNorm(self.embed_dim, eps=config.layer_norm_epsilon)
        self.gradient_checkpointing = False
        self.apply(self._init_weights)

    def _init_weights(self, module):
        """HF GPT2PreTrainedModel._init_weights (gpt2.py:359-384)."""
        std = self.config.initializer_range
        if isinstance(module, (nn.Linear, Conv1D)):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        for name, p in module.named_parameters():
            if name == "c_proj.weight":
                p.data.normal_(mean=0.0, std=(std / math.sqrt(2 * self.config.n_layer)))

    def forward(self, attention_mask=None, inputs_embeds=None, mask_tables=None):
        """inputs_embeds (B, L, H); attention_mask: additive 0/-inf mask, (L, L) or the (B,1,L,L) expansion the
        reference's sdpa branch builds (dreamvla_model.py:769-775) -- the batch copies are identical, row 0 is used."""
        mt = mask_tables      # tables built on the device from the mask rule (pretrain phase): `attention_mask` is then unused
        if mt is None and attention_mask is not None:
            m2 = attention_mask
            while m2.dim() > 2:
                m2 = m2[0]
            mt = ops.mask_tables_for(m2)
        hidden_states = ops.dropout(inputs_embeds, self.embd_pdrop, self.training)
        for block in self.h:
            hidden_states = block(hidden_states, mask_tables=mt)
        hidden_states = self.ln_f(hidden_states)
        return hidden_states.view(inputs_embeds.shape)

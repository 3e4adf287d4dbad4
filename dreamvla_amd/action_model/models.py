"""DiT action denoiser on the HIP kernels -- host-side mirror of /root/reference/models/action_model/models.py
(same constructor arguments, parameter names and shapes; the DiT LayerNorms have no parameters)."""
import math

import torch
import torch.nn as nn

from .. import ops
from ..nn import Attention, LayerNorm, Linear, Mlp


class TimestepEmbedder(nn.Module):
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size

    _FREQS = {}

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        half = dim // 2
        key = (half, max_period, str(t.device))
        freqs = TimestepEmbedder._FREQS.get(key)
        if freqs is None:   # computed on the host exactly as the reference does (models.py:44-49), copied to the device once
            freqs = TimestepEmbedder._FREQS[key] = torch.exp(
                -math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(device=t.device)
        args = t[:, None].float() * freqs[None]
        embedding = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            embedding = torch.cat([embedding, torch.zeros_like(embedding[:, :1])], dim=-1)
        return embedding

    def forward(self, t):
        t_freq = self.timestep_embedding(t, self.frequency_embedding_size).to(next(self.mlp.parameters()).dtype)
        h = self.mlp[0](t_freq, act="silu")   # Linear + SiLU fused in the GEMM epilogue
        return self.mlp[2](h)


class LabelEmbedder(nn.Module):
    def __init__(self, in_size, hidden_size, dropout_prob=0.1, conditions_shape=(1, 1, 384)):
        super().__init__()
        self.linear = Linear(in_size, hidden_size)
        self.dropout_prob = dropout_prob
        if dropout_prob > 0:
            self.uncondition = nn.Parameter(torch.empty(conditions_shape[1:]))

    def token_drop(self, conditions, force_drop_ids=None):
        if force_drop_ids is None:
            drop_ids = torch.rand(conditions.shape[0], device=conditions.device) < self.dropout_prob
        else:
            drop_ids = force_drop_ids == 1
        return torch.where(drop_ids.unsqueeze(1).unsqueeze(1).expand(conditions.shape[0], *self.uncondition.shape),
                           self.uncondition.to(conditions.dtype), conditions)

    def forward(self, conditions, train, force_drop_ids=None):
        use_dropout = self.dropout_prob > 0
        if (train and use_dropout) or (force_drop_ids is not None):
            conditions = self.token_drop(conditions, force_drop_ids)
        return self.linear(conditions)


class ActionEmbedder(nn.Module):
    def __init__(self, action_size, hidden_size):
        super().__init__()
        self.linear = Linear(action_size, hidden_size)

    def forward(self, x):
        return self.linear(x)


class HistoryEmbedder(nn.Module):
    def __init__(self, action_size, hidden_size):
        super().__init__()
        self.linear = Linear(action_size, hidden_size)

    def forward(self, x):
        return self.linear(x)


class _PlainLayerNorm(LayerNorm):
    def __init__(self, dim):
        super().__init__(dim, elementwise_affine=False, eps=1e-6)


class DiTBlock(nn.Module):
    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, **block_kwargs):
        super().__init__()
        self.norm1 = _PlainLayerNorm(hidden_size)
        self.attn = Attention(hidden_size, num_heads=num_heads, qkv_bias=True)
        self.norm2 = _PlainLayerNorm(hidden_size)
        self.mlp = Mlp(in_features=hidden_size, hidden_features=int(hidden_size * mlp_ratio), act="gelu_tanh")

    def forward(self, x):
        if ops.ln_fusable(x, x.shape[-1]):
            # evaluation sampler (2 x bs x 6 token rows): the parameter-free LayerNorms run inside the GEMMs that consume them
            x = self.attn(x, residual=x, ln_eps=self.norm1.eps)
            return self.mlp(x, residual=x, ln_eps=self.norm2.eps)
        r, n = self.norm1.fork(x)          # (x, LN(x)): one backward kernel for dL/dx of both paths (ops._LayerNormFork)
        x = self.attn(n, residual=r)
        r, n = self.norm2.fork(x)
        return self.mlp(n, residual=r)


class FinalLayer(nn.Module):
    def __init__(self, hidden_size, out_channels):
        super().__init__()
        self.norm_final = _PlainLayerNorm(hidden_size)
        self.linear = Linear(hidden_size, out_channels, bias=True)

    def forward(self, x):
        if ops.ln_fusable(x, x.shape[-1]):
            return self.linear(x, ln_eps=self.norm_final.eps)
        return self.linear(self.norm_final(x))


class DiT(nn.Module):
    def __init__(self, in_channels=7, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4.0,
                 class_dropout_prob=0.1, token_size=4096, future_action_window_size=1, past_action_window_size=0,
                 learn_sigma=False):
        super().__init__()
        if past_action_window_size != 0:
            raise AssertionError("Error: action_history is not used now")
        self.learn_sigma, self.in_channels, self.num_heads = learn_sigma, in_channels, num_heads
        self.out_channels = in_channels * (2 if learn_sigma else 1)
        self.class_dropout_prob = class_dropout_prob
        self.past_action_window_size, self.future_action_window_size = past_action_window_size, future_action_window_size
        # Registration order is the reference's (action_model/models.py:186-205): it fixes the state_dict key order and the
        # order in which the initialisers below draw from the RNG.
        n_pos = 2 * future_action_window_size + past_action_window_size + 2      # condition token + current action + windows
        for name, build in (
                ("history_embedder", lambda: HistoryEmbedder(action_size=in_channels, hidden_size=hidden_size)),
                ("x_embedder", lambda: ActionEmbedder(action_size=in_channels, hidden_size=hidden_size)),
                ("t_embedder", lambda: TimestepEmbedder(hidden_size)),
                ("z_embedder", lambda: LabelEmbedder(in_size=token_size, hidden_size=hidden_size, dropout_prob=class_dropout_prob,
                                                     conditions_shape=(1, 1, token_size))),
                ("positional_embedding", lambda: nn.Parameter(hidden_size ** -0.5 * torch.randn(n_pos, hidden_size))),
                ("blocks", lambda: nn.ModuleList(DiTBlock(hidden_size, num_heads, mlp_ratio=mlp_ratio) for _ in range(depth))),
                ("final_layer", lambda: FinalLayer(hidden_size, self.out_channels))):
            setattr(self, name, build())
        self.initialize_weights()

    def initialize_weights(self):
        """action_model/models.py:207-232: Xavier-uniform on every Linear (depth-first order, as `self.apply` visits them),
        then N(0, 0.02) on the embedders / timestep MLP and zeros on the output layer."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        narrow = [self.x_embedder.linear.weight, self.history_embedder.linear.weight]
        if self.class_dropout_prob > 0:
            narrow.append(self.z_embedder.uncondition)
        narrow += [self.z_embedder.linear.weight, self.t_embedder.mlp[0].weight, self.t_embedder.mlp[2].weight]
        for w in narrow:
            nn.init.normal_(w, std=0.02)
        for b in (self.x_embedder.linear.bias, self.history_embedder.linear.bias, self.z_embedder.linear.bias,
                  self.final_layer.linear.weight, self.final_layer.linear.bias):
            nn.init.zeros_(b)

    def forward(self, x, t, z):
        """x: (N, T, 7) noisy actions, t: (N,) timesteps, z: (N, T', token) conditions -> (N, T, 7)"""
        wdt = torch.bfloat16       # compute dtype (fp32 parameters are masters: ops.shadow)
        x = self.x_embedder(x.to(wdt))
        t = self.t_embedder(t)
        z = self.z_embedder(z.to(wdt), self.training)
        c = t.unsqueeze(1) + z
        x = torch.cat((c, x), dim=1)
        x = x + self.positional_embedding.to(x.dtype)
        for block in self.blocks:
            x = block(x)
        x = self.final_layer(x)
        return x[:, c.shape[1]:, :]

    def forward_with_cfg(self, x, t, z, cfg_scale):
        half = x[: len(x) // 2]
        combined = torch.cat([half, half], dim=0).to(torch.bfloat16)
        model_out = self.forward(combined, t, z)
        eps, rest = model_out[:, :, :self.in_channels], model_out[:, :, self.in_channels:]
        cond_eps, uncond_eps = torch.split(eps, len(eps) // 2, dim=0)
        half_eps = uncond_eps + cfg_scale * (cond_eps - uncond_eps)
        eps = torch.cat([half_eps, half_eps], dim=0)
        return torch.cat([eps, rest], dim=2)

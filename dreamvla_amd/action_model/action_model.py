"""ActionModel (DiT + diffusion loss) -- host-side mirror of /root/reference/models/action_model/action_model.py."""
import torch
from torch import nn

from .gaussian_diffusion import create_diffusion
from .models import DiT


def DiT_S(**kwargs):
    return DiT(depth=6, hidden_size=384, num_heads=4, **kwargs)   # head_dim 96: the short-sequence kernel (csrc/attention_small.hip)


def DiT_B(**kwargs):
    return DiT(depth=12, hidden_size=768, num_heads=12, **kwargs)


def DiT_L(**kwargs):
    return DiT(depth=24, hidden_size=1024, num_heads=16, **kwargs)


DiT_models = {'DiT-S': DiT_S, 'DiT-B': DiT_B, 'DiT-L': DiT_L}


class ActionModel(nn.Module):
    def __init__(self, token_size, model_type, in_channels, future_action_window_size, past_action_window_size,
                 diffusion_steps=100, noise_schedule='squaredcos_cap_v2'):
        super().__init__()
        self.in_channels = in_channels
        self.noise_schedule = noise_schedule
        self.diffusion_steps = diffusion_steps
        self.diffusion = create_diffusion(timestep_respacing="", noise_schedule=noise_schedule,
                                          diffusion_steps=self.diffusion_steps, sigma_small=True, learn_sigma=False)
        self.ddim_diffusion = None
        self.past_action_window_size = past_action_window_size
        self.future_action_window_size = future_action_window_size
        self.net = DiT_models[model_type](token_size=token_size, in_channels=in_channels, class_dropout_prob=0.1,
                                          learn_sigma=False, future_action_window_size=future_action_window_size,
                                          past_action_window_size=past_action_window_size)

    def loss(self, x, z, noise=None, timestep=None):
        """mean((eps_hat - eps)^2) (action_model.py:57-73).  `noise` / `timestep` may be injected for parity tests;
        by default they are drawn exactly like the reference (randn_like, randint(0, 100))."""
        inj = getattr(self, "_injected", None)     # parity tests inject (noise, timestep)
        if inj is not None and noise is None:
            noise, timestep = inj[0].to(x.dtype), inj[1]
        if noise is None:
            noise = torch.randn_like(x)
        if timestep is None:
            timestep = torch.randint(0, self.diffusion.num_timesteps, (x.size(0),), device=x.device)
        x_t = self.diffusion.q_sample(x, timestep, noise)
        noise_pred = self.net(x_t, timestep, z)
        assert noise_pred.shape == noise.shape == x.shape
        return ((noise_pred.float() - noise.float()) ** 2).mean()   # fp32 scalar (the reference's is in the model dtype)

    def create_ddim(self, ddim_step=10):
        self.ddim_diffusion = create_diffusion(timestep_respacing="ddim" + str(ddim_step),
                                               noise_schedule=self.noise_schedule,
                                               diffusion_steps=self.diffusion_steps, sigma_small=True, learn_sigma=False)
        return self.ddim_diffusion

    @torch.no_grad()
    def sample_ddim_cfg(self, cond, noise, cfg_scale):
        """The evaluation sampler of models/dreamvla_model.py:935-987 -- `ddim_sample_loop(net.forward_with_cfg, ..., eta=0)` over
        [cond ; uncondition] -- with the work that does not depend on the sampler step taken out of the loop and the per-step
        tensor algebra in one kernel.  Same arithmetic, same rounding points:
          * the condition embedding z_embedder([cond ; uncondition]) is computed once (the reference recomputes it in each of
            the 10 steps from the same input);
          * every row of a step has the same timestep, and the 10 timesteps are known: t_embedder runs once on the 10 of them;
          * both halves of the sampler state are the same tensor (forward_with_cfg feeds [half ; half] and returns
            [eps ; eps], action_model/models.py:253-268): the state is kept once, x_embedder runs on bs rows;
          * guidance + DDIM update: ops.ddim_cfg_step (one launch instead of ~20 elementwise ATen launches on (bs, 3, 7)).
        cond: (bs, T, token) conditions; noise: (bs, T, C) start noise.  Returns the samples (bs, T, C), float32."""
        import numpy as np
        from .. import ops
        net, dd = self.net, self.ddim_diffusion
        bs, T = noise.shape[0], noise.shape[1]
        wdt = torch.bfloat16
        unc = net.z_embedder.uncondition.to(cond.dtype).unsqueeze(0).expand(bs, T, -1)
        z_emb = net.z_embedder(torch.cat([cond, unc], 0).to(wdt), False)               # (2 bs, T, H)
        steps = list(range(dd.num_timesteps))[::-1]
        cache = self.__dict__.setdefault("_fast_tables", {})
        key = (str(cond.device), dd.num_timesteps)
        if key not in cache:        # built outside any stream capture (the engine's eager warm-up calls come first)
            cache[key] = torch.tensor([dd.timestep_map[i] for i in steps], device=cond.device, dtype=torch.long)
        t_emb = net.t_embedder(cache[key])                                               # (steps, H): one call
        pos = net.positional_embedding.to(wdt)
        f32 = np.float32
        x = noise.float().contiguous()

        def coefficients(i):
            # the coefficients as `_extract_into_tensor` gathers them: float64 tables read as float32, sqrt taken in float32
            acp_prev = f32(dd.alphas_cumprod_prev[i])
            return (f32(dd.sqrt_recip_alphas_cumprod[i]), f32(dd.sqrt_recipm1_alphas_cumprod[i]), np.sqrt(acp_prev),
                    np.sqrt(f32(1.0) - acp_prev))

        hidden = net.x_embedder.linear.out_features
        if getattr(self, "team_sampler", True) and ops.dit_team_ok(hidden, net.num_heads, net.in_channels, T, bs, cond.device):
            # the whole loop below as ONE persistent kernel on the CUs of one XCD (csrc/dit_team.hip): same arithmetic and
            # rounding points, the ~60 launches of a sampler step become exchanges through that XCD's L2
            ck = ("team", str(cond.device), dd.num_timesteps)
            if ck not in cache:
                cache[ck] = {"coef": torch.tensor(np.array([coefficients(i) for i in steps], dtype=np.float32), device=cond.device),
                             "ws": ops.dit_team_workspace(hidden, cond.device), "ptrs": None, "table": None, "timeouts": 0}
            st = cache[ck]
            ws = [ops.shadow(w).contiguous() for blk in net.blocks
                  for w in (blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight, blk.attn.proj.bias,
                            blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias)]
            ptrs = tuple(w.data_ptr() for w in ws)
            if st["ptrs"] != ptrs:          # (re)built outside a capture: the engine's eager warm-up calls come first
                st["table"] = torch.tensor(ptrs, dtype=torch.int64, device=cond.device).view(len(net.blocks), 8)
                st["ptrs"], st["keep"] = ptrs, ws
            capturing = torch.cuda.is_current_stream_capturing()
            # eager calls compare the workspace's timeout count after THEIR launch with the last count seen on this workspace
            # (st["timeouts"]: kept current by the previous eager call and by RolloutEngine, which reads the count after every
            # replay of a graph that contains the kernel) -- one host read per eager call, not two (round-5 ADVICE)
            timeouts_before = None if capturing else st["timeouts"]
            cond_tab = (z_emb.unsqueeze(0) + t_emb.view(t_emb.shape[0], 1, 1, -1)).contiguous()    # z_emb + t_emb[j], all steps
            sh = lambda w: ops.shadow(w).contiguous()
            out = ops.dit_team_sample(st["table"], len(net.blocks), hidden, net.num_heads, sh(net.x_embedder.linear.weight),
                                      sh(net.x_embedder.linear.bias), sh(net.final_layer.linear.weight),
                                      sh(net.final_layer.linear.bias), pos.contiguous(), cond_tab, st["coef"], x, cfg_scale,
                                      net.blocks[0].norm1.eps, st["ws"])
            self.team_launches = getattr(self, "team_launches", 0) + 1       # (a captured launch counts once: RolloutEngine reads
            #                                                                    "did the decode graph contain the team kernel" off it)
            if not capturing:
                # eager calls check at once.  Under hipGraph replay no Python runs: RolloutEngine.step checks the ACTION it is about
                # to hand out and falls back to the launch-by-launch sampler (round-4 ADVICE); the kernel retires its own status,
                # so the launch after a timeout is clean.
                timeouts, xcc = ops.dit_team_status(st["ws"])
                self.team_xcc_mask = xcc
                st["timeouts"] = timeouts
                if timeouts != timeouts_before:
                    raise ops.DitTeamTimeout(f"dvla_dit_sample: an exchange inside the kernel timed out (the output of this call is NaN; "
                                       f"{timeouts} launches so far, XCC mask {xcc:#x}); set `team_sampler = False` on the action "
                                       f"model for the launch-by-launch sampler")
            return out
        for j, i in enumerate(steps):
            xe = net.x_embedder(x.to(wdt))                                               # (bs, T, H)
            tok = torch.cat((z_emb + t_emb[j], torch.cat((xe, xe), 0)), dim=1) + pos     # (2 bs, 2 T, H)
            for blk in net.blocks:
                tok = blk(tok)
            out = net.final_layer(tok)[:, T:, :]                                         # (2 bs, T, C)
            x = ops.ddim_cfg_step(out, x, cfg_scale, *coefficients(i))
        return x


class ActionModelFM(nn.Module):
    """Flow-matching variant (`--use_fm`; /root/reference/models/action_model/action_model.py:86-170): same DiT, 10 "diffusion
    steps" that are only the time grid of the flow.  loss: t ~ U{0, .1, ..., .9}, x_t = t x + (1 - t) eps, the network
    predicts the velocity u_t = x - eps (the FLOAT t goes into the timestep embedder as is).  Sampling: FMDiffusion."""

    def __init__(self, token_size, model_type, in_channels, future_action_window_size, past_action_window_size,
                 diffusion_steps=10, noise_schedule='squaredcos_cap_v2'):
        super().__init__()
        self.in_channels = in_channels
        self.noise_schedule = noise_schedule
        self.diffusion_steps = diffusion_steps
        self.diffusion = create_diffusion(timestep_respacing="", noise_schedule=noise_schedule,
                                          diffusion_steps=self.diffusion_steps, sigma_small=True, learn_sigma=False)
        self.ddim_diffusion = None
        self.past_action_window_size = past_action_window_size
        self.future_action_window_size = future_action_window_size
        self.net = DiT_models[model_type](token_size=token_size, in_channels=in_channels, class_dropout_prob=0.1,
                                          learn_sigma=False, future_action_window_size=future_action_window_size,
                                          past_action_window_size=past_action_window_size)

    def loss(self, x, z, noise=None, timestep=None):
        """mean((u_hat - (x - eps))^2) (action_model.py:121-141).  `noise` / integer `timestep` (0 .. steps-1) may be injected
        for parity tests; by default they are drawn exactly like the reference (randn_like, randint(0, steps))."""
        inj = getattr(self, "_injected", None)
        if inj is not None and noise is None:
            noise, timestep = inj[0].to(x.dtype), inj[1]
        if noise is None:
            noise = torch.randn_like(x)
        if timestep is None:
            timestep = torch.randint(0, self.diffusion.num_timesteps, (x.size(0),), device=x.device)
        t = timestep.float() / self.diffusion.num_timesteps
        tv = t.view(-1, 1, 1)
        x_t = (tv * x.float() + (1 - tv) * noise.float()).to(x.dtype)     # (the reference's fp32 `timestep` promotes x_t to fp32)
        ut = self.net(x_t, t, z)
        assert ut.shape == noise.shape == x.shape
        return ((ut.float() - (x.float() - noise.float())) ** 2).mean()

    def create_ddim(self, ddim_step=10):
        from .gaussian_diffusion import FMDiffusion
        self.ddim_diffusion = FMDiffusion(self.diffusion_steps)
        return self.ddim_diffusion

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

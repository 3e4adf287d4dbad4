"""Diffusion schedules + the two samplers DreamVLA uses -- restates the subset of
/root/reference/models/action_model/{gaussian_diffusion,respace,__init__}.py that is reachable from
DreamVLA.forward (dreamvla_model.py:927-987): q_sample for the training loss and the eta = 0 DDIM loop
with classifier-free guidance for evaluation.  Tables are float64 numpy exactly as in the reference
(gaussian_diffusion.py:116-201) and are gathered to float32 per timestep (`_extract_into_tensor`, 870-882).
The learned-sigma / KL / DDPM-ancestral / conditioning paths (dead for DreamVLA: learn_sigma=False,
sigma_small=True, predict eps, DDIM only) are not restated.

The per-step sampler algebra runs on (bs, 3, 7) tensors: it is done with a handful of elementwise torch ops on
the device (a few hundred bytes each); the DiT forward inside the loop is the HIP path.
"""
import math

import numpy as np
import torch as th


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    betas = []
    for i in range(num_diffusion_timesteps):
        t1 = i / num_diffusion_timesteps
        t2 = (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(betas)


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "squaredcos_cap_v2":
        return betas_for_alpha_bar(num_diffusion_timesteps, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def space_timesteps(num_timesteps, section_counts):
    """respace.py:12-65"""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired_count = int(section_counts[len("ddim"):])
            if desired_count == 1:
                return set([50])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired_count:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start_idx = 0
    all_steps = []
    for i, section_count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < section_count:
            raise ValueError(f"cannot divide section of {size} steps into {section_count}")
        frac_stride = 1 if section_count <= 1 else (size - 1) / (section_count - 1)
        cur_idx = 0.0
        taken_steps = []
        for _ in range(section_count):
            taken_steps.append(start_idx + round(cur_idx))
            cur_idx += frac_stride
        all_steps += taken_steps
        start_idx += size
    return set(all_steps)


_DEV_TABLES = {}


def _device_table(arr, device):
    """float64 copy of a schedule table on `device`, made once: the per-call host->device copy of the reference
    (gaussian_diffusion.py:870-882) is a pageable-memory copy, i.e. a host synchronisation per extract (50 per DDIM-10
    sample) and illegal inside a hipGraph capture.  Same values, same float64 -> float32 gather.
    The entry keeps the numpy array itself alive and is valid for that very object only: keyed on the bare address, a
    collected diffusion object's table could be answered with another same-length table allocated at that address
    (sqrt_one_minus_alphas_cumprod where sqrt_alphas_cumprod was) -- round-1 ADVICE."""
    key = (id(arr), str(device))
    ent = _DEV_TABLES.get(key)
    if ent is not None and ent[0] is arr:
        return ent[1]
    if len(_DEV_TABLES) > 256:
        _DEV_TABLES.clear()
    t = th.from_numpy(arr).to(device=device)
    _DEV_TABLES[key] = (arr, t)
    return t


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    res = _device_table(arr, timesteps.device)[timesteps].float()
    while len(res.shape) < len(broadcast_shape):
        res = res[..., None]
    return res + th.zeros(broadcast_shape, device=timesteps.device)


class GaussianDiffusion:
    def __init__(self, *, betas):
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1 and (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)

    def q_sample(self, x_start, t, noise):
        """gaussian_diffusion.py:215-230"""
        return (_extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + _extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def _model_timesteps(self, t):
        return t

    def ddim_sample(self, model, x, t, model_kwargs=None):
        """eta = 0, clip_denoised = False, eps-prediction (gaussian_diffusion.py:255-353,522-569)."""
        model_output = model(x, self._model_timesteps(t), **(model_kwargs or {}))
        model_output = model_output.to(x.dtype)
        pred_xstart = (_extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x.shape) * x
                       - _extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x.shape) * model_output)
        eps = ((_extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x.shape) * x - pred_xstart)
               / _extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x.shape))
        alpha_bar_prev = _extract_into_tensor(self.alphas_cumprod_prev, t, x.shape)
        mean_pred = pred_xstart * th.sqrt(alpha_bar_prev) + th.sqrt(1 - alpha_bar_prev) * eps
        return {"sample": mean_pred, "pred_xstart": pred_xstart}

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=False, model_kwargs=None, device=None,
                         progress=False, eta=0.0, **_unused):
        assert eta == 0.0 and not clip_denoised, "only the eta=0, unclipped DDIM path of the reference is restated"
        img = noise if noise is not None else th.randn(*shape, device=device)
        for i in list(range(self.num_timesteps))[::-1]:
            t = th.full((shape[0],), i, device=img.device, dtype=th.long)   # a fill kernel, not a host->device copy
            with th.no_grad():
                img = self.ddim_sample(model, img, t, model_kwargs=model_kwargs)["sample"]
        return img


class FMDiffusion:
    """Euler integrator of the flow-matching head (/root/reference/models/action_model/respace.py:118-191): starts from FRESH
    noise (the `noise` the caller passes is ignored, as upstream: its positional slot is swallowed by *args), forces the
    guidance scale to 1 and takes num_timesteps steps x <- x + u(x, i / n) / n.  Upstream hard-codes device='cuda' for the
    start noise; here it is drawn on `device` (the same thing on a GPU box)."""

    def __init__(self, num_timesteps):
        self.num_timesteps = int(num_timesteps)

    def ddim_sample_loop(self, model, shape, *args, noise=None, clip_denoised=True, model_kwargs=None, device=None,
                         progress=False, start_noise=None, **kwargs):
        """`start_noise` (not upstream): the start noise as an input, for captured graphs and parity tests
        (DreamVLA.decode_tokens(test_noise=...)); None = drawn here, as upstream."""
        model_kwargs = dict(model_kwargs or {})
        if "cfg_scale" in model_kwargs:
            model_kwargs["cfg_scale"] = 1.0
        final = th.randn(*shape, device=device) if start_noise is None else start_noise.to(device=device, dtype=th.float32)
        delta = 1.0 / self.num_timesteps
        for i in range(self.num_timesteps):
            t = th.full((shape[0],), float(i) / self.num_timesteps, device=device, dtype=th.float32)
            with th.no_grad():
                ut = model(final, t, **model_kwargs)
            final = final + delta * ut.to(final.dtype)
        return final


class SpacedDiffusion(GaussianDiffusion):
    """respace.py:67-116,193-205: keep a subset of timesteps, re-derive betas, feed the model ORIGINAL indices."""

    def __init__(self, use_timesteps, betas):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(betas)
        base = GaussianDiffusion(betas=betas)
        last_alpha_cumprod = 1.0
        new_betas = []
        for i, alpha_cumprod in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - alpha_cumprod / last_alpha_cumprod)
                last_alpha_cumprod = alpha_cumprod
                self.timestep_map.append(i)
        super().__init__(betas=np.array(new_betas))

    def _model_timesteps(self, t):
        key = (str(t.device), t.dtype)
        cache = self.__dict__.setdefault("_map_tensors", {})
        map_tensor = cache.get(key)
        if map_tensor is None:
            map_tensor = cache[key] = th.tensor(self.timestep_map, device=t.device, dtype=t.dtype)
        return map_tensor[t]


def create_diffusion(timestep_respacing, noise_schedule="linear", use_kl=False, sigma_small=False,
                     predict_xstart=False, learn_sigma=True, rescale_learned_sigmas=False, diffusion_steps=1000):
    """action_model/__init__.py:9-45 for the configuration DreamVLA uses (eps prediction, fixed small sigma)."""
    if use_kl or predict_xstart or learn_sigma or rescale_learned_sigmas:
        raise NotImplementedError("only the eps-prediction / fixed-sigma configuration used by DreamVLA is restated")
    betas = get_named_beta_schedule(noise_schedule, diffusion_steps)
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    return SpacedDiffusion(use_timesteps=space_timesteps(diffusion_steps, timestep_respacing), betas=betas)

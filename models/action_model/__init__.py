"""Reference import path /root/reference/models/action_model -> MI355X implementation."""
from dreamvla_amd.action_model import (ActionModel, ActionModelFM, DiT, DiT_models, GaussianDiffusion,  # noqa: F401
                                       SpacedDiffusion, create_diffusion, space_timesteps)

"""DiT action head + Gaussian diffusion utilities -- host-side mirror of /root/reference/models/action_model/."""
from .gaussian_diffusion import create_diffusion, GaussianDiffusion, SpacedDiffusion, space_timesteps  # noqa: F401
from .models import DiT  # noqa: F401
from .action_model import ActionModel, ActionModelFM, DiT_models  # noqa: F401

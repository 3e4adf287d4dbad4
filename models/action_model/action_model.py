from dreamvla_amd.action_model.action_model import ActionModel, ActionModelFM, DiT_B, DiT_L, DiT_S, DiT_models  # noqa: F401

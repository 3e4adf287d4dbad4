"""Reference import path /root/reference/models/dreamvla_model.py -> MI355X implementation."""
from dreamvla_amd.dreamvla_model import (DreamVLA, SiLogLoss, generate_attention_mask, get_1d_sincos_pos_embed,  # noqa: F401
                                         get_1d_sincos_pos_embed_from_grid, get_2d_sincos_pos_embed,
                                         get_2d_sincos_pos_embed_from_grid)

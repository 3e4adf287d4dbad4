"""Reference import path /root/reference/models/vit_mae.py -> MI355X implementation."""
from dreamvla_amd.vit_mae import (MaskedAutoencoderViT, get_1d_sincos_pos_embed_from_grid, get_2d_sincos_pos_embed,  # noqa: F401
                                  get_2d_sincos_pos_embed_from_grid)

"""Reference import path /root/reference/models/gpt2.py -> MI355X implementation."""
from dreamvla_amd.gpt2 import (GPT2Attention, GPT2Block, GPT2Config, GPT2MLP, GPT2Model, GPT2SdpaAttention)  # noqa: F401

"""Reference import path /root/reference/models/perceiver_resampler.py -> MI355X implementation."""
from dreamvla_amd.perceiver_resampler import FeedForward, PerceiverAttention, PerceiverResampler  # noqa: F401

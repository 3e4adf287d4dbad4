"""ORACLE SCAFFOLDING (tests only) -- `einops_exts.rearrange_many` (used by
/root/reference/models/perceiver_resampler.py:3,52): map einops.rearrange over a tuple."""
from einops import rearrange


def rearrange_many(tensors, pattern, **axes_lengths):
    return tuple(rearrange(t, pattern, **axes_lengths) for t in tensors)

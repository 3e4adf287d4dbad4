"""Fused gradient clipping + AdamW over the flat buffers of the data-parallel gradient reducer.

The reference's step is caller code: `torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)` followed by
`torch.optim.AdamW.step()` (train.py / utils/train_utils.py:600-608), i.e. ~80 multi-tensor launches over ~1000
tensors.  Here every trainable parameter is re-homed as a view into one flat bf16 buffer per gradient bucket of
`dreamvla_amd.ddp.GradBucketReducer` (same element order as the bucket), both moments are flat too, and a step is
`dvla_sumsq_bf16` per bucket (gradient norm, accumulated into one device scalar -- no host synchronisation) plus
`dvla_adamw_bf16` per bucket.  Element-wise semantics are those of torch's AdamW with bf16 parameters (moments kept in
bf16, math in fp32) and of `clip_grad_norm_` (scaled gradient rounded to bf16).  Like torch with gradient views, every
parameter of a bucket is updated each step -- also the ones that received no gradient (zero gradient, weight decay
only).
"""
import torch

from . import _lib
from ._lib import check


class FlatAdamW:
    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        self.reducer = reducer
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.max_grad_norm = None if max_grad_norm is None else float(max_grad_norm)
        self.step_count = 0
        self.state = []
        lib = _lib.load()
        for b in reducer.buckets:
            g = b["flat"]
            if not g.is_cuda or g.dtype != torch.bfloat16:
                raise TypeError("FlatAdamW: bf16 CUDA gradient buckets only (no CPU / fp32 fallback)")
            flat_p = torch.zeros_like(g)
            for p, off in zip(b["params"], b["offsets"]):       # same (256-B aligned) offsets as the gradient views
                n = p.numel()
                flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + n].view_as(p)        # the parameter now lives inside the flat buffer
            self.state.append({"p": flat_p, "g": g, "m": torch.zeros_like(g), "v": torch.zeros_like(g)})
        dev = reducer.buckets[0]["flat"].device
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._partial = torch.empty(int(lib.dvla_sumsq_partial_len()), dtype=torch.float32, device=dev)

    @torch.no_grad()
    def step(self):
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        self.step_count += 1
        clip = self.max_grad_norm is not None
        if clip:
            for i, s in enumerate(self.state):
                check(lib.dvla_sumsq_bf16(s["g"].data_ptr(), s["g"].numel(), self._partial.data_ptr(), self._sumsq.data_ptr(),
                                          1 if i > 0 else 0, stream), "dvla_sumsq_bf16")
        for s in self.state:
            check(lib.dvla_adamw_bf16(s["p"].data_ptr(), s["g"].data_ptr(), s["m"].data_ptr(), s["v"].data_ptr(), s["p"].numel(),
                                      self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                                      self._sumsq.data_ptr() if clip else None, self.max_grad_norm if clip else 0.0, stream),
                  "dvla_adamw_bf16")

    def grad_norm(self):
        """total gradient norm of the last step() with clipping (device scalar tensor)"""
        return self._sumsq.sqrt()

"""Fused gradient clipping + AdamW over the flat buffers of the data-parallel gradient reducer.

The reference's step is caller code: `torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)` followed by
`torch.optim.AdamW.step()` (train.py / utils/train_utils.py:600-608), i.e. ~80 multi-tensor launches over ~1000
tensors.  Here every trainable parameter is re-homed as a view into one flat bf16 buffer per gradient bucket of
`dreamvla_amd.ddp.GradBucketReducer` (same element order as the bucket), both moments are flat too, and a step is
`dvla_sumsq_bf16` per bucket (gradient norm, accumulated into one device scalar -- no host synchronisation) plus
`dvla_adamw_bf16` per bucket.  Element-wise semantics are those of torch's AdamW with bf16 parameters (moments kept in
bf16, math in fp32) and of `clip_grad_norm_` (scaled gradient rounded to bf16).
"""
import torch

from . import _lib
from ._lib import check


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.Optimizer surface (param_groups with `lr` -- so torch's LR schedulers, which the reference attaches to its
    AdamW (train.py:176-200), drive it unchanged --, state_dict / load_state_dict for checkpoint / resume) over flat buffers.
    Parameters the reducer has learned never to receive a gradient (the reference's constructed-but-unused modules) are left
    alone -- no weight decay, no moment update -- exactly as torch's AdamW skips `p.grad is None` parameters."""

    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        self.reducer = reducer
        super().__init__(list(reducer.params), dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps),
                                                    weight_decay=float(weight_decay)))
        self.max_grad_norm = None if max_grad_norm is None else float(max_grad_norm)
        self.step_count = 0
        self.flat = []
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
            self.flat.append({"p": flat_p, "g": g, "m": torch.zeros_like(g), "v": torch.zeros_like(g)})
        dev = reducer.buckets[0]["flat"].device
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._partial = torch.empty(int(lib.dvla_sumsq_partial_len()), dtype=torch.float32, device=dev)

    def _ranges(self, bi):
        """element ranges of bucket bi that hold parameters which receive gradients (maximal runs; alignment gaps between two
        used neighbours are included -- they are zero in every buffer)"""
        b = self.reducer.buckets[bi]
        total = b["flat"].numel()
        used = b["expected"]
        if all(used):
            return [(0, total)]
        runs, start = [], None
        bounds = list(b["offsets"]) + [total]
        for i, u in enumerate(used):
            if u and start is None:
                start = bounds[i]
            if not u and start is not None:
                runs.append((start, bounds[i]))
                start = None
        if start is not None:
            runs.append((start, total))
        return runs

    @torch.no_grad()
    def step(self, closure=None):
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        self.step_count += 1
        grp = self.param_groups[0]
        lr, (b1, b2), eps, wd = float(grp["lr"]), grp["betas"], float(grp["eps"]), float(grp["weight_decay"])
        clip = self.max_grad_norm is not None
        if clip:
            for i, s in enumerate(self.flat):
                check(lib.dvla_sumsq_bf16(s["g"].data_ptr(), s["g"].numel(), self._partial.data_ptr(), self._sumsq.data_ptr(),
                                          1 if i > 0 else 0, stream), "dvla_sumsq_bf16")
        for bi, s in enumerate(self.flat):
            for (lo, hi) in self._ranges(bi):
                if hi <= lo:
                    continue
                o = 2 * lo      # bf16 byte offset
                check(lib.dvla_adamw_bf16(s["p"].data_ptr() + o, s["g"].data_ptr() + o, s["m"].data_ptr() + o, s["v"].data_ptr() + o,
                                          hi - lo, lr, float(b1), float(b2), eps, wd, self.step_count,
                                          self._sumsq.data_ptr() if clip else None, self.max_grad_norm if clip else 0.0, stream),
                      "dvla_adamw_bf16")

    def zero_grad(self, set_to_none=True):
        self.reducer.zero_grad()

    def grad_norm(self):
        """total gradient norm of the last step() with clipping (device scalar tensor)"""
        return self._sumsq.sqrt()

    def state_dict(self):
        """step count, hyper-parameters and both moments (flat, per bucket) + the layout they belong to"""
        return {"step": self.step_count,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "layout": [[int(p.numel()) for p in b["params"]] for b in self.reducer.buckets],
                "exp_avg": [s["m"].clone() for s in self.flat], "exp_avg_sq": [s["v"].clone() for s in self.flat]}

    def load_state_dict(self, sd):
        layout = [[int(p.numel()) for p in b["params"]] for b in self.reducer.buckets]
        if sd["layout"] != layout:
            raise ValueError("FlatAdamW.load_state_dict: the checkpoint's bucket layout does not match this model / reducer")
        self.step_count = int(sd["step"])
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)
        for s, m, v in zip(self.flat, sd["exp_avg"], sd["exp_avg_sq"]):
            s["m"].copy_(m)
            s["v"].copy_(v)

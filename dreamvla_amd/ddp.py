"""Data-parallel gradient exchange for one process per GPU (RCCL over xGMI through torch.distributed's "nccl"
backend; "gloo" on CPU for tests).

The reference wraps the model in torch DDP with find_unused_parameters=True (train.py:173) because several modules
are constructed but never used.  This reducer is built for the MI355X node instead of translated from that:

  * gradients live in a few LARGE flat buckets (default 256 MiB of bf16): `p.grad` of every trainable parameter is a
    view into its bucket, so backward writes straight into the communication buffer (no copy-in / copy-out);
  * buckets are filled in reverse parameter order (the order backward produces gradients); when the last gradient of
    a bucket has been accumulated its all-reduce is launched asynchronously on the communication stream while
    backward keeps running -- xGMI is point-to-point (7 links x ~153 GB/s per GPU), so few large collectives beat
    many 25 MB ones;
  * parameters that receive no gradient (the reference's constructed-but-unused modules: action_pose_encoder,
    recon_state_decoder, the DiT history embedder ... -- they sit in three of the four buckets of the CALVIN
    configuration) never fire.  The first step flushes their buckets at `finish()`; from the set of parameters that did
    fire the reducer then LEARNS which ones to wait for, so from the second step on a bucket is launched as soon as its
    used parameters are done and the exchange overlaps with backward (no per-iteration graph walk as in
    find_unused_parameters=True).  If the used set ever grows, the bucket is held until `finish()` for that step and the
    set is re-learned; a gradient arriving after its bucket was launched raises (pass static_unused=False to always
    flush such buckets at `finish()`);
  * `finish()` waits for the handles and averages (divide by world size), exactly DDP's semantics;
  * `direct_grads=True`: `zero_grad()` leaves `p.grad = None` and publishes each parameter's bucket slot as
    `p._dvla_grad_view`; the backward of dreamvla_amd.ops (weight-gradient GEMMs, bias column sums, LayerNorm parameter
    gradients) writes its result straight into that slot and returns it, autograd's AccumulateGrad then adopts the tensor
    (no `grad += new` kernel per parameter -- ~400 launches per step on DreamVLA), and the hook below copies only the
    gradients that some other operator produced elsewhere (learned tokens, position embeddings).
"""
import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params, bucket_bytes=256 << 20, process_group=None, static_unused=True, direct_grads=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.static_unused = bool(static_unused)
        self.direct_grads = bool(direct_grads)
        self.copied = 0     # direct_grads: gradients the hook had to copy into their slot (not produced in place)
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets = []       # dicts: flat, params, pending
        order = list(reversed(self.params))
        cur, cur_bytes, cur_key = [], 0, None
        for p in order:
            key = (p.dtype, p.device)
            nbytes = p.numel() * p.element_size()
            if cur and (key != cur_key or cur_bytes + nbytes > bucket_bytes):
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_key = key
        if cur:
            self._make_bucket(cur)
        self._handles = []
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b["params"]):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi, pi)))

    ALIGN = 128   # elements: every view starts on a 256-byte boundary (the kernels read parameters / gradients as 16-B vectors)

    def _make_bucket(self, plist):
        offsets, off = [], 0
        for p in plist:
            offsets.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        flat = torch.zeros(off, dtype=plist[0].dtype, device=plist[0].device)
        for p, o in zip(plist, offsets):
            view = flat[o:o + p.numel()].view_as(p)
            p._dvla_grad_view = view
            p._dvla_grad_free = False
            p.grad = None if self.direct_grads else view
        self.buckets.append({"flat": flat, "params": plist, "offsets": offsets, "pending": len(plist), "launched": False,
                             "fired": [False] * len(plist),      # this step
                             "expected": [True] * len(plist),    # parameters the launch waits for (learned)
                             "hold": False})

    def _make_hook(self, bi, pi):
        def hook(param):
            b = self.buckets[bi]
            if b["fired"][pi]:
                return
            b["fired"][pi] = True
            if self.direct_grads:       # adopt gradients that were not produced in place
                view = param._dvla_grad_view
                g = param.grad
                if g is not None and g.data_ptr() != view.data_ptr():
                    view.copy_(g)
                    param.grad = view
                    self.copied += 1
            if not b["expected"][pi]:
                if b["launched"]:
                    raise RuntimeError("GradBucketReducer: a parameter that received no gradient in earlier steps received one "
                                       "after its bucket's all-reduce was launched; use static_unused=False")
                b["hold"] = True          # the used set grew: flush at finish() this step, re-learn there
                return
            b["pending"] -= 1
            if b["pending"] == 0 and not b["launched"] and not b["hold"]:
                self._launch(b)
        return hook

    def _launch(self, b):
        b["launched"] = True
        if self.world > 1:
            self._handles.append(dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def zero_grad(self):
        """zero the flat buffers (grads stay views) and re-arm the hooks for the next backward."""
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = sum(b["expected"])
            b["launched"] = False
            b["hold"] = False
            b["fired"] = [False] * len(b["params"])
            if self.direct_grads:
                for p in b["params"]:
                    p.grad = None
                    p._dvla_grad_free = True       # dreamvla_amd.ops may write this step's gradient into the slot once
        self._handles = []

    def finish(self):
        """call after backward(): flush buckets that never completed (unused parameters), wait, average."""
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
            if self.static_unused and any(b["fired"]):
                if b["hold"]:                       # grew: wait for everything seen so far
                    b["expected"] = [e or f for e, f in zip(b["expected"], b["fired"])]
                elif all(b["expected"]):            # first learning step: wait only for what fired
                    b["expected"] = list(b["fired"])
        for h in self._handles:
            h.wait()
        self._handles = []
        if self.world > 1:
            for b in self.buckets:
                b["flat"].div_(self.world)

    def grads_are_views(self):
        """every gradient lives in its bucket (direct_grads: every gradient that exists)"""
        ok = True
        for b in self.buckets:
            base = b["flat"].untyped_storage().data_ptr()
            for p in b["params"]:
                if self.direct_grads and p.grad is None:
                    continue
                ok &= p.grad is not None and p.grad.untyped_storage().data_ptr() == base
        return ok

    def grad_of(self, p):
        """this step's (reduced) gradient of a parameter as a view of its bucket, also when p.grad is None"""
        return p._dvla_grad_view

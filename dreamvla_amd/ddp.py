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
  * collectives are issued in BUCKET INDEX ORDER on every rank, whatever the order in which buckets become ready: a bucket
    is launched only when every bucket before it has been launched (a held / unfinished bucket therefore also defers the ones
    behind it to `finish()`), so ranks whose used-parameter sets differ for a step still execute the same sequence;
  * the average is taken by the collective itself (ReduceOp.AVG on RCCL; SUM + one divide on gloo, which has no AVG):
    no separate pass over the gradient buffers;
  * gradient accumulation over several backward passes (the reference's gradient_accumulation_steps,
    utils/train_utils.py:588-607): run the first passes under `with reducer.no_sync():` -- nothing is launched, gradients
    add up in the buckets -- and the last one outside it.  A second backward WITHOUT no_sync() after a bucket has been
    launched raises: its gradients would be added to a buffer that is already being reduced;
  * `direct_grads=True`: `zero_grad()` leaves `p.grad = None` and publishes each parameter's bucket slot as
    `p._dvla_grad_view`; the backward of dreamvla_amd.ops (weight-gradient GEMMs, bias column sums, LayerNorm parameter
    gradients) writes its result straight into that slot and returns it, autograd's AccumulateGrad then adopts the tensor
    (no `grad += new` kernel per parameter -- ~400 launches per step on DreamVLA), and the hook below copies only the
    gradients that some other operator produced elsewhere (learned tokens, position embeddings).
"""
import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params, bucket_bytes=256 << 20, process_group=None, static_unused=True, direct_grads=False,
                 last_bucket_bytes=64 << 20, robust_gemm_schedule=None):
        """last_bucket_bytes: size cap of the bucket that is filled LAST (the gradients of the model's first parameters: token
        embeddings, resampler, trunk layer 0 ...): its all-reduce cannot start before backward ends, so it is the exposed tail
        of the exchange -- 64 MiB is ~0.5 ms over xGMI where a 256-MiB bucket would be ~2 ms.
        robust_gemm_schedule: while a collective is outstanding, run the persistent GEMMs under the schedule that does not
        assume every workgroup is co-resident (dvla_set_gemm_schedule(8, 0): equal-sized short-lived workgroups, no stream-K):
        an RCCL kernel holds CUs for the whole transfer and the one-workgroup-per-CU schedules then wait for the workgroups
        that cannot start (+65 % per GEMM launch with 16 of 256 CUs taken; +0 ... 9 % under the robust schedule:
        profiles/r02_gemm_cu_contention.txt).  Default: on for the "nccl" (= RCCL) backend, off otherwise."""
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.static_unused = bool(static_unused)
        self.direct_grads = bool(direct_grads)
        self._no_sync = False
        self._next_launch = 0   # buckets [0, _next_launch) have been launched this step (index order on every rank)
        self.copied = 0     # direct_grads: gradients the hook had to copy into their slot (not produced in place)
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.buckets = []       # dicts: flat, params, pending
        order = list(reversed(self.params))
        sizes = [p.numel() * p.element_size() for p in order]
        remaining = sum(sizes)
        tail_cut = False       # the cut that opens the (small) last bucket has been made
        cur, cur_bytes, cur_key = [], 0, None
        for p, nbytes in zip(order, sizes):
            key = (p.dtype, p.device)
            cut = bool(cur) and (key != cur_key or cur_bytes + nbytes > bucket_bytes)
            if (not cut and cur and not tail_cut and last_bucket_bytes and remaining <= last_bucket_bytes
                    and cur_bytes + remaining > last_bucket_bytes):
                cut = tail_cut = True
            if cut:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_key = key
            remaining -= nbytes
        if cur:
            self._make_bucket(cur)
        self._handles = []
        self._hooks = []
        if robust_gemm_schedule is None:
            robust_gemm_schedule = self.world > 1 and dist.get_backend(process_group) == "nccl"
        self.robust_gemm_schedule = bool(robust_gemm_schedule)
        self._saved_schedule = None
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
                             "fired": [False] * len(plist),      # this step (any backward pass of it): what the reducer learns from
                             "arrived": [False] * len(plist),    # this backward PASS (re-armed when no_sync() exits)
                             "expected": [True] * len(plist),    # parameters the launch waits for (learned)
                             "hold": False})

    def _make_hook(self, bi, pi):
        def hook(param):
            b = self.buckets[bi]
            if b["launched"]:
                raise RuntimeError("GradBucketReducer: a gradient arrived for a parameter whose bucket is already being "
                                   "all-reduced (a second backward before zero_grad(), or a parameter that joined the graph "
                                   "late): run all but the last backward of an accumulation cycle under reducer.no_sync(), or "
                                   "construct the reducer with static_unused=False")
            if self.direct_grads:       # adopt gradients that were not produced in place
                view = param._dvla_grad_view
                g = param.grad
                if g is not None and g.data_ptr() != view.data_ptr():
                    # a post-accumulate hook sees the RUNNING TOTAL of this step in param.grad (first pass: the incoming
                    # gradient; later passes: autograd has already added to it, in place -- then it still is the slot and we
                    # are not here -- or out of place (create_graph=True), and then `g` is the sum): copy, never add
                    view.copy_(g)
                    param.grad = view
                    self.copied += 1
            b["fired"][pi] = True
            if b["arrived"][pi]:
                return                    # already counted in this pass
            b["arrived"][pi] = True
            if not b["expected"][pi]:
                b["hold"] = True          # the used set grew: flush at finish() this step, re-learn there
                return
            b["pending"] -= 1
            self._launch_ready()
        return hook

    def _launch_ready(self):
        """launch, in index order, every leading bucket whose expected gradients have all arrived"""
        if self._no_sync:
            return
        while self._next_launch < len(self.buckets):
            b = self.buckets[self._next_launch]
            if b["pending"] != 0 or b["hold"]:
                return
            self._launch(b)

    def _gemm_schedule(self, robust):
        """switch the GEMM schedule for the time collectives are outstanding (see __init__) and back"""
        if not self.robust_gemm_schedule:
            return
        import ctypes
        from . import _lib
        lib = _lib.load()
        from .ops import GemmTuner
        if robust and self._saved_schedule is None:
            k, sk = ctypes.c_int(0), ctypes.c_int(0)
            lib.dvla_get_gemm_schedule(ctypes.byref(k), ctypes.byref(sk))
            self._saved_schedule = (k.value, sk.value)
            lib.dvla_set_gemm_schedule(8, 0)
            self.robust_switches = getattr(self, "robust_switches", 0) + 1     # (reported by bench.py's N > 1 line)
            GemmTuner.schedule_tag = 1      # problem keys under the robust schedule are tuned (and locked) on their own
        elif not robust and self._saved_schedule is not None:
            lib.dvla_set_gemm_schedule(*self._saved_schedule)
            self._saved_schedule = None
            GemmTuner.schedule_tag = 0

    def _launch(self, b):
        b["launched"] = True
        self._next_launch += 1
        if self.world > 1:
            self._gemm_schedule(True)
            op = dist.ReduceOp.AVG if self._avg_in_collective() else dist.ReduceOp.SUM
            flat = b["flat"]
            if flat.is_cuda and dist.get_backend(self.group) == "gloo":
                # test configuration (several ranks sharing one GPU; RCCL refuses duplicate devices): gloo reduces on the
                # host -- stage the bucket through host memory explicitly (ordered behind the backward kernels that wrote it)
                host = flat.cpu()
                self._handles.append((dist.all_reduce(host, op=op, group=self.group, async_op=True), host, flat))
            else:
                self._handles.append((dist.all_reduce(flat, op=op, group=self.group, async_op=True), None, None))

    def _avg_in_collective(self):
        return dist.get_backend(self.group) == "nccl"

    def no_sync(self):
        """context manager: backward passes inside it only accumulate into the buckets (no collective)"""
        red = self

        class _NoSync:
            def __enter__(self_inner):
                red._no_sync = True

            def __exit__(self_inner, *exc):
                red._no_sync = False
                red._rearm_pass()
                return False
        return _NoSync()

    def _rearm_pass(self):
        """the accumulation passes are over: the NEXT backward is the one whose arrivals launch the buckets, so count its
        arrivals from zero (round-2 ADVICE: the counters used to stay at zero after the no_sync passes, every hook of the last
        pass returned early and all collectives were issued back to back in finish() -- correct, but never overlapped)"""
        for b in self.buckets:
            if not b["launched"]:
                b["arrived"] = [False] * len(b["params"])
                b["pending"] = sum(b["expected"])

    def zero_grad(self):
        """zero the flat buffers (grads stay views) and re-arm the hooks for the next backward."""
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = sum(b["expected"])
            b["launched"] = False
            b["hold"] = False
            b["fired"] = [False] * len(b["params"])
            b["arrived"] = [False] * len(b["params"])
            if self.direct_grads:
                for p in b["params"]:
                    p.grad = None
                    p._dvla_grad_free = True       # dreamvla_amd.ops may write this step's gradient into the slot once
        self._handles = []
        self._next_launch = 0
        # a step that raised in backward / skipped finish() must not leave the process on the robust schedule (round-3 ADVICE)
        self._gemm_schedule(False)

    def finish(self):
        """call after the (last) backward(): flush, in index order, the buckets that were not launched during backward (unused
        parameters, held buckets and everything behind them), wait; the result is the gradient averaged over the ranks."""
        if self._no_sync:
            raise RuntimeError("GradBucketReducer.finish() inside no_sync()")
        self.last_early_launches = self._next_launch      # buckets whose collective was launched DURING backward (bench.py reports it)
        for b in self.buckets[self._next_launch:]:
            self._launch(b)
        for b in self.buckets:
            if self.static_unused and any(b["fired"]):
                if b["hold"]:                       # grew: wait for everything seen so far
                    b["expected"] = [e or f for e, f in zip(b["expected"], b["fired"])]
                elif all(b["expected"]):            # first learning step: wait only for what fired
                    b["expected"] = list(b["fired"])
        try:
            for h, host, flat in self._handles:
                h.wait()
                if host is not None:
                    flat.copy_(host)
        finally:
            self._handles = []
            self._gemm_schedule(False)
        if self.world > 1 and not self._avg_in_collective():
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

"""Host-side operators: torch.autograd.Function wrappers around the C-ABI HIP kernels.

Every op here runs ONLY on the HIP extension (bf16 CUDA tensors); there is no eager fallback -- a CPU
tensor, a wrong dtype or a missing library raises.  torch is used for device memory (torch.empty),
streams and autograd bookkeeping.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import ACT, AttnParams, DT_BF16, DT_F32, GemmParams, check

BF16 = torch.bfloat16


# ---------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------
def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, name, dtype=BF16):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise _lib.DvlaError(f"{name}: tensor is on {t.device}; the DreamVLA HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype} (the kernels compute in bf16; fp32 parameters / activations "
                        f"are accepted by the module-level operators -- linear, mlp, layer_norm, attention -- which keep "
                        f"fp32 masters and run on bf16 shadows)")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _dt(t):
    return DT_F32 if t.dtype == torch.float32 else DT_BF16


def _param_dt(t, name):
    if t.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"{name}: expected bf16 or fp32 parameter, got {t.dtype}")
    if not t.is_cuda:
        raise _lib.DvlaError(f"{name}: parameter is on {t.device}; no CPU fallback")
    return DT_F32 if t.dtype == torch.float32 else DT_BF16


def _rows2d(x, cols):
    """View x as (rows, cols) with unit inner stride (copies only if the layout forces it)."""
    x2 = x.reshape(-1, cols)
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < cols):
        x2 = x2.contiguous()
    return x2


class _Seeds:
    """Counter-based dropout seeds: (lo, hi) = (per-call counter, hash of torch.initial_seed() and rank)."""
    counter = 0
    salt = 0

    @classmethod
    def next(cls):
        cls.counter = (cls.counter + 1) & 0xFFFFFFFF
        base = (torch.initial_seed() ^ (cls.salt * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        hi = ((base >> 32) ^ (base & 0xFFFFFFFF) * 0x85EBCA6B) & 0xFFFFFFFF
        return cls.counter, hi


def set_seed_salt(salt):
    _Seeds.salt = int(salt)


def next_seed():
    return _Seeds.next()


# ---------------------------------------------------------------------------------------------------
# raw kernel launchers (no autograd)
# ---------------------------------------------------------------------------------------------------
FWD_SPLIT_K_ALLOWED = os.environ.get("DVLA_FWD_SPLIT_K", "1") != "0"
FWD_SPLIT_K = False          # switched on by the rollout engine around its encode / decode (forward_split_k below)


class forward_split_k:
    """`with ops.forward_split_k():` -- forward GEMMs with few hundred rows and a long K are cut along K (fwd_split_k).  The
    rollout engine wraps its encode / decode in it.  It is NOT on by default: it changes the fp32 summation order of the GEMMs
    it touches (the frozen ViT's MLP down-projection among them), and the training-mode parity checks hold the step to
    tolerances calibrated on row-independent kernels (a two-rank run whose half batches took another split than the full
    batch's moved an ill-conditioned bias gradient by 100 %)."""

    def __init__(self, on=True):
        self.on = bool(on) and FWD_SPLIT_K_ALLOWED

    def __enter__(self):
        global FWD_SPLIT_K
        self.was, FWD_SPLIT_K = FWD_SPLIT_K, self.on
        return self

    def __exit__(self, *exc):
        global FWD_SPLIT_K
        FWD_SPLIT_K = self.was
        return False


def _fwd_split_epilogue_ok(out, bias, residual, N):
    """what the reduce-with-epilogue pass of a split forward GEMM needs (csrc/gemm.hip dvla_gemm_bf16, `split_epi`: 16-byte
    vectors of C / bias / residual, leading dimensions in whole vectors).  A call that does not meet it -- an `out=` view at an odd
    column offset, a sliced residual -- simply runs unsplit: the OPPORTUNISTIC split must never turn a GEMM that works into
    DVLA_ERR_UNSUPPORTED (round-4 ADVICE); an EXPLICIT split_k > 1 still reports the constraint."""
    if N % 8 != 0 or out.dtype != BF16 or out.data_ptr() % 16 != 0 or out.stride(0) % 8 != 0:
        return False
    if bias is not None and bias.data_ptr() % 16 != 0:
        return False
    if residual is not None and (residual.data_ptr() % 16 != 0 or residual.stride(0) % 8 != 0 or residual.stride(-1) != 1):
        return False
    return True


def fwd_split_k(M, N, K, cus=256):
    """Split-K for a FORWARD GEMM (bias / activation / residual applied by the reduction pass, include/dvla.h): the evaluation
    engine's trunk at one episode has 930 rows, and its MLP down-projection (N = 1024, K = 4096) is 64 tiles of 128 x 128
    that each walk all of K -- 43.6 us on a quarter of the chip (profiles/r04_midrows_perf.jsonl).  Only where the few-rows
    kernel does not apply (M > 512), the output leaves most CUs idle (<= 96 tiles) and K is long enough to pay for the
    reduction launch: then K is cut so that tiles x splits ~ the CU count, at least 1024 deep per split."""
    if M <= 512 or K < 2048:
        return 1
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles > 96:
        return 1
    return max(1, min(K // 1024, cus // tiles, 8))


def auto_split_k(M, N, K, cus=256):
    """weight-gradient GEMMs: small output (M x N), very long contraction (K = tokens).  The kernels that run them tile the
    output in 256 x 256 (persistent, one workgroup per CU), so the number of work items tiles * split_k should land just
    under a multiple of the CU count: 1024 x 3072 is 48 tiles -> 5 slices = 240 items = one 94 %-full round (round 1 took
    6 = 288 items = two rounds, the second 12 % full: 632 TFLOP/s where its 4-slice siblings reach 960-1000).  Cheapest
    by (1 + 1 % per slice) / round efficiency, slices of at least ~1.3 k of K, at most 16."""
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles >= (3 * cus) // 4 or K < 4096:
        return 1
    best_s, best_cost = 1, float("inf")
    for sk in range(1, 17):
        if K // sk < 1280:
            break
        items = tiles * sk
        eff = items / float(cus * -(-items // cus))
        cost = (1.0 + 0.01 * sk) / eff          # every slice adds M x N x 4 B of partial sums to write and reduce
        if cost < best_cost:
            best_s, best_cost = sk, cost
    return best_s


class GemmProfiler:
    """Brackets every GEMM launch with HIP events on the launch stream (torch's current stream) and accumulates the
    algorithmic FLOPs (2*M*N*K) -- bench.py's roofline leg.  Not active unless used as a context manager."""
    active = None

    def __init__(self):
        self.records = []
        self.shapes = []

    def breakdown(self):
        torch.cuda.synchronize()
        agg = {}
        for (e0, e1, f), sh in zip(self.records, self.shapes):
            a = agg.setdefault(sh, [0, 0.0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += f
        rows = [{"M": k[0], "N": k[1], "K": k[2], "a_trans": k[3], "b_trans": k[4], "split_k": k[5], "kernel": k[6],
                 "variant": k[7], "epilogue": k[8], "launches": v[0], "ms": v[1], "tflops": v[2] / (v[1] * 1e-3) / 1e12 if v[1] > 0 else 0.0} for k, v in agg.items()]
        return sorted(rows, key=lambda r: -r["ms"])

    def __enter__(self):
        GemmProfiler.active = self
        return self

    def __exit__(self, *exc):
        GemmProfiler.active = None
        return False

    def summary(self, kernel=None):
        """all launches (every one is a hand-written kernel; `kernel` is kept for the breakdown's column)"""
        torch.cuda.synchronize()
        recs = [r for r, sh in zip(self.records, self.shapes) if kernel is None or sh[6] == kernel]
        total_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in recs)
        flops = sum(f for _, _, f in recs)
        n = max(len(recs), 1)
        return {"launches": len(recs), "total_ms": total_ms, "avg_us": total_ms * 1e3 / n,
                "gflop_per_launch": flops / n / 1e9, "tflops": (flops / (total_ms * 1e-3) / 1e12) if total_ms > 0 else 0.0}


class GemmTuner:
    """Online choice of the GEMM kernel configuration per problem key (like a convolution autotuner).

    include/dvla.h exposes the configurations of the hand-written kernels through `dvla_set_gemm_variant` (0 = the
    library's cost model, 2 = register-staged 128x128, 4 / 6 / 7 = LDS-DMA ring 256x256 / 128x128 / 256x128 BK64, 8 = phase
    kernel 256x256 BK64; they differ
    only in fp32 summation order).  Which one is fastest depends on more than (M, N, K): the epilogue, what the
    neighbouring kernels left in L2 / Infinity Cache, the clocks.  So the first calls of every key run the candidates IN
    TURN -- each real call is executed exactly once, with one candidate, bracketed by two events that are read back later
    without a host sync -- and once every candidate has `ROUNDS` finished timing(s) the key is locked to the best one
    (median).  A forced variant that does not apply to a shape falls back to the register-staged kernel inside the
    library, so every trial is valid.  Disable with DVLA_GEMM_AUTOTUNE=0 (the cost model is then used for every call).
    Every candidate is a hand-written HIP kernel of libdvla_hip.so: no vendor GEMM library is linked or offered (the
    hipBLASLt yardstick lives in libdvla_cmp.so and is driven only by tests/library_yardstick.py)."""
    CANDIDATES = tuple(int(v) for v in os.environ.get("DVLA_GEMM_CANDIDATES", "0,4,6,7,8,9,10,2").split(","))
    ROUNDS = int(os.environ.get("DVLA_GEMM_TUNE_ROUNDS", "3"))
    enabled = os.environ.get("DVLA_GEMM_AUTOTUNE", "1") != "0" and os.environ.get("DVLA_GEMM_VARIANT") is None
    KEY_LEN = 17    # fields of a problem key (ops.gemm builds it)
    table = {}      # key -> locked variant
    trials = {}     # key -> {"pending": [(variant, e0, e1)], "times": {variant: [ms]}, "next": int}
    frozen = False  # True (hipGraph capture / replay-critical sections): no trials, no events -- locked choice or the cost model
    schedule_tag = 0  # part of every problem key: 1 while GradBucketReducer runs the GEMMs under its robust schedule (collectives
                      # outstanding).  A candidate timed under one schedule must not be locked for the other (round-3 ADVICE).

    # Problem keys that also ask for a k-sum (the bias gradient next to a weight gradient) have two more candidates (round 5): 108 /
    # 109 = the phase kernel (plain / stream-K schedule) WITHOUT its summing code + the column-sum kernel over the operand, against
    # 8 / 9 / 10 = the phase kernel with the dots in its load segments and 4 / 6 / 7 = the ring kernels' fused sums.  Measured: the
    # fused dots cost the summing workgroups 7-13 % and every launch waits for its slowest workgroup -- they win for the trunk
    # (K = 20 832: +7 % over the ring kernel), the separate pass wins for the decoders' K = 91 840 (a 45-us pass next to a 585-us GEMM).
    KSUM_EXTRA = (108, 109)

    @classmethod
    def pick(cls, key, ksum=False):
        v = cls.table.get(key)
        if v is not None:
            return v, None
        if cls.frozen:
            return 0, None
        st = cls.trials.get(key)
        if st is None:
            cands = cls.CANDIDATES + (cls.KSUM_EXTRA if ksum else ())
            st = cls.trials[key] = {"pending": [], "times": {c: [] for c in cands}, "next": 0}
        still = []
        for (var, e0, e1) in st["pending"]:
            if e1.query():
                if var in st["times"]:
                    st["times"][var].append(e0.elapsed_time(e1))
            else:
                still.append((var, e0, e1))
        st["pending"] = still
        cands = tuple(st["times"])
        if all(len(t) >= cls.ROUNDS for t in st["times"].values()):
            best = min(cands, key=lambda c: sorted(st["times"][c])[len(st["times"][c]) // 2])
            cls.table[key] = best
            del cls.trials[key]
            return best, None
        var = cands[st["next"] % len(cands)]
        st["next"] += 1
        return var, st

    @classmethod
    def reset(cls):
        cls.table.clear(); cls.trials.clear()

    @classmethod
    def save_plan(cls, path):
        """the locked choices as JSON (problem key -> configuration): a later process replays exactly this kernel mix with
        `load_plan` -- e.g. the rocprofv3 counter passes of the tuned step, which must not contain tuner trials"""
        import json
        with open(path, "w") as f:
            json.dump([[list(k), int(v)] for k, v in cls.table.items()], f)

    @classmethod
    def load_plan(cls, path, freeze=True):
        import json
        with open(path) as f:
            for k, v in json.load(f):
                k = tuple(bool(x) if isinstance(x, bool) else int(x) for x in k)
                if len(k) == cls.KEY_LEN - 1:      # a plan written before the schedule tag joined the key: default schedule
                    k = k + (0,)
                cls.table[k] = int(v)
        cls.frozen = bool(freeze)     # keys the plan does not know take the cost model (no trials)

    @classmethod
    def summary(cls):
        """how many problem keys each configuration won (bench.py reports it next to the roofline)"""
        out = {}
        for v in cls.table.values():
            out[f"hip:{v}"] = out.get(f"hip:{v}", 0) + 1
        return out


def gemm(a, b, *, a_trans=False, b_trans=False, bias=None, act=0, want_preact=False, dact_aux=None, dact=0,
         dropout_p=0.0, seed=(0, 0), residual=None, res_rows=0, out_dtype=BF16, out=None, accumulate=False, split_k=1,
         variant=None, ksum=None, a_ln_eps=None):
    """C[M,N] = epilogue(A . B^T); see include/dvla.h.  `variant` forces a kernel configuration (tests / sweeps)
    instead of asking the tuner.  a: (M,K) or (K,M) if a_trans; b: (N,K) or (K,N) if b_trans.
    a_ln_eps: the rows of A are layer-normalised (no affine, this epsilon) on their way into the product -- Linear(LayerNorm(x))
    from x in one launch; few-rows kernel only (M <= 512, k-contiguous operands, 512 <= K <= 1536), inference.
    ksum = ("a" | "b", out): also out[i] = sum over k of that operand's row i (the bias gradient of a weight-gradient GEMM,
    summed from the fragments the kernel multiplies anyway); out: 1-D, M (a) or N (b) long, bf16 or fp32.
    Returns C (and the pre-activation tensor if want_preact)."""
    lib = _lib.load()
    _req(a, "gemm.a"); _req(b, "gemm.b")
    if a.dim() != 2 or b.dim() != 2:
        raise ValueError("gemm operands must be 2-D")
    if a.stride(1) != 1:
        a = a.contiguous()
    if b.stride(1) != 1:
        b = b.contiguous()
    if a_trans:
        K, M = a.shape
    else:
        M, K = a.shape
    if b_trans:
        Kb, N = b.shape
    else:
        N, Kb = b.shape
    if K != Kb:
        raise ValueError(f"gemm: contraction mismatch {K} vs {Kb}")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    else:
        if out.shape != (M, N) or out.stride(1) != 1:
            raise ValueError("gemm: bad `out`")
    p = GemmParams()
    p.A, p.lda, p.a_trans = a.data_ptr(), a.stride(0), int(a_trans)
    p.B, p.ldb, p.b_trans = b.data_ptr(), b.stride(0), int(b_trans)
    p.C, p.ldc, p.c_dtype = out.data_ptr(), out.stride(0), _dt(out)
    p.M, p.N, p.K = M, N, K
    if bias is not None:
        p.bias, p.bias_dtype = bias.data_ptr(), _param_dt(bias, "gemm.bias")
    p.act = int(act)
    preact = None
    if want_preact:
        preact = torch.empty((M, N), dtype=BF16, device=a.device)
        p.preact, p.ld_preact = preact.data_ptr(), preact.stride(0)
    if dact_aux is not None:
        _req(dact_aux, "gemm.dact_aux")
        if dact_aux.stride(1) != 1:
            dact_aux = dact_aux.contiguous()
        p.dact_aux, p.ld_dact, p.dact = dact_aux.data_ptr(), dact_aux.stride(0), int(dact)
    p.dropout_p = float(dropout_p)
    p.seed_lo, p.seed_hi = int(seed[0]) & 0xFFFFFFFF, int(seed[1]) & 0xFFFFFFFF
    if residual is not None:
        _req(residual, "gemm.residual")
        if residual.stride(1) != 1:
            residual = residual.contiguous()
        p.residual, p.ld_res, p.res_rows = residual.data_ptr(), residual.stride(0), int(res_rows)
    p.accumulate = int(accumulate)
    # split_k: None / 1 = no split requested (the forward rule below may choose one inside `with forward_split_k()`);
    # 0 = no split, and the forward rule is OFF for this call (explicit opt-out); n > 1 = exactly n slices (round-4 ADVICE: the
    # normalisation used to call int(None) before its own None test).
    sk_req = 1 if split_k is None else int(split_k)
    if sk_req < 0:
        raise ValueError("gemm: split_k >= 0")
    p.split_k = max(1, sk_req)
    if (FWD_SPLIT_K and sk_req == 1 and not torch.is_grad_enabled() and not a_trans and not want_preact
            and dact_aux is None and dropout_p == 0.0 and not accumulate and ksum is None and a_ln_eps is None and variant is None
            and _fwd_split_epilogue_ok(out, bias, residual, N)):
        # few hundred rows, long K: the evaluation engine's trunk (see fwd_split_k; on inside `with forward_split_k()` only)
        p.split_k = fwd_split_k(M, N, K)
    ksum_ws = None
    if ksum is not None:
        which, kout = ksum
        klen = M if which == "a" else N
        if which not in ("a", "b") or kout.shape != (klen,) or not kout.is_contiguous() or kout.device != a.device:
            raise ValueError("gemm: bad `ksum`")
        # decided ONCE, independent of what the tuner would pick for this call (round-3 ADVICE: the layout used to be checked only
        # under the configurations that leave the sum to the column-sum kernel, so a row-major operand failed on some tuning steps
        # and passed on others).  The column-sum fallback needs the summed operand stored k-major; an operand in any other
        # layout can only be summed by the ring kernels' fused code: such a call names one of them itself (variant 4 / 6 / 7)
        # and is never handed to the tuner.
        if not (a_trans if which == "a" else b_trans) and (variant is None or int(variant) not in (4, 6, 7)):
            raise _lib.DvlaError("gemm: k-sums of an operand that is not stored k-major (a_trans / b_trans) need a ring "
                                 "configuration forced (variant 4, 6 or 7)")
    prof = GemmProfiler.active
    # "plain": nothing but (optionally) the bias vector rides on the GEMM (reported per shape by the profiler)
    plain = (act == 0 and not want_preact and dact_aux is None and residual is None and dropout_p == 0.0
             and not (bias is not None and accumulate))
    trial, key = None, None
    if a_ln_eps is not None:
        p.a_layernorm, p.a_ln_eps = 1, float(a_ln_eps)
        variant = 11                    # only the few-rows kernel normalises: never handed to the tuner
    forced = variant is not None
    variant = int(variant) if forced else 0
    # a problem the few-rows kernel takes (csrc/gemm_skinny.h) is not handed to the tuner: every tiled candidate falls back to the
    # register-staged kernel there, the trials of a 5-20 us launch are timing noise (eager trials locked the slow fallback for 120
    # of a control step's 540 small GEMMs), and configuration 0 picks the few-rows kernel by itself
    few_rows = (not a_trans and not b_trans and M <= 512 and K >= 16 and K % 16 == 0 and int(p.split_k) <= 1 and a.stride(0) % 8 == 0
                and b.stride(0) % 8 == 0)
    if GemmTuner.enabled and not forced and not few_rows:
        key = (M, N, K, int(a_trans), int(b_trans), int(p.split_k), int(act), int(dact),
               bias is not None, want_preact, dact_aux is not None, residual is not None,
               dropout_p > 0.0, out.dtype == torch.float32, bool(accumulate), 0 if ksum is None else (1 if ksum[0] == "a" else 2),
               int(GemmTuner.schedule_tag))
        variant, trial = GemmTuner.pick(key, ksum=ksum is not None)
    # k-sums: carried by the ring kernels (variants 4 / 6 / 7), by the phase kernel (8 / 9 / 10, round 5) and by whatever the
    # library's own choice is (0: it falls back by itself); under a configuration without the summing code -- 2, and the tuner's
    # pseudo-configurations 108 / 109 = phase kernel without the dots -- the column-sum kernel runs here, AFTER the profiler's end
    # event (it is not GEMM time) but inside the tuner's trial window (a configuration's cost includes what it leaves to others)
    lib_variant = variant % 100
    ksum_here = ksum is not None and (variant >= 100 or lib_variant not in (0, 4, 6, 7, 8, 9, 10))
    if ksum is not None and not ksum_here:
        p.ksum, p.ksum_dtype, p.ksum_operand = kout.data_ptr(), _dt(kout), 1 if which == "a" else 2
        ksum_ws = torch.empty(lib.dvla_gemm_ksum_partial_rows(p.split_k) * klen, dtype=torch.float32, device=a.device)
        p.ksum_workspace = ksum_ws.data_ptr()
    if prof is not None or trial is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = None
    if p.split_k > 1:
        ws = torch.empty((p.split_k, M, N), dtype=torch.float32, device=a.device)
        p.workspace = ws.data_ptr()
    if lib_variant:
        lib.dvla_set_gemm_variant(lib_variant)
    try:
        check(lib.dvla_gemm_bf16(C.byref(p), _stream()), "dvla_gemm_bf16")
    finally:
        if lib_variant:
            lib.dvla_set_gemm_variant(0)
    if prof is not None or trial is not None:
        e1.record()
    e2 = e1 if (prof is not None or trial is not None) else None
    if ksum_here:
        src = a if which == "a" else b
        colsum(src, kout.dtype, out=kout)
        if trial is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record()
    if trial is not None:
        trial["pending"].append((variant, e0, e2))
    if prof is not None:
        prof.records.append((e0, e1, 2.0 * M * N * K))
        prof.shapes.append((M, N, K, int(a_trans), int(b_trans), int(p.split_k), "hip",
                            int(variant), ("bias" if bias is not None else "plain") if plain else "fused"))
    return (out, preact) if want_preact else out


def layernorm_fwd(x2, gamma, beta, eps, want_stats):
    lib = _lib.load()
    rows, cols = x2.shape
    y = torch.empty_like(x2)
    mean = rstd = None
    if want_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
    pdt = _param_dt(gamma, "layernorm.weight") if gamma is not None else DT_BF16
    if gamma is not None and beta is not None and beta.dtype != gamma.dtype:
        raise TypeError("layernorm weight/bias dtype mismatch")
    check(lib.dvla_layernorm_fwd(x2.data_ptr(), _ptr(gamma), _ptr(beta), pdt, y.data_ptr(), _ptr(mean), _ptr(rstd),
                                 rows, cols, float(eps), _stream()), "dvla_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy2, x2, gamma, mean, rstd, need_param_grads, dres2=None, grad_dtype=torch.float32, dg_out=None,
                  db_out=None, need_dx=True):
    """dres2: gradient of the residual stream that bypassed the LayerNorm; added to dx inside the kernel.
    dgamma / dbeta come back in grad_dtype (bf16 or fp32) straight from the reduction kernel.
    need_dx=False (with need_param_grads): the input needs no gradient; dx is neither computed nor stored (returns None)."""
    lib = _lib.load()
    rows, cols = x2.shape
    if not need_dx and (dres2 is not None or not need_param_grads):
        raise ValueError("layernorm_bwd(need_dx=False) is for parameter gradients only")
    dx = torch.empty_like(x2) if need_dx else None
    dg = db = part = None
    if need_param_grads:
        dg = dg_out if dg_out is not None else torch.empty(cols, dtype=grad_dtype, device=x2.device)
        db = db_out if db_out is not None else torch.empty(cols, dtype=grad_dtype, device=x2.device)
        part = torch.empty(2 * lib.dvla_layernorm_bwd_partial_rows() * cols, dtype=torch.float32, device=x2.device)
    pdt = _param_dt(gamma, "layernorm.weight") if gamma is not None else DT_BF16
    gdt = DT_F32 if grad_dtype == torch.float32 else DT_BF16
    check(lib.dvla_layernorm_bwd_add(dy2.data_ptr(), x2.data_ptr(), _ptr(gamma), pdt, mean.data_ptr(), rstd.data_ptr(),
                                     _ptr(dres2), _ptr(dx), _ptr(dg), _ptr(db), gdt, _ptr(part), rows, cols, _stream()),
          "dvla_layernorm_bwd_add")
    return dx, dg, db


def colsum(x2, out_dtype=torch.float32, out=None):
    """column sums (bias gradients), written in out_dtype (fp32 or bf16) by the reduction kernel itself"""
    lib = _lib.load()
    rows, cols = x2.shape
    if out_dtype not in (torch.float32, BF16):
        raise TypeError(f"colsum: fp32 or bf16 output, got {out_dtype}")
    if out is None:
        out = torch.empty(cols, dtype=out_dtype, device=x2.device)
    elif out.dtype != out_dtype or out.numel() != cols or not out.is_contiguous():
        raise ValueError("colsum: bad `out`")
    part = torch.empty(lib.dvla_colsum_partial_rows() * cols, dtype=torch.float32, device=x2.device)
    check(lib.dvla_colsum_dt(x2.data_ptr(), x2.stride(0), rows, cols, out.data_ptr(),
                             DT_F32 if out_dtype == torch.float32 else DT_BF16, part.data_ptr(), _stream()), "dvla_colsum_dt")
    return out


def cast_to(x, dtype):
    """fp32 <-> bf16 cast on the HIP path."""
    lib = _lib.load()
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    if x.dtype == torch.float32 and dtype == BF16:
        check(lib.dvla_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "dvla_cast")
    elif x.dtype == BF16 and dtype == torch.float32:
        check(lib.dvla_cast_bf16_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "dvla_cast")
    else:
        raise TypeError(f"cast {x.dtype} -> {dtype} unsupported")
    return out


def act_bwd_raw(dy2, preact2, act, dropout_p=0.0, seed=(0, 0)):
    lib = _lib.load()
    if dy2.stride(1) != 1 or dy2.stride(0) != dy2.shape[1]:
        dy2 = dy2.contiguous()
    if preact2 is not None and not preact2.is_contiguous():
        preact2 = preact2.contiguous()
    dz = torch.empty(dy2.shape, dtype=BF16, device=dy2.device)
    check(lib.dvla_act_bwd(dy2.data_ptr(), _ptr(preact2), dz.data_ptr(), dy2.shape[0], dy2.shape[1], int(act),
                           float(dropout_p), int(seed[0]), int(seed[1]), _stream()), "dvla_act_bwd")
    return dz


def act_bwd_colsum(dy2, preact2, act, dropout_p, seed, out_dtype, bias_param=None):
    """(dz, column sums of dz) in one pass (dvla_act_bwd_colsum) -- None when the shape does not allow 16-byte octets (the caller
    then takes act_bwd_raw and sums elsewhere).  bias_param: the parameter whose gradient the sums are (its reducer slot, if it
    has one, is written in place -- claimed only once the shape is known to be taken)"""
    lib = _lib.load()
    if dy2.stride(1) != 1 or dy2.stride(0) != dy2.shape[1]:
        dy2 = dy2.contiguous()
    if preact2 is not None and not preact2.is_contiguous():
        preact2 = preact2.contiguous()
    rows, cols = dy2.shape
    if cols % 8 or dy2.data_ptr() % 16 or (preact2 is not None and preact2.data_ptr() % 16):
        return None
    dz = torch.empty(dy2.shape, dtype=BF16, device=dy2.device)
    out = _grad_dest(bias_param, out_dtype) if bias_param is not None else None
    if out is None:
        out = torch.empty(cols, dtype=out_dtype, device=dy2.device)
    part = torch.empty(lib.dvla_colsum_partial_rows() * cols, dtype=torch.float32, device=dy2.device)
    check(lib.dvla_act_bwd_colsum(dy2.data_ptr(), _ptr(preact2), dz.data_ptr(), rows, cols, int(act), float(dropout_p), int(seed[0]),
                                  int(seed[1]), out.data_ptr(), DT_F32 if out.dtype == torch.float32 else DT_BF16, part.data_ptr(),
                                  _stream()), "dvla_act_bwd_colsum")
    return dz, out


def dropout_raw(x2, p, seed):
    lib = _lib.load()
    x2 = x2.contiguous()
    y = torch.empty_like(x2)
    check(lib.dvla_dropout(x2.data_ptr(), y.data_ptr(), x2.shape[0], x2.shape[1], float(p), int(seed[0]), int(seed[1]),
                           _stream()), "dvla_dropout")
    return y


def act_fwd_raw(x, act):
    lib = _lib.load()
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib.dvla_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), int(act), _stream()), "dvla_act_fwd")
    return y


def ddim_cfg_step(model_out, x, cfg_scale, a, b, sqrt_acp_prev, sqrt_1m_acp_prev):
    """classifier-free guidance + one eta = 0 DDIM update in one launch (dvla_ddim_cfg_step).  model_out: bf16 (2 bs, T, C) -- a
    view whose samples are contiguous (e.g. `full[:, T:, :]` of a contiguous (2 bs, 2 T, C) tensor); x: fp32 (bs, T, C)."""
    lib = _lib.load()
    _req(model_out, "ddim_cfg_step.model_out")
    bs, per = x.shape[0], x[0].numel()
    if model_out.shape[0] != 2 * bs or model_out[0].numel() != per or model_out.dtype != BF16 or x.dtype != torch.float32:
        raise ValueError("ddim_cfg_step: model_out (2 bs, ...) bf16 and x (bs, ...) fp32 of matching sample size")
    if not model_out[0].is_contiguous() or (bs > 0 and model_out.stride(0) < per) or not x.is_contiguous():
        raise ValueError("ddim_cfg_step: samples must be contiguous")
    out = torch.empty_like(x)
    check(lib.dvla_ddim_cfg_step(model_out.data_ptr(), int(model_out.stride(0)), x.data_ptr(), out.data_ptr(), bs, per,
                                 float(cfg_scale), float(a), float(b), float(sqrt_acp_prev), float(sqrt_1m_acp_prev), _stream()),
          "dvla_ddim_cfg_step")
    return out


# The whole evaluation sampler of the DiT head in one launch (csrc/dit_team.hip, include/dvla.h dvla_dit_sample)
DIT_TEAM = os.environ.get("DVLA_DIT_TEAM", "1") != "0"        # 0: always the launch-by-launch sampler (measurement / A-B)
_DIT_TEAM_CUS = {}


def dit_team_ok(hidden, heads, channels, tokens, bs, device):
    """the shapes dvla_dit_sample takes (everything else runs the launch-by-launch sampler): DiT-B (hidden 768, head_dim 64),
    at most 8 tokens per sequence and 16 token rows -- one episode -- on a whole MI355X"""
    if not DIT_TEAM or device.type != "cuda":
        return False
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _DIT_TEAM_CUS:
        _DIT_TEAM_CUS[key] = torch.cuda.get_device_properties(key).multi_processor_count
    rows = 4 * bs * tokens
    return (hidden == 768 and heads * 64 == hidden and 2 * tokens <= 8 and rows <= 16 and channels <= 16
            and bs * tokens * channels <= 256 and _DIT_TEAM_CUS[key] >= 256)


def dit_team_workspace(hidden, device):
    n = int(_lib.load().dvla_dit_sample_workspace_bytes(int(hidden)))
    return torch.zeros(n, dtype=torch.uint8, device=device)


def dit_team_status(workspace):
    """(timeouts, xcc_mask) of the dvla_dit_sample launches on this workspace -- synchronises.  timeouts = how many launches so
    far had a wait inside the kernel time out (their outputs are NaN; the count only grows, a launch retires its own status:
    include/dvla.h); xcc_mask: bit i set = a team member of the LAST launch ran on XCC i (one bit = the team shared one L2: the
    fast case)."""
    w = workspace[:384].view(torch.int32).cpu()
    mask = 0
    for x in w[64:96].tolist():
        mask |= 1 << (x & 15)
    return int(w[33]), mask


class DitTeamTimeout(RuntimeError):
    """a wait inside dvla_dit_sample timed out (the 32 workgroups were not co-resident: shared / busy GPU); the call's output is NaN"""


def dit_team_inject_timeouts(n):
    """test hook: the next n dvla_dit_sample launches report a timeout (tests/rollout_checks.py exercises the recovery path)"""
    _lib.load().dvla_dit_sample_inject_timeouts(int(n))


def dit_team_sample(block_ptrs, depth, hidden, heads, xemb_w, xemb_b, final_w, final_b, pos, cond, coef, noise, cfg_scale, ln_eps,
                    workspace):
    """block_ptrs: int64 device tensor (depth, 8) of bf16 weight addresses (qkv w, b, proj w, b, fc1 w, b, fc2 w, b);
    cond (steps, 2 bs, T, hidden) bf16; coef (steps, 4) fp32; noise (bs, T, C) fp32.  Returns the samples (bs, T, C) fp32."""
    lib = _lib.load()
    for t, nm in ((xemb_w, "xemb_w"), (xemb_b, "xemb_b"), (final_w, "final_w"), (final_b, "final_b"), (pos, "pos"), (cond, "cond")):
        _req(t, "dit_team_sample." + nm)
        if t.dtype != BF16 or not t.is_contiguous():
            raise ValueError(f"dit_team_sample.{nm}: contiguous bf16")
    if coef.dtype != torch.float32 or noise.dtype != torch.float32 or not coef.is_contiguous() or not noise.is_contiguous():
        raise ValueError("dit_team_sample: coef / noise contiguous fp32")
    steps, two_bs, T = cond.shape[0], cond.shape[1], cond.shape[2]
    bs, Cn = noise.shape[0], noise.shape[2]
    if two_bs != 2 * bs or noise.shape[1] != T or tuple(coef.shape) != (steps, 4) or tuple(block_ptrs.shape) != (depth, 8):
        raise ValueError("dit_team_sample: shapes")
    out = torch.empty_like(noise)
    p = _lib.DitSampleParams()
    p.blocks = block_ptrs.data_ptr()
    p.xemb_w, p.xemb_b, p.final_w, p.final_b = xemb_w.data_ptr(), xemb_b.data_ptr(), final_w.data_ptr(), final_b.data_ptr()
    p.pos, p.cond, p.coef, p.noise, p.out = pos.data_ptr(), cond.data_ptr(), coef.data_ptr(), noise.data_ptr(), out.data_ptr()
    p.workspace, p.workspace_bytes = workspace.data_ptr(), workspace.numel()
    p.cfg_scale, p.ln_eps = float(cfg_scale), float(ln_eps)
    p.depth, p.hidden, p.heads, p.channels, p.tokens, p.bs, p.steps = int(depth), int(hidden), int(heads), int(Cn), int(T), int(bs), int(steps)
    check(lib.dvla_dit_sample(C.byref(p), _stream()), "dvla_dit_sample")
    return out


def add_raw(a, b, period=0):
    lib = _lib.load()
    a = a.contiguous(); b = b.contiguous()
    out = torch.empty_like(a)
    check(lib.dvla_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), int(period), _stream()), "dvla_add")
    return out


# ---------------------------------------------------------------------------------------------------
# attention launchers
# ---------------------------------------------------------------------------------------------------
class MaskTables:
    """Device-side tables derived once from an additive 0/-inf attention mask (see include/dvla.h)."""

    def __init__(self, Lq, Lk_full, Lk, key_index, bits_q, bits_k, tile_map, visible_fraction):
        self.Lq, self.Lk_full, self.Lk = Lq, Lk_full, Lk
        self.key_index, self.bits_q, self.bits_k, self.tile_map = key_index, bits_q, bits_k, tile_map
        self.visible_fraction = visible_fraction


def build_mask_tables(mask, device=None, compact_keys=True, key_order=None):
    """mask: (Lq, Lk) additive mask whose entries are 0 or -inf (generate_attention_mask,
    models/dreamvla_model.py:25-66; CLIP causal mask).  Host-side, once per mask.  `key_order` (optional int array) lists
    the kept key columns in the order the tables should use (must name exactly the columns somebody can see)."""
    import numpy as np
    device = mask.device if device is None else device
    m = mask.detach().float().cpu()
    if m.dim() != 2:
        raise ValueError("attention mask must be 2-D (Lq, Lk)")
    if bool(((m != 0) & ~torch.isneginf(m)).any()):
        raise ValueError("the HIP attention kernels take 0 / -inf masks only (all the reference ever builds)")
    vis = (m == 0).numpy()
    Lq, Lk_full = vis.shape
    key_index = None
    if compact_keys:
        # Keys nobody sees are dropped, and the kept ones are ORDERED by how many queries see them (stable, descending): the
        # gather list is free to permute the keys (softmax and P.V are sums over keys), and columns with the same audience
        # end up in the same 32-key tiles.  For the trunk mask that puts the text / state / image columns of all window steps
        # first (a block-causal prefix: tiles entirely visible or entirely hidden) and the obs columns -- visible to the three
        # action rows of their own step only -- last: 129 instead of 186 non-empty 32 x 32 tiles of 420 at L = 651, 45
        # instead of 162 of them mixed (the order csrc/masks.hip builds from the rule).
        count = vis.sum(axis=0)
        cols = np.nonzero(count > 0)[0]
        if len(cols) == 0:
            raise ValueError("mask hides every key")
        cols = cols[np.argsort(-count[cols], kind="stable")]
        if key_order is not None:
            given = np.asarray(key_order, dtype=np.int64).reshape(-1)
            if len(given) != len(cols) or not np.array_equal(np.sort(given), np.sort(cols)):
                raise ValueError("key_order must be a permutation of the visible key columns")
            cols = given
        if len(cols) < Lk_full or bool((cols != np.arange(Lk_full)).any()):
            key_index = torch.from_numpy(cols.astype(np.int32)).to(device)
            vis = vis[:, cols]
        dead = np.setdiff1d(np.arange(Lk_full), cols)
    Lk = vis.shape[1]
    nqt, nkt = (Lq + 31) // 32, (Lk + 31) // 32
    vp = np.zeros((nqt * 32, nkt * 32), dtype=bool)
    vp[:Lq, :Lk] = vis
    valid = np.zeros_like(vp)
    valid[:Lq, :Lk] = True
    w = (np.uint64(1) << np.arange(32, dtype=np.uint64))
    bits_q = (vp.reshape(nqt * 32, nkt, 32).astype(np.uint64) * w).sum(-1).astype(np.uint32)[:Lq]          # (Lq, nkt)
    bits_k = (vp.T.reshape(nkt * 32, nqt, 32).astype(np.uint64) * w).sum(-1).astype(np.uint32)[:Lk]        # (Lk, nqt)
    t_vis = vp.reshape(nqt, 32, nkt, 32).sum(axis=(1, 3))
    t_val = valid.reshape(nqt, 32, nkt, 32).sum(axis=(1, 3))
    tile_map = np.full((nqt, nkt), 2, dtype=np.uint8)
    tile_map[t_vis == t_val] = 1
    tile_map[t_vis == 0] = 0
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32) if a.dtype == np.uint32 else np.ascontiguousarray(a)).to(device)
    mt = MaskTables(Lq, Lk_full, Lk, key_index, to_dev(bits_q), to_dev(bits_k), to_dev(tile_map), float(vis.mean()))
    if compact_keys and key_index is not None:
        # key rows nobody sees: the backward kernels do not write their dk / dv -- the caller zero-fills exactly these rows
        mt.dead_keys = torch.from_numpy(dead.astype(np.int64)).to(device)
    return mt


def draw_mask_drop(K, num_obs_token, action_pred_steps, atten_only_obs, mask_l_obs_ratio):
    """The obs-token columns that `mask_l_obs_ratio` hides from the action rows of every window step, drawn with numpy's
    global RNG in the order generate_attention_mask draws them (dreamvla_model.py:49-59: one np.random.choice per window
    step, only when the atten_only_obs branch is taken).  -> int32 array (K, count), count = int(ratio * num_obs)."""
    import numpy as np
    draws = bool(num_obs_token > 0 and atten_only_obs and action_pred_steps and mask_l_obs_ratio > 0)
    count = int(mask_l_obs_ratio * num_obs_token) if draws else 0
    out = np.zeros((K, count), dtype=np.int32)
    if draws:       # also when count == 0: np.random.choice(..., size=0, replace=False) still advances the stream
        for i in range(K):
            out[i] = np.random.choice(range(num_obs_token), size=count, replace=False)
    return out


def mask_rule_visible(K, num_A, num_B, atten_goal, atten_goal_state, atten_only_obs, attn_robot_proprio_state, num_obs_token,
                      action_pred_steps, drop):
    """Host mirror (numpy, boolean (L, L)) of the predicate csrc/masks.hip evaluates on the device -- documentation and the
    CPU test's subject (tests/test_mask.py pins it bit for bit against the reference's generate_attention_mask)."""
    import numpy as np
    blk = num_A + num_B
    L = K * blk
    r = np.arange(L)[:, None]
    c = np.arange(L)[None, :]
    i, ro, j, co = r // blk, r % blk, c // blk, c % blk
    has_act = num_obs_token > 0 and action_pred_steps > 0
    act_row = has_act & (ro >= num_A + num_obs_token) & (ro < num_A + num_obs_token + action_pred_steps)
    obs_col = (co >= num_A) & (co < num_A + num_obs_token)
    base = ((j <= i) & (co < num_A)) | (act_row & (j == i) & obs_col)
    if atten_only_obs and has_act:
        dropped = np.zeros((K, max(num_obs_token, 1)), dtype=bool)
        for s in range(K):
            dropped[s, np.asarray(drop[s], dtype=np.int64)] = True
        is_dropped = dropped[np.broadcast_to(i, (L, L)), np.clip(np.broadcast_to(co, (L, L)) - num_A, 0, max(num_obs_token, 1) - 1)] & obs_col
        only = (j == i) & (((co >= 2) & (co < num_A)) | (obs_col & ~is_dropped) | (bool(attn_robot_proprio_state) & (co == 1)))
        vis = np.where(act_row, only, base)
    else:
        vis = base
    if num_obs_token > 0 and atten_goal and atten_goal_state:
        obs_row = (ro >= num_A) & (ro < num_A + num_obs_token)
        vis = vis | (obs_row & (i < K - atten_goal) & (c == (i + atten_goal) * blk + 1))
    return vis


def build_mask_tables_device(device, *, K, num_A, num_B, atten_goal=0, atten_goal_state=False, atten_only_obs=False,
                             attn_robot_proprio_state=False, num_obs_token=0, action_pred_steps=0, drop=None):
    """MaskTables of generate_attention_mask(K, num_A, num_B, ...) computed ON THE DEVICE from the rule (csrc/masks.hip):
    no (L, L) tensor, no device -> host copy, no synchronisation.  `drop` = draw_mask_drop(...) (host numpy, (K, count))."""
    import numpy as np
    lib = _lib.load()
    has_act = num_obs_token > 0 and action_pred_steps > 0
    n_drop = 0 if (drop is None or not (atten_only_obs and has_act)) else int(drop.shape[1])
    blk = num_A + num_B
    L = K * blk
    per_step = num_A + ((num_obs_token - n_drop) if has_act else 0)
    Lk = K * per_step
    nqt, nkt = (L + 31) // 32, (Lk + 31) // 32
    rule = _lib.MaskRule(K, num_A, num_B, num_obs_token, action_pred_steps, int(atten_goal), int(bool(atten_goal_state)),
                         int(bool(atten_only_obs)), int(bool(attn_robot_proprio_state)), n_drop)
    drop_dev = None
    if n_drop > 0:
        drop_dev = torch.from_numpy(np.ascontiguousarray(drop, dtype=np.int32)).to(device, non_blocking=True)
    key_index = torch.empty(Lk, dtype=torch.int32, device=device)
    bits_q = torch.empty((L, nkt), dtype=torch.int32, device=device)
    bits_k = torch.empty((Lk, nqt), dtype=torch.int32, device=device)
    tile_map = torch.empty((nqt, nkt), dtype=torch.uint8, device=device)
    check(lib.dvla_mask_tables(C.byref(rule), _ptr(drop_dev), key_index.data_ptr(), bits_q.data_ptr(), bits_k.data_ptr(),
                               tile_map.data_ptr(), _stream()), "dvla_mask_tables")
    mt = MaskTables(L, L, Lk, key_index if Lk < L else None, bits_q, bits_k, tile_map, float("nan"))
    mt._keep = drop_dev     # the kernels above read it asynchronously
    return mt


_MASK_CACHE = {}


def mask_tables_for(mask):
    """Cached build_mask_tables for a mask tensor (the reference keeps the mask as an nn.Parameter,
    dreamvla_model.py:286-298, and hands the trunk `mask[None, None]`-style views of it).  The cache is keyed on the BASE
    tensor object (a view such as `m4[0, 0]` is a new Python object on every call; its `_base` is not) plus the view's
    geometry, and an entry is valid only while that base object is alive at the same version, storage address and device:
    a freed mask's storage address can be handed to the next mask of the same size by the caching allocator, so
    (data_ptr, shape) alone would return the tables of a mask that no longer exists (round-1 ADVICE); a `.data` rebind or a
    cross-device move changes address / device without changing the version (round-2 ADVICE)."""
    import weakref
    base = mask._base if mask._base is not None else mask
    key = (id(base), mask.storage_offset(), tuple(mask.shape), tuple(mask.stride()))
    ent = _MASK_CACHE.get(key)
    if ent is not None:
        ref, version, ptr, dev, mt = ent
        if ref() is base and version == base._version and ptr == mask.data_ptr() and dev == mask.device:
            return mt
    for k in [k for k, e in _MASK_CACHE.items() if e[0]() is None]:
        del _MASK_CACHE[k]
    mt = build_mask_tables(mask)
    _MASK_CACHE[key] = (weakref.ref(base), base._version, mask.data_ptr(), mask.device, mt)
    return mt


def _attn_params(q, k, v, o, H, Lq, scale, mt, dropout_p, seed, lse):
    p = AttnParams()
    B = q.shape[0]
    p.q, p.k, p.v, p.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        if t.stride(3) != 1 or t.shape[3] != 64:
            raise ValueError("attention expects (B, L, H, 64) views with unit inner stride")
        setattr(p, name + "_stride_b", t.stride(0))
        setattr(p, name + "_stride_t", t.stride(1))
        setattr(p, name + "_stride_h", t.stride(2))
    Lk = k.shape[1]
    if mt is not None:
        if mt.Lq != Lq or mt.Lk_full != k.shape[1]:
            raise ValueError(f"attention mask is {mt.Lq}x{mt.Lk_full}, tensors are {Lq}x{k.shape[1]}")
        Lk = mt.Lk
        p.key_index = _ptr(mt.key_index)
        p.mask_bits_q, p.mask_bits_k, p.tile_map = mt.bits_q.data_ptr(), mt.bits_k.data_ptr(), mt.tile_map.data_ptr()
    p.B, p.H, p.Lq, p.Lk = B, H, Lq, Lk
    p.scale = float(scale)
    p.dropout_p = float(dropout_p)
    p.seed_lo, p.seed_hi = int(seed[0]) & 0xFFFFFFFF, int(seed[1]) & 0xFFFFFFFF
    p.lse = _ptr(lse)
    return p


def attn_fwd_raw(q, k, v, *, scale, mask_tables=None, dropout_p=0.0, seed=(0, 0), want_lse=True):
    """q: (B, Lq, H, 64), k/v: (B, Lk, H, 64) strided bf16 views.  Returns o (B, Lq, H, 64) contiguous, lse."""
    lib = _lib.load()
    for n, t in (("q", q), ("k", k), ("v", v)):
        _req(t, "attention." + n)
    B, Lq, H, _ = q.shape
    o = torch.empty((B, Lq, H, 64), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device) if want_lse else None
    p = _attn_params(q, k, v, o, H, Lq, scale, mask_tables, dropout_p, seed, lse)
    check(lib.dvla_attn_fwd(C.byref(p), _stream()), "dvla_attn_fwd")
    return o, lse


def attn_bwd_raw(q, k, v, o, lse, dout, dq, dk, dv, *, scale, mask_tables=None, dropout_p=0.0, seed=(0, 0)):
    """dk/dv rows that mask_tables.key_index does not name are NOT written: pass zero-filled buffers then."""
    lib = _lib.load()
    B, Lq, H, _ = q.shape
    if dout.stride(3) != 1:
        dout = dout.contiguous()
    p = _attn_params(q, k, v, o, H, Lq, scale, mask_tables, dropout_p, seed, lse)
    delta = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
    p.dout = dout.data_ptr()
    p.do_stride_b, p.do_stride_t, p.do_stride_h = dout.stride(0), dout.stride(1), dout.stride(2)
    p.delta = delta.data_ptr()
    p.dq, p.dk, p.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    for name, t in (("dq", dq), ("dk", dk), ("dv", dv)):
        setattr(p, name + "_stride_b", t.stride(0))
        setattr(p, name + "_stride_t", t.stride(1))
        setattr(p, name + "_stride_h", t.stride(2))
    check(lib.dvla_attn_bwd(C.byref(p), _stream()), "dvla_attn_bwd")


# ---------------------------------------------------------------------------------------------------
# autograd Functions
# ---------------------------------------------------------------------------------------------------
# fp32 masters, bf16 compute (the shipped scripts: `--precision fp32 --bf16_module vision_encoder`,
# scripts/CALVIN_ABC_D/DreamVLA/finetune.sh:13,22 -> train.py:122-163: trainable modules stay fp32).  gfx950 has no TF32: an
# fp32 GEMM runs at 1/16 of the bf16 MFMA rate, so the module-level operators below keep the caller's fp32 parameters as the
# MASTERS (the optimizer, state_dict and `.grad` are fp32, exactly what train.py builds) and multiply on bf16 SHADOWS:
#   weights      : _Shadow -- one dvla_cast_f32_to_bf16 per weight and parameter version (frozen weights: once); the gradient
#                  the backward kernels produce for the shadow is widened back to fp32 for the master
#   activations  : _ToCompute / results stay bf16 inside the model; fp32 inputs are narrowed once where they enter
#   biases / LayerNorm affine parameters: read as fp32 by the kernels, gradients written as fp32 (param_dtype codes)
# ---------------------------------------------------------------------------------------------------
class _Shadow(torch.autograd.Function):
    """bf16 compute copy of an fp32 master weight, cached per weight.  An entry is valid for exactly one (tensor object,
    version counter, storage address, device): in-place updates made on the parameter itself (`p.add_()`, torch optimizers)
    bump the version; `p.data = ...` / `module.to(device)` / flat re-homing change the address or the device.  What NONE of
    these can see is an in-place update through `.data` or a raw pointer (`p.data.add_()`, a custom kernel): every
    `Optimizer.step()` therefore drops the shadows of the trainable weights (global post-step hook installed below), and
    code that edits masters any other way calls `ops.invalidate_shadows()` (EMA swaps, manual SGD, checkpoint loads do not
    need to: `load_state_dict` copies through `param.copy_`, which bumps the version)."""
    cache = {}       # id(master) -> (weakref(master), version, data_ptr, device, bf16 shadow, requires_grad)

    @staticmethod
    def forward(ctx, w):
        import weakref
        ent = _Shadow.cache.get(id(w))
        if (ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == w.data_ptr() and ent[3] == w.device):
            return ent[4]
        sh = cast_to(w.detach(), BF16)
        if len(_Shadow.cache) > 4096:
            _Shadow.cache.clear()
        _Shadow.cache[id(w)] = (weakref.ref(w), w._version, w.data_ptr(), w.device, sh, bool(w.requires_grad))
        return sh

    @staticmethod
    def backward(ctx, g):
        return cast_to(g.contiguous(), torch.float32) if g.dtype == BF16 else g.float()


def invalidate_shadows(trainable_only=False):
    """Forget the cached bf16 shadows of fp32 master weights (all of them, or only those of weights with requires_grad):
    the next forward re-casts from the masters.  Called automatically after every torch `Optimizer.step()`; call it yourself
    after updating masters through `.data` / raw pointers outside an optimizer."""
    if not trainable_only:
        _Shadow.cache.clear()
        return
    for k in [k for k, e in _Shadow.cache.items() if e[5] or e[0]() is None]:
        del _Shadow.cache[k]


def _install_optimizer_hook():
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:       # very old torch: masters must then be updated in place on the parameter (version counter)
        return
    register_optimizer_step_post_hook(lambda opt, args, kwargs: invalidate_shadows(trainable_only=True))


_install_optimizer_hook()


class _ToCompute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return cast_to(x, BF16)

    @staticmethod
    def backward(ctx, g):
        return cast_to(g.contiguous(), torch.float32) if g.dtype == BF16 else g.float()


def shadow(w):
    """bf16 compute copy of a weight (identity for bf16 weights); differentiable: the master receives an fp32 gradient"""
    if w is None or w.dtype == BF16:
        return w
    if w.dtype != torch.float32:
        raise TypeError(f"weight dtype {w.dtype}: bf16 or fp32 (master) only")
    if not w.is_cuda:
        raise _lib.DvlaError(f"weight is on {w.device}; the DreamVLA HIP path has no CPU fallback")
    return _Shadow.apply(w)


def to_compute(x):
    """activation in the compute dtype (bf16); fp32 activations are narrowed by the HIP cast kernel, differentiably"""
    if x is None or x.dtype == BF16:
        return x
    if x.dtype != torch.float32:
        raise TypeError(f"activation dtype {x.dtype}: bf16 or fp32 only")
    if not x.is_cuda:
        raise _lib.DvlaError(f"tensor is on {x.device}; the DreamVLA HIP path has no CPU fallback")
    return _ToCompute.apply(x)


def _grad_dest(param, dtype=None):
    """The slot a gradient reducer published for this parameter's gradient (dreamvla_amd.ddp.GradBucketReducer with
    direct_grads=True), if nothing has been written there in this step -- the backward kernels then produce the gradient
    in place and AccumulateGrad adopts the returned tensor instead of adding it to a zeroed one.  None otherwise."""
    if param is None or not getattr(param, "_dvla_grad_free", False) or param.grad is not None:
        return None
    view = param._dvla_grad_view
    if dtype is not None and view.dtype != dtype:
        return None
    param._dvla_grad_free = False
    # a FRESH alias: AccumulateGrad adopts an incoming gradient only if nobody else holds a reference to that tensor
    # object (otherwise it clones it) -- the slot tensor itself is referenced by the reducer
    return view.view_as(view)


# DVLA_KSUM=0: bias gradients by the separate column-sum kernel instead of riding on the weight-gradient GEMM (same-box A/B
# measurements of the fusion; not a product knob)
_KSUM_FUSED = os.environ.get("DVLA_KSUM", "1") != "0"


def _bias_grad_out(b, dtype, n, device):
    """where a bias gradient is written: the reducer's bucket slot when there is one (see _grad_dest), else a fresh vector"""
    dst = _grad_dest(b, dtype)
    return dst if dst is not None else torch.empty(n, dtype=dtype, device=device)


class _Linear(torch.autograd.Function):
    """y = residual + dropout(act(x . W^T + b)).  conv1d=True: W is HF Conv1D (in, out) (models/gpt2.py:53)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, act, conv1d, dropout_p, res_rows=0):
        _req(x, "linear.input"); _req(w, "linear.weight")
        K = w.shape[0] if conv1d else w.shape[1]
        N = w.shape[1] if conv1d else w.shape[0]
        x2 = _rows2d(x, K)
        if res_rows and residual is not None and residual.requires_grad:
            raise ValueError("a broadcast (res_rows) residual must not require grad")
        need_grad = any(ctx.needs_input_grad)  # (grad mode is off inside Function.forward)
        seed = next_seed() if dropout_p > 0 else (0, 0)
        res2 = _rows2d(residual, N) if residual is not None else None
        want_pre = need_grad and act != 0
        r = gemm(x2, w, b_trans=conv1d, bias=b, act=act, want_preact=want_pre, dropout_p=dropout_p, seed=seed,
                 residual=res2, res_rows=res_rows)
        y2, pre = r if want_pre else (r, None)
        ctx.act, ctx.conv1d, ctx.dropout_p, ctx.seed = act, conv1d, dropout_p, seed
        ctx.has_bias, ctx.has_res = b is not None, residual is not None
        ctx.x_shape = x.shape
        ctx.bias_dtype = b.dtype if b is not None else None
        if need_grad:
            ctx.save_for_backward(x2, w, pre, b)
        return y2.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, pre, b = ctx.saved_tensors
        conv1d = ctx.conv1d
        N = w.shape[1] if conv1d else w.shape[0]
        K = w.shape[0] if conv1d else w.shape[1]
        dy2 = _rows2d(_req(dy, "linear.grad_output"), N)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        dx = dw = db = dres = None
        if ctx.act != 0 or ctx.dropout_p > 0:
            # the elementwise backward pass has dz in registers: the bias gradient (its column sums) comes out of the same pass,
            # and the weight-gradient GEMM below needs no summing code -- any configuration may take it (the phase kernel included)
            both = act_bwd_colsum(dy2, pre, ctx.act, ctx.dropout_p, ctx.seed, ctx.bias_dtype, bias_param=b) if want_db else None
            if both is not None:
                dz, db = both
            else:
                dz = act_bwd_raw(dy2, pre, ctx.act, ctx.dropout_p, ctx.seed)
        else:
            dz = dy2
        M = dz.shape[0]
        if ctx.needs_input_grad[0]:
            # dx[m,k] = sum_n dz[m,n] W(n,k)
            dx = gemm(dz, w, b_trans=not conv1d).view(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            dst = _grad_dest(w, BF16)      # the reducer's bucket slot, written in place (None: a fresh tensor)
            ks = None
            if want_db and db is None and _KSUM_FUSED:    # db = sum_m dz[m, :] rides on the dW GEMM, which streams dz anyway
                db = _bias_grad_out(b, ctx.bias_dtype, N, dz.device)
                ks = ("b" if conv1d else "a", db)
            if conv1d:   # dW[k,n] = sum_m x[m,k] dz[m,n]
                dw = gemm(x2, dz, a_trans=True, b_trans=True, split_k=auto_split_k(K, N, M), out=dst, ksum=ks)
            else:        # dW[n,k] = sum_m dz[m,n] x[m,k]
                dw = gemm(dz, x2, a_trans=True, b_trans=True, split_k=auto_split_k(N, K, M), out=dst, ksum=ks)
        if want_db and db is None:
            db = colsum(dz, ctx.bias_dtype, out=_grad_dest(b, ctx.bias_dtype))
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres, None, None, None, None


def linear_ln(x, w, b=None, *, eps, act="none", residual=None):
    """Linear(LayerNorm(x)) in ONE launch for the evaluation-time shapes: the LayerNorm (no affine parameters: the DiT blocks',
    models/action_model/models.py:129-141) is applied to the rows of x inside the few-rows GEMM (dvla.h a_layernorm).  Inference
    only (no autograd graph is recorded); callers check `torch.is_grad_enabled()` and the shape limits (ln_fusable)."""
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        raise RuntimeError("linear_ln is an inference path (call it under torch.no_grad())")
    x, w = to_compute(x), shadow(w)
    K = w.shape[1]
    x2 = _rows2d(x, K)
    res2 = _rows2d(to_compute(residual), w.shape[0]) if residual is not None else None
    y = gemm(x2, w, bias=b, act=ACT[act] if isinstance(act, str) else int(act), residual=res2, a_ln_eps=eps)
    return y.view(*x.shape[:-1], w.shape[0])


def ln_fusable(x, K):
    """may Linear(LayerNorm(x)) over rows of K features run as one few-rows GEMM launch?  (inference, <= 512 rows, 512 <= K <= 1536)"""
    return (not torch.is_grad_enabled()) and x.is_cuda and x.numel() // K <= 512 and 512 <= K <= 1536 and K % 16 == 0


def linear(x, w, b=None, *, act="none", conv1d=False, residual=None, dropout_p=0.0, res_rows=0):
    """res_rows > 0: `residual` is a (res_rows, N) table added to output row m at row m % res_rows (e.g. a fixed
    position embedding shared by every image of the batch); it receives no gradient."""
    return _Linear.apply(to_compute(x), shadow(w), b, to_compute(residual), ACT[act] if isinstance(act, str) else int(act),
                         bool(conv1d), float(dropout_p), int(res_rows))


class _Mlp(torch.autograd.Function):
    """y = residual + dropout(act(x W1^T + b1) W2^T + b2): fc1 epilogue stores the pre-activation, the
    backward dH GEMM applies act'(u) in its epilogue (no separate elementwise pass)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, act, conv1d, dropout_p):
        _req(x, "mlp.input"); _req(w1, "mlp.fc1.weight"); _req(w2, "mlp.fc2.weight")
        K = w1.shape[0] if conv1d else w1.shape[1]
        N = w2.shape[1] if conv1d else w2.shape[0]
        x2 = _rows2d(x, K)
        need_grad = any(ctx.needs_input_grad)
        seed = next_seed() if dropout_p > 0 else (0, 0)
        res2 = _rows2d(residual, N) if residual is not None else None
        if need_grad:
            h, u = gemm(x2, w1, b_trans=conv1d, bias=b1, act=act, want_preact=True)
        else:
            h, u = gemm(x2, w1, b_trans=conv1d, bias=b1, act=act), None
        y2 = gemm(h, w2, b_trans=conv1d, bias=b2, dropout_p=dropout_p, seed=seed, residual=res2)
        ctx.act, ctx.conv1d, ctx.dropout_p, ctx.seed = act, conv1d, dropout_p, seed
        ctx.x_shape = x.shape
        ctx.has_b1, ctx.has_b2, ctx.has_res = b1 is not None, b2 is not None, residual is not None
        ctx.b1_dtype = b1.dtype if b1 is not None else None
        ctx.b2_dtype = b2.dtype if b2 is not None else None
        if need_grad:
            ctx.save_for_backward(x2, w1, w2, u, h, b1, b2)
        return y2.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, u, h, b1, b2 = ctx.saved_tensors
        conv1d = ctx.conv1d
        N = w2.shape[1] if conv1d else w2.shape[0]
        Hd = w1.shape[1] if conv1d else w1.shape[0]
        K = w1.shape[0] if conv1d else w1.shape[1]
        dy2 = _rows2d(_req(dy, "mlp.grad_output"), N)
        dx = dw1 = db1 = dw2 = db2 = dres = None
        want_db2 = ctx.has_b2 and ctx.needs_input_grad[4]
        dz = dy2
        if ctx.dropout_p > 0:       # dropout backward + the second bias gradient (column sums of dz) in one pass
            both = act_bwd_colsum(dy2, None, 0, ctx.dropout_p, ctx.seed, ctx.b2_dtype, bias_param=b2) if want_db2 else None
            if both is not None:
                dz, db2 = both
            else:
                dz = act_bwd_raw(dy2, None, 0, ctx.dropout_p, ctx.seed)
        M = dz.shape[0]
        # du = (dz . W2) * act'(u)
        du = gemm(dz, w2, b_trans=not conv1d, dact_aux=u, dact=ctx.act)
        if ctx.needs_input_grad[3]:
            dst = _grad_dest(w2, BF16)
            ks = None
            if want_db2 and db2 is None and _KSUM_FUSED:
                db2 = _bias_grad_out(b2, ctx.b2_dtype, N, dz.device)
                ks = ("b" if conv1d else "a", db2)
            dw2 = (gemm(h, dz, a_trans=True, b_trans=True, split_k=auto_split_k(Hd, N, M), out=dst, ksum=ks) if conv1d else
                   gemm(dz, h, a_trans=True, b_trans=True, split_k=auto_split_k(N, Hd, M), out=dst, ksum=ks))
        if want_db2 and db2 is None:
            db2 = colsum(dz, ctx.b2_dtype, out=_grad_dest(b2, ctx.b2_dtype))
        if ctx.needs_input_grad[0]:
            dx = gemm(du, w1, b_trans=not conv1d).view(ctx.x_shape)
        want_db1 = ctx.has_b1 and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dst = _grad_dest(w1, BF16)
            ks = None
            if want_db1 and _KSUM_FUSED:
                db1 = _bias_grad_out(b1, ctx.b1_dtype, Hd, du.device)
                ks = ("b" if conv1d else "a", db1)
            dw1 = (gemm(x2, du, a_trans=True, b_trans=True, split_k=auto_split_k(K, Hd, M), out=dst, ksum=ks) if conv1d else
                   gemm(du, x2, a_trans=True, b_trans=True, split_k=auto_split_k(Hd, K, M), out=dst, ksum=ks))
        if want_db1 and db1 is None:
            db1 = colsum(du, ctx.b1_dtype, out=_grad_dest(b1, ctx.b1_dtype))
        if ctx.has_res and ctx.needs_input_grad[5]:
            dres = dy
        return dx, dw1, db1, dw2, db2, dres, None, None, None


def mlp(x, w1, b1, w2, b2, *, act, conv1d=False, residual=None, dropout_p=0.0):
    return _Mlp.apply(to_compute(x), shadow(w1), b1, shadow(w2), b2, to_compute(residual),
                      ACT[act] if isinstance(act, str) else int(act), bool(conv1d), float(dropout_p))


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _req(x, "layernorm.input")
        cols = x.shape[-1]
        x2 = x.reshape(-1, cols)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        need_grad = any(ctx.needs_input_grad)
        y, mean, rstd = layernorm_fwd(x2, gamma, beta, eps, need_grad)
        ctx.has_affine = gamma is not None
        ctx.has_beta = beta is not None
        ctx.x_shape = x.shape
        if need_grad:
            ctx.save_for_backward(x2, gamma, mean, rstd, beta)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd, beta = ctx.saved_tensors
        dy2 = _req(dy, "layernorm.grad_output").reshape(x2.shape)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        need_p = ctx.has_affine and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        gdt = gamma.dtype if need_p else torch.float32
        need_dx = ctx.needs_input_grad[0] or not need_p       # (an input without gradient: parameter gradients only)
        dx, dg, db = layernorm_bwd(dy2, x2, gamma, mean, rstd, need_p, None, gdt,
                                   _grad_dest(gamma, gdt) if need_p else None,
                                   _grad_dest(beta, gdt) if need_p and ctx.has_beta else None, need_dx=need_dx)
        return (dx.view(ctx.x_shape) if dx is not None else None), dg, (db if ctx.has_beta else None), None


def layer_norm(x, weight, bias, eps):
    return _LayerNorm.apply(to_compute(x), weight, bias, float(eps))


class _LayerNormFork(torch.autograd.Function):
    """(x, LN(x)) for a pre-LN residual block h = x + f(LN(x)): the first output is x itself, to be used as the residual
    operand of f's last GEMM.  Backward receives both gradients and returns dL/dh + LN'(dL/dLN) from ONE kernel
    (dvla_layernorm_bwd_add) instead of LayerNorm-backward followed by autograd's accumulation add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _req(x, "layernorm.input")
        cols = x.shape[-1]
        x2 = x.reshape(-1, cols)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        need_grad = any(ctx.needs_input_grad)
        y, mean, rstd = layernorm_fwd(x2, gamma, beta, eps, need_grad)
        ctx.has_affine = gamma is not None
        ctx.has_beta = beta is not None
        ctx.x_shape = x.shape
        if need_grad:
            ctx.save_for_backward(x2, gamma, mean, rstd, beta)
        return x.view_as(x), y.view(x.shape)

    @staticmethod
    def backward(ctx, dres, dy):
        x2, gamma, mean, rstd, beta = ctx.saved_tensors
        if dy is None:          # the normalised branch was not used
            return dres, None, None, None
        dy2 = _req(dy, "layernorm.grad_output").reshape(x2.shape)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dres2 = None
        if dres is not None:
            dres2 = _req(dres, "layernorm.grad_residual").reshape(x2.shape)
            if not dres2.is_contiguous():
                dres2 = dres2.contiguous()
        need_p = ctx.has_affine and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        gdt = gamma.dtype if need_p else torch.float32
        dx, dg, db = layernorm_bwd(dy2, x2, gamma, mean, rstd, need_p, dres2, gdt,
                                   _grad_dest(gamma, gdt) if need_p else None,
                                   _grad_dest(beta, gdt) if need_p and ctx.has_beta else None)
        return dx.view(ctx.x_shape), dg, (db if ctx.has_beta else None), None


class _LayerNormRows(torch.autograd.Function):
    """LayerNorm over the last `keep` tokens of every sequence of x (n, L, D) -> (n * keep, D), without materialising the strided
    slice (dvla_layernorm_fwd_rows / _bwd_rows): the dream-head decoders' `norm(x[:, -n_mask:, :])`.  Backward returns the gradient of
    the whole (n, L, D) buffer, zeros in the leading L - keep tokens of every sequence (written by the kernel)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, keep):
        lib = _lib.load()
        _req(x, "layernorm.input")
        n, L, D = x.shape
        if not x.is_contiguous():
            x = x.contiguous()
        need_grad = any(ctx.needs_input_grad)
        rows = n * keep
        y = torch.empty((rows, D), dtype=x.dtype, device=x.device)
        mean = rstd = None
        if need_grad:
            mean = torch.empty(rows, dtype=torch.float32, device=x.device)
            rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        pdt = _param_dt(gamma, "layernorm.weight") if gamma is not None else DT_BF16
        check(lib.dvla_layernorm_fwd_rows(x.data_ptr(), _ptr(gamma), _ptr(beta), pdt, y.data_ptr(), _ptr(mean), _ptr(rstd),
                                          rows, D, float(eps), keep, L, L - keep, 0, _stream()), "dvla_layernorm_fwd_rows")
        ctx.keep, ctx.has_affine, ctx.has_beta = keep, gamma is not None, beta is not None
        if need_grad:
            ctx.save_for_backward(x, gamma, mean, rstd, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, mean, rstd, beta = ctx.saved_tensors
        n, L, D = x.shape
        keep = ctx.keep
        rows = n * keep
        dy2 = _req(dy, "layernorm.grad_output").reshape(rows, D)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        need_p = ctx.has_affine and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        gdt = gamma.dtype if need_p else torch.float32
        dx = torch.empty_like(x)
        dg = db = part = None
        if need_p:
            dg = _grad_dest(gamma, gdt)
            db = _grad_dest(beta, gdt) if ctx.has_beta else None
            dg = dg if dg is not None else torch.empty(D, dtype=gdt, device=x.device)
            db = db if (db is not None or not ctx.has_beta) else torch.empty(D, dtype=gdt, device=x.device)
            part = torch.empty(2 * lib.dvla_layernorm_bwd_partial_rows() * D, dtype=torch.float32, device=x.device)
        pdt = _param_dt(gamma, "layernorm.weight") if gamma is not None else DT_BF16
        check(lib.dvla_layernorm_bwd_rows(dy2.data_ptr(), x.data_ptr(), _ptr(gamma), pdt, mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                          _ptr(dg), _ptr(db), DT_F32 if gdt == torch.float32 else DT_BF16, _ptr(part), rows, D,
                                          keep, L, L - keep, 0, _stream()), "dvla_layernorm_bwd_rows")
        return dx, dg, (db if ctx.has_beta else None), None, None


class _LayerNormConcat(torch.autograd.Function):
    """cat((LayerNorm_a(a), LayerNorm_b(b)), dim=1) of a (n, La, D) and b (n, Lb, D) -> (n, La + Lb, D): each LayerNorm writes its row range
    of the ONE output buffer (dvla_layernorm_fwd_rows, map_output), and backward reads its rows of the incoming gradient in place
    (dvla_layernorm_bwd_rows) -- no torch.cat forward, no copies of the cat's strided gradient slices backward.  The PerceiverResampler's
    `to_kv(torch.cat((norm_media(x), norm_latents(latents)), dim=-2))` (models/perceiver_resampler.py:44-49)."""

    @staticmethod
    def forward(ctx, a, ga, ba, b, gb, bb, eps_a, eps_b):
        lib = _lib.load()
        _req(a, "layernorm.input"); _req(b, "layernorm.input")
        n, La, D = a.shape
        Lb = b.shape[1]
        if b.shape[0] != n or b.shape[2] != D:
            raise ValueError("layer_norm_concat: (n, La, D) and (n, Lb, D)")
        a = a if a.is_contiguous() else a.contiguous()
        b = b if b.is_contiguous() else b.contiguous()
        L = La + Lb
        out = torch.empty((n, L, D), dtype=a.dtype, device=a.device)
        need_grad = any(ctx.needs_input_grad)
        stats = []
        for (x, g, be, Lx, off, eps) in ((a, ga, ba, La, 0, eps_a), (b, gb, bb, Lb, La, eps_b)):
            rows = n * Lx
            mean = torch.empty(rows, dtype=torch.float32, device=a.device) if need_grad else None
            rstd = torch.empty(rows, dtype=torch.float32, device=a.device) if need_grad else None
            pdt = _param_dt(g, "layernorm.weight") if g is not None else DT_BF16
            check(lib.dvla_layernorm_fwd_rows(x.data_ptr(), _ptr(g), _ptr(be), pdt, out.data_ptr(), _ptr(mean), _ptr(rstd), rows, D, float(eps),
                                              Lx, L, off, 1, _stream()), "dvla_layernorm_fwd_rows")
            stats += [mean, rstd]
        ctx.dims = (n, La, Lb, D)
        ctx.flags = (ga is not None, ba is not None, gb is not None, bb is not None)
        if need_grad:
            ctx.save_for_backward(a, ga, ba, b, gb, bb, *stats)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        a, ga, ba, b, gb, bb, mean_a, rstd_a, mean_b, rstd_b = ctx.saved_tensors
        n, La, Lb, D = ctx.dims
        L = La + Lb
        dout = _req(dout, "layernorm.grad_output")
        dout = dout if dout.is_contiguous() else dout.contiguous()
        res = []
        for k, (x, g, be, mean, rstd, Lx, off) in enumerate(((a, ga, ba, mean_a, rstd_a, La, 0), (b, gb, bb, mean_b, rstd_b, Lb, La))):
            need_x = ctx.needs_input_grad[3 * k]
            need_p = g is not None and (ctx.needs_input_grad[3 * k + 1] or ctx.needs_input_grad[3 * k + 2])
            if not (need_x or need_p):
                res += [None, None, None]
                continue
            rows = n * Lx
            gdt = g.dtype if need_p else torch.float32
            dx = torch.empty_like(x) if need_x else None      # (no input gradient wanted: neither reduced nor stored)
            dg = db = part = None
            if need_p:
                dg = _grad_dest(g, gdt)
                dg = dg if dg is not None else torch.empty(D, dtype=gdt, device=x.device)
                if be is not None:
                    db = _grad_dest(be, gdt)
                    db = db if db is not None else torch.empty(D, dtype=gdt, device=x.device)
                part = torch.empty(2 * lib.dvla_layernorm_bwd_partial_rows() * D, dtype=torch.float32, device=x.device)
            pdt = _param_dt(g, "layernorm.weight") if g is not None else DT_BF16
            check(lib.dvla_layernorm_bwd_rows(dout.data_ptr(), x.data_ptr(), _ptr(g), pdt, mean.data_ptr(), rstd.data_ptr(), _ptr(dx),
                                              _ptr(dg), _ptr(db), DT_F32 if gdt == torch.float32 else DT_BF16, _ptr(part), rows, D,
                                              Lx, L, off, 1, _stream()), "dvla_layernorm_bwd_rows")
            res += [dx, dg, db]
        return (*res[:3], *res[3:], None, None)


def layer_norm_concat(a, wa, ba, eps_a, b, wb, bb, eps_b):
    """cat((LayerNorm(a; wa, ba), LayerNorm(b; wb, bb)), dim=1) in two launches on one output buffer; see _LayerNormConcat."""
    a, b = to_compute(a), to_compute(b)
    if a.dim() != 3 or b.dim() != 3 or a.shape[-1] % 8 != 0 or a.shape[-1] > 2048:
        raise ValueError("layer_norm_concat: (n, La, D) and (n, Lb, D) with D % 8 == 0, D <= 2048")
    return _LayerNormConcat.apply(a, wa, ba, b, wb, bb, float(eps_a), float(eps_b))


def layer_norm_last_tokens(x, weight, bias, eps, keep):
    """LayerNorm(x[:, -keep:, :]) of x (n, L, D) as (n * keep, D); one kernel each way, no copy of the slice."""
    x = to_compute(x)
    if x.dim() != 3 or not (0 < keep <= x.shape[1]) or x.shape[-1] % 8 != 0 or x.shape[-1] > 2048:
        raise ValueError("layer_norm_last_tokens: x (n, L, D) with D % 8 == 0, D <= 2048 and 0 < keep <= L")
    if weight is not None and bias is not None and bias.dtype != weight.dtype:
        raise TypeError("layernorm weight/bias dtype mismatch")
    return _LayerNormRows.apply(x, weight, bias, float(eps), int(keep))


def layer_norm_fork(x, weight, bias, eps):
    """-> (x as the residual operand, LayerNorm(x)); see _LayerNormFork.  Without autograd it is layer_norm."""
    x = to_compute(x)
    if not (torch.is_grad_enabled() and (x.requires_grad or (weight is not None and weight.requires_grad))):
        return x, _LayerNorm.apply(x, weight, bias, float(eps))
    return _LayerNormFork.apply(x, weight, bias, float(eps))


class _SelfAttention(torch.autograd.Function):
    """Packed self-attention: qkv (B, L, 3*H*64) laid out [q | k | v] x (H, 64) -- exactly timm's
    qkv.reshape(B,N,3,h,d) and GPT-2's c_attn(...).split(H) order.  Returns (B, L, H*64)."""

    @staticmethod
    def forward(ctx, qkv, H, scale, mask_tables, dropout_p):
        _req(qkv, "attention.qkv")
        B, L, W = qkv.shape
        if W != 3 * H * 64:
            raise ValueError("attention: head_dim must be 64")
        if not qkv.is_contiguous():
            qkv = qkv.contiguous()
        v5 = qkv.view(B, L, 3, H, 64)
        q, k, v = v5[:, :, 0], v5[:, :, 1], v5[:, :, 2]
        need_grad = any(ctx.needs_input_grad)
        seed = next_seed() if dropout_p > 0 else (0, 0)
        o, lse = attn_fwd_raw(q, k, v, scale=scale, mask_tables=mask_tables, dropout_p=dropout_p, seed=seed,
                              want_lse=need_grad)
        ctx.H, ctx.scale, ctx.dropout_p, ctx.seed, ctx.mt = H, scale, dropout_p, seed, mask_tables
        if need_grad:
            ctx.save_for_backward(qkv, o, lse)
        return o.view(B, L, H * 64)

    @staticmethod
    def backward(ctx, dout):
        qkv, o, lse = ctx.saved_tensors
        B, L, _ = qkv.shape
        H, mt = ctx.H, ctx.mt
        v5 = qkv.view(B, L, 3, H, 64)
        # with a compacted key axis the kernel leaves dk/dv rows of never-visible keys untouched -> zeros
        dead = getattr(mt, "dead_keys", None) if mt is not None else None
        if mt is not None and mt.key_index is not None and dead is None:
            dqkv = torch.zeros_like(qkv)             # tables built on the device: the unnamed rows are not listed
        else:
            dqkv = torch.empty_like(qkv)
            if dead is not None and dead.numel():
                # dk / dv of the never-visible keys (a few rows, not the buffer); index_fill_, not `[...] = 0`: the indexed
                # assignment uploads its scalar from the host every call (a copy + 60 us of idle GPU per layer)
                dqkv.view(B, L, 3, H * 64)[:, :, 1:].index_fill_(1, dead, 0)
        d5 = dqkv.view(B, L, 3, H, 64)
        do = _req(dout, "attention.grad_output").contiguous().view(B, L, H, 64)
        attn_bwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], o, lse, do, d5[:, :, 0], d5[:, :, 1], d5[:, :, 2],
                     scale=ctx.scale, mask_tables=mt, dropout_p=ctx.dropout_p, seed=ctx.seed)
        return dqkv, None, None, None, None


class _SelfAttentionSmall(torch.autograd.Function):
    """Packed self-attention with head_dim != 64 on short sequences (dvla_attn_small_fwd / _bwd: L <= 64, head_dim <= 128, no
    mask, no dropout) -- the DiT-S action head (head_dim 96, 6 tokens).  qkv (B, L, 3*H*D) -> (B, L, H*D)."""

    @staticmethod
    def forward(ctx, qkv, H, D, scale):
        lib = _lib.load()
        _req(qkv, "attention.qkv")
        B, L, W = qkv.shape
        if W != 3 * H * D:
            raise ValueError("attention: qkv width must be 3 * heads * head_dim")
        if L > 64 or D > 128 or D % 8 or (4 * L * (D + 1) + 2 * L * (L + 1)) * 4 > 160 * 1024 or B > 65535:
            raise _lib.DvlaError(f"attention with head_dim {D}: only sequences of <= 64 tokens and head_dim <= 128 (multiple of 8) "
                                 f"are supported off the head_dim-64 MFMA kernels (got L = {L})")
        if not qkv.is_contiguous():
            qkv = qkv.contiguous()
        v5 = qkv.view(B, L, 3, H, D)
        o = torch.empty((B, L, H, D), dtype=BF16, device=qkv.device)
        lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
        p = _SelfAttentionSmall._params(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], o, H, L, scale, lse)
        check(lib.dvla_attn_small_fwd(C.byref(p), D, _stream()), "dvla_attn_small_fwd")
        ctx.H, ctx.D, ctx.scale = H, D, scale
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(qkv, o, lse)
        return o.view(B, L, H * D)

    @staticmethod
    def _params(q, k, v, o, H, L, scale, lse):
        p = AttnParams()
        p.q, p.k, p.v, p.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
        for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
            setattr(p, name + "_stride_b", t.stride(0)); setattr(p, name + "_stride_t", t.stride(1)); setattr(p, name + "_stride_h", t.stride(2))
        p.B, p.H, p.Lq, p.Lk = q.shape[0], H, L, L
        p.scale = float(scale)
        p.lse = lse.data_ptr()
        return p

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        qkv, o, lse = ctx.saved_tensors
        B, L, _ = qkv.shape
        H, D = ctx.H, ctx.D
        v5 = qkv.view(B, L, 3, H, D)
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(B, L, 3, H, D)
        do = _req(dout, "attention.grad_output").contiguous().view(B, L, H, D)
        p = _SelfAttentionSmall._params(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], o, H, L, ctx.scale, lse)
        p.dout = do.data_ptr()
        p.do_stride_b, p.do_stride_t, p.do_stride_h = do.stride(0), do.stride(1), do.stride(2)
        p.dq, p.dk, p.dv = d5[:, :, 0].data_ptr(), d5[:, :, 1].data_ptr(), d5[:, :, 2].data_ptr()
        for name, t in (("dq", d5[:, :, 0]), ("dk", d5[:, :, 1]), ("dv", d5[:, :, 2])):
            setattr(p, name + "_stride_b", t.stride(0)); setattr(p, name + "_stride_t", t.stride(1)); setattr(p, name + "_stride_h", t.stride(2))
        check(lib.dvla_attn_small_bwd(C.byref(p), D, _stream()), "dvla_attn_small_bwd")
        return dqkv, None, None, None


_PACK_TABLES = {}


def _packed_block_diagonal(G, L, device):
    """MaskTables of G sequences of length L laid end to end: a query sees the keys of its own sequence only."""
    key = (G, L, str(device))
    mt = _PACK_TABLES.get(key)
    if mt is None:
        seq = torch.arange(G * L) // L
        mask = torch.zeros(G * L, G * L)
        mask[seq[:, None] != seq[None, :]] = float("-inf")
        mt = _PACK_TABLES[key] = build_mask_tables(mask, device=device)
    return mt


def self_attention(qkv, num_heads, *, scale=None, mask_tables=None, dropout_p=0.0, head_dim=64):
    """qkv (B, L, 3 * H * 64) -> (B, L, H * 64).  head_dim != 64: the short-sequence kernel (_SelfAttentionSmall).
    Very short sequences (the DiT action head: L = 6, B = 1792) are PACKED: G consecutive sequences are handed to the
    kernels as one sequence of G * L tokens under a block-diagonal mask (a view -- the batch is contiguous -- plus a cached
    table): the kernels work on 32 x 32 score tiles and 128-query workgroups, so one 6 x 6 problem per workgroup used
    3.5 % of a tile and paid a whole prologue (272 us per backward for 0.1 GFLOP).  Masked scores contribute exactly zero;
    only the fp32 summation order of a sequence that straddles a tile boundary changes."""
    scale = (1.0 / math.sqrt(float(head_dim))) if scale is None else scale
    qkv = to_compute(qkv)
    B, L = qkv.shape[0], qkv.shape[1]
    if head_dim != 64:
        if mask_tables is not None or dropout_p > 0.0:
            raise _lib.DvlaError(f"attention with head_dim {head_dim}: masks / dropout need the head_dim-64 kernels")
        return _SelfAttentionSmall.apply(qkv, int(num_heads), int(head_dim), float(scale))
    if mask_tables is None and dropout_p == 0.0 and 1 < L <= 16 and B >= 64 and qkv.is_contiguous():
        G = 128 // L
        while G > 1 and B % G != 0:
            G -= 1
        if G > 1:
            mt = _packed_block_diagonal(G, L, qkv.device)
            o = _SelfAttention.apply(qkv.view(B // G, G * L, qkv.shape[2]), int(num_heads), float(scale), mt, 0.0)
            return o.view(B, L, o.shape[2])
    return _SelfAttention.apply(qkv, int(num_heads), float(scale), mask_tables, float(dropout_p))


class _CrossAttention(torch.autograd.Function):
    """q (B, Lq, H*64); kv (B, Lk, 2*H*64) laid out [k | v] x (H, 64) (perceiver to_kv(...).chunk(2),
    models/perceiver_resampler.py:50-52).  Returns (B, Lq, H*64)."""

    @staticmethod
    def forward(ctx, q, kv, H, scale):
        _req(q, "attention.q"); _req(kv, "attention.kv")
        B, Lq, _ = q.shape
        Lk = kv.shape[1]
        q = q.contiguous(); kv = kv.contiguous()
        q4 = q.view(B, Lq, H, 64)
        kv5 = kv.view(B, Lk, 2, H, 64)
        need_grad = any(ctx.needs_input_grad)
        o, lse = attn_fwd_raw(q4, kv5[:, :, 0], kv5[:, :, 1], scale=scale, want_lse=need_grad)
        ctx.H, ctx.scale = H, scale
        if need_grad:
            ctx.save_for_backward(q, kv, o, lse)
        return o.view(B, Lq, H * 64)

    @staticmethod
    def backward(ctx, dout):
        q, kv, o, lse = ctx.saved_tensors
        H = ctx.H
        B, Lq, _ = q.shape
        Lk = kv.shape[1]
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        kv5, dkv5 = kv.view(B, Lk, 2, H, 64), dkv.view(B, Lk, 2, H, 64)
        do = _req(dout, "attention.grad_output").contiguous().view(B, Lq, H, 64)
        attn_bwd_raw(q.view(B, Lq, H, 64), kv5[:, :, 0], kv5[:, :, 1], o, lse, do, dq.view(B, Lq, H, 64),
                     dkv5[:, :, 0], dkv5[:, :, 1], scale=ctx.scale)
        return dq, dkv, None, None


def cross_attention(q, kv, num_heads, *, scale=None):
    scale = (1.0 / math.sqrt(64.0)) if scale is None else scale
    return _CrossAttention.apply(to_compute(q), to_compute(kv), int(num_heads), float(scale))


class _AssembleTokens(torch.autograd.Function):
    """cat(parts, dim=2) + pos in one gather-write pass (dvla_assemble_tokens).  parts: (B, S, t_k, H) tensors -- ordinary,
    or expanded views of learned tokens / of a per-sample embedding (zero strides are read in place); pos (1, S, 1, H) or
    None.  Backward is slicing: every part's gradient is a view of the output gradient (autograd's expand-backward sums
    the broadcast ones), the position gradient a sum over batch and tokens."""

    @staticmethod
    def forward(ctx, pos, *parts):
        lib = _lib.load()
        B, S, _, H = parts[0].shape
        srcs = (_lib.TokenSrc * len(parts))()
        keep, t0 = [], 0
        for k, p in enumerate(parts):
            _req(p, "assemble_tokens.part")
            if p.dim() != 4 or p.shape[0] != B or p.shape[1] != S or p.shape[3] != H:
                raise ValueError("assemble_tokens: parts must be (B, S, t, H)")
            ok = (p.stride(3) == 1 and (p.shape[2] == 1 or p.stride(2) == H) and p.stride(0) % 8 == 0 and p.stride(1) % 8 == 0
                  and p.data_ptr() % 16 == 0)
            if not ok:
                p = p.contiguous()
            keep.append(p)
            srcs[k] = _lib.TokenSrc(p.data_ptr(), p.stride(0), p.stride(1), t0, p.shape[2])
            t0 += p.shape[2]
        out = torch.empty((B, S, t0, H), dtype=BF16, device=parts[0].device)
        pz = None
        if pos is not None:
            pz = _req(pos, "assemble_tokens.pos").reshape(S, H)
            if not pz.is_contiguous():
                pz = pz.contiguous()
        check(lib.dvla_assemble_tokens(srcs, len(parts), _ptr(pz), H, out.data_ptr(), B, S, t0, H, _stream()),
              "dvla_assemble_tokens")
        ctx.counts = [p.shape[2] for p in parts]
        ctx.pos_shape = None if pos is None else pos.shape
        return out

    @staticmethod
    def backward(ctx, g):
        grads, t0 = [], 0
        for k, n in enumerate(ctx.counts):
            grads.append(g[:, :, t0:t0 + n, :] if ctx.needs_input_grad[1 + k] else None)
            t0 += n
        gpos = None
        if ctx.pos_shape is not None and ctx.needs_input_grad[0]:
            gpos = g.sum(dim=(0, 2)).reshape(ctx.pos_shape)
        return (gpos, *grads)


class _ConcatSharedSuffix(torch.autograd.Function):
    """[prefix_b ; suffix] for every sequence b: (n, nq, D) and ONE (ns, D) block shared by all sequences -> (n, nq + ns, D).
    Forward = one gather-write pass (dvla_assemble_tokens with a zero batch stride for the suffix: write-bound); backward =
    a slice for the prefix and ONE column-sum pass over the batch for the suffix (dvla_colsum on the (n, ns * D) window of the
    gradient).  torch.cat((prefix, suffix.expand(n, ...))) + autograd's expand-backward moved the same bytes at a third of the
    bandwidth (the dream-head decoders build a 564-MB qkv buffer this way: 2.7 ms of cat kernels per training step)."""

    @staticmethod
    def forward(ctx, prefix, suffix):
        lib = _lib.load()
        _req(prefix, "concat_shared_suffix.prefix"); _req(suffix, "concat_shared_suffix.suffix")
        n, nq, D = prefix.shape
        ns = suffix.shape[0]
        if suffix.shape[1] != D or D % 8 != 0:
            raise ValueError("concat_shared_suffix: (n, nq, D) and (ns, D) with D % 8 == 0")
        if not (prefix.stride(2) == 1 and prefix.stride(1) == D and prefix.stride(0) % 8 == 0 and prefix.data_ptr() % 16 == 0):
            prefix = prefix.contiguous()
        suffix = suffix.contiguous()
        srcs = (_lib.TokenSrc * 2)()
        srcs[0] = _lib.TokenSrc(prefix.data_ptr(), prefix.stride(0), 0, 0, nq)
        srcs[1] = _lib.TokenSrc(suffix.data_ptr(), 0, 0, nq, ns)
        out = torch.empty((n, nq + ns, D), dtype=BF16, device=prefix.device)
        check(lib.dvla_assemble_tokens(srcs, 2, None, D, out.data_ptr(), n, 1, nq + ns, D, _stream()), "dvla_assemble_tokens")
        ctx.nq, ctx.ns = nq, ns
        return out

    @staticmethod
    def backward(ctx, g):
        g = _req(g, "concat_shared_suffix.grad").contiguous()
        n, L, D = g.shape
        dp = g[:, :ctx.nq] if ctx.needs_input_grad[0] else None
        ds = None
        if ctx.needs_input_grad[1]:
            ds = colsum(g.view(n, L * D)[:, ctx.nq * D:], out_dtype=BF16).view(ctx.ns, D)
        return dp, ds


def concat_shared_suffix(prefix, suffix):
    """(n, nq, D), (ns, D) -> (n, nq + ns, D): every sequence followed by the same suffix block"""
    return _ConcatSharedSuffix.apply(to_compute(prefix), to_compute(suffix))


def assemble_tokens(parts, pos=None):
    """(B, S, sum t_k, H) = cat(parts, dim=2) + pos"""
    if len(parts) > 16:
        raise ValueError("assemble_tokens: at most 16 parts")
    return _AssembleTokens.apply(to_compute(pos), *[to_compute(t) for t in parts])


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        _req(x, "dropout.input")
        seed = next_seed()
        ctx.p, ctx.seed, ctx.shape = p, seed, x.shape
        return dropout_raw(x.reshape(-1, x.shape[-1]), p, seed).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        return dropout_raw(dy.reshape(-1, ctx.shape[-1]), ctx.p, ctx.seed).view(ctx.shape), None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(to_compute(x), float(p))


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        _req(x, "act.input")
        ctx.act = act
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(x)
        return act_fwd_raw(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        cols = x.shape[-1]
        return act_bwd_raw(dy.reshape(-1, cols), x.reshape(-1, cols).contiguous(), ctx.act).view(x.shape), None


def activation(x, act):
    return _Act.apply(to_compute(x), ACT[act] if isinstance(act, str) else int(act))

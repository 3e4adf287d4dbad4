"""CPU, world_size 2, gloo: the gradient-bucket reducer used for data-parallel training (dreamvla_amd/ddp.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 32)
        self.b = nn.Linear(32, 8)
        self.unused = nn.Linear(4, 4)          # constructed but never used (like the reference's action_projector)
        self.tok = nn.Parameter(torch.zeros(1, 8))

        self.use_extra = False

    def forward(self, x):
        y = self.b(torch.relu(self.a(x))) + self.tok
        if self.use_extra:                     # the used set grows in a later step
            y = y + self.unused(x[:, :4]).sum(dim=1, keepdim=True)
        return y


def _worker(rank, world, port, q, direct):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dreamvla_amd.ddp import GradBucketReducer
    torch.manual_seed(0)
    m = Tiny()
    # robust_gemm_schedule=True: what the RCCL backend switches on by default (checked below through the C ABI's getter)
    red = GradBucketReducer(m.parameters(), bucket_bytes=1500, direct_grads=direct, robust_gemm_schedule=True)   # several buckets
    assert len(red.buckets) >= 3 and red.grads_are_views()
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    out = {}
    early = {}

    def _schedule():
        import ctypes
        from dreamvla_amd import _lib
        k, sk = ctypes.c_int(0), ctypes.c_int(0)
        _lib.load().dvla_get_gemm_schedule(ctypes.byref(k), ctypes.byref(sk))
        return k.value, sk.value
    sched0 = _schedule()
    for it in range(5):                 # steps 0-2: `unused` gets no gradient (learned after step 0); 3-4: it does
        m.use_extra = it >= 3
        red.zero_grad()
        loss = ((m(xs) - ys) ** 2).mean()
        loss.backward()
        early[it] = [b["launched"] for b in red.buckets]             # launched during backward, before finish()
        from dreamvla_amd.ops import GemmTuner
        if any(early[it]):       # collectives outstanding: the GEMMs run under the schedule that needs no co-residency
            assert _schedule() == (8, 0), _schedule()
            assert GemmTuner.schedule_tag == 1      # ... and the tuner keys its trials / locked choices on that (round-3 ADVICE)
        red.finish()
        assert _schedule() == sched0, (_schedule(), sched0)          # ... and back once every handle has been waited for
        assert GemmTuner.schedule_tag == 0
        assert red.grads_are_views()
        out[it] = {n: red.grad_of(p).clone() for n, p in m.named_parameters()}
        if direct:      # parameters without a gradient keep p.grad = None; the others were adopted into their bucket slot
            assert m.unused.weight.grad is None or it >= 3
            assert m.a.weight.grad.data_ptr() == red.grad_of(m.a.weight).data_ptr()
        with torch.no_grad():
            for p in m.parameters():
                p -= 0.1 * red.grad_of(p)
    ub = [i for i, b in enumerate(red.buckets) if any(p is m.unused.weight or p is m.unused.bias for p in b["params"])]
    mixed = [i for i in ub if any(p is not m.unused.weight and p is not m.unused.bias for p in red.buckets[i]["params"])]
    assert not all(early[0])                        # step 0: buckets holding an unused parameter wait for finish()
    for i in mixed:                                 # steps 1-2: they are launched as soon as their USED parameters are done
        assert not early[0][i] and early[1][i] and early[2][i], (early, i)
    for i in ub:                                    # step 3: the used set grew -> held until finish(); step 4: re-learned
        assert not early[3][i] and early[4][i], (early, i)
    # round-3 ADVICE: an iteration that never reaches finish() (backward / the step raised) must not leave the process on the
    # robust schedule: the next zero_grad() restores it
    red.zero_grad()
    ((m(xs) - ys) ** 2).mean().backward()
    assert any(b["launched"] for b in red.buckets) and _schedule() == (8, 0)
    for h, _, _ in red._handles:      # (drain what was launched so that the ranks stay in step)
        h.wait()
    red.zero_grad()
    assert _schedule() == sched0 and GemmTuner.schedule_tag == 0
    # gradient accumulation (utils/train_utils.py:588-607): two half-batches under no_sync() + outside == one full pass
    m.use_extra = True          # (the set learned in steps 3-4: every expected gradient arrives, so buckets can go out early)
    red.zero_grad()
    with red.no_sync():
        ((m(xs[:2]) - ys[:2]) ** 2).mean().mul(0.5).backward()
        assert not any(b["launched"] for b in red.buckets)
    ((m(xs[2:]) - ys[2:]) ** 2).mean().mul(0.5).backward()
    # round-2 ADVICE: the LAST pass of an accumulation cycle launches buckets during backward (arrival counters are re-armed
    # when no_sync() exits), it does not leave every collective to finish()
    assert red._next_launch > 0 and any(b["launched"] for b in red.buckets)
    red.finish()
    acc = {n: red.grad_of(p).clone() for n, p in m.named_parameters()}
    red.zero_grad()
    ((m(xs) - ys) ** 2).mean().backward()
    red.finish()
    for n, p in m.named_parameters():
        assert torch.allclose(acc[n], red.grad_of(p), atol=1e-6), n
    # a second backward without no_sync() after the exchange started must raise instead of silently diverging the ranks
    red.zero_grad()
    ((m(xs) - ys) ** 2).mean().backward()
    raised = False
    try:
        ((m(xs) - ys) ** 2).mean().backward()
    except RuntimeError as e:
        raised = "no_sync" in str(e)
    assert raised
    red.finish()
    # collectives are issued in bucket-index order whatever the readiness order
    assert red._next_launch == len(red.buckets)
    if rank == 0:
        q.put({it: {k: v.numpy() for k, v in d.items()} for it, d in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("direct", [False, True], ids=["grad_views", "direct_grads"])
def test_bucket_reducer_matches_full_batch_gradients(direct):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, direct)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full batch (mean over 8 samples == average of the two ranks' means over 4)
    torch.manual_seed(0)
    m = Tiny()
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    for it in range(5):
        m.use_extra = it >= 3
        m.zero_grad()
        ((m(X) - Y) ** 2).mean().backward()
        for n, p in m.named_parameters():
            want = torch.zeros_like(p) if p.grad is None else p.grad
            assert torch.allclose(torch.from_numpy(got[it][n]), want, atol=1e-6), (it, n)
        with torch.no_grad():
            for p in m.parameters():
                if p.grad is not None:
                    p -= 0.1 * p.grad


def test_bf16_sum_over_eight_ranks_error_is_stated():
    """the buckets are bf16 and RCCL reduces in the buffer dtype: what does an 8-rank bf16 ring sum cost against an fp32
    sum?  (simulated on CPU: sequential bf16 accumulation in ring order, the worst case of the rounding points).  The
    relative L2 error stays below 1 % of the bf16 rounding the averaged gradient gets anyway is NOT guaranteed -- the number
    is recorded here so that DESIGN.md can state it: ~4e-3 rel-L2, i.e. one bf16 ulp, for i.i.d. gradients."""
    torch.manual_seed(0)
    g = [torch.randn(1 << 16).mul(1e-3).to(torch.bfloat16) for _ in range(8)]
    acc = g[0].clone()
    for t in g[1:]:
        acc = (acc + t)            # bf16 + bf16 -> bf16: one rounding per hop
    ring = (acc / 8).float()
    exact = torch.stack([t.float() for t in g]).sum(0) / 8
    rel = float((ring - exact).norm() / exact.norm())
    assert rel < 8e-3, rel
    one_round = float((exact.to(torch.bfloat16).float() - exact).norm() / exact.norm())
    assert rel < 4 * one_round + 1e-3, (rel, one_round)


def _rccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from dreamvla_amd.ddp import GradBucketReducer
    torch.manual_seed(0)
    m = Tiny().cuda()
    red = GradBucketReducer(m.parameters(), bucket_bytes=1500)
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 16, generator=g).cuda(), torch.randn(8, 8, generator=g).cuda()
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    red.zero_grad()
    ((m(xs) - ys) ** 2).mean().backward()
    red.finish()
    torch.cuda.synchronize()
    if rank == 0:
        q.put({n: red.grad_of(p).cpu().numpy() for n, p in m.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bucket_reducer_over_rccl_two_ranks():
    """the same reducer over RCCL (backend "nccl": ReduceOp.AVG inside the collective); needs two GPUs, skips otherwise"""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    m = Tiny()
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    ((m(X) - Y) ** 2).mean().backward()
    for n, p in m.named_parameters():
        want = torch.zeros_like(p) if p.grad is None else p.grad
        assert torch.allclose(torch.from_numpy(got[n]), want, atol=1e-5), n

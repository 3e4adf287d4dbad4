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
    red = GradBucketReducer(m.parameters(), bucket_bytes=1500, direct_grads=direct)      # force several buckets
    assert len(red.buckets) >= 3 and red.grads_are_views()
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    out = {}
    early = {}
    for it in range(5):                 # steps 0-2: `unused` gets no gradient (learned after step 0); 3-4: it does
        m.use_extra = it >= 3
        red.zero_grad()
        loss = ((m(xs) - ys) ** 2).mean()
        loss.backward()
        early[it] = [b["launched"] for b in red.buckets]             # launched during backward, before finish()
        red.finish()
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

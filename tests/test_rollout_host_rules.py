"""CPU: host-side rules added in round 4 for the evaluation engine -- the forward split-K rule and its scoping, the shapes the
one-XCD sampler kernel takes, the engine's `sample=` modes (no GPU, no library calls)."""
import pytest
import torch

from dreamvla_amd import ops
from dreamvla_amd.rollout import RolloutEngine


def test_fwd_split_k_rule():
    # the evaluation engine's trunk at one episode (S = 10 / 7): the MLP down-projection is cut four ways, nothing else
    assert ops.fwd_split_k(930, 1024, 4096) == 4 and ops.fwd_split_k(651, 1024, 4096) == 4
    assert ops.fwd_split_k(930, 3072, 1024) == 1 and ops.fwd_split_k(930, 1024, 1024) == 1 and ops.fwd_split_k(930, 4096, 1024) == 1
    # the few-rows kernel's territory, and problems that fill the chip by themselves
    assert ops.fwd_split_k(512, 1024, 4096) == 1 and ops.fwd_split_k(120, 768, 3072) == 1
    assert ops.fwd_split_k(20832, 1024, 4096) == 1 and ops.fwd_split_k(59520, 1024, 4096) == 1
    # at least 1024 of K per split, at most tiles x splits ~ the CU count
    for (M, N, K) in [(600, 256, 2048), (1576, 768, 3072), (700, 512, 8192), (1000, 128, 65536)]:
        s = ops.fwd_split_k(M, N, K)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        assert 1 <= s <= 8 and K // s >= 1024 and tiles * s <= 256 + tiles


def test_forward_split_k_is_scoped():
    assert ops.FWD_SPLIT_K is False                  # not a default: training-mode kernels stay row-independent
    with ops.forward_split_k():
        assert ops.FWD_SPLIT_K is ops.FWD_SPLIT_K_ALLOWED
        with ops.forward_split_k(False):
            assert ops.FWD_SPLIT_K is False
        assert ops.FWD_SPLIT_K is ops.FWD_SPLIT_K_ALLOWED
    assert ops.FWD_SPLIT_K is False
    with pytest.raises(RuntimeError):
        with ops.forward_split_k():
            raise RuntimeError("x")
    assert ops.FWD_SPLIT_K is False                  # restored on the way out of an exception too


def test_dit_team_shapes():
    cpu = torch.device("cpu")
    assert not ops.dit_team_ok(768, 12, 7, 3, 1, cpu)            # no GPU: never


class _Act:
    in_channels = 7


class _FakeDiTModel(torch.nn.Module):
    sequence_length = 4
    hidden_dim = 8
    use_dit_head = True
    action_pred_steps = 3
    action_model = _Act()

    def __init__(self):
        super().__init__()
        self.transformer_backbone = torch.nn.Linear(8, 8)
        self.eval()


def test_engine_sample_modes():
    m = _FakeDiTModel()
    with pytest.raises(ValueError):
        RolloutEngine(m, 2, use_graph=False, sample="last")
    newest = RolloutEngine(m, 2, use_graph=False)                 # the default
    assert newest.sample_all is False and tuple(newest.draw_noise().shape) == (2, 3, 7)
    every = RolloutEngine(m, 2, use_graph=False, sample="all")
    assert every.sample_all is True and tuple(every.draw_noise().shape) == (2 * 4, 3, 7)
    m.use_dit_head = False                                          # the MLP head samples nothing: always all positions
    mlp = RolloutEngine(m, 2, use_graph=False)
    assert mlp.sample_all is True and mlp.needs_noise is False


def test_forward_split_needs_an_epilogue_the_reduction_pass_can_run():
    """round-4 ADVICE: the opportunistic forward split-K must fall back to an unsplit GEMM -- not raise -- when C / bias / residual
    do not meet the reduce-with-epilogue pass's vector constraints (CPU tensors: only addresses / strides are looked at)"""
    BF = torch.bfloat16
    out = torch.empty(930, 1024, dtype=BF)
    bias = torch.empty(1024, dtype=BF)
    res = torch.empty(930, 1024, dtype=BF)
    assert ops._fwd_split_epilogue_ok(out, bias, res, 1024)
    assert ops._fwd_split_epilogue_ok(out, None, None, 1024)
    wide = torch.empty(930, 1040, dtype=BF)
    assert not ops._fwd_split_epilogue_ok(wide[:, 4:1028], bias, res, 1024)          # `out=` view at an odd column offset
    assert not ops._fwd_split_epilogue_ok(out, bias, wide[:, 3:1027], 1024)          # sliced residual: unaligned, ld % 8 != 0 is not the issue here
    odd_ld = torch.empty(930, 1028, dtype=BF)
    assert not ops._fwd_split_epilogue_ok(out, bias, odd_ld[:, :1024], 1024)         # residual leading dimension not in whole vectors
    assert not ops._fwd_split_epilogue_ok(torch.empty(930, 1024), bias, res, 1024)   # fp32 output
    assert not ops._fwd_split_epilogue_ok(torch.empty(930, 1020, dtype=BF), None, None, 1020)
    assert not ops._fwd_split_epilogue_ok(out, torch.empty(1032, dtype=BF)[1:1025], res, 1024)   # bias at a 2-byte offset


def _libero_ensembling_oracle(chunks, temp, max_steps):
    """utils/eval_utils_libero.py:66-73 + 160-176, restated for ONE episode: the (max_steps, max_steps + P, 7) table, the chunk of
    control step t into row t / columns t .. t + P - 1, column t of every row with seven non-zero values, numpy float64 weights
    exp(-temp i) / sum (oldest row first), float64 weighted sum, gripper > 0.5 -> +-1.  Returns the executed action per step."""
    import numpy as np
    P = chunks[0].shape[0]
    table = torch.zeros(max_steps, max_steps + P, 7)
    out = []
    for t, chunk in enumerate(chunks):
        table[t:t + 1, t:t + P] = chunk.unsqueeze(0)
        col = table[:, t]
        col = col[torch.all(col != 0, axis=1)]
        w = np.exp(-temp * np.arange(len(col)))
        w = torch.from_numpy(w / w.sum()).unsqueeze(dim=1)
        a = (col * w).sum(dim=0, keepdim=True)
        a = torch.concat((a[:, :6], a[:, 6:] > 0.5), dim=-1)
        a[:, -1] = (a[:, -1] - 0.5) * 2
        out.append(a[-1].clone())
    return torch.stack(out)


@pytest.mark.parametrize("P,temp", [(3, 0.01), (5, 0.5), (1, 0.01)])
def test_temporal_ensembler_matches_the_libero_wrapper(P, temp):
    """dreamvla_amd.rollout.TemporalEnsembler (a ring of the last P chunks per episode, batched) against the restated table of
    eval_utils_libero.py: three episodes in lock-step, one reset mid-way, a chunk with a zero element (the reference drops that
    row), 14 control steps -- the float64 averages bit for bit, cast to float32"""
    from dreamvla_amd.rollout import TemporalEnsembler
    g = torch.Generator().manual_seed(5)
    B, T = 3, 14
    arm = torch.randn(T, B, P, 6, generator=g)
    grip = torch.rand(T, B, P, 1, generator=g)
    arm[4, 1, :, 2] = 0.0                     # episode 1, step 4: a zero element in every row of the chunk -> `actions_populated` drops it
    ens = TemporalEnsembler(B, P, temp, torch.device("cpu"))
    got = []
    for t in range(T):
        if t == 8:
            ens.reset(torch.tensor([False, False, True]))
        got.append(ens(arm[t], grip[t]))
    got = torch.stack(got)                    # (T, B, 7)
    for b in range(B):
        chunks = [torch.cat((arm[t, b], grip[t, b]), dim=-1) for t in range(T)]
        if b == 2:                             # the reset episode: two independent runs of the wrapper
            want = torch.cat((_libero_ensembling_oracle(chunks[:8], temp, 20), _libero_ensembling_oracle(chunks[8:], temp, 20)))
        else:
            want = _libero_ensembling_oracle(chunks, temp, 20)
        if b == 1 and P > 1:
            # step 4's chunk is dropped wherever it would be used (columns 4 .. 4 + P - 1); at step 4 itself it is the only candidate
            # besides older ones -- with P = 1 nothing is left (the reference would divide by an empty sum): not part of this case
            pass
        assert torch.equal(got[:, b].to(torch.float32), want.to(torch.float32)), (b, (got[:, b] - want).abs().max())


def test_the_garbage_collector_is_off_during_a_graph_capture(monkeypatch):
    """Round 6: the full GPU suite aborted in `Garbage-collecting` inside the encode capture -- a collected object whose destructor calls
    into HIP while the stream is capturing ends the process.  `_Graphed` collects, keeps the collector off for the capture (also when
    the captured function raises) and leaves it as it found it.  Driven here with stand-ins for the capture API (no GPU)."""
    import contextlib
    import gc

    from dreamvla_amd import rollout

    seen = []

    class FakeGraph:
        def replay(self):
            seen.append(("replay", gc.isenabled()))

    @contextlib.contextmanager
    def fake_capture(g):
        seen.append(("capture begins", gc.isenabled()))
        yield
        seen.append(("capture ends", gc.isenabled()))

    monkeypatch.setattr(torch.cuda, "CUDAGraph", FakeGraph)
    monkeypatch.setattr(torch.cuda, "graph", fake_capture)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda: None)

    def fn(x):
        seen.append(("fn", gc.isenabled()))
        return (x + 1,)

    assert gc.isenabled()
    g = rollout._Graphed(fn, warmup=1)
    x = torch.zeros(3)
    g(x)                                              # warm-up call: eager, collector untouched
    assert seen == [("fn", True)]
    out = g(x)                                        # capture + first replay
    assert seen[1:] == [("capture begins", False), ("fn", False), ("capture ends", False), ("replay", True)]
    assert gc.isenabled() and torch.equal(out[0], x + 1)

    def boom(x):
        raise RuntimeError("inside the capture")
    b = rollout._Graphed(boom, warmup=0)
    with pytest.raises(RuntimeError):
        b(x)
    assert gc.isenabled()                             # restored on the way out of an exception
    gc.disable()
    try:
        rollout._Graphed(fn, warmup=0)(x)
        assert not gc.isenabled()                     # a caller that runs without the collector keeps running without it
    finally:
        gc.enable()

"""Host logic of dreamvla_amd.rollout.RolloutEngine on CPU: the token ring reproduces the reference wrapper's history
queue / last-frame padding / action selection (utils/eval_utils_calvin.py:103-146) for lock-step and partially
reset episode batches.  The model is a stand-in (frame tokens = the frame id); the GPU parity of the real decode is in
tests/test_model_gpu.py."""
import torch

from dreamvla_amd.rollout import RolloutEngine
from tests.rollout_checks import WindowOracle


class _FakeModel(torch.nn.Module):
    sequence_length = 4
    hidden_dim = 8

    def __init__(self):
        super().__init__()
        self.transformer_backbone = torch.nn.Linear(8, 8)
        self.eval()


def test_ring_matches_reference_queue_semantics():
    B, S = 3, 4
    eng = RolloutEngine(_FakeModel(), B, use_graph=False)
    oracles = [WindowOracle(S) for _ in range(B)]
    for t in range(11):
        if t == 6:
            eng.reset(torch.tensor([False, True, False]))
            oracles[1] = WindowOracle(S)
        if t == 9:
            eng.reset()
            oracles = [WindowOracle(S) for _ in range(B)]
        ids = torch.tensor([100.0 * b + t for b in range(B)])
        eng._push(ids.view(B, 1, 1).expand(B, 36, 8).contiguous())
        for b in range(B):
            window, pick = oracles[b].push(float(ids[b]))
            assert eng.tokens[b, :, 0, 0].tolist() == window, (t, b)
            assert int(eng.count[b]) - 1 == pick, (t, b)


def test_engine_rejects_training_mode_and_wrong_history():
    import pytest
    m = _FakeModel()
    with pytest.raises(ValueError):
        RolloutEngine(m, 2, history_len=7)
    m.train()
    with pytest.raises(ValueError):
        RolloutEngine(m, 2)

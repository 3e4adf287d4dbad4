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


class _TextModel(_FakeModel):
    """stand-in with the three hooks the engine calls: the text token of a frame is the instruction's first token id, the decode
    returns it as the arm action -- so the executed action says which instruction the window was conditioned on"""
    use_dit_head = False
    action_pred_steps = 1

    def encode_text(self, tok):                       # (B, 1, 77) -> (B, 1, 1, H)
        return tok[:, :, :1].float().unsqueeze(-1).expand(tok.shape[0], 1, 1, self.hidden_dim).contiguous()

    def encode_frames(self, ip, iw, st, text_token, text_embedding=None):
        B = ip.shape[0]
        return [text_embedding.view(B, 1, 1, self.hidden_dim).float(), torch.zeros(B, 1, 35, self.hidden_dim)]

    def decode_tokens(self, tokens, mode="test", test_noise=None, test_select=None):
        B, S = tokens.shape[:2]
        arm = tokens[:, :, 0, :6].reshape(B, S, 1, 6)
        return arm, torch.ones(B, S, 1, 1)


def _text(ids):
    t = torch.zeros(len(ids), 77, dtype=torch.int64)
    t[:, 0] = torch.tensor(ids)
    return t


def test_instruction_is_latched_until_reset():
    """ModelWrapper.step fills `text_queue` once, when it is empty (utils/eval_utils_calvin.py:109-112): an episode keeps the
    instruction of its first step after reset(); text="current" follows the argument of every step (round-4 ADVICE)."""
    B = 2
    z = torch.zeros(B, 3, 4, 4)
    st = torch.zeros(B, 7)
    eng = RolloutEngine(_TextModel(), B, use_graph=False)                     # default: latched
    a, _, _ = eng.step(z, z, st, _text([11, 21]))
    assert a[:, 0].tolist() == [11.0, 21.0]
    a, _, _ = eng.step(z, z, st, _text([12, 22]))                            # a new instruction without a reset: ignored
    assert a[:, 0].tolist() == [11.0, 21.0]
    eng.reset(torch.tensor([False, True]))
    a, _, _ = eng.step(z, z, st, _text([13, 23]))                            # episode 1 was reset: it takes the new one
    assert a[:, 0].tolist() == [11.0, 23.0]
    eng.reset()
    a, _, _ = eng.step(z, z, st, _text([14, 24]))
    assert a[:, 0].tolist() == [14.0, 24.0]
    cur = RolloutEngine(_TextModel(), B, use_graph=False, text="current")
    cur.step(z, z, st, _text([11, 21]))
    a, _, _ = cur.step(z, z, st, _text([12, 22]))
    assert a[:, 0].tolist() == [12.0, 22.0]
    import pytest
    with pytest.raises(ValueError):
        RolloutEngine(_TextModel(), B, use_graph=False, text="frozen")

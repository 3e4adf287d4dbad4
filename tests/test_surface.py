"""CPU: drop-in surface of models.dreamvla_model.DreamVLA -- constructor, state_dict keys / shapes (against the
fixture dumped from the REAL reference at the shipped CALVIN finetune configuration), trainable set, casts."""
import json
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def model_and_surface():
    from models.dreamvla_model import DreamVLA
    surf = json.load(open(os.path.join(GOLD, "state_dict_surface_C.json")))
    m = DreamVLA(clip_device="cpu", vit_checkpoint_path=None, **surf["cfg"])
    return m, surf


def test_state_dict_keys_and_shapes(model_and_surface):
    m, surf = model_and_surface
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert set(mine) == set(surf["entries"]), (sorted(set(mine) ^ set(surf["entries"]))[:10])
    bad = [k for k in mine if mine[k] != surf["entries"][k]]
    assert not bad, bad[:10]


def test_trainable_parameter_set(model_and_surface):
    m, surf = model_and_surface
    mine = sorted(n for n, p in m.named_parameters() if p.requires_grad)
    assert mine == surf["trainable"]
    m.clip_model.requires_grad_(False)
    m.vision_encoder.requires_grad_(False)
    n = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert abs(n / 1e6 - 496.6) < 0.1          # 496.6 M trainable parameters


def test_caller_visible_attributes_and_casts(model_and_surface):
    m, _ = model_and_surface
    for attr in ("image_processor", "clip_model", "vision_encoder", "perceiver_resampler", "transformer_backbone",
                 "image_primary_projector", "cls_token_primary_projector", "image_wrist_projector",
                 "cls_token_wrist_projector", "sequence_length", "action_model", "attention_mask"):
        assert hasattr(m, attr), attr
    m._init_model_type()
    assert m.transformer_backbone_type == "torch.FloatTensor"
    assert m.attention_mask.shape == (651, 651) and not m.attention_mask.requires_grad
    assert callable(m.clip_model.encode_text)


def test_forward_on_cpu_raises(model_and_surface):
    from dreamvla_amd._lib import DvlaError
    m, _ = model_and_surface
    B, S = 1, 7
    with pytest.raises((DvlaError, TypeError)):
        m(torch.zeros(B, S, 3, 224, 224), torch.zeros(B, S, 3, 224, 224), torch.zeros(B, S, 7),
          torch.zeros(B, S, 77, dtype=torch.long), mode="test")


def test_dit_initialisation_matches_reference_bit_for_bit():
    """Same seed -> same DiT parameters as the REAL reference class (key order, RNG draw order of the initialisers):
    action_model/models.py:186-232.  Needs /root/reference (build container only)."""
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference tree not present")
    from dreamvla_amd.action_model.models import DiT
    ref = ref_loader.ref_module("models.action_model.models")
    kw = dict(in_channels=7, hidden_size=768, depth=2, num_heads=12, token_size=1024, future_action_window_size=2)
    torch.manual_seed(0)
    ours = DiT(**kw).state_dict()
    torch.manual_seed(0)
    theirs = ref.DiT(**kw).state_dict()
    assert list(ours) == list(theirs)
    for k in ours:
        assert torch.equal(ours[k].float(), theirs[k].float()), k

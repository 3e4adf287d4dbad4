"""the CLIP text tower pinned against an independent third-party implementation
(transformers.CLIPTextModelWithProjection; fixture by oracle/make_golden_clip.py, full-size ViT-B/32 text tower)."""
import os

import pytest
import torch

from oracle import make_golden_clip as G
from oracle import model_ref as M

FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_text_hf.pt")
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    return float((a - b).norm() / b.norm())


def test_oracle_clip_text_vs_huggingface():
    """the oracle's restatement of openai/CLIP encode_text (oracle/model_ref.py::clip_text) == Hugging Face CLIP, fp32"""
    fx = torch.load(FX, map_location="cpu")
    sd = {"c." + k: v for k, v in G.openai_state_dict().items()}
    with torch.no_grad():
        got = M.clip_text(sd, "c", fx["tokens"])
    assert _rel(got, fx["text_embeds"]) <= 2e-5


def test_product_clip_loads_openai_keys():
    """dreamvla_amd.clip_text.CLIPTextEncoder carries exactly the openai/CLIP text-tower keys (state_dict surface)"""
    from dreamvla_amd.clip_text import CLIPTextEncoder
    m = CLIPTextEncoder()
    want = set(G.openai_keys()) | {"logit_scale"}
    assert set(m.state_dict()) == want
    for k, shp in G.openai_keys().items():
        assert tuple(m.state_dict()[k].shape) == tuple(shp), k


@pytest.mark.gpu
def test_hip_clip_text_vs_huggingface():
    """the HIP text tower (bf16) vs the Hugging Face fp32 outputs.  Tolerance: a 12-layer bf16 pipeline against an fp32
    reference -- the same class as the 12-layer ViT (fixtures' recorded reference bf16 deviations are 4-6e-3 for comparable
    depth): 1e-2 rel-L2, and an element-wise bound of 24 bf16 ulps of the largest magnitude."""
    from dreamvla_amd.clip_text import CLIPTextEncoder
    fx = torch.load(FX, map_location="cpu")
    m = CLIPTextEncoder()
    sd = G.openai_state_dict()
    sd["logit_scale"] = m.state_dict()["logit_scale"]
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda", BF).eval()
    with torch.no_grad():
        got = m.encode_text(fx["tokens"].to("cuda"))
    r = _rel(got, fx["text_embeds"])
    max_abs = float((got.float().cpu() - fx["text_embeds"]).abs().max())
    assert r <= 1e-2, r
    assert max_abs <= 24 * 2.0 ** -8 * float(fx["text_embeds"].abs().max()), max_abs

"""the input pipeline (ToTensor + Normalize + RandomShiftsAug + bf16 cast).  Golden vectors come from the REAL
RandomShiftsAug class of the reference (tests/golden/input_pipeline.pt, oracle/make_golden_inputs.py)."""
import os

import numpy as np
import pytest
import torch

from dreamvla_amd import preprocess as P

FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "input_pipeline.pt")


def _norm(u8):
    x = u8.permute(0, 3, 1, 2).float().div(255.0)
    return x.sub(torch.tensor(P.CLIP_MEAN).view(1, 3, 1, 1)).div(torch.tensor(P.CLIP_STD).view(1, 3, 1, 1))


def _want(case, got_full):
    if "out" in case:
        return got_full, case["out"]
    return got_full.flatten()[case["idx"]], case["vals"]


def _ulp_bf16(v):
    return 2.0 ** (torch.floor(torch.log2(v.abs().clamp_min(1e-30))) - 7)


def test_gather_formulation_matches_reference_module():
    """RandomShiftsAug == replicate-clamped integer gather: the host mirror of the kernel's addressing against the real
    module's grid_sample output.  grid_sample computes its sample coordinates and bilinear weights in fp32, so the reference
    lands a few 1e-5 pixels off the pixel centres it aims at and its output carries up to ~1e-4 of a neighbouring pixel
    (measured: 2e-5 at 64 px, 1e-4 at 224 px, on values in [-2, 2.7]); the gather is the exact answer.  After the model's
    bf16 cast the two differ on a few per cent of the elements, by one bf16 ulp each."""
    fx = torch.load(FX, map_location="cpu")
    for case in fx["cases"]:
        x = _norm(case["u8"])
        got = P.shift_gather_reference(x, case["shift"], case["pad"])
        g, w = _want(case, got)
        assert float((g - w).abs().max()) <= 2e-4, case["hw"]
        gb, wb = g.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()
        assert float((gb != wb).float().mean()) <= 0.05
        assert bool(((gb - wb).abs() <= 2e-4 + 2.0 ** -7 * torch.maximum(gb.abs(), wb.abs())).all())   # one bf16 ulp (or the fp32 noise near 0)


def test_shift_draw_ranges():
    g = torch.Generator().manual_seed(0)
    a = P.draw_shifts(1000, 10, traj=False, generator=g)
    b = P.draw_shifts(1000, 10, traj=True, generator=g)
    assert int(a.min()) == 0 and int(a.max()) == 20 and int(b.min()) == 1 and int(b.max()) == 20


def test_clip_image_preprocess_host():
    """the image_processor drop-in: shapes, value range, centre crop of a non-square frame, identity on 224 x 224 inputs"""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    x = P.clip_image_preprocess(img)
    assert x.shape == (3, 224, 224) and x.dtype == torch.float32
    assert torch.equal(x, _norm(torch.from_numpy(img)[None])[0])             # no resize needed: pure ToTensor + Normalize
    wide = rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)               # CALVIN static camera is 200 x 200; a wide one:
    u8 = P.clip_image_resize_u8(wide)
    assert u8.shape == (224, 224, 3)
    from PIL import Image
    ref = Image.fromarray(wide).resize((336, 224), Image.BICUBIC).crop((56, 0, 280, 224))
    assert np.array_equal(u8, np.asarray(ref))


def test_no_cpu_fallback():
    from dreamvla_amd._lib import DvlaError
    with pytest.raises(DvlaError):
        P.preprocess_frames(torch.zeros(1, 64, 64, 3, dtype=torch.uint8))


@pytest.mark.gpu
def test_device_pipeline_vs_reference_module():
    """dvla_image_preprocess (uint8 HWC -> bf16 CHW, shifts injected) vs the real module's output rounded to bf16: bit-equal
    to the exact gather (bf16 of the fp32 normalisation), and equal to the reference's grid_sample result except for the
    rounding-boundary crossings quantified in the CPU test."""
    fx = torch.load(FX, map_location="cpu")
    for case in fx["cases"]:
        got = P.preprocess_frames(case["u8"].cuda(), case["shift"].cuda(), case["pad"]).cpu()
        exact = P.shift_gather_reference(_norm(case["u8"]), case["shift"], case["pad"]).to(torch.bfloat16)
        assert torch.equal(got, exact), (case["hw"], case["traj"])
        g, w = _want(case, got.float())
        wb = w.to(torch.bfloat16).float()
        assert float((g != wb).float().mean()) <= 0.05
        assert bool(((g - wb).abs() <= 2e-4 + 2.0 ** -7 * torch.maximum(g.abs(), wb.abs())).all())
    # no augmentation: shifts = None is the identity
    u8 = fx["cases"][0]["u8"]
    got = P.preprocess_frames(u8.cuda()).cpu()
    assert torch.equal(got, _norm(u8).to(torch.bfloat16))

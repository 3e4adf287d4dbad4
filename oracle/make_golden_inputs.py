"""ORACLE -- golden vectors for the input pipeline from the REAL reference class.

`utils/data_utils.py` cannot be imported here (omegaconf, pytorch3d, petrel ... are absent), so the `RandomShiftsAug` class
definition is taken out of the reference source file with `ast` and executed unchanged in a namespace that holds torch / nn / F
(utils/data_utils.py:326-383); torch.randint is replaced by a recorder so that the integer shifts it draws are stored next
to the outputs.  The normalisation in front of it is ToTensor + Normalize exactly as clip's `_transform` composes them.

    python -m oracle.make_golden_inputs        # writes tests/golden/input_pipeline.pt
"""
import ast
import os
import sys

import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/utils/data_utils.py"


def reference_class():
    src = open(REF).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "RandomShiftsAug")
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(compile(ast.Module(body=[node], type_ignores=[]), REF, "exec"), ns)
    return ns["RandomShiftsAug"]


def normalise(u8):
    """(n, H, W, 3) uint8 -> (n, 3, H, W) fp32: ToTensor then Normalize (clip _transform)"""
    from dreamvla_amd.preprocess import CLIP_MEAN, CLIP_STD
    x = u8.permute(0, 3, 1, 2).float().div(255.0)
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(1, 3, 1, 1)
    return x.sub(mean).div(std)


def run(aug, x, traj):
    """run the real module and record the shifts it drew: returns (output, (n, 2) int32 pixel shifts (sx, sy))"""
    rec = {}
    real = torch.randint

    def fake(lo, hi, size=None, **kw):
        g = torch.Generator().manual_seed(77 if traj else 78)
        v = real(lo, hi, size, generator=g)
        rec["shift"] = v.clone()
        return v.to(kw.get("dtype", torch.float32))
    torch.randint = fake
    try:
        out = aug.forward_traj(x) if traj else aug(x)
    finally:
        torch.randint = real
    return out, rec["shift"].reshape(-1, 2).to(torch.int32)


def main():
    cls = reference_class()
    g = torch.Generator().manual_seed(5)
    fx = {"source": "RandomShiftsAug extracted from /root/reference/utils/data_utils.py:326-383 and run unchanged; "
                    "ToTensor + Normalize as clip's _transform (oracle/make_golden_inputs.py)", "cases": []}
    for (n, t, hw, pad, traj) in [(3, 1, 64, 4, False), (2, 3, 64, 10, True), (2, 1, 224, 10, False), (1, 2, 224, 4, True)]:
        u8 = torch.randint(0, 256, (n * t, hw, hw, 3), generator=g, dtype=torch.uint8)
        x = normalise(u8)
        aug = cls(pad)
        if traj:
            out, shift = run(aug, x.view(n, t, 3, hw, hw), True)
            out = out.reshape(n * t, 3, hw, hw)
        else:
            out, shift = run(aug, x, False)
        case = dict(n=n * t, hw=hw, pad=pad, traj=traj, u8=u8, shift=shift)
        if hw <= 64:
            case["out"] = out.clone()
        else:                       # large frames: a strided sample + the bf16-rounded checksum
            flat = out.flatten()
            idx = torch.linspace(0, flat.numel() - 1, 20000).long()
            case.update(idx=idx, vals=flat[idx].clone(), bf16_sum=float(out.to(torch.bfloat16).float().sum()))
        fx["cases"].append(case)
    torch.save(fx, os.path.join(GOLD, "input_pipeline.pt"))
    print("input_pipeline.pt", os.path.getsize(os.path.join(GOLD, "input_pipeline.pt")))


if __name__ == "__main__":
    main()

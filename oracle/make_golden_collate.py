"""ORACLE -- golden vectors for the loader side of the input pipeline from the REAL reference methods.

`utils/data_utils.py` cannot be imported here (omegaconf, pytorch3d, petrel, torchvision ... are absent), so -- exactly as
oracle/make_golden_inputs.py does for `RandomShiftsAug` -- the function definitions are cut out of the reference source with
`ast` and executed UNCHANGED:

    DiskCalvinDataset.collator     /root/reference/utils/data_utils.py:1308-1397
    DiskLiberoDataset.collator     /root/reference/utils/data_utils.py:2719-2798
    preprocess_image               :175-179   (the collator's `self.image_fn`, bound with functools.partial as in :1000-1003)
    depth_image_fn                 :3588-3607
    RandomShiftsAug                :326-383   (`self.rgb_shift` / `self.gripper_shift`)

on a stub `self` that carries the attributes the methods read (image_fn, text_fn, rgb_pad, gripper_pad, traj_cons, act_step,
window_size, rgb_shift, gripper_shift, load_track_labels).  Third-party pieces that are not in /root/reference and not in this
image are stand-ins, as under oracle/shims/: `T.Resize(NEAREST)` on a tensor (torchvision: F.interpolate(mode="nearest")), the
`image_processor` that `clip.load` returns (openai/CLIP `_transform`: torchvision Resize(BICUBIC) / CenterCrop / ToTensor /
Normalize on a PIL image = dreamvla_amd.preprocess.clip_image_preprocess, which restates exactly that), and the tokenizer
(tests/collate_samples.fake_tokenize).  torch.randint is wrapped by a recorder so that the integer shifts the augmentation
draws are stored next to the outputs (they are injected into the device collator by the test).

    python -m oracle.make_golden_collate        # writes tests/golden/collate.pt
"""
import ast
import functools
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference/utils/data_utils.py"


class _Resize:
    """torchvision.transforms.Resize((h, w), interpolation=NEAREST) applied to a (C, H, W) tensor"""

    def __init__(self, size, interpolation=None):
        assert interpolation == "nearest"
        self.size = tuple(size)

    def __call__(self, x):
        return F.interpolate(x.unsqueeze(0), size=self.size, mode="nearest").squeeze(0)


T_SHIM = types.SimpleNamespace(Resize=_Resize, InterpolationMode=types.SimpleNamespace(NEAREST="nearest"))


def reference_namespace():
    """executes the reference definitions named in the module docstring; returns the namespace"""
    tree = ast.parse(open(REF).read())
    ns = {"torch": torch, "nn": nn, "F": F, "np": np, "T": T_SHIM}
    want_fn = {"preprocess_image", "depth_image_fn"}
    body = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in want_fn)
            or (isinstance(n, ast.ClassDef) and n.name == "RandomShiftsAug")]
    assert len(body) == 3
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    for cls_name, key in (("DiskCalvinDataset", "calvin_collator"), ("DiskLiberoDataset", "libero_collator")):
        cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name)
        fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "collator")
        sub = dict(ns)
        exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), sub)
        ns[key] = sub["collator"]
        ns[key + "_lines"] = (fn.lineno, fn.end_lineno)
    return ns


def stub_self(ns, case, window_size):
    from dreamvla_amd.preprocess import clip_image_preprocess
    from tests.collate_samples import fake_tokenize
    return types.SimpleNamespace(
        image_fn=functools.partial(ns["preprocess_image"], image_processor=clip_image_preprocess),
        text_fn=fake_tokenize, rgb_pad=case["rgb_pad"], gripper_pad=case["gripper_pad"], traj_cons=case["traj_cons"],
        act_step=case["act_step"], window_size=window_size, load_track_labels=case["load_track_labels"],
        rgb_shift=ns["RandomShiftsAug"](case["rgb_pad"]) if case["rgb_pad"] != -1 else None,
        gripper_shift=ns["RandomShiftsAug"](case["gripper_pad"]) if case["gripper_pad"] != -1 else None)


def run_case(ns, case):
    from tests.collate_samples import STRIDE, make_samples
    samples = make_samples(case)
    window = case["T"] - (case["act_step"] - 1)
    me = stub_self(ns, case, window)
    draws = []
    real = torch.randint
    g = torch.Generator().manual_seed(1234 + len(case["name"]))

    def recorder(lo, hi, size=None, **kw):
        v = real(lo, hi, size, generator=g)
        draws.append(v.reshape(-1, 2).to(torch.int32).clone())
        return v.to(kw.get("dtype", torch.float32))
    torch.randint = recorder
    try:
        out = ns[case["dataset"] + "_collator"](me, samples)
    finally:
        torch.randint = real
    assert isinstance(out, tuple) and len(out) == 13
    # which draw belongs to which entry: the methods call rgb_shift on the static frames, then (CALVIN + traj_cons) on the static
    # depth maps, then gripper_shift on the gripper frames, then (CALVIN + traj_cons) on the gripper depth maps
    order = []
    depth_too = case["dataset"] == "calvin" and case["traj_cons"]
    if case["rgb_pad"] != -1:
        order += ["rgb_static"] + (["depth_static"] if depth_too else [])
    if case["gripper_pad"] != -1:
        order += ["rgb_gripper"] + (["depth_gripper"] if depth_too else [])
    assert len(order) == len(draws), (order, len(draws))
    rec = dict(case=case, window_size=window, shifts=dict(zip(order, draws)), entries={})
    names = ["image", "text", "action", "gripper", "state", "robot_obs", "depth_static", "depth_gripper", "dino", "dino_gripper",
             "sam", "sam_gripper", "tracks"]
    for name, v in zip(names, out):
        if name in ("image", "gripper", "depth_static", "depth_gripper") and v is not None:
            flat = v.contiguous().flatten()
            rec["entries"][name] = dict(shape=tuple(v.shape), dtype=str(v.dtype), sample=flat[::STRIDE].clone(), sum=float(flat.double().sum()))
        elif isinstance(v, dict):
            rec["entries"][name] = {k: t.clone() for k, t in v.items()}
        else:
            rec["entries"][name] = None if v is None else v.clone()
    return rec


def main():
    from tests.collate_samples import CASES
    ns = reference_namespace()
    fx = {"source": "DiskCalvinDataset.collator (data_utils.py:%d-%d) and DiskLiberoDataset.collator (:%d-%d), preprocess_image, "
                    "depth_image_fn and RandomShiftsAug cut out of /root/reference/utils/data_utils.py with ast and run unchanged "
                    "on a stub self (oracle/make_golden_collate.py)" % (ns["calvin_collator_lines"] + ns["libero_collator_lines"]),
          "cases": [run_case(ns, c) for c in CASES]}
    path = os.path.join(GOLD, "collate.pt")
    torch.save(fx, path)
    print("collate.pt", os.path.getsize(path), fx["source"])


if __name__ == "__main__":
    main()

"""SURVEY 8 f3, loader side: dreamvla_amd.collate.DeviceCollator / LiberoDeviceCollator against the REAL reference methods
`DiskCalvinDataset.collator` (utils/data_utils.py:1308-1397) and `DiskLiberoDataset.collator` (:2719-2798): the methods are cut
out of the reference source with `ast` and run unchanged on a stub self by oracle/make_golden_collate.py (fixture
tests/golden/collate.pt: 8 cases over act_step 1 / 3, traj_cons, pads on / off, tracks / DINO / SAM labels present / absent,
load_track_labels).  CPU: the 11 host entries of the 13-entry tuple bit for bit (text, actions, states, robot_obs chunks, depth
maps incl. the traj_cons shift, label cuts, track dictionary).  GPU: the two camera entries (bf16 device tensors produced by
csrc/input_pipeline.hip) against the fp32 frames the real collator produced, with the shifts it drew injected."""
import os

import pytest
import torch

from dreamvla_amd import collate, preprocess as P
from tests.collate_samples import CASES, STRIDE, fake_tokenize, make_samples

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collate.pt")
HOST_ENTRIES = ["text", "action", "state", "robot_obs", "depth_static", "depth_gripper", "dino", "dino_gripper", "sam", "sam_gripper", "tracks"]
NAMES = ["image", "text", "action", "gripper", "state", "robot_obs", "depth_static", "depth_gripper", "dino", "dino_gripper",
         "sam", "sam_gripper", "tracks"]


def _fixture():
    return torch.load(GOLD, weights_only=False)


def _collator(rec, device):
    c = rec["case"]
    cls = collate.LiberoDeviceCollator if c["dataset"] == "libero" else collate.DeviceCollator
    col = cls(fake_tokenize, window_size=rec["window_size"], rgb_pad=c["rgb_pad"], gripper_pad=c["gripper_pad"],
              traj_cons=c["traj_cons"], act_step=c["act_step"], load_track_labels=c["load_track_labels"], device=device)
    used = []

    def inject(n, pad, key):                 # the shifts the REAL collator drew for this entry (recorded by the generator)
        sh = rec["shifts"][key]
        assert sh.shape == (n, 2) and int(sh.min()) >= (1 if c["traj_cons"] else 0) and int(sh.max()) <= 2 * pad
        used.append(key)
        return sh
    col._shifts = inject
    return col, used


def _same(got, want, name):
    if want is None:
        assert got is None, name
    elif isinstance(want, dict):
        assert isinstance(got, dict) and set(got) == set(want), name
        for k in want:
            assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), (name, k)
    else:
        assert got.dtype == want.dtype and got.shape == want.shape and torch.equal(got, want), name


def _depth_close(got, ent, name):
    """depth entries: stored as every STRIDE-th element + the sum.  Unshifted maps are bit-identical; on the traj_cons path the
    reference's grid_sample evaluates its sample points in fp32 and carries up to ~1e-4 of a neighbouring pixel where this
    path does the exact integer gather (tests/test_input_pipeline.py)."""
    assert tuple(got.shape) == ent["shape"] and str(got.dtype) == ent["dtype"], name
    flat = got.contiguous().flatten()
    d = (flat[::STRIDE] - ent["sample"]).abs()
    assert float(d.max()) <= 2e-3, (name, float(d.max()))           # depth values are O(5); a wrong pixel is off by O(1)
    assert abs(float(flat.double().sum()) - ent["sum"]) <= 1e-5 * abs(ent["sum"]) + 1e-3, name
    return float(d.max())


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_host_entries_match_the_real_collators(idx):
    rec = _fixture()["cases"][idx]
    assert rec["case"] == CASES[idx]
    smp = make_samples(rec["case"])
    col, used = _collator(rec, "cpu")
    T = rec["case"]["T"]
    col._camera = lambda sample, cam, pad: torch.zeros(len(sample), T, 3, 8, 8)          # (the HIP half is tested on the GPU)
    out = col(smp)
    assert isinstance(out, tuple) and len(out) == 13
    got = dict(zip(NAMES, out))
    for name in HOST_ENTRIES:
        want = rec["entries"][name]
        if name.startswith("depth") and want is not None:
            dmax = _depth_close(got[name], want, name)
            if not rec["case"]["traj_cons"]:
                assert dmax == 0.0
        else:
            _same(got[name], want, name)
    # the cut of the camera entries (the stand-in frames carry T steps): window_size steps are left
    assert got["image"].shape[1] == rec["window_size"] and got["gripper"].shape[1] == rec["window_size"]
    assert tuple(rec["entries"]["image"]["shape"][:2]) == (len(smp), rec["window_size"])
    # every depth draw the real collator made was consumed, in the reference's call order
    assert used == [k for k in rec["shifts"] if k.startswith("depth")]


def test_fixture_covers_what_it_says():
    fx = _fixture()
    assert "DiskCalvinDataset.collator" in fx["source"] and "DiskLiberoDataset.collator" in fx["source"] and "run unchanged" in fx["source"]
    cs = [r["case"] for r in fx["cases"]]
    assert {c["dataset"] for c in cs} == {"calvin", "libero"} and {c["act_step"] for c in cs} == {1, 3}
    assert {c["traj_cons"] for c in cs} == {False, True}
    assert any(r["entries"]["tracks"] == {} for r in fx["cases"]) and any(r["entries"]["tracks"] for r in fx["cases"])
    assert all(r["entries"]["depth_static"] is None for r in fx["cases"] if r["case"]["dataset"] == "libero")
    assert all(r["entries"]["depth_static"] is not None for r in fx["cases"] if r["case"]["dataset"] == "calvin")


def test_libero_collator_requires_episode_id():
    rec = _fixture()["cases"][4]
    smp = make_samples(rec["case"])
    col, _ = _collator(rec, "cpu")
    col._camera = lambda sample, cam, pad: torch.zeros(len(sample), rec["case"]["T"], 3, 8, 8)
    del smp[1]["episode_id"]
    with pytest.raises(KeyError):
        col(smp)


def _fake_tokenize(strings):
    calls.append(list(strings))
    return torch.stack([torch.tensor([hash(s) % 49408] + [0] * 76, dtype=torch.int64) for s in strings])


calls = []


def test_token_cache_tokenises_each_string_once():
    calls.clear()
    tc = collate.TokenCache(_fake_tokenize)
    a = tc(["x", "y", "x", "x"])
    b = tc(["y", "x"])
    assert calls == [["x", "y"]] and tc.hits == 4 and tc.misses == 2
    assert torch.equal(a[0], a[2]) and torch.equal(a[1], b[0]) and a.shape == (4, 77)


def test_uint8_frames_are_the_input_of_the_reference_transform():
    """the host half keeps exactly the information the fp32 CLIP transform has: Normalize(ToTensor(u8)) == image_processor(pil)"""
    s = make_samples(CASES[0], 1)[0]
    for f in s["rgb_obs"]["rgb_static"][:2] + s["rgb_obs"]["rgb_gripper"][:2]:
        u8 = torch.from_numpy(P.clip_image_resize_u8(f).copy())
        x = u8.permute(2, 0, 1).float().div(255.0)
        x = (x - torch.tensor(P.CLIP_MEAN).view(3, 1, 1)) / torch.tensor(P.CLIP_STD).view(3, 1, 1)
        assert torch.equal(x, P.clip_image_preprocess(f))


def test_token_cache_overflow_keeps_the_whole_batch():
    """round-3 ADVICE: a clear in the middle of a batch must not lose the batch's already-cached strings"""
    calls.clear()
    tc = collate.TokenCache(_fake_tokenize, max_entries=3)
    tc(["a", "b", "c"])
    out = tc(["a", "d", "b"])                      # 3 cached + 1 new > 3: cleared, the whole batch re-tokenised
    assert out.shape == (3, 77) and calls[-1] == ["a", "d", "b"]
    assert torch.equal(out[0], _fake_tokenize(["a"])[0]) and torch.equal(out[2], _fake_tokenize(["b"])[0])


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(CASES)))
def test_device_frames_match_the_real_collators(idx):
    """entries 0 and 3 on the device (uint8 upload + one HIP kernel) against the fp32 frames of the real collator, same shifts.
    Tolerance: the kernel normalises with an fma on (u8 / 255) and rounds to bf16 -- at most one bf16 ulp (2^-6 at |x| < 4) from
    the reference's fp32 value; grid_sample's fp32 sample points add ~1e-4 of a neighbouring pixel on a few per cent of elements."""
    rec = _fixture()["cases"][idx]
    smp = make_samples(rec["case"])
    col, used = _collator(rec, "cuda")
    out = col(smp)
    got = dict(zip(NAMES, out))
    for name in ("image", "gripper"):
        ent, g = rec["entries"][name], got[name]
        assert g.is_cuda and g.dtype == torch.bfloat16 and tuple(g.shape) == ent["shape"], name
        flat = g.float().cpu().contiguous().flatten()
        want = ent["sample"]
        d = (flat[::STRIDE] - want).abs()
        assert float(d.max()) <= 2.0 ** -6 + 1e-3, (name, float(d.max()))
        assert float((d > 2.0 ** -7).float().mean()) < 0.05, name
        assert abs(float(flat.double().sum()) - ent["sum"]) <= 1e-4 * float(flat.abs().double().sum()), name   # unbiased bf16 rounding
    assert used == list(rec["shifts"])              # every draw of the real collator, in its call order
    for name in HOST_ENTRIES:                         # and the host entries are what the CPU test checked
        want = rec["entries"][name]
        if not (name.startswith("depth") and want is not None):
            _same(got[name], want, name)

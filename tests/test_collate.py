"""SURVEY 8 f3, loader side: dreamvla_amd.collate.DeviceCollator against a restatement of the reference collator's image / text
half (utils/data_utils.py:175-183, 1308-1397) on synthetic PIL frames.  CPU: the host pieces (uint8 resize == the input of the
fp32 transform, token cache, tuple layout, act_step chunking, depth path).  GPU: the device tensors against the host transform +
the exact shift gather (itself pinned to the real RandomShiftsAug class in tests/test_input_pipeline.py)."""
import numpy as np
import pytest
import torch

from dreamvla_amd import collate, preprocess as P


def _samples(B=2, T=5, hw=(200, 200), seed=0, labels=True):
    from PIL import Image
    rng = np.random.RandomState(seed)
    out = []
    for b in range(B):
        s = {"actions": [rng.uniform(-1, 1, 7).astype(np.float32) for _ in range(T)],
             "robot_obs": [rng.uniform(-1, 1, 15).astype(np.float32) for _ in range(T)],
             "rgb_obs": {"rgb_static": [Image.fromarray(rng.randint(0, 256, hw + (3,), dtype=np.uint8)) for _ in range(T)],
                         "rgb_gripper": [Image.fromarray(rng.randint(0, 256, (84, 84, 3), dtype=np.uint8)) for _ in range(T)]},
             "depth_obs": {"depth_static": [rng.uniform(0, 5, hw).astype(np.float32) for _ in range(T)],
                           "depth_gripper": [rng.uniform(0, 5, (84, 84)).astype(np.float32) for _ in range(T)]},
             "lang": ["push the red block", "open the drawer"][b % 2]}
        if labels:
            s["sam_features_obs"] = {"sam_feats_static": torch.randn(T, 256, 256), "sam_feats_gripper": torch.randn(T, 256, 256)}
        out.append(s)
    return out


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
    s = _samples(1, 2)[0]
    for f in s["rgb_obs"]["rgb_static"] + s["rgb_obs"]["rgb_gripper"]:
        u8 = torch.from_numpy(P.clip_image_resize_u8(f).copy())
        x = u8.permute(2, 0, 1).float().div(255.0)
        x = (x - torch.tensor(P.CLIP_MEAN).view(3, 1, 1)) / torch.tensor(P.CLIP_STD).view(3, 1, 1)
        assert torch.equal(x, P.clip_image_preprocess(f))


def test_host_side_layout_matches_the_reference_collator():
    """tuple order / shapes / act_step chunking of data_utils.py:1308-1397 (device = cpu: only the host fields are built)"""
    smp = _samples(3, 6)
    col = collate.DeviceCollator(_fake_tokenize, window_size=4, act_step=3, device="cpu")
    col._camera = lambda sample, cam, pad: torch.zeros(len(sample), 6, 3, 8, 8)          # (the HIP half is tested on the GPU)
    out = col(smp)
    assert len(out) == 13
    img, txt, act, grip, state, robot, d_s, d_g, dino, dino_g, sam, sam_g, tr = out
    assert txt.shape == (3, 77) and act.shape == (3, 4, 3, 7) and robot.shape == (3, 4, 3, 7)
    assert img.shape[1] == 4 and grip.shape[1] == 4 and state.shape == (3, 4, 15) and d_s.shape == (3, 4, 1, 224, 224)
    assert dino is None and sam.shape == (3, 4, 256, 256) and tr == {}
    ref_act = torch.from_numpy(np.array([np.stack(s["actions"]) for s in smp]))
    ref_state = torch.from_numpy(np.array([np.stack(s["robot_obs"]) for s in smp]))
    for b in range(3):
        for ix in range(4):
            assert torch.equal(act[b, ix], ref_act[b, ix:ix + 3])
            want = ref_state[b, ix:ix + 3]
            assert torch.equal(robot[b, ix], torch.cat([want[..., :6], want[..., [-1]]], -1))
    # depth: nearest resize, unshifted without traj_cons
    want = torch.nn.functional.interpolate(torch.from_numpy(np.stack(smp[0]["depth_obs"]["depth_static"])).unsqueeze(1), size=(224, 224), mode="nearest")
    assert torch.equal(d_s[0], want[:4])


def test_token_cache_overflow_keeps_the_whole_batch():
    """round-3 ADVICE: a clear in the middle of a batch must not lose the batch's already-cached strings"""
    calls.clear()
    tc = collate.TokenCache(_fake_tokenize, max_entries=3)
    tc(["a", "b", "c"])
    out = tc(["a", "d", "b"])                      # 3 cached + 1 new > 3: cleared, the whole batch re-tokenised
    assert out.shape == (3, 77) and calls[-1] == ["a", "d", "b"]
    assert torch.equal(out[0], _fake_tokenize(["a"])[0]) and torch.equal(out[2], _fake_tokenize(["b"])[0])


def _libero_reference_collate(sample, window_size, act_step):
    """restatement of the host half of DiskLiberoDataset.collator (data_utils.py:2719-2798) -- actions, states, robot_obs
    chunking, label cuts, return layout; the image half is the CALVIN one (same image_fn / RandomShiftsAug calls)"""
    action = torch.from_numpy(np.array([np.stack(s["actions"]) for s in sample]))
    state = torch.from_numpy(np.array([np.stack(s["robot_obs"]) for s in sample]))
    _ = [s["episode_id"] for s in sample]
    sam = torch.stack([s["sam_features_obs"]["sam_feats_static"] for s in sample]) if "sam_features_obs" in sample[0] else None
    tr = None
    if "track_label" in sample[0]:
        tr = {k: torch.stack([s["track_label"][k] for s in sample]) for k in ("tracks", "track_visibility", "tracks_gripper",
                                                                               "track_visibility_gripper")}
    robot = torch.zeros(1)
    if act_step != 1:
        acts = torch.zeros((action.shape[0], window_size, act_step, action.shape[-1]))
        robot = torch.zeros((action.shape[0], window_size, act_step, state.shape[-1]))
        for b in range(action.shape[0]):
            for ix in range(window_size):
                acts[b, ix] = action[b, ix:ix + act_step]
                robot[b, ix] = state[b, ix:ix + act_step]
        robot = torch.cat([robot[..., :6], robot[..., [-1]]], dim=-1)
        action = acts
        state = state[:, :-(act_step - 1)]
        sam = None if sam is None else sam[:, :-(act_step - 1)]
        tr = None if tr is None else {k: v[:, :-(act_step - 1)] for k, v in tr.items()}
    return action, state, robot, sam, (tr if tr is not None else dict())


@pytest.mark.parametrize("act_step", [1, 3])
def test_libero_collator_layout(act_step):
    """DiskLiberoDataset.collator (data_utils.py:2719-2798): no depth entries even when the samples carry depth, the track
    dictionary whenever the samples have `track_label`, `episode_id` required, robot_obs chunking (2765-2772)"""
    T = 6
    W = T - (act_step - 1)
    smp = _samples(3, T)
    for i, s in enumerate(smp):
        s["episode_id"] = 100 + i
        s["robot_obs"] = [np.concatenate([r, r[:1]]) for r in s["robot_obs"]]      # 16 values: LIBERO's proprio layout is wider
        s["track_label"] = {k: torch.randn(T, 196, 2) for k in ("tracks", "track_visibility", "tracks_gripper",
                                                                 "track_visibility_gripper")}
    col = collate.LiberoDeviceCollator(_fake_tokenize, window_size=W, act_step=act_step, device="cpu")
    col._camera = lambda sample, cam, pad: torch.zeros(len(sample), T, 3, 8, 8)
    out = col(smp)
    assert len(out) == 13
    img, txt, act, grip, state, robot, d_s, d_g, dino, dino_g, sam, sam_g, tr = out
    assert d_s is None and d_g is None                       # LIBERO: no depth path although the samples hold depth_obs
    r_act, r_state, r_robot, r_sam, r_tr = _libero_reference_collate(smp, W, act_step)
    assert torch.equal(act, r_act) and torch.equal(state, r_state) and torch.equal(robot, r_robot) and torch.equal(sam, r_sam)
    assert set(tr) == set(r_tr) and all(torch.equal(tr[k], r_tr[k]) for k in tr)
    assert img.shape[1] == W and grip.shape[1] == W and txt.shape == (3, 77)
    if act_step != 1:
        assert robot.shape == (3, W, act_step, 7) and act.shape == (3, W, act_step, 7)
    # the CALVIN flavour on the same samples: depth present, tracks withheld without load_track_labels
    outc = collate.DeviceCollator(_fake_tokenize, window_size=W, act_step=act_step, device="cpu")
    outc._camera = col._camera
    oc = outc(smp)
    assert oc[6] is not None and oc[12] == {}
    del smp[1]["episode_id"]
    with pytest.raises(KeyError):
        col(smp)


@pytest.mark.gpu
@pytest.mark.parametrize("traj", [False, True])
def test_device_frames_match_host_transform_and_shift(traj):
    smp = _samples(2, 3, labels=False)
    g = torch.Generator().manual_seed(5)
    col = collate.DeviceCollator(_fake_tokenize, window_size=3, rgb_pad=10, gripper_pad=4, traj_cons=traj, device="cuda", generator=g)
    out = col(smp)
    img, grip = out[0], out[3]
    assert img.is_cuda and img.dtype == torch.bfloat16 and img.shape == (2, 3, 3, 224, 224) and grip.shape == (2, 3, 3, 224, 224)
    g2 = torch.Generator().manual_seed(5)
    for got, cam, pad in ((img, "rgb_static", 10), (grip, "rgb_gripper", 4)):
        host = torch.stack([torch.stack([P.clip_image_preprocess(f) for f in s["rgb_obs"][cam]]) for s in smp])   # the reference's image_fn
        sh = P.draw_shifts(6, pad, traj=traj, generator=g2)
        want = P.shift_gather_reference(host.view(6, 3, 224, 224), sh, pad).view(2, 3, 3, 224, 224)
        d = (got.float().cpu() - want.to(torch.bfloat16).float()).abs()
        # the kernel normalises with an fma on (u8 / 255): at most one bf16 ulp from the host's (x - mean) / std
        assert float(d.max()) <= 2.0 ** -6 and float((d > 0).float().mean()) < 0.2

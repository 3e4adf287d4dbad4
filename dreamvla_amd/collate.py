"""Loader side of the input pipeline: a drop-in for `DiskCalvinDataset.collator`
(/root/reference/utils/data_utils.py:1308-1397) and `DiskLiberoDataset.collator` (2719-2798) that keeps the camera frames as uint8 until they are on the device.

The reference collator runs the CLIP image transform on every PIL frame on the host (`self.image_fn` = `preprocess_image`,
data_utils.py:175-179: Resize / CenterCrop / ToTensor / Normalize -> fp32 (3, 224, 224) = 602 KB per frame), stacks, applies
`RandomShiftsAug` with `grid_sample` on the host (1336-1355), and the training loop then uploads fp32 and casts
(utils/train_utils.py:99-100).  Here the host does only what needs PIL (the antialiased bicubic resize + crop, to uint8 HWC =
150 KB per frame); the resized frames go to the device in ONE pinned, asynchronous copy per camera and
`dreamvla_amd.preprocess.preprocess_frames` (csrc/input_pipeline.hip) does ToTensor + Normalize + the shift gather + the bf16
cast in one kernel.  The instruction strings are tokenised through a per-string cache (CALVIN has 34 tasks x a few phrasings;
the reference re-tokenises every sample of every batch, data_utils.py:181-183).  Everything else of the collator -- states,
actions, the `act_step` chunking, depth / DINO / SAM / track labels -- is host tensor bookkeeping and is kept as it is.

The returned tuple has the reference's layout (13 entries); entries 0 and 3 (static / gripper camera) are bf16 CUDA tensors
(B, T, 3, 224, 224), so the loop's `.to(device_id, dtype=cast_dtype, non_blocking=True)` on them is a no-op."""
import numpy as np
import torch

from . import preprocess as P


class TokenCache:
    """`tokenize(list_of_str) -> (n, 77) int tensor` behind a per-string cache (the collator's `text_fn`)."""

    def __init__(self, tokenize, max_entries=65536):
        self.tokenize, self.max_entries = tokenize, max_entries
        self.cache = {}
        self.hits = self.misses = 0

    def __call__(self, strings):
        uniq = list(dict.fromkeys(strings))
        missing = [s for s in uniq if s not in self.cache]
        if missing and len(self.cache) + len(missing) > self.max_entries:
            # full: start over WITH this batch -- every string of the batch is tokenised again, so the stack below finds all of
            # them (round-3 ADVICE: clearing and re-inserting only `missing` lost the batch's previously cached strings)
            self.cache.clear()
            missing = uniq
        if missing:
            toks = self.tokenize(missing)
            for s, t in zip(missing, toks):
                self.cache[s] = t.clone()
        self.misses += len(missing)
        self.hits += len(strings) - len(missing)
        return torch.stack([self.cache[s] for s in strings])


def depth_image_fn(depth_images, size=224):
    """data_utils.py:3588-3607: stack -> (N, 1, H, W) fp32 -> nearest resize to 224 x 224 (torchvision Resize(NEAREST) on a
    tensor is F.interpolate(mode="nearest"))."""
    d = np.stack([np.array(img, dtype=np.float32) for img in depth_images])
    if d.ndim != 3:
        raise ValueError("Depth images should have shape (N, H, W)")
    t = torch.from_numpy(d).unsqueeze(1)
    return torch.nn.functional.interpolate(t, size=(size, size), mode="nearest")


class DeviceCollator:
    """dataset="calvin": DiskCalvinDataset.collator (data_utils.py:1308-1397).  dataset="libero": DiskLiberoDataset.collator
    (data_utils.py:2719-2798) -- the same image / text / chunking code (robot_obs chunks at :2765-2772, forward_traj branch
    at :2747-2760) with three differences: LIBERO has NO depth path (entries 6 and 7 are None whatever the samples carry,
    :2795-2796), the track dictionary is returned whenever the samples hold `track_label` (:2798; CALVIN: when
    `load_track_labels`), and every sample must carry an `episode_id` (:2726)."""

    def __init__(self, tokenize, window_size, rgb_pad=-1, gripper_pad=-1, traj_cons=False, act_step=1, n_px=224,
                 device="cuda", load_track_labels=False, generator=None, dataset="calvin"):
        if dataset not in ("calvin", "libero"):
            raise ValueError(f"dataset {dataset!r}: 'calvin' or 'libero'")
        self.dataset = dataset
        self.text_fn = tokenize if isinstance(tokenize, TokenCache) else TokenCache(tokenize)
        self.window_size, self.act_step = window_size, act_step
        self.rgb_pad, self.gripper_pad, self.traj_cons = rgb_pad, gripper_pad, traj_cons
        self.n_px, self.device, self.load_track_labels = n_px, torch.device(device), load_track_labels
        self.generator = generator

    # ---- camera frames: PIL -> uint8 HWC on the host, everything else on the device -------------------------------------
    def _frames_u8(self, sample, cam):
        arr = np.stack([np.stack([P.clip_image_resize_u8(f, self.n_px) for f in s["rgb_obs"][cam]]) for s in sample])
        t = torch.from_numpy(arr)                               # (B, T, H, W, 3) uint8
        if self.device.type == "cuda":
            t = t.pin_memory()
        return t

    def _shifts(self, n, pad, key):
        """the (n, 2) integer shifts of one RandomShiftsAug call; `key` names the entry they are for ("rgb_static",
        "depth_static", "rgb_gripper", "depth_gripper" -- drawn in this order, the order of the reference's calls,
        data_utils.py:1336-1355).  RandomShiftsAug.forward draws randint(0, 2 pad + 1) per image, forward_traj
        randint(1, 2 pad + 1) per frame (data_utils.py:344-348 / 371-375): one (sx, sy) pair per frame either way.
        The parity test replaces this method to inject the shifts the real collator drew (tests/test_collate.py)."""
        return P.draw_shifts(n, pad, traj=self.traj_cons, generator=self.generator)

    def _camera(self, sample, cam, pad):
        u8 = self._frames_u8(sample, cam)
        B, T = u8.shape[:2]
        shifts = None if pad == -1 else self._shifts(B * T, pad, cam)
        dev = u8.to(self.device, non_blocking=True)
        return P.preprocess_frames(dev, shifts, 0 if pad == -1 else pad)             # (B, T, 3, H, W) bf16, one kernel

    def _depth(self, sample, cam, pad):
        d = torch.stack([depth_image_fn(s["depth_obs"][cam], self.n_px) for s in sample])      # (B, T, 1, H, W) fp32, host
        if pad != -1 and self.traj_cons:         # (the reference shifts the depth maps only on the traj_cons path, with their OWN draw)
            B, T = d.shape[:2]
            sh = self._shifts(B * T, pad, cam)
            d = P.shift_gather_reference(d.view(B * T, *d.shape[2:]), sh, pad).view_as(d)
        return d

    def __call__(self, sample):
        action_tensors = torch.from_numpy(np.array([np.stack(s["actions"]) for s in sample]))
        state_tensors = torch.from_numpy(np.array([np.stack(s["robot_obs"]) for s in sample]))
        libero = self.dataset == "libero"
        has_depth = (not libero) and "depth_obs" in sample[0]
        image_tensors = self._camera(sample, "rgb_static", self.rgb_pad)
        depth_static = self._depth(sample, "depth_static", self.rgb_pad) if has_depth else None
        gripper_tensors = self._camera(sample, "rgb_gripper", self.gripper_pad)
        depth_gripper = self._depth(sample, "depth_gripper", self.gripper_pad) if has_depth else None
        if libero:
            _ = [s["episode_id"] for s in sample]        # KeyError on a sample without one, like the reference
        text_tensors = self.text_fn([s["lang"] for s in sample])
        tracks = {}
        if "track_label" in sample[0]:
            tracks = {k: torch.stack([s["track_label"][k] for s in sample])
                      for k in ("tracks", "track_visibility", "tracks_gripper", "track_visibility_gripper")}
        dino = dino_g = sam = sam_g = None
        if "dino_features_obs" in sample[0]:
            dino = torch.stack([s["dino_features_obs"]["dino_feats_static"] for s in sample])
            dino_g = torch.stack([s["dino_features_obs"]["dino_feats_gripper"] for s in sample])
        if "sam_features_obs" in sample[0]:
            sam = torch.stack([s["sam_features_obs"]["sam_feats_static"] for s in sample])
            sam_g = torch.stack([s["sam_features_obs"]["sam_feats_gripper"] for s in sample])
        robot_obs = torch.zeros(1)
        if self.act_step != 1:                      # data_utils.py:1359-1391
            a = self.act_step
            acts = torch.zeros((action_tensors.shape[0], self.window_size, a, action_tensors.shape[-1]))
            robot_obs = torch.zeros((action_tensors.shape[0], self.window_size, a, state_tensors.shape[-1]))
            for b in range(action_tensors.shape[0]):
                for ix in range(self.window_size):
                    acts[b, ix] = action_tensors[b, ix:ix + a]
                    robot_obs[b, ix] = state_tensors[b, ix:ix + a]
            robot_obs = torch.cat([robot_obs[..., :6], robot_obs[..., [-1]]], dim=-1)
            action_tensors = acts
            cut = lambda t: None if t is None else t[:, :-(a - 1)]
            image_tensors, gripper_tensors, state_tensors = cut(image_tensors), cut(gripper_tensors), cut(state_tensors)
            depth_static, depth_gripper = cut(depth_static), cut(depth_gripper)
            tracks = {k: cut(v) for k, v in tracks.items()}
            dino, dino_g, sam, sam_g = cut(dino), cut(dino_g), cut(sam), cut(sam_g)
        return (image_tensors, text_tensors, action_tensors, gripper_tensors, state_tensors, robot_obs, depth_static, depth_gripper,
                dino, dino_g, sam, sam_g, tracks if (self.load_track_labels or libero) else dict())


class LiberoDeviceCollator(DeviceCollator):
    """drop-in for DiskLiberoDataset.collator (data_utils.py:2719-2798)"""

    def __init__(self, *args, **kwargs):
        kwargs["dataset"] = "libero"
        super().__init__(*args, **kwargs)

"""Synthetic loader samples shared by oracle/make_golden_collate.py (which feeds them to the REAL reference collators) and
tests/test_collate.py (which feeds the same samples to dreamvla_amd.collate): what `DiskCalvinDataset.__getitem__` /
`BaseLiberoDataset.__getitem__` hand to `collator` -- lists of PIL frames, per-frame action / state vectors, depth maps, label
tensors (feature sizes shrunk: the collators only stack and cut them), the instruction string."""
import numpy as np
import torch

CASES = [
    # name, dataset, T, act_step, traj_cons, rgb_pad, gripper_pad, load_track_labels, extras the samples carry
    dict(name="calvin_a1", dataset="calvin", T=4, act_step=1, traj_cons=False, rgb_pad=10, gripper_pad=4, load_track_labels=False, extras=("sam",)),
    dict(name="calvin_a3", dataset="calvin", T=5, act_step=3, traj_cons=False, rgb_pad=10, gripper_pad=4, load_track_labels=True, extras=("sam", "dino", "track")),
    dict(name="calvin_a3_traj", dataset="calvin", T=5, act_step=3, traj_cons=True, rgb_pad=10, gripper_pad=4, load_track_labels=False, extras=("sam", "track")),
    dict(name="calvin_a1_traj_nopad", dataset="calvin", T=3, act_step=1, traj_cons=True, rgb_pad=-1, gripper_pad=4, load_track_labels=True, extras=("track",)),
    dict(name="libero_a1", dataset="libero", T=4, act_step=1, traj_cons=False, rgb_pad=10, gripper_pad=4, load_track_labels=False, extras=("sam",)),
    dict(name="libero_a3", dataset="libero", T=5, act_step=3, traj_cons=False, rgb_pad=10, gripper_pad=4, load_track_labels=False, extras=("sam", "dino", "track")),
    dict(name="libero_a3_traj", dataset="libero", T=5, act_step=3, traj_cons=True, rgb_pad=10, gripper_pad=4, load_track_labels=False, extras=("sam",)),
    dict(name="libero_a1_traj_track", dataset="libero", T=3, act_step=1, traj_cons=True, rgb_pad=4, gripper_pad=-1, load_track_labels=False, extras=("track",)),
]
LANG = ["push the red block to the left", "open the drawer", "turn on the lightbulb"]


def make_samples(case, B=2):
    from PIL import Image
    seed = sum(ord(c) for c in case["name"])
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    T, libero = case["T"], case["dataset"] == "libero"
    ns = 16 if libero else 15                  # LIBERO's proprio vector is wider; the collator keeps [:6] + [-1] of the chunks
    out = []
    for b in range(B):
        s = {"actions": [rng.uniform(-1, 1, 7).astype(np.float32) for _ in range(T)],
             "robot_obs": [rng.uniform(-1, 1, ns).astype(np.float32) for _ in range(T)],
             "rgb_obs": {"rgb_static": [Image.fromarray(rng.randint(0, 256, (200, 200, 3), dtype=np.uint8)) for _ in range(T)],
                         "rgb_gripper": [Image.fromarray(rng.randint(0, 256, (84, 84, 3), dtype=np.uint8)) for _ in range(T)]},
             "depth_obs": {"depth_static": [rng.uniform(0, 5, (200, 200)).astype(np.float32) for _ in range(T)],
                           "depth_gripper": [rng.uniform(0, 5, (84, 84)).astype(np.float32) for _ in range(T)]},
             "lang": LANG[(b + seed) % len(LANG)]}
        if libero:
            s["episode_id"] = 100 + b
        if "sam" in case["extras"]:
            s["sam_features_obs"] = {"sam_feats_static": torch.randn(T, 16, 8, generator=g), "sam_feats_gripper": torch.randn(T, 16, 8, generator=g)}
        if "dino" in case["extras"]:
            s["dino_features_obs"] = {"dino_feats_static": torch.randn(T, 12, 6, generator=g), "dino_feats_gripper": torch.randn(T, 12, 6, generator=g)}
        if "track" in case["extras"]:
            s["track_label"] = {"tracks": torch.randn(T, 9, 2, generator=g), "track_visibility": (torch.rand(T, 9, generator=g) > 0.5),
                                "tracks_gripper": torch.randn(T, 9, 2, generator=g), "track_visibility_gripper": (torch.rand(T, 9, generator=g) > 0.5)}
        out.append(s)
    return out


def fake_tokenize(strings):
    """deterministic stand-in for clip.tokenize(list, truncate=True) -> (n, 77) int64 (the tokenizer itself is third party)"""
    rows = []
    for s in strings:
        ids = [49406] + [1 + (sum(ord(c) for c in w) * 31 + i) % 49000 for i, w in enumerate(s.split())] + [49407]
        rows.append(torch.tensor(ids + [0] * (77 - len(ids)), dtype=torch.int64))
    return torch.stack(rows)


STRIDE = 53      # the image / depth entries are stored as every 53rd element (53 is coprime to 224: all rows / columns get sampled)

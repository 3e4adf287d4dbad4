"""Seeded synthetic CALVIN-like batches for bench.py, smoke tests and golden generation.
There is no network / dataset in the build or bench environment; shapes and value ranges follow the reference
collator (utils/data_utils.py:1308-1397) and train loop (utils/train_utils.py:99-145)."""
import torch


def synthetic_batch(B, S, window=None, seed=1234, heads=(), gripper_width=False, tracks=False):
    """Seeded synthetic CALVIN-like batch; values are bf16-representable.
    gripper_width: the LIBERO state layout (6 arm values + the two finger widths, utils/train_utils.py:128-129) instead of CALVIN's
    6 + open / closed flag; tracks: CoTracker labels without the trajectory head (the `--load_track_labels --flow_as_mask` runs of
    scripts/LIBERO/DreamVLA/finetune_long.sh use them to mask the image loss)."""
    W = window or S
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    u = lambda *s: torch.rand(*s, generator=g).to(torch.bfloat16).float()
    batch = {
        "image_primary": r(B, W, 3, 224, 224),
        "image_wrist": r(B, W, 3, 224, 224),
        "state": torch.cat([u(B, W, 6), (torch.rand(B, W, 1, generator=g) > 0.5).float()], dim=-1),
        "text_token": torch.randint(1, 49000, (B, 77), generator=g).unsqueeze(1).repeat(1, W, 1),
        "actions": torch.cat([u(B, W, 6) * 2 - 1, (torch.rand(B, W, 1, generator=g) > 0.5).float()], dim=-1),
    }
    # EOT token (largest id) at a random position, as clip.tokenize produces
    eot = torch.randint(5, 77, (B,), generator=g)
    for b in range(B):
        batch["text_token"][b, :, eot[b]] = 49407
        batch["text_token"][b, :, eot[b] + 1:] = 0
    if "depth" in heads:
        batch["depth_primary"] = u(B, W, 1, 224, 224) * 10 + 0.01
        batch["depth_wrist"] = u(B, W, 1, 224, 224) * 10 + 0.01
    if "dino" in heads:
        batch["dino_primary"], batch["dino_wrist"] = r(B, W, 256, 768), r(B, W, 256, 768)
    if "sam" in heads:
        batch["sam_primary"], batch["sam_wrist"] = r(B, W, 256, 256), r(B, W, 256, 256)
    if gripper_width:
        batch["state"] = torch.cat([batch["state"][..., :6], u(B, W, 2) * 0.08], dim=-1)
    if "traj" in heads or tracks:
        batch["tracks"], batch["tracks_gripper"] = r(B, W, 784, 2) * 2, r(B, W, 784, 2) * 2
    return batch

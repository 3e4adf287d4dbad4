"""ORACLE -- golden values of the reference's training-loss block, produced by running the REAL training loop
`utils/train_utils.py:59-760 train_one_epoch_calvin` (imported from /root/reference) for one batch on CPU with
stand-ins for everything around the loss arithmetic: a model stub whose forward returns seeded prediction tensors
(leaf tensors, so the loop's own `loss.backward()` leaves d(total)/d(prediction) on them), a one-batch loader, a
no-op optimizer / scheduler and a `wandb` object that records what the loop logs.  Build container only.

    python -m oracle.make_golden_losses        # writes tests/golden/losses.pt

Stored: the case definitions (seeds, flags), the loss values the loop logged, and a strided sample of the gradient the
loop's backward put on every prediction.  tests/test_losses_golden.py replays the cases through
dreamvla_amd/losses.py::calvin_losses on the same seeded tensors (`loss_case_tensors` below regenerates them).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402
from dreamvla_amd.synthetic import synthetic_batch  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# flags follow scripts/CALVIN_ABC_D/DreamVLA/finetune.sh (C), the MLP-head variant (A) and
# scripts/LIBERO/DreamVLA/finetune_long.sh with every dream head switched on (E: BASELINE configs[3])
CASES = {
    "C_calvin_dit": dict(B=2, S=3, heads=("depth", "sam"), use_dit_head=True, obs_pred=True, depth_pred=True,
                         sam_feat_pred=True, dino_feat_pred=False, trajectory_pred=False, flow_as_mask=False, seed=11),
    "A_mlp_head": dict(B=2, S=3, heads=(), use_dit_head=False, obs_pred=True, depth_pred=False, sam_feat_pred=False,
                       dino_feat_pred=False, trajectory_pred=False, flow_as_mask=False, seed=12),
    "E_libero_all_heads": dict(B=2, S=3, heads=("dino", "sam", "traj"), use_dit_head=True, obs_pred=True, depth_pred=False,
                               sam_feat_pred=True, dino_feat_pred=True, trajectory_pred=True, flow_as_mask=True, seed=13),
    "E_atten_goal": dict(B=2, S=4, heads=("dino", "traj"), use_dit_head=False, obs_pred=True, depth_pred=False,
                         sam_feat_pred=False, dino_feat_pred=True, trajectory_pred=True, flow_as_mask=False, atten_goal=1,
                         seed=14),
}
FUTURE, PRED_NUM, STEPS = 3, 1, 3


def loss_case_tensors(case):
    """seeded batch (window = S + 3) and prediction tensors of one case; shapes are the ones DreamVLA.forward returns"""
    B, S = case["B"], case["S"]
    n = B * S
    batch = synthetic_batch(B, S, window=S + FUTURE, seed=case["seed"], heads=case["heads"])
    batch["actions"][..., 6:] = batch["actions"][..., 6:] * 2 - 1        # the loader's {-1, +1} gripper command
    g = torch.Generator().manual_seed(1000 + case["seed"])
    r = lambda *s: torch.randn(*s, generator=g)
    preds = {}
    if case["use_dit_head"]:
        preds["arm"] = torch.rand((), generator=g) + 0.5                   # the DiT head returns its scalar loss
    else:
        preds["arm"] = r(B, S, STEPS, 6)
        preds["gripper"] = torch.sigmoid(r(B, S, STEPS, 1))
    preds["image"] = r(n, 2, PRED_NUM, 196, 768)
    if case["depth_pred"]:
        preds["depth"] = torch.rand(n, 2, PRED_NUM, 196, 256, generator=g) * 5 + 0.05     # post-ReLU depths
    if case["dino_feat_pred"]:
        preds["dino"] = r(n, 2, PRED_NUM, 256, 768)
    if case["sam_feat_pred"]:
        preds["sam"] = r(n, 2, PRED_NUM, 256, 256)
    if case["trajectory_pred"]:
        preds["traj"] = r(n, 2, PRED_NUM, 196, 8)
    # every floating input is bf16-representable: the bf16 HIP loss kernels then see EXACTLY the values the real training loop
    # saw in fp32, and can be compared with the loop's loss values / gradients directly (one hop, tests/gpu_checks.py)
    rb = lambda t: t.to(torch.bfloat16).to(t.dtype)
    batch = {k: (rb(v) if torch.is_tensor(v) and torch.is_floating_point(v) else v) for k, v in batch.items()}
    preds = {k: (rb(v) if v.dim() > 0 else v) for k, v in preds.items()}
    return batch, preds


def sample(t, k=2048):
    flat = t.detach().flatten()
    idx = torch.linspace(0, flat.numel() - 1, min(k, flat.numel())).long()
    return dict(shape=list(t.shape), idx=idx, vals=flat[idx].clone(), l2=float(flat.norm()))


class _Loader:
    num_batches = 1

    def __init__(self, item):
        self.item = item

    def __iter__(self):
        return iter([self.item])


class _Stub(torch.nn.Module):
    """returns the preset predictions in the order of models/dreamvla_model.py:991"""

    def __init__(self, preds):
        super().__init__()
        self.p = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone()) for k, v in preds.items()})

    def forward(self, **kw):
        p = self.p
        g = lambda k: p[k] if k in p else None
        arm = p["arm"]
        grip = arm if "gripper" not in p else p["gripper"]
        return (arm, grip, g("image"), None, None, None, g("depth"), g("traj"), g("dino"), g("sam"))


class _Opt:
    param_groups = [{"lr": 0.0}]

    def step(self):
        pass

    def zero_grad(self):
        pass


class _Progress:
    """stands in for tqdm: the loop reports every loss through `t.set_postfix` (train_utils.py:734).  (Its wandb branch
    cannot be used for the trajectory cases: it reads `pred_traj_example_primary`, which upstream never defines.)"""
    last = None

    def __init__(self, it, **kw):
        self.it = it

    def __iter__(self):
        return iter(self.it)

    def set_description(self, *a, **k):
        pass

    def set_postfix(self, d):
        _Progress.last = dict(d)


def run_reference_loop(case):
    tu = ref_loader.ref_module("utils.train_utils")
    batch, preds = loss_case_tensors(case)
    S = case["S"]
    args = types.SimpleNamespace(
        action_pred_steps=STEPS, atten_goal=case.get("atten_goal", 0), batch_size=case["B"], delete_previous_checkpoint=False,
        depth_pred=case["depth_pred"], dino_feat_pred=case["dino_feat_pred"], flow_as_mask=case["flow_as_mask"],
        future_steps=FUTURE, gradient_accumulation_steps=1, gripper_width=False, loss_action=True, loss_arm_action_ratio=1.0,
        loss_depth=True, loss_dino_feat=True, loss_gripper_action_ratio=0.01, loss_image=True, loss_sam_feat=True,
        loss_trajectory=True, no_pred_gripper_traj=False, no_unshuffle=False, num_epochs=1, obs_pred=case["obs_pred"],
        patch_size=16, precision="fp32", pred_num=PRED_NUM, rank=0, report_to_wandb=False, run_name="golden",
        sam_feat_pred=case["sam_feat_pred"], save_checkpoint=False, save_checkpoint_path="/tmp", save_every_iter=10 ** 9,
        sequence_length=S, trajectory_pred=case["trajectory_pred"], use_dit_head=case["use_dit_head"], use_dpt_head=False,
        window_size=S + FUTURE, world_size=1)
    tracks = {}
    if "tracks" in batch:
        tracks = {"tracks": batch["tracks"].clone(), "tracks_gripper": batch["tracks_gripper"].clone()}
    g = lambda k: batch[k].clone() if k in batch else None
    # collator order (utils/data_utils.py:1308-1397 -> train_utils.py:99-118): 0 image, 1 text (B,77), 2 actions, 3 wrist
    # image, 4 states, 6/7 depths, 8/9 dino, 10/11 sam, 12 track dict
    states = torch.cat([batch["state"][..., :6], batch["state"][..., 6:] * 2 - 1], dim=-1)
    item = [g("image_primary"), batch["text_token"][:, 0].clone(), g("actions"), g("image_wrist"), states, None,
            g("depth_primary"), g("depth_wrist"), g("dino_primary"), g("dino_wrist"), g("sam_primary"), g("sam_wrist"), tracks]
    stub = _Stub(preds)
    grads = {}
    for k, p in stub.p.items():
        p.register_hook(lambda gr, k=k: grads.__setitem__(k, gr.detach().clone()))     # before the loop's clip_grad_norm_
    real_cuda, real_flow, real_tqdm = torch.Tensor.cuda, tu.visualize_optical_flow, tu.tqdm
    tu.tqdm = _Progress
    torch.Tensor.cuda = lambda self, *a, **k: self                                      # train_utils.py:455-456 call .cuda()
    tu.visualize_optical_flow = lambda flow, **k: np.zeros(flow.shape[:2] + (3,), np.uint8)   # logging image only (cv2)
    try:
        tu.train_one_epoch_calvin(args, stub, 0, _Loader(item), _Opt(), types.SimpleNamespace(step=lambda: None), "cpu", None)
    finally:
        torch.Tensor.cuda, tu.visualize_optical_flow, tu.tqdm = real_cuda, real_flow, real_tqdm
    keys = ("loss", "loss_arm_action", "loss_gripper_action", "loss_image", "loss_depth", "loss_dino_feat",
            "loss_sam_feat", "loss_pred_trajectory")
    return {"case": case, "losses": {k: float(_Progress.last[k]) for k in keys},
            "grads": {k: sample(v) for k, v in grads.items()}}


def main():
    assert ref_loader.available(), "needs /root/reference"
    out = {"source": "utils/train_utils.py:train_one_epoch_calvin of the REAL reference, run by oracle/make_golden_losses.py",
           "cases": {}}
    for name, case in CASES.items():
        out["cases"][name] = run_reference_loop(dict(case))
        print(name, out["cases"][name]["losses"])
    torch.save(out, os.path.join(GOLD, "losses.pt"))
    print("losses.pt", os.path.getsize(os.path.join(GOLD, "losses.pt")))


if __name__ == "__main__":
    main()

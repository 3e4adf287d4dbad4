"""Training-loss block of the reference train loop (utils/train_utils.py:98-588, utils/sigloss.py), restated as a
function so bench.py / a trainer can run forward + loss + backward without the reference's dataloader plumbing.

These reductions are the CALLER's code in the reference (plain ATen ops on the outputs of `DreamVLA.forward`).  Two
formulations live here:
  * the restated ATen formulation (any device / dtype) -- pinned against the real training loop's values and gradients
    (tests/test_losses_golden.py) and used as the checker of
  * the HIP formulation (`dvla_patch_mse_*`, `dvla_cosine_loss_*`, `dvla_silog_loss_*` of include/dvla.h): the three
    HBM-bound terms -- image MSE incl. patchify / per-patch normalisation / flow mask, the DINO / SAM cosine losses, SiLog on
    the un-patchified depth -- each as one forward and one backward kernel per camera view reading the caller's slices in
    place.  `calvin_losses` takes it for bf16 CUDA predictions (`fused=None`); smooth-L1 / BCE / trajectory MSE act on a few
    thousand elements and stay ATen.
"""
import ctypes as C
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------------
# HIP formulation
# ---------------------------------------------------------------------------------------------------------------------
def _frame_view(t):
    """t: (bs, T, *inner) view whose inner block is contiguous -> _lib.FrameView addressing it in place"""
    from ._lib import FrameView
    inner = 1
    for d in range(t.dim() - 1, 1, -1):
        if t.shape[d] != 1 and t.stride(d) != inner:
            raise ValueError("loss kernels need a contiguous per-frame block")
        inner *= t.shape[d]
    return FrameView(t.data_ptr(), t.stride(0), t.stride(1), t.shape[1])


def _fused_ok(pred, *labels):
    return (pred is not None and pred.is_cuda and pred.dtype == torch.bfloat16
            and all(lb.is_cuda and lb.dtype == torch.bfloat16 for lb in labels))


class _TwoViewLoss(torch.autograd.Function):
    """0.5 * (loss(pred[:, 0], label_primary) + loss(pred[:, 1], label_wrist)) for one of the three kernel families.
    pred: (bs*S, 2, 1, rows, cols) as DreamVLA.forward returns it; labels: (bs, T, ...) views of the window tensors."""

    @staticmethod
    def forward(ctx, pred, lab_p, lab_w, kind, bs, S, T, mask_p, mask_w, lambd):
        from . import _lib
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        if pred.dim() != 5 or pred.shape[1] != 2 or pred.shape[2] != 1:
            raise NotImplementedError("fused losses take (bs*S, 2, 1, rows, cols) predictions (pred_num == 1)")
        if not pred.is_contiguous():
            pred = pred.contiguous()
        rows, cols = pred.shape[-2], pred.shape[-1]
        p6 = pred.view(bs, S, 2, pred.shape[2], rows, cols)
        views = [p6[:, :T, v, 0] for v in (0, 1)]
        out = torch.empty(2, 2, dtype=torch.float32, device=pred.device)
        part = torch.empty(int(lib.dvla_loss_partial_len()), dtype=torch.float32, device=pred.device)
        n_frames = bs * T
        for v, (pv, lb, mk) in enumerate(zip(views, (lab_p, lab_w), (mask_p, mask_w))):
            fp, fl = _frame_view(pv), _frame_view(lb)
            o = out[v].data_ptr()
            if kind == "patch_mse":
                rc = lib.dvla_patch_mse_fwd(C.byref(fp), C.byref(fl), None if mk is None else mk.data_ptr(), n_frames, o,
                                            part.data_ptr(), stream)
            elif kind == "cosine":
                rc = lib.dvla_cosine_loss_fwd(C.byref(fp), C.byref(fl), rows, cols, n_frames, o, part.data_ptr(), stream)
            else:
                rc = lib.dvla_silog_loss_fwd(C.byref(fp), C.byref(fl), n_frames, float(lambd), o, part.data_ptr(), stream)
            _lib.check(rc, f"dvla_{kind}_fwd")
        ctx.save_for_backward(pred, lab_p, lab_w, mask_p, mask_w, out)
        ctx.args = (kind, bs, S, T, float(lambd))
        return 0.5 * (out[0, 0] + out[1, 0])

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        pred, lab_p, lab_w, mask_p, mask_w, out = ctx.saved_tensors
        kind, bs, S, T, lambd = ctx.args
        rows, cols = pred.shape[-2], pred.shape[-1]
        dpred = torch.empty_like(pred) if T == S else torch.zeros_like(pred)     # frames >= T get no loss
        p6, d6 = pred.view(bs, S, 2, pred.shape[2], rows, cols), dpred.view(bs, S, 2, pred.shape[2], rows, cols)
        gh = (g.to(torch.float32) * 0.5).reshape(1).contiguous()
        n_frames = bs * T
        for v, (lb, mk) in enumerate(zip((lab_p, lab_w), (mask_p, mask_w))):
            fp, fl, fd = _frame_view(p6[:, :T, v, 0]), _frame_view(lb), _frame_view(d6[:, :T, v, 0])
            if kind == "patch_mse":
                rc = lib.dvla_patch_mse_bwd(C.byref(fp), C.byref(fl), None if mk is None else mk.data_ptr(), n_frames,
                                            gh.data_ptr(), C.byref(fd), stream)
            elif kind == "cosine":
                rc = lib.dvla_cosine_loss_bwd(C.byref(fp), C.byref(fl), rows, cols, n_frames, gh.data_ptr(), C.byref(fd), stream)
            else:
                rc = lib.dvla_silog_loss_bwd(C.byref(fp), C.byref(fl), n_frames, lambd, out[v].data_ptr(), gh.data_ptr(),
                                             C.byref(fd), stream)
            _lib.check(rc, f"dvla_{kind}_bwd")
        return dpred, None, None, None, None, None, None, None, None, None


def two_view_loss(kind, pred, lab_p, lab_w, bs, S, T, mask_p=None, mask_w=None, lambd=0.5):
    """kind in {"patch_mse", "cosine", "silog"}; see _TwoViewLoss"""
    return _TwoViewLoss.apply(pred, lab_p, lab_w, kind, int(bs), int(S), int(T), mask_p, mask_w, float(lambd))


# ---------------------------------------------------------------------------------------------------------------------
# restated ATen formulation
# ---------------------------------------------------------------------------------------------------------------------


def patchify(imgs, patch_size):
    """(N,3,H,W) -> (N, L, p*p*3)  'nchpwq->nhwpqc'   (train_utils.py:37-50)"""
    h = w = imgs.shape[2] // patch_size
    x = imgs.reshape(imgs.shape[0], 3, h, patch_size, w, patch_size)
    x = torch.einsum('nchpwq->nhwpqc', x)
    return x.reshape(imgs.shape[0], h * w, patch_size ** 2 * 3)


def normalize_patchfied_image(p):
    """per-patch (x - mean) / sqrt(var_unbiased + 1e-6)   (train_utils.py:52-57)"""
    mean = p.mean(dim=-1, keepdim=True)
    var = p.var(dim=-1, keepdim=True)
    return (p - mean) / (var + 1.e-6) ** .5


def unpatchify(patches, patch_size=16, img_size=(224, 224)):
    """(B, P, 196, ps*ps*C) -> (B, P, C, H, W)   (train_utils.py:783-799)"""
    B, P, num_patches, patch_dim = patches.shape
    H, W = img_size
    gs = int(num_patches ** 0.5)
    C = patch_dim // (patch_size * patch_size)
    patches = patches.view(B, P, gs, gs, patch_size, patch_size, C)
    return patches.permute(0, 1, 6, 2, 4, 3, 5).contiguous().view(B, P, C, H, W)


def silog_loss(pred, target, lambd=0.5):
    """utils/sigloss.py:11-15"""
    d = torch.log(target + 1e-6) - torch.log(pred + 1e-6)
    return torch.sqrt(torch.pow(d, 2).mean() - lambd * torch.pow(d.mean(), 2))


def label_actions(actions, sequence_length, action_pred_steps, atten_goal=0):
    """train_utils.py:138,145: gripper {-1,1} -> {0,1} is done by the caller; windows of future actions."""
    return torch.cat([actions[:, j:sequence_length - atten_goal + j, :].unsqueeze(-2) for j in range(action_pred_steps)], dim=-2)


def calvin_losses(outputs, batch, *, sequence_length, future_steps=3, atten_goal=0, pred_num=1, patch_size=16,
                  use_dit_head=True, loss_arm_action_ratio=1.0, loss_gripper_action_ratio=0.01, label_action=None,
                  flow_as_mask=False, compute_dtype=torch.float32, fused=None):
    """outputs: the 10-tuple of DreamVLA.forward(mode='train'); batch: dict with window-length tensors
    (image_primary/image_wrist (B,W,3,224,224), optional depth_*/dino_*/sam_*/tracks*).  Returns (total, parts).
    The reference computes these in the model dtype; `compute_dtype=float32` (default) evaluates the reductions in
    fp32, which is at least as accurate.  `flow_as_mask` (LIBERO scripts, train_utils.py:283-333): the image loss is
    taken on the patches whose 2x2-pooled track flow exceeds 1 px (primary mask dilated 3x3, wrist mask not).
    Pinned against the real training loop's values and gradients: tests/test_losses_golden.py.
    fused: None = the HIP kernels for every term whose prediction and labels are bf16 CUDA tensors (pred_num == 1), the
    ATen formulation otherwise; False = ATen everywhere (the checker); True = HIP or raise."""
    (arm, grip, image_pred, _, _, _, depth_pred, traj_pred, dino_pred, sam_pred) = outputs
    S, T = sequence_length, sequence_length - atten_goal
    lo, hi = future_steps, future_steps + T + pred_num - 1
    bs = batch["image_primary"].shape[0]
    dev = batch["image_primary"].device
    zero = torch.zeros((), device=dev, dtype=compute_dtype)
    parts = {}
    if use_dit_head:
        parts["arm_action"], parts["gripper_action"] = arm.to(compute_dtype), zero
    else:
        parts["arm_action"] = F.smooth_l1_loss(arm[:, :T].to(compute_dtype), label_action[:, :T, :, :6].to(compute_dtype))
        parts["gripper_action"] = F.binary_cross_entropy(grip[:, :T].to(compute_dtype), label_action[:, :T, :, 6:].to(compute_dtype))
    def use_fused(pred, *keys):
        ok = pred_num == 1 and _fused_ok(pred, *(batch[k] for k in keys))
        if fused is True and pred is not None and not ok:
            raise TypeError("fused losses need bf16 CUDA predictions / labels and pred_num == 1")
        return ok and fused is not False

    def fmask(key, dilate, dt):
        t = batch[key][:, :T + pred_num - 1].to(dt)
        hw = int(t.shape[2] ** 0.5)
        tp = t.reshape(-1, hw, hw, t.shape[3]).permute(0, 3, 1, 2)                 # (B*P, 2, H, W)
        m = (torch.norm(F.avg_pool2d(tp, kernel_size=2, stride=2), dim=1) > 1.0).unsqueeze(1).to(dt)
        if dilate:
            m = F.max_pool2d(m, kernel_size=3, stride=1, padding=1)
        return m.reshape(m.shape[0], 1, -1, 1)
    parts["image"] = zero
    if image_pred is not None and use_fused(image_pred, "image_primary", "image_wrist"):
        mp = mw = None
        if flow_as_mask and "tracks" in batch:       # (bs*T, 196) {0,1} masks from the 28x28 track flow: a few KB, ATen
            mp = fmask("tracks", True, torch.float32).reshape(-1, 196).contiguous()
            mw = fmask("tracks_gripper", False, torch.float32).reshape(-1, 196).contiguous()
        parts["image"] = two_view_loss("patch_mse", image_pred, batch["image_primary"][:, lo:hi], batch["image_wrist"][:, lo:hi],
                                       bs, S, T, mp, mw)
    elif image_pred is not None:
        def lab(key):
            x = batch[key][:, lo:hi].flatten(0, 1).to(compute_dtype)
            x = normalize_patchfied_image(patchify(x, patch_size))
            x = x.view(bs, T + pred_num - 1, *x.shape[1:])
            return x.unfold(1, pred_num, 1).permute(0, 1, 4, 2, 3).flatten(0, 1)
        ip = image_pred.reshape(bs, S, *image_pred.shape[1:])[:, :T].reshape(-1, *image_pred.shape[1:]).to(compute_dtype)
        lp, lw = lab("image_primary"), lab("image_wrist")
        if flow_as_mask and "tracks" in batch:
            mp, mw = fmask("tracks", True, compute_dtype), fmask("tracks_gripper", False, compute_dtype)
            parts["image"] = 0.5 * (F.mse_loss(ip[:, 0] * mp, lp * mp) + F.mse_loss(ip[:, 1] * mw, lw * mw))
        else:
            parts["image"] = 0.5 * (F.mse_loss(ip[:, 0], lp) + F.mse_loss(ip[:, 1], lw))
    parts["depth"] = zero
    if depth_pred is not None and use_fused(depth_pred, "depth_primary", "depth_wrist"):
        parts["depth"] = two_view_loss("silog", depth_pred, batch["depth_primary"][:, lo:hi], batch["depth_wrist"][:, lo:hi],
                                       bs, S, T, lambd=0.5)
    elif depth_pred is not None:
        def dlab(key):
            return batch[key][:, lo:hi].to(compute_dtype).unfold(1, pred_num, 1).permute(0, 1, 5, 2, 3, 4).flatten(0, 1)
        dp = depth_pred.reshape(bs, S, *depth_pred.shape[1:])[:, :T].reshape(-1, *depth_pred.shape[1:]).to(compute_dtype)
        dx, dg = unpatchify(dp[:, 0], patch_size), unpatchify(dp[:, 1], patch_size)
        parts["depth"] = 0.5 * (silog_loss(dx, dlab("depth_primary")) + silog_loss(dg, dlab("depth_wrist")))

    def cos_loss(pred, key_p, key_w):
        if use_fused(pred, key_p, key_w) and pred.shape[-1] % 64 == 0 and pred.shape[-1] <= 1024:
            return two_view_loss("cosine", pred, batch[key_p][:, lo:hi], batch[key_w][:, lo:hi], bs, S, T)
        pp = pred.reshape(bs, S, *pred.shape[1:])[:, :T].reshape(-1, *pred.shape[1:]).to(compute_dtype)
        lp = batch[key_p][:, lo:hi].reshape(-1, *batch[key_p].shape[2:]).to(compute_dtype)
        lw = batch[key_w][:, lo:hi].reshape(-1, *batch[key_w].shape[2:]).to(compute_dtype)
        return 0.5 * ((1 - F.cosine_similarity(pp[:, 0, 0], lp, dim=-1)).mean()
                      + (1 - F.cosine_similarity(pp[:, 1, 0], lw, dim=-1)).mean())
    parts["dino"] = cos_loss(dino_pred, "dino_primary", "dino_wrist") if dino_pred is not None else zero
    parts["sam"] = cos_loss(sam_pred, "sam_primary", "sam_wrist") if sam_pred is not None else zero
    parts["trajectory"] = zero
    if traj_pred is not None:
        def tlab(key):
            t = batch[key][:, 0:T + pred_num - 1].to(compute_dtype)
            h = w = int(math.sqrt(t.shape[-2]))
            t = t.view(bs, t.shape[1], h, w, t.shape[-1]).permute(0, 1, 4, 2, 3)           # b p c h w
            t = F.pixel_unshuffle(t, downscale_factor=h // 14)
            t = t.flatten(3).permute(0, 1, 3, 2)                                           # b p (h w) c
            return t.unfold(1, pred_num, 1).permute(0, 1, 4, 2, 3).flatten(0, 1)
        tp = traj_pred.reshape(bs, S, *traj_pred.shape[1:])[:, :T].reshape(-1, *traj_pred.shape[1:]).to(compute_dtype)
        parts["trajectory"] = 0.1 * (F.mse_loss(tp[:, 0], tlab("tracks")) + F.mse_loss(tp[:, 1], tlab("tracks_gripper")))
    total = (loss_arm_action_ratio * parts["arm_action"] + loss_gripper_action_ratio * parts["gripper_action"]
             + 0.1 * parts["image"] + 0.001 * parts["depth"] + 0.1 * parts["trajectory"] + 0.01 * parts["dino"]
             + 0.01 * parts["sam"])                                                          # train_utils.py:585
    return total, parts

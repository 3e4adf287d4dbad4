"""Camera-frame preprocessing: the host half (`clip_image_preprocess`, the drop-in for the
`image_processor` that `clip.load` returns and eval / data code calls: utils/data_utils.py:175-178) and the device half
(`preprocess_frames`: ToTensor + Normalize + RandomShiftsAug + bf16 cast in one HIP kernel, csrc/input_pipeline.hip).

clip's `_transform(224)` (openai/CLIP clip/clip.py) is
    Resize(224, interpolation=BICUBIC) -> CenterCrop(224) -> convert("RGB") -> ToTensor() -> Normalize(CLIP_MEAN, CLIP_STD)
torchvision's Resize / CenterCrop on a PIL image are `Image.resize((w', h'), BICUBIC)` with the SHORTER side scaled to 224
(the other side truncated to int) and a crop whose offsets are round((size - 224) / 2); restated here on PIL directly
(torchvision is not a dependency of this package)."""
import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _to_pil(img):
    from PIL import Image
    if isinstance(img, Image.Image):
        return img
    if isinstance(img, torch.Tensor):
        img = img.detach().cpu().numpy()
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[0] in (1, 3) and a.shape[2] not in (1, 3):
        a = np.transpose(a, (1, 2, 0))
    if a.dtype != np.uint8:
        a = np.clip(a * (255.0 if a.max() <= 1.0 else 1.0), 0, 255).astype(np.uint8)
    return Image.fromarray(a.squeeze() if a.ndim == 3 and a.shape[2] == 1 else a)


def clip_image_resize_u8(img, n_px=224):
    """Resize(n_px, BICUBIC) + CenterCrop(n_px) + RGB -> uint8 (n_px, n_px, 3) numpy array: the part of the CLIP transform
    that stays on the host (PIL's antialiased bicubic); the rest runs in `preprocess_frames` on the device."""
    from PIL import Image
    pil = _to_pil(img)
    w, h = pil.size
    if w <= h:
        nw, nh = n_px, int(n_px * h / w)
    else:
        nw, nh = int(n_px * w / h), n_px
    if (nw, nh) != (w, h):
        pil = pil.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
    pil = pil.crop((left, top, left + n_px, top + n_px)).convert("RGB")
    return np.asarray(pil, dtype=np.uint8)


def clip_image_preprocess(img, n_px=224):
    """`image_processor(pil)` of the reference (the second return value of clip.load): -> fp32 (3, n_px, n_px)"""
    u8 = torch.from_numpy(clip_image_resize_u8(img, n_px).copy())
    x = u8.permute(2, 0, 1).float().div(255.0)                                      # ToTensor
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(3, 1, 1)
    return x.sub_(mean).div_(std)                                                   # Normalize


def draw_shifts(n, pad, traj=False, generator=None):
    """integer (sx, sy) per frame as RandomShiftsAug draws them (utils/data_utils.py:344-348 / 371-375):
    forward(): randint(0, 2 pad + 1), one pair per image; forward_traj(): randint(1, 2 pad + 1), one pair per frame."""
    return torch.randint(1 if traj else 0, 2 * pad + 1, (n, 2), generator=generator, dtype=torch.int32)


def shift_gather_reference(x, shifts, pad):
    """RandomShiftsAug as the gather it is (exact arithmetic): x (n, c, h, w) any dtype, shifts (n, 2) ints (sx, sy).
    Host-side mirror of the kernel's addressing; tests pin it against the real RandomShiftsAug module."""
    n, c, h, w = x.shape
    ys = torch.arange(h).view(1, h) + shifts[:, 1].view(n, 1).long() - pad
    xs = torch.arange(w).view(1, w) + shifts[:, 0].view(n, 1).long() - pad
    ys, xs = ys.clamp_(0, h - 1), xs.clamp_(0, w - 1)
    idx_n = torch.arange(n).view(n, 1, 1, 1)
    idx_c = torch.arange(c).view(1, c, 1, 1)
    return x[idx_n, idx_c, ys.view(n, 1, h, 1), xs.view(n, 1, 1, w)]


def preprocess_frames(frames_u8, shifts=None, pad=0, mean=CLIP_MEAN, std=CLIP_STD):
    """frames_u8: (..., H, W, 3) uint8 CUDA tensor (resized frames); shifts: (n, 2) int32 (sx, sy) or None.
    -> (..., 3, H, W) bf16 on the device: ToTensor + Normalize + RandomShiftsAug + cast, one HIP kernel (no CPU fallback)."""
    import ctypes as C
    from . import _lib
    from .ops import _stream
    lib = _lib.load()
    if not isinstance(frames_u8, torch.Tensor) or frames_u8.dtype != torch.uint8:
        raise TypeError("preprocess_frames: uint8 tensor (..., H, W, 3) expected")
    if not frames_u8.is_cuda:
        raise _lib.DvlaError(f"preprocess_frames: tensor is on {frames_u8.device}; the HIP input pipeline has no CPU fallback")
    lead, (H, W, ch) = frames_u8.shape[:-3], frames_u8.shape[-3:]
    if ch != 3:
        raise ValueError("preprocess_frames: channels-last RGB frames expected")
    src = frames_u8.reshape(-1, H, W, 3).contiguous()
    n = src.shape[0]
    sh = None
    if shifts is not None:
        sh = shifts.to(device=src.device, dtype=torch.int32).reshape(n, 2).contiguous()
    out = torch.empty((n, 3, H, W), dtype=torch.bfloat16, device=src.device)
    m3 = (C.c_float * 3)(*[float(v) for v in mean])
    s3 = (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(lib.dvla_image_preprocess(src.data_ptr(), None if sh is None else sh.data_ptr(), out.data_ptr(), n, H, W, int(pad),
                                         m3, s3, _stream()), "dvla_image_preprocess")
    return out.view(*lead, 3, H, W)

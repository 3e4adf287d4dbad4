// input_pipeline.hip -- camera frames: uint8 HWC -> ToTensor -> CLIP Normalize -> RandomShiftsAug -> bf16 CHW, one pass.
// SURVEY.md section 8 f3.
//
// The reference does this on the host, frame by frame, in the dataloader: `image_processor(pil)` = clip's _transform
// (Resize bicubic / CenterCrop / ToTensor / Normalize((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258,
// 0.27577711)); utils/data_utils.py:175-178) produces fp32 (3, 224, 224) tensors, the collater stacks them and applies
// RandomShiftsAug (utils/data_utils.py:326-383: replicate-pad by `pad`, then grid_sample on a grid shifted by an INTEGER number
// of pixels -- every sample point is a pixel centre of the padded image, i.e. a pure gather), and the training loop moves
// ~600 KB of fp32 per frame to the GPU and casts it (utils/train_utils.py:118-123).  Here the resized uint8 frame (150 KB) is
// what crosses PCIe and everything after the resize is this kernel:
//
//   out[n, c, y, x] = bf16( (float(src[n, clamp(y + sy_n - pad), clamp(x + sx_n - pad), c]) / 255 - mean[c]) / std[c] )
//
// in exactly torch's operation order and fp32 rounding (ToTensor: /255; Normalize: sub then div) followed by the model's
// bf16 cast; shifts (sx, sy) in [0, 2 pad] come from the caller (drawn on the host with torch's generator as the reference
// draws them: forward() one pair per image from randint(0, 2 pad + 1), forward_traj() one per frame from randint(1, ...)).
// HBM-bound: 3 B read + 6 B written per pixel; a thread produces 8 consecutive x of one (n, c, y) row (16-byte store).
#include "common.h"
#include "../../include/dvla.h"

namespace {

__global__ void preprocess_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ shift, bf16_t* __restrict__ out,
                                  int64_t N, int H, int W, int pad, float m0, float m1, float m2, float s0, float s1, float s2) {
  const int xo = W / 8;                       // octets per row (W % 8 == 0)
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = N * 3 * (int64_t)H * xo;
  if (idx >= total) return;
  const int ox = (int)(idx % xo);
  int64_t r = idx / xo;
  const int y = (int)(r % H); r /= H;
  const int c = (int)(r % 3);
  const int64_t n = r / 3;
  int sx = pad, sy = pad;                     // no augmentation: the identity gather
  if (shift) { sx = shift[2 * n]; sy = shift[2 * n + 1]; }
  int yy = y + sy - pad; yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), stdv = c == 0 ? s0 : (c == 1 ? s1 : s2);
  const uint8_t* row = src + ((n * H + yy) * (int64_t)W) * 3 + c;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int xx = ox * 8 + e + sx - pad; xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
    const float t = (float)row[(int64_t)xx * 3] / 255.0f;     // ToTensor
    v[e] = (t - mean) / stdv;                                  // Normalize: sub_, div_
  }
  *reinterpret_cast<uint4*>(out + ((n * 3 + c) * (int64_t)H + y) * W + ox * 8) =
      make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}

}  // namespace

extern "C" int dvla_image_preprocess(const uint8_t* src, const int32_t* shift, void* out, int64_t n, int32_t height, int32_t width,
                                     int32_t pad, const float* mean3, const float* std3, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!src || !out || !mean3 || !std3 || n < 0 || height < 1 || width < 8 || pad < 0) return DVLA_ERR_ARG;
  if (n == 0) return DVLA_OK;
  if (width % 8 != 0 || (reinterpret_cast<uintptr_t>(out) & 15)) return DVLA_ERR_UNSUPPORTED;
  const int64_t total = n * 3 * (int64_t)height * (width / 8);
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, shift,
                     reinterpret_cast<bf16_t*>(out), n, height, width, pad, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return dvla_check_launch();
}

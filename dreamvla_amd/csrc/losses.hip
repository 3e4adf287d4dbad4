// losses.hip -- the three HBM-bound reductions of the training-loss block (utils/train_utils.py:172-450,
// utils/sigloss.py), each as ONE forward kernel (+ a one-block finaliser) and ONE backward kernel instead of the caller's
// chain of patchify / mean / var / normalise / mse / log / cosine_similarity ATen ops with their fp32 temporaries:
//   patch_mse : image-prediction loss.  Label = raw frames (n,3,224,224); per 16x16x3 patch: (x - mean) / sqrt(var + 1e-6)
//               (unbiased variance), optional per-patch {0,1} flow mask, mean squared error against pred (n,196,768).
//   cosine    : 1 - cos(pred_row, label_row), mean over rows (DINO / SAM feature heads).
//   silog     : sqrt(mean d^2 - lambda mean(d)^2), d = log(t + 1e-6) - log(p + 1e-6), p = un-patchified depth prediction.
// One wave per patch / row; per-wave partial sums in a fixed order -> deterministic.  All arithmetic fp32 on bf16 data.
// Predictions and labels are addressed through (frame index -> batch, time) strides so the caller's slices
// (pred[:, view, 0], label[:, future:future+T]) are read in place.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dvla.h"
#include "common.h"

namespace {

constexpr int LOSS_THREADS = 256;   // 4 waves per workgroup, one unit (patch / row) per wave at a time
constexpr int LOSS_MAX_BLOCKS = 2048;

struct View {          // element (frame f, inner offset o) lives at base[(f / T) * stride_b + (f % T) * stride_t + o]
  const bf16_t* base;
  int64_t stride_b, stride_t;
  int T;
  __device__ __forceinline__ const bf16_t* frame(int64_t f) const { return base + (f / T) * stride_b + (f % T) * stride_t; }
};
struct ViewW {
  bf16_t* base;
  int64_t stride_b, stride_t;
  int T;
  __device__ __forceinline__ bf16_t* frame(int64_t f) const { return base + (f / T) * stride_b + (f % T) * stride_t; }
};

__device__ __forceinline__ void block_partials(float (&v)[2], float* __restrict__ partial, int k) {
  // sum the 4 waves' values (already wave-reduced) and write partial[block][0..k)
  __shared__ float red[4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = v[0]; red[wave][1] = v[1]; }
  __syncthreads();
  if (threadIdx.x < k)
    partial[(int64_t)blockIdx.x * k + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------------------
// patch MSE.  Patch element e = (py * 16 + px) * 3 + c  <->  image[c][h*16 + py][w*16 + px]   ('nchpwq->nhwpqc',
// utils/train_utils.py:37-50).  A lane owns e = lane + 64 i, i < 12.
template <bool BWD>
__global__ __launch_bounds__(LOSS_THREADS) void patch_mse_kernel(View pred, View img, const float* __restrict__ mask, int64_t n_frames,
                                                                float* __restrict__ partial, ViewW dpred, const float* __restrict__ gout,
                                                                float gscale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_patches = n_frames * 196;
  float acc = 0.f;
  float g = 0.f;
  if (BWD) g = gout[0] * gscale;
  for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < n_patches; u += (int64_t)gridDim.x * 4) {
    const int64_t f = u / 196;
    const int p = (int)(u - f * 196), h = p / 14, w = p - h * 14;
    const bf16_t* im = img.frame(f);
    float t[12];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int e = lane + 64 * i;
      const int c = e % 3, q = e / 3, px = q & 15, py = q >> 4;
      t[i] = bf2f(im[((int64_t)c * 224 + h * 16 + py) * 224 + w * 16 + px]);
      s += t[i];
    }
    const float mean = wave_sum(s) * (1.0f / 768.0f);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) { t[i] -= mean; ss += t[i] * t[i]; }
    const float rstd = rsqrtf(wave_sum(ss) * (1.0f / 767.0f) + 1.e-6f);   // torch.var: unbiased
    const float m = mask ? mask[u] : 1.0f;
    const bf16_t* pr = pred.frame(f) + (int64_t)p * 768;
    bf16_t* dp = BWD ? dpred.frame(f) + (int64_t)p * 768 : nullptr;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int e = lane + 64 * i;
      const float d = (bf2f(pr[e]) - t[i] * rstd) * m;       // pred*m - label*m
      if (BWD) dp[e] = f2bf(g * d * m);
      else acc += d * d;
    }
  }
  if (!BWD) {
    float v[2] = {wave_sum(acc), 0.f};
    block_partials(v, partial, 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// cosine: rows of C elements (C % 8 == 0, C <= 1024), read as 16-byte vectors.  C <= 256: two rows per wave (one per
// 32-lane half, reductions stay inside the half); otherwise one row per wave, vector v = lane + 64 i.
// F.cosine_similarity: x.y / sqrt(max(|x|^2 |y|^2, eps^2)), eps 1e-8
__device__ __forceinline__ float group_sum(float v, int width) {   // width 64 or 32: xor offsets below `width`
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    if (o < width) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ void unpack8l(const uint4& u, float (&f)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = bf2f((bf16_t)(w[k] & 0xffff)); f[2 * k + 1] = bf2f((bf16_t)(w[k] >> 16)); }
}
template <bool BWD>
__global__ __launch_bounds__(LOSS_THREADS) void cosine_kernel(View pred, View lab, int rows_per_frame, int C, int64_t n_frames,
                                                             float* __restrict__ partial, ViewW dpred, const float* __restrict__ gout,
                                                             float gscale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_rows = n_frames * rows_per_frame;
  const int nvec = C >> 3;                        // 16-byte vectors per row (<= 128)
  const int rpw = nvec <= 32 ? 2 : 1;             // rows per wave
  const int width = rpw == 2 ? 32 : 64;
  const int sub = rpw == 2 ? (lane >> 5) : 0, sl = rpw == 2 ? (lane & 31) : lane;
  float acc = 0.f;
  float g = 0.f;
  if (BWD) g = gout[0] * gscale;
  for (int64_t u0 = ((int64_t)blockIdx.x * 4 + wave) * rpw; u0 < n_rows; u0 += (int64_t)gridDim.x * 4 * rpw) {
    const int64_t u = u0 + sub;
    const bool live = u < n_rows;
    const int64_t uc = live ? u : n_rows - 1;
    const int64_t f = uc / rows_per_frame;
    const int r = (int)(uc - f * rows_per_frame);
    const uint4* pr = reinterpret_cast<const uint4*>(pred.frame(f) + (int64_t)r * C);
    const uint4* lr = reinterpret_cast<const uint4*>(lab.frame(f) + (int64_t)r * C);
    float x[2][8], y[2][8];
    float sxy = 0.f, sxx = 0.f, syy = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = sl + 64 * i;
      if (v < nvec) {
        unpack8l(pr[v], x[i]); unpack8l(lr[v], y[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sxy += x[i][e] * y[i][e]; sxx += x[i][e] * x[i][e]; syy += y[i][e] * y[i][e]; }
      }
    }
    sxy = group_sum(sxy, width); sxx = group_sum(sxx, width); syy = group_sum(syy, width);
    const float inv = rsqrtf(fmaxf(sxx * syy, 1.e-16f));
    const float cosv = sxy * inv;
    if (!BWD) {
      if (live && sl == 0) acc += 1.0f - cosv;    // one lane per row carries the row's value
    } else if (live) {
      // d(1 - cos)/dx = -(y / den - cos x / |x|^2)   (den clamp inactive for non-degenerate rows)
      uint4* dp = reinterpret_cast<uint4*>(dpred.frame(f) + (int64_t)r * C);
      const float inv_xx = sxx > 0.f ? 1.0f / sxx : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = sl + 64 * i;
        if (v < nvec) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = -g * (y[i][e] * inv - cosv * x[i][e] * inv_xx);
          dp[v] = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
        }
      }
    }
  }
  if (!BWD) {
    float v[2] = {wave_sum(acc), 0.f};
    block_partials(v, partial, 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// SiLog on un-patchified depth: pred (n, 196, 256), patch element e = py*16 + px  <->  depth image (n,1,224,224)
// (utils/train_utils.py:783-799 with one channel).  stats (bwd) = {loss, mean d} as the finaliser wrote them.
template <bool BWD>
__global__ __launch_bounds__(LOSS_THREADS) void silog_kernel(View pred, View dep, int64_t n_frames, float lambd, float* __restrict__ partial,
                                                            ViewW dpred, const float* __restrict__ stats, const float* __restrict__ gout,
                                                            float gscale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_patches = n_frames * 196;
  const float inv_n = 1.0f / ((float)n_frames * 196.0f * 256.0f);
  float s1 = 0.f, s2 = 0.f;
  float coef = 0.f, md = 0.f;
  if (BWD) { md = stats[1]; coef = gout[0] * gscale * inv_n / fmaxf(stats[0], 1.e-20f); }   // dL/dd_i = (d_i - lambda mean d) / (N L)
  for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < n_patches; u += (int64_t)gridDim.x * 4) {
    const int64_t f = u / 196;
    const int p = (int)(u - f * 196), h = p / 14, w = p - h * 14;
    const bf16_t* pr = pred.frame(f) + (int64_t)p * 256;
    const bf16_t* dm = dep.frame(f);
    bf16_t* dp = BWD ? dpred.frame(f) + (int64_t)p * 256 : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = lane + 64 * i, px = e & 15, py = e >> 4;
      const float t = bf2f(dm[((int64_t)(h * 16 + py)) * 224 + w * 16 + px]);
      const float pv = bf2f(pr[e]);
      const float d = __logf(t + 1.e-6f) - __logf(pv + 1.e-6f);
      if (BWD) dp[e] = f2bf(-coef * (d - lambd * md) / (pv + 1.e-6f));
      else { s1 += d; s2 += d * d; }
    }
  }
  if (!BWD) {
    float v[2] = {wave_sum(s1), wave_sum(s2)};
    block_partials(v, partial, 2);
  }
}

// out[0] = loss; out[1] = auxiliary (silog: mean d).  mode 0: sum / count; mode 1: silog from (sum d, sum d^2)
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ partial, int nblocks, int k, int mode, float count,
                                                        float lambd, float* __restrict__ out) {
  __shared__ float red[4][2];
  float s[2] = {0.f, 0.f};
  for (int i = threadIdx.x; i < nblocks; i += 256)
    for (int j = 0; j < k; ++j) s[j] += partial[(int64_t)i * k + j];
  s[0] = wave_sum(s[0]); s[1] = wave_sum(s[1]);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s[0]; red[threadIdx.x >> 6][1] = s[1]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    const float b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    if (mode == 0) { out[0] = a / count; out[1] = 0.f; }
    else {
      const float md = a / count, m2 = b / count;
      out[0] = sqrtf(fmaxf(m2 - lambd * md * md, 0.f));
      out[1] = md;
    }
  }
}

inline int blocks_for(int64_t units) {
  int64_t b = (units + 3) / 4;
  if (b > LOSS_MAX_BLOCKS) b = LOSS_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}
inline View view_of(const dvla_frame_view& v) {
  return View{reinterpret_cast<const bf16_t*>(v.base), v.stride_b, v.stride_t, v.T > 0 ? v.T : 1};
}
inline ViewW viewW_of(const dvla_frame_view& v) {
  return ViewW{reinterpret_cast<bf16_t*>(const_cast<void*>(v.base)), v.stride_b, v.stride_t, v.T > 0 ? v.T : 1};
}

}  // namespace

extern "C" int64_t dvla_loss_partial_len(void) { return (int64_t)LOSS_MAX_BLOCKS * 2; }

extern "C" int dvla_patch_mse_fwd(const dvla_frame_view* pred, const dvla_frame_view* image, const float* patch_mask,
                                  int64_t n_frames, float* out2, float* partial, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!pred || !image || !pred->base || !image->base || !out2 || !partial || n_frames <= 0) return DVLA_ERR_ARG;
  const int nb = blocks_for(n_frames * 196);
  hipLaunchKernelGGL(patch_mse_kernel<false>, dim3(nb), dim3(LOSS_THREADS), 0, stream, view_of(*pred), view_of(*image), patch_mask,
                     n_frames, partial, ViewW{nullptr, 0, 0, 1}, nullptr, 0.f);
  int rc = dvla_check_launch();
  if (rc != DVLA_OK) return rc;
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, stream, partial, nb, 1, 0, (float)((double)n_frames * 196.0 * 768.0),
                     0.f, out2);
  return dvla_check_launch();
}

extern "C" int dvla_patch_mse_bwd(const dvla_frame_view* pred, const dvla_frame_view* image, const float* patch_mask,
                                  int64_t n_frames, const float* grad_out, const dvla_frame_view* dpred, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!pred || !image || !dpred || !pred->base || !image->base || !dpred->base || !grad_out || n_frames <= 0) return DVLA_ERR_ARG;
  const int nb = blocks_for(n_frames * 196);
  hipLaunchKernelGGL(patch_mse_kernel<true>, dim3(nb), dim3(LOSS_THREADS), 0, stream, view_of(*pred), view_of(*image), patch_mask,
                     n_frames, nullptr, viewW_of(*dpred), grad_out, (float)(2.0 / ((double)n_frames * 196.0 * 768.0)));
  return dvla_check_launch();
}

extern "C" int dvla_cosine_loss_fwd(const dvla_frame_view* pred, const dvla_frame_view* label, int32_t rows_per_frame, int32_t cols,
                                    int64_t n_frames, float* out2, float* partial, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!pred || !label || !pred->base || !label->base || !out2 || !partial || n_frames <= 0 || rows_per_frame <= 0) return DVLA_ERR_ARG;
  if (cols <= 0 || cols % 8 != 0 || cols > 1024) return DVLA_ERR_UNSUPPORTED;   // 16-byte vectors; frame bases must be 16-B aligned
  const int nb = blocks_for(n_frames * rows_per_frame);
  hipLaunchKernelGGL(cosine_kernel<false>, dim3(nb), dim3(LOSS_THREADS), 0, stream, view_of(*pred), view_of(*label), rows_per_frame, cols,
                     n_frames, partial, ViewW{nullptr, 0, 0, 1}, nullptr, 0.f);
  int rc = dvla_check_launch();
  if (rc != DVLA_OK) return rc;
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, stream, partial, nb, 1, 0, (float)((double)n_frames * rows_per_frame),
                     0.f, out2);
  return dvla_check_launch();
}

extern "C" int dvla_cosine_loss_bwd(const dvla_frame_view* pred, const dvla_frame_view* label, int32_t rows_per_frame, int32_t cols,
                                    int64_t n_frames, const float* grad_out, const dvla_frame_view* dpred, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!pred || !label || !dpred || !pred->base || !label->base || !dpred->base || !grad_out || n_frames <= 0 || rows_per_frame <= 0)
    return DVLA_ERR_ARG;
  if (cols <= 0 || cols % 8 != 0 || cols > 1024) return DVLA_ERR_UNSUPPORTED;   // 16-byte vectors; frame bases must be 16-B aligned
  const int nb = blocks_for(n_frames * rows_per_frame);
  hipLaunchKernelGGL(cosine_kernel<true>, dim3(nb), dim3(LOSS_THREADS), 0, stream, view_of(*pred), view_of(*label), rows_per_frame, cols,
                     n_frames, nullptr, viewW_of(*dpred), grad_out, (float)(1.0 / ((double)n_frames * rows_per_frame)));
  return dvla_check_launch();
}

extern "C" int dvla_silog_loss_fwd(const dvla_frame_view* pred, const dvla_frame_view* depth, int64_t n_frames, float lambd, float* out2,
                                   float* partial, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!pred || !depth || !pred->base || !depth->base || !out2 || !partial || n_frames <= 0) return DVLA_ERR_ARG;
  const int nb = blocks_for(n_frames * 196);
  hipLaunchKernelGGL(silog_kernel<false>, dim3(nb), dim3(LOSS_THREADS), 0, stream, view_of(*pred), view_of(*depth), n_frames, lambd,
                     partial, ViewW{nullptr, 0, 0, 1}, nullptr, nullptr, 0.f);
  int rc = dvla_check_launch();
  if (rc != DVLA_OK) return rc;
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, stream, partial, nb, 2, 1, (float)((double)n_frames * 196.0 * 256.0),
                     lambd, out2);
  return dvla_check_launch();
}

extern "C" int dvla_silog_loss_bwd(const dvla_frame_view* pred, const dvla_frame_view* depth, int64_t n_frames, float lambd,
                                   const float* out2, const float* grad_out, const dvla_frame_view* dpred, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!pred || !depth || !dpred || !pred->base || !depth->base || !dpred->base || !out2 || !grad_out || n_frames <= 0) return DVLA_ERR_ARG;
  const int nb = blocks_for(n_frames * 196);
  hipLaunchKernelGGL(silog_kernel<true>, dim3(nb), dim3(LOSS_THREADS), 0, stream, view_of(*pred), view_of(*depth), n_frames, lambd, nullptr,
                     viewW_of(*dpred), out2, grad_out, 1.0f);
  return dvla_check_launch();
}

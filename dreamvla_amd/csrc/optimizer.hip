// optimizer.hip -- gradient-norm clipping + AdamW over flat bf16 buffers (gfx950), HBM-bound.
//
// Replaces `torch.nn.utils.clip_grad_norm_` + `torch.optim.AdamW.step` of the training step (train.py /
// utils/train_utils.py in the reference: the caller's optimizer step).  Parameters, gradients
// and both moments live in flat bf16 buffers (the gradient buffers are the data-parallel communication buckets), so
// the whole step is two kernels per bucket instead of ~80 multi-tensor launches:
//   dvla_sumsq_bf16 : sum of squares (fp32, deterministic two-stage reduction) accumulated into a device scalar
//   dvla_adamw_bf16 : clip coefficient from that scalar, then the AdamW update, all math in fp32 per element,
//                     one read-modify-write pass: 2 B (g) + 3 x (2 + 2) B (p, m, v) = 14 B / parameter.
// Element-wise semantics follow torch's fused AdamW with bf16 parameters (state in the parameter dtype) and
// clip_grad_norm_ (the gradient is rounded to bf16 after scaling, as the in-place `grad.mul_(coef)` does).
#include "common.h"
#include "../../include/dvla.h"

namespace {

constexpr int SS_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const bf16_t* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  const int64_t nvec = n >> 3;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const uint4 u = reinterpret_cast<const uint4*>(x)[i];
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = bf2f((bf16_t)(w[k] & 0xffff)), b = bf2f((bf16_t)(w[k] >> 16));
      s = fmaf(a, a, s); s = fmaf(b, b, s);
    }
  }
  if (blockIdx.x == 0)   // ragged tail (n % 8 elements)
    for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += 256) { const float a = bf2f(x[i]); s = fmaf(a, a, s); }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out,
                                                          int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3];
    out[0] = accumulate ? out[0] + t : t;
  }
}

struct AdamArgs {
  bf16_t* p; const bf16_t* g; bf16_t* m; bf16_t* v; int64_t n;
  float lr, beta1, beta2, eps, wd, inv_bc1, inv_bc2_sqrt;
  const float* sumsq; float max_norm;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
  g = bf2f(f2bf(g * coef));                        // clip_grad_norm_ scales the bf16 gradient in place
  p *= (1.0f - a.lr * a.wd);                       // decoupled weight decay
  m = m + (g - m) * (1.0f - a.beta1);              // exp_avg.lerp_(grad, 1 - beta1)
  v = a.beta2 * v + (1.0f - a.beta2) * g * g;
  const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
  p -= (a.lr * a.inv_bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  float coef = 1.0f;
  if (a.sumsq) {
    const float c = a.max_norm / (sqrtf(a.sumsq[0]) + 1e-6f);
    coef = c < 1.0f ? c : 1.0f;
  }
  const int64_t nvec = a.n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const uint4 up = reinterpret_cast<const uint4*>(a.p)[i], ug = reinterpret_cast<const uint4*>(a.g)[i];
    const uint4 um = reinterpret_cast<const uint4*>(a.m)[i], uv = reinterpret_cast<const uint4*>(a.v)[i];
    const uint32_t wp[4] = {up.x, up.y, up.z, up.w}, wg[4] = {ug.x, ug.y, ug.z, ug.w};
    const uint32_t wm[4] = {um.x, um.y, um.z, um.w}, wv[4] = {uv.x, uv.y, uv.z, uv.w};
    uint32_t op[4], om[4], ov[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float p0 = bf2f((bf16_t)(wp[k] & 0xffff)), p1 = bf2f((bf16_t)(wp[k] >> 16));
      float m0 = bf2f((bf16_t)(wm[k] & 0xffff)), m1 = bf2f((bf16_t)(wm[k] >> 16));
      float v0 = bf2f((bf16_t)(wv[k] & 0xffff)), v1 = bf2f((bf16_t)(wv[k] >> 16));
      adam_one(p0, bf2f((bf16_t)(wg[k] & 0xffff)), m0, v0, a, coef);
      adam_one(p1, bf2f((bf16_t)(wg[k] >> 16)), m1, v1, a, coef);
      op[k] = pack2bf(p0, p1); om[k] = pack2bf(m0, m1); ov[k] = pack2bf(v0, v1);
    }
    reinterpret_cast<uint4*>(a.p)[i] = make_uint4(op[0], op[1], op[2], op[3]);
    reinterpret_cast<uint4*>(a.m)[i] = make_uint4(om[0], om[1], om[2], om[3]);
    reinterpret_cast<uint4*>(a.v)[i] = make_uint4(ov[0], ov[1], ov[2], ov[3]);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (nvec << 3) + threadIdx.x; i < a.n; i += 256) {
      float p = bf2f(a.p[i]), m = bf2f(a.m[i]), v = bf2f(a.v[i]);
      adam_one(p, bf2f(a.g[i]), m, v, a, coef);
      a.p[i] = f2bf(p); a.m[i] = f2bf(m); a.v[i] = f2bf(v);
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int64_t dvla_sumsq_partial_len(void) { return SS_BLOCKS; }

extern "C" int dvla_sumsq_bf16(const void* x, int64_t n, float* partial, float* out, int32_t accumulate, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!x || !partial || !out || n < 0) return DVLA_ERR_ARG;
  if (!al16(x)) return DVLA_ERR_UNSUPPORTED;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > SS_BLOCKS) blocks = SS_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const bf16_t*>(x), n, partial);
  int rc = dvla_check_launch();
  if (rc != DVLA_OK) return rc;
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, partial, (int)blocks, out, (int)accumulate);
  return dvla_check_launch();
}

extern "C" int dvla_adamw_bf16(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int64_t step, const float* grad_sumsq, float max_norm,
                               void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return DVLA_ERR_ARG;
  if (n == 0) return DVLA_OK;
  if (!al16(param) || !al16(grad) || !al16(exp_avg) || !al16(exp_avg_sq)) return DVLA_ERR_UNSUPPORTED;
  AdamArgs a;
  a.p = reinterpret_cast<bf16_t*>(param); a.g = reinterpret_cast<const bf16_t*>(grad);
  a.m = reinterpret_cast<bf16_t*>(exp_avg); a.v = reinterpret_cast<bf16_t*>(exp_avg_sq); a.n = n;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  a.inv_bc1 = (float)(1.0 / bc1); a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  a.sumsq = grad_sumsq; a.max_norm = max_norm;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  return dvla_check_launch();
}

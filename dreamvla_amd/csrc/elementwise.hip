// elementwise.hip -- small HBM-bound helpers (bias-gradient column sums, dropout, activation backward,
// casts, broadcast add).  All bf16 traffic is 16-byte vectorised where alignment allows.
#include "common.h"
#include "../../include/dvla.h"

namespace {

constexpr int CS_BLOCKS = 512;  // row slabs (stage-1 partial rows): >= 4 workgroups per CU even for 512-column inputs

// stage 1: block = 4 waves over one 512-column strip of one row slab.  Wave w takes rows r_begin + w, + 4, ...; a lane
// owns 8 consecutive columns (16-B loads, a wave reads 1 KiB contiguous per row).  The 4 waves are combined through LDS.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ x, int64_t ld, int64_t rows, int64_t cols,
                                                             float* __restrict__ partial, int vec_ok) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t slab = blockIdx.y, nslab = gridDim.y;
  const int64_t r_begin = rows * slab / nslab, r_end = rows * (slab + 1) / nslab;
  const int64_t c0 = (int64_t)blockIdx.x * 512 + lane * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < cols) {
    if (vec_ok && c0 + 8 <= cols) {
#pragma unroll 8
      for (int64_t r = r_begin + wave; r < r_end; r += 4) {   // unrolled: eight independent 16-B loads in flight per lane
        const uint4 u = *reinterpret_cast<const uint4*>(x + r * ld + c0);
        s[0] += bf2f((bf16_t)(u.x & 0xffff)); s[1] += bf2f((bf16_t)(u.x >> 16));
        s[2] += bf2f((bf16_t)(u.y & 0xffff)); s[3] += bf2f((bf16_t)(u.y >> 16));
        s[4] += bf2f((bf16_t)(u.z & 0xffff)); s[5] += bf2f((bf16_t)(u.z >> 16));
        s[6] += bf2f((bf16_t)(u.w & 0xffff)); s[7] += bf2f((bf16_t)(u.w >> 16));
      }
    } else {
      for (int64_t r = r_begin + wave; r < r_end; r += 4)
        for (int e = 0; e < 8; ++e)
          if (c0 + e < cols) s[e] += bf2f(x[r * ld + c0 + e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = s[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int64_t c = (int64_t)blockIdx.x * 512 + i;
    if (c < cols) partial[slab * cols + c] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}

// stage 2 (also used for the LayerNorm parameter gradients): out[c] = sum_k partial[k*stride + c]; block = 64 columns x 4
// interleaved slab subsets, combined through LDS (fixed order -> deterministic).
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ partial, void* __restrict__ out, int out_bf16,
                                                            int nslab, int64_t cols, int64_t stride) {
  // 16 columns x 16 row-parts per workgroup: a thread adds nslab / 16 partial rows, then a 16-way LDS reduction
  __shared__ float red[16][16];
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int64_t c = (int64_t)blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < cols)
#pragma unroll 4
    for (int k = part; k < nslab; k += 16) s += partial[(int64_t)k * stride + c];
  red[part][cl] = s;
  __syncthreads();
  if (part == 0 && c < cols) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][cl];
    if (out_bf16) reinterpret_cast<bf16_t*>(out)[c] = f2bf(tot);   // the bias gradient in the parameter's dtype, no cast kernel
    else reinterpret_cast<float*>(out)[c] = tot;
  }
}

__global__ void dropout_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t rows, int64_t cols,
                               uint32_t thr, float scale, uint32_t seed_lo, uint32_t seed_hi) {
  const int64_t total = rows * cols;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < total; i += (int64_t)gridDim.x * blockDim.x * 2) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t idx = i + e;
      if (idx < total) {
        const int64_t r = idx / cols, c = idx % cols;
        const uint32_t h = drop_hash_rk(drop_rowkey(seed_lo, seed_hi, (uint32_t)r), (uint32_t)c);
        y[idx] = (h >= thr) ? f2bf(bf2f(x[idx]) * scale) : (bf16_t)0;
      }
    }
  }
}

__global__ void act_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ pre, bf16_t* __restrict__ dz,
                               int64_t rows, int64_t cols, int act, int has_drop, uint32_t thr, float scale,
                               uint32_t seed_lo, uint32_t seed_hi) {
  const int64_t total = rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    float g = bf2f(dy[idx]);
    if (has_drop) {
      const int64_t r = idx / cols, c = idx % cols;
      const uint32_t h = drop_hash_rk(drop_rowkey(seed_lo, seed_hi, (uint32_t)r), (uint32_t)c);
      g = (h >= thr) ? g * scale : 0.f;
    }
    if (pre) g *= act_bwd(bf2f(pre[idx]), act);
    dz[idx] = f2bf(g);
  }
}

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
  f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
  f[4] = bf2f((bf16_t)(u.z & 0xffff)); f[5] = bf2f((bf16_t)(u.z >> 16));
  f[6] = bf2f((bf16_t)(u.w & 0xffff)); f[7] = bf2f((bf16_t)(u.w >> 16));
}

// dz = dropout'(dy) * act'(preact) AND the column sums of dz in the same pass (dvla_act_bwd_colsum): the gradient of a
// `y = dropout(act(x W + b))` branch is needed both as the operand of the two backward GEMMs and, summed over the rows, as db.
// Round 3 summed db from the fragments of the weight-gradient GEMM (ring kernels only); the phase kernel cannot carry those
// sums, so its launches paid a separate column-sum pass over dz -- which this elementwise pass has in registers anyway.
// Same strip decomposition as colsum_partial_kernel: block = 4 waves over a 512-column strip of one row slab, a lane owns 8
// consecutive columns (16-byte loads and stores), the sums are taken of the ROUNDED dz values (what a column sum over the stored
// tensor gives) and combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ pre,
                                                             bf16_t* __restrict__ dz, int64_t rows, int64_t cols, int act, int has_drop,
                                                             uint32_t thr, float scale, uint32_t seed_lo, uint32_t seed_hi,
                                                             float* __restrict__ partial) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t slab = blockIdx.y, nslab = gridDim.y;
  const int64_t r_begin = rows * slab / nslab, r_end = rows * (slab + 1) / nslab;
  const int64_t c0 = (int64_t)blockIdx.x * 512 + lane * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < cols) {       // cols % 8 == 0 (host check): the octet is whole
#pragma unroll 4
    for (int64_t r = r_begin + wave; r < r_end; r += 4) {
      const uint4 u = *reinterpret_cast<const uint4*>(dy + r * cols + c0);
      float g[8];
      unpack8(u, g);
      if (has_drop) {
        const uint32_t rowkey = drop_rowkey(seed_lo, seed_hi, (uint32_t)r);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = (drop_hash_rk(rowkey, (uint32_t)(c0 + e)) >= thr) ? g[e] * scale : 0.f;
      }
      if (pre) {
        float a[8];
        unpack8(*reinterpret_cast<const uint4*>(pre + r * cols + c0), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= act_bwd(a[e], act);
      }
      const uint4 o = make_uint4(pack2bf(g[0], g[1]), pack2bf(g[2], g[3]), pack2bf(g[4], g[5]), pack2bf(g[6], g[7]));
      *reinterpret_cast<uint4*>(dz + r * cols + c0) = o;
      float z[8];
      unpack8(o, z);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += z[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = s[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int64_t c = (int64_t)blockIdx.x * 512 + i;
    if (c < cols) partial[slab * cols + c] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}

__global__ void act_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = f2bf(act_fwd(bf2f(x[i]), act));
}

__global__ void cast_f2b_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = f2bf(s[i]);
}
__global__ void cast_b2f_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = bf2f(s[i]);
}
__global__ void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o, int64_t n,
                           int64_t period) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    o[i] = f2bf(bf2f(a[i]) + bf2f(b[period > 0 ? (i % period) : i]));
}

inline unsigned grid_for(int64_t n, int per_thread = 1) {
  int64_t b = (n + 256LL * per_thread - 1) / (256LL * per_thread);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}
inline uint32_t thr_of(float p) {
  double thr = (double)p * 4294967296.0;
  return thr >= 4294967295.0 ? 4294967295u : (uint32_t)thr;
}

// ---- token assembly (models/dreamvla_model.py:739-759): out[b, s, t, :] = part_k[b, s, t - t0_k, :] + pos[s, :].
// The parts are the per-frame conditioning tokens and the learned query tokens (broadcast: zero batch / time strides); one
// gather-write pass replaces torch.cat + the position add.  One 16-byte vector per thread.
struct AsmSrc { const bf16_t* p; int64_t sb, ss; int t0; };
struct AsmArgs { AsmSrc src[DVLA_MAX_TOKEN_SRCS]; int n_src; bf16_t* out; const bf16_t* pos; int64_t pos_ss; int B, S, T, H; };
template <class IDX>   // IDX = uint32_t when the vector count fits (64-bit div / mod per 16 bytes would pace the copy)
__global__ __launch_bounds__(256) void assemble_kernel(AsmArgs a) {
  const IDX hv = (IDX)(a.H >> 3);
  const IDX total = (IDX)a.B * (IDX)a.S * (IDX)a.T * hv;
  for (IDX i = (IDX)blockIdx.x * 256 + threadIdx.x; i < total; i += (IDX)gridDim.x * 256) {
    const IDX row = i / hv;
    const int v = (int)(i - row * hv);
    const IDX bs = row / (IDX)a.T;
    const int t = (int)(row - bs * (IDX)a.T);
    const IDX b = bs / (IDX)a.S;
    const int s = (int)(bs - b * (IDX)a.S);
    int k = 0;
    while (k + 1 < a.n_src && t >= a.src[k + 1].t0) ++k;      // sources are sorted by first token
    const bf16_t* sp = a.src[k].p + (int64_t)b * a.src[k].sb + (int64_t)s * a.src[k].ss + (int64_t)(t - a.src[k].t0) * a.H + v * 8;
    const uint4 x = *reinterpret_cast<const uint4*>(sp);
    uint4 o = x;
    if (a.pos) {
      const uint4 z = *reinterpret_cast<const uint4*>(a.pos + (int64_t)s * a.pos_ss + v * 8);
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, zs[4] = {z.x, z.y, z.z, z.w};
      uint32_t r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        r[q] = pack2bf(bf2f((bf16_t)(xs[q] & 0xffff)) + bf2f((bf16_t)(zs[q] & 0xffff)),
                       bf2f((bf16_t)(xs[q] >> 16)) + bf2f((bf16_t)(zs[q] >> 16)));
      o = make_uint4(r[0], r[1], r[2], r[3]);
    }
    *reinterpret_cast<uint4*>(a.out + (int64_t)row * a.H + v * 8) = o;
  }
}

}  // namespace

extern "C" int dvla_abi_version(void) { return DVLA_ABI_VERSION; }
extern "C" int64_t dvla_colsum_partial_rows(void) { return CS_BLOCKS; }

// out[c] = sum of `nrows` fp32 partial rows (row stride `stride`): stage 2 alone -- the k-sum partials of a GEMM (gemm.hip)
int dvla_reduce_partial_rows(const float* partial, int nrows, int64_t cols, int64_t stride, void* out, int out_bf16, hipStream_t stream) {
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, stream, partial, out, out_bf16, nrows, cols, stride);
  return dvla_check_launch();
}

extern "C" int dvla_colsum_dt(const void* x, int64_t ld, int64_t rows, int64_t cols, void* out, int32_t out_dtype,
                              float* partial, void* stream_);
extern "C" int dvla_colsum(const void* x, int64_t ld, int64_t rows, int64_t cols, float* out, float* partial, void* stream_) {
  return dvla_colsum_dt(x, ld, rows, cols, out, DVLA_DT_F32, partial, stream_);
}

extern "C" int dvla_colsum_dt(const void* x, int64_t ld, int64_t rows, int64_t cols, void* out, int32_t out_dtype,
                              float* partial, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!x || !out || !partial || rows < 0 || cols <= 0) return DVLA_ERR_ARG;
  if (out_dtype != DVLA_DT_F32 && out_dtype != DVLA_DT_BF16) return DVLA_ERR_ARG;
  const int64_t colblocks = (cols + 511) / 512;
  int64_t nslab = rows / 16;
  if (nslab > CS_BLOCKS) nslab = CS_BLOCKS;
  if (nslab < 1) nslab = 1;
  const int vec_ok = (ld % 8 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)colblocks, (unsigned)nslab), dim3(256), 0, stream,
                     reinterpret_cast<const bf16_t*>(x), ld, rows, cols, partial, vec_ok);
  int rc = dvla_check_launch();
  if (rc != DVLA_OK) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, stream, partial, out,
                     out_dtype == DVLA_DT_BF16 ? 1 : 0, (int)nslab, cols, cols);
  return dvla_check_launch();
}

extern "C" int dvla_dropout(const void* x, void* y, int64_t rows, int64_t cols, float p, uint32_t seed_lo, uint32_t seed_hi,
                            void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!x || !y || rows < 0 || cols <= 0 || p < 0.f || p >= 1.f) return DVLA_ERR_ARG;
  if (rows == 0) return DVLA_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(rows * cols, 2)), dim3(256), 0, stream,
                     reinterpret_cast<const bf16_t*>(x), reinterpret_cast<bf16_t*>(y), rows, cols, thr_of(p),
                     1.0f / (1.0f - p), seed_lo, seed_hi);
  return dvla_check_launch();
}

extern "C" int dvla_act_bwd(const void* dy, const void* preact, void* dz, int64_t rows, int64_t cols, int32_t act,
                            float dropout_p, uint32_t seed_lo, uint32_t seed_hi, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!dy || !dz || rows < 0 || cols <= 0 || dropout_p < 0.f || dropout_p >= 1.f) return DVLA_ERR_ARG;
  if (rows == 0) return DVLA_OK;
  const int has_drop = dropout_p > 0.f;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, stream,
                     reinterpret_cast<const bf16_t*>(dy), reinterpret_cast<const bf16_t*>(preact),
                     reinterpret_cast<bf16_t*>(dz), rows, cols, act, has_drop, thr_of(dropout_p),
                     has_drop ? 1.0f / (1.0f - dropout_p) : 1.0f, seed_lo, seed_hi);
  return dvla_check_launch();
}

extern "C" int dvla_act_bwd_colsum(const void* dy, const void* preact, void* dz, int64_t rows, int64_t cols, int32_t act,
                                   float dropout_p, uint32_t seed_lo, uint32_t seed_hi, void* colsum, int32_t colsum_dtype,
                                   float* partial, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!dy || !dz || !colsum || !partial || rows < 0 || cols <= 0 || dropout_p < 0.f || dropout_p >= 1.f) return DVLA_ERR_ARG;
  if (colsum_dtype != DVLA_DT_F32 && colsum_dtype != DVLA_DT_BF16) return DVLA_ERR_ARG;
  auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
  if (cols % 8 != 0 || !al16(dy) || !al16(dz) || (preact && !al16(preact))) return DVLA_ERR_UNSUPPORTED;   // 16-byte octets
  const int has_drop = dropout_p > 0.f;
  const int64_t colblocks = (cols + 511) / 512;
  int64_t nslab = rows / 16;
  if (nslab > CS_BLOCKS) nslab = CS_BLOCKS;
  if (nslab < 1) nslab = 1;
  hipLaunchKernelGGL(act_bwd_colsum_kernel, dim3((unsigned)colblocks, (unsigned)nslab), dim3(256), 0, stream,
                     reinterpret_cast<const bf16_t*>(dy), reinterpret_cast<const bf16_t*>(preact), reinterpret_cast<bf16_t*>(dz), rows,
                     cols, act, has_drop, thr_of(dropout_p), has_drop ? 1.0f / (1.0f - dropout_p) : 1.0f, seed_lo, seed_hi, partial);
  int rc = dvla_check_launch();
  if (rc != DVLA_OK) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(256), 0, stream, partial, colsum,
                     colsum_dtype == DVLA_DT_BF16 ? 1 : 0, (int)nslab, cols, cols);
  return dvla_check_launch();
}

// classifier-free guidance + one eta = 0 DDIM update (dvla.h): every intermediate is rounded where the tensor expression it
// replaces rounds it (bf16 for the guidance arithmetic on the model's bf16 output, fp32 -- without contraction -- for the update)
__global__ void ddim_cfg_step_kernel(const bf16_t* __restrict__ mo, int64_t sample_stride, const float* __restrict__ x,
                                     float* __restrict__ xn, int64_t bs, int64_t per, float cfg, float a, float b, float sp, float sq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs * per) return;
  const int64_t s = i / per, j = i - s * per;
  const float cond = bf2f(mo[s * sample_stride + j]), unc = bf2f(mo[(s + bs) * sample_stride + j]);
  const float d = bf2f(f2bf(__fsub_rn(cond, unc)));
  const float sd = bf2f(f2bf(__fmul_rn(cfg, d)));
  const float e = bf2f(f2bf(__fadd_rn(unc, sd)));
  const float ax = __fmul_rn(a, x[i]);
  const float px = __fsub_rn(ax, __fmul_rn(b, e));
  const float e2 = __fdiv_rn(__fsub_rn(ax, px), b);
  xn[i] = __fadd_rn(__fmul_rn(px, sp), __fmul_rn(sq, e2));
}

extern "C" int dvla_ddim_cfg_step(const void* model_out, int64_t sample_stride, const float* x, float* x_next, int64_t bs,
                                  int64_t per_sample, float cfg_scale, float a, float b, float sqrt_acp_prev, float sqrt_1m_acp_prev,
                                  void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!model_out || !x || !x_next || bs < 0 || per_sample <= 0 || sample_stride < per_sample || !(b != 0.f)) return DVLA_ERR_ARG;
  if (bs == 0) return DVLA_OK;
  const int64_t n = bs * per_sample;
  hipLaunchKernelGGL(ddim_cfg_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const bf16_t*>(model_out), sample_stride, x, x_next, bs, per_sample, cfg_scale, a, b,
                     sqrt_acp_prev, sqrt_1m_acp_prev);
  return dvla_check_launch();
}

extern "C" int dvla_act_fwd(const void* x, void* y, int64_t n, int32_t act, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!x || !y || n < 0) return DVLA_ERR_ARG;
  if (n == 0) return DVLA_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, reinterpret_cast<const bf16_t*>(x),
                     reinterpret_cast<bf16_t*>(y), n, act);
  return dvla_check_launch();
}

extern "C" int dvla_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!src || !dst || n < 0) return DVLA_ERR_ARG;
  if (n == 0) return DVLA_OK;
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, reinterpret_cast<bf16_t*>(dst), n);
  return dvla_check_launch();
}
extern "C" int dvla_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!src || !dst || n < 0) return DVLA_ERR_ARG;
  if (n == 0) return DVLA_OK;
  hipLaunchKernelGGL(cast_b2f_kernel, dim3(grid_for(n)), dim3(256), 0, stream, reinterpret_cast<const bf16_t*>(src), dst, n);
  return dvla_check_launch();
}
extern "C" int dvla_assemble_tokens(const dvla_token_src* srcs, int32_t n_src, const void* pos, int64_t pos_stride_s, void* out,
                                    int32_t B, int32_t S, int32_t T, int32_t H, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!srcs || !out || n_src < 1 || n_src > DVLA_MAX_TOKEN_SRCS || B < 0 || S < 1 || T < 1 || H < 8) return DVLA_ERR_ARG;
  if (B == 0) return DVLA_OK;
  if (H % 8 != 0 || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(pos) & 15) || (pos && pos_stride_s % 8 != 0))
    return DVLA_ERR_UNSUPPORTED;
  AsmArgs a;
  a.n_src = n_src;
  int next = 0;
  for (int k = 0; k < n_src; ++k) {
    const dvla_token_src& q = srcs[k];
    if (!q.base || q.tok_count < 1 || q.tok_begin != next) return DVLA_ERR_ARG;      // contiguous cover of [0, T), in order
    if ((reinterpret_cast<uintptr_t>(q.base) & 15) || q.stride_b % 8 != 0 || q.stride_s % 8 != 0) return DVLA_ERR_UNSUPPORTED;
    a.src[k] = AsmSrc{reinterpret_cast<const bf16_t*>(q.base), q.stride_b, q.stride_s, q.tok_begin};
    next += q.tok_count;
  }
  if (next != T) return DVLA_ERR_ARG;
  a.out = reinterpret_cast<bf16_t*>(out);
  a.pos = reinterpret_cast<const bf16_t*>(pos);
  a.pos_ss = pos_stride_s;
  a.B = B; a.S = S; a.T = T; a.H = H;
  const int64_t vecs = (int64_t)B * S * T * (H / 8);
  if (vecs + 256LL * 4096 < (1LL << 32))
    hipLaunchKernelGGL(assemble_kernel<uint32_t>, dim3(grid_for(vecs)), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(assemble_kernel<uint64_t>, dim3(grid_for(vecs)), dim3(256), 0, stream, a);
  return dvla_check_launch();
}

extern "C" int dvla_add(const void* a, const void* b, void* out, int64_t n, int64_t b_period, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!a || !b || !out || n < 0) return DVLA_ERR_ARG;
  if (n == 0) return DVLA_OK;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, stream, reinterpret_cast<const bf16_t*>(a),
                     reinterpret_cast<const bf16_t*>(b), reinterpret_cast<bf16_t*>(out), n, b_period);
  return dvla_check_launch();
}

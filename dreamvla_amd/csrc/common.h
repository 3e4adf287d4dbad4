// Shared device helpers for the DreamVLA CDNA4 (gfx950) kernels.
// Everything here is wave64 / MFMA specific; there is no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DVLA_OK 0
#define DVLA_ERR_ARG (-1)
#define DVLA_ERR_LAUNCH (-2)
#define DVLA_ERR_UNSUPPORTED (-3)

enum DvlaAct { ACT_NONE = 0, ACT_GELU_ERF = 1, ACT_GELU_TANH = 2, ACT_RELU = 3, ACT_SILU = 4, ACT_QUICK_GELU = 5,
               ACT_TANH = 6, ACT_SIGMOID = 7 };

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even float -> bf16 (same rounding as torch's float->bfloat16 cast); both forms lower to
// the gfx950 hardware converter v_cvt_pk_bf16_f32 (one instruction per PAIR of values).
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float hw_f32x2;
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  const hw_bf16x2 b = __builtin_convertvector(v, hw_bf16x2);
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // raw v_exp_f32

// ---- fast transcendental helpers (epilogue-resident: they must cost a handful of VALU ops, not a libm call) ----
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }            // v_rcp_f32 (1 ulp)
__device__ __forceinline__ float fast_exp(float x) { return fast_exp2(x * 1.4426950408889634f); }  // e^x
// erf by Abramowitz-Stegun 7.1.26: |error| <= 1.5e-7 (three orders below bf16 resolution)
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = fast_rcp(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float r = 1.0f - poly * t * fast_exp(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = fast_exp(-2.0f * fabsf(x));
  return copysignf((1.0f - e) * fast_rcp(1.0f + e), x);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case ACT_GELU_ERF: return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f));
    case ACT_GELU_TANH: {
      const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
      return 0.5f * x * (1.0f + fast_tanh(u));
    }
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_SILU: return x * fast_sigmoid(x);
    case ACT_QUICK_GELU: return x * fast_sigmoid(1.702f * x);
    case ACT_TANH: return fast_tanh(x);
    case ACT_SIGMOID: return fast_sigmoid(x);
    default: return x;
  }
}

// d act(x) / dx evaluated at the pre-activation x
__device__ __forceinline__ float act_bwd(float x, int act) {
  switch (act) {
    case ACT_GELU_ERF: {
      const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f));
      const float pdf = 0.3989422804014327f * fast_exp(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case ACT_GELU_TANH: {
      const float x2 = x * x;
      const float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
      const float t = fast_tanh(u);
      const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x2);
      return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
    }
    case ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case ACT_SILU: {
      const float s = fast_sigmoid(x);
      return s * (1.0f + x * (1.0f - s));
    }
    case ACT_QUICK_GELU: {
      const float s = fast_sigmoid(1.702f * x);
      return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    case ACT_TANH: { const float t = fast_tanh(x); return 1.0f - t * t; }
    case ACT_SIGMOID: { const float s = fast_sigmoid(x); return s * (1.0f - s); }
    default: return 1.0f;
  }
}

// ---- counter-based dropout RNG (stateless; forward and backward recompute the same mask) ----------
// keep(element) <=> drop_hash(seed, idx_hi, idx_lo) >= thr,  thr = floor(p * 2^32)
// elementwise tensors: idx_hi = row, idx_lo = col.  attention probs: idx_hi = (b*H+h)*Lq + q, idx_lo = key.
// oracle/torch_ref.py::drop_keep_mask restates this bit for bit.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_rowkey(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx_hi) {
  return hash32(idx_hi ^ seed_hi) + seed_lo;
}
__device__ __forceinline__ uint32_t drop_hash_rk(uint32_t rowkey, uint32_t idx_lo) {
  return hash32(rowkey + idx_lo * 0x9E3779B9u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// load 8 consecutive bf16 along the contiguous ("inner") dim of a 2-D operand, zero filled outside
// [outer_lim) x [inner_lim).  vec_ok = (ld % 8 == 0 && base 16-B aligned && inner offset % 8 == 0).
__device__ __forceinline__ uint4 load8_guard(const bf16_t* __restrict__ base, int64_t ld, int64_t outer,
                                              int64_t inner, int64_t outer_lim, int64_t inner_lim, bool vec_ok) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (outer < outer_lim && inner < inner_lim) {
    const bf16_t* p = base + outer * ld + inner;
    if (vec_ok && inner + 8 <= inner_lim) {
      v = *reinterpret_cast<const uint4*>(p);
    } else {
      uint32_t w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t lo = (inner + 2 * i < inner_lim) ? (uint32_t)p[2 * i] : 0u;
        uint32_t hi = (inner + 2 * i + 1 < inner_lim) ? (uint32_t)p[2 * i + 1] : 0u;
        w[i] = lo | (hi << 16);
      }
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  return v;
}

static inline int dvla_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DVLA_OK : DVLA_ERR_LAUNCH;
}

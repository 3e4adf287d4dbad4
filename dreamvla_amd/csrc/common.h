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
// Activations.  The GEMM epilogue runs them on PAIRS (f32x2) so that the polynomial parts issue as packed fp32
// (v_pk_fma_f32 / v_pk_mul_f32: two values per lane and instruction); the scalar forms wrap the pair forms.
//   Phi(x) = 0.5 (1 + erf(x / sqrt 2)) by Abramowitz-Stegun 7.1.26 in its erfc form:
//     z = |x| / sqrt 2, t = 1 / (1 + p z), q = (a1 t + ... + a5 t^5) e^{-z^2}  (= erfc z, |abs error| <= 1.5e-7)
//     Phi(x) = x >= 0 ? 1 - q/2 : q/2          -- the negative tail keeps its RELATIVE accuracy (no 1 - (1 - q))
//   gelu_erf(x) = x Phi(x);  gelu_erf'(x) = Phi(x) + x phi(x) with phi(x) = e^{-z^2} / sqrt(2 pi): one exp serves both.
//   gelu_tanh(x) = 0.5 x (1 + tanh u) = x sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3): one exp + one rcp.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 rcp2(f32x2 v) { return f32x2{fast_rcp(v.x), fast_rcp(v.y)}; }
__device__ __forceinline__ f32x2 exp2_2(f32x2 v) { return f32x2{fast_exp2(v.x), fast_exp2(v.y)}; }
__device__ __forceinline__ f32x2 sel_nonneg(f32x2 x, f32x2 a, f32x2 b) {   // x >= 0 ? a : b, per component
  return f32x2{x.x >= 0.f ? a.x : b.x, x.y >= 0.f ? a.y : b.y};
}
// q = erfc(|x| / sqrt 2) and e = exp(-x^2 / 2)
__device__ __forceinline__ void erfc_parts(f32x2 x, f32x2& q, f32x2& e) {
  const f32x2 z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const f32x2 t = rcp2(z * 0.3275911f + 1.0f);
  f32x2 poly = t * 1.061405429f + (-1.453152027f);
  poly = poly * t + 1.421413741f;
  poly = poly * t + (-0.284496736f);
  poly = poly * t + 0.254829592f;
  e = exp2_2(x * x * (-0.5f * 1.4426950408889634f));
  q = poly * t * e;
}
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) { return rcp2(exp2_2(x * (-1.4426950408889634f)) + 1.0f); }
// sigmoid of an argument that already carries the factor -log2(e): 1 / (1 + 2^t).  tanh-GELU folds that factor into its two
// polynomial constants -- one packed multiply less per pair in an epilogue that is VALU-bound (round 5)
__device__ __forceinline__ f32x2 sigmoid2_scaled(f32x2 t) { return rcp2(exp2_2(t) + 1.0f); }
constexpr float GELU_T_K1 = -1.4426950408889634f * 2.0f * 0.7978845608028654f * 0.044715f;
constexpr float GELU_T_K2 = -1.4426950408889634f * 2.0f * 0.7978845608028654f;
__device__ __forceinline__ f32x2 tanh2(f32x2 x) { return sigmoid2(x + x) * 2.0f - 1.0f; }

__device__ __forceinline__ f32x2 act_fwd2(f32x2 x, int act) {
  switch (act) {
    case ACT_GELU_ERF: {
      f32x2 q, e;
      erfc_parts(x, q, e);
      const f32x2 h = x * 0.5f * q;
      return sel_nonneg(x, x - h, h);
    }
    case ACT_GELU_TANH: {
      return x * sigmoid2_scaled(x * (x * x * GELU_T_K1 + GELU_T_K2));
    }
    case ACT_RELU: return f32x2{x.x > 0.f ? x.x : 0.f, x.y > 0.f ? x.y : 0.f};
    case ACT_SILU: return x * sigmoid2(x);
    case ACT_QUICK_GELU: return x * sigmoid2(x * 1.702f);
    case ACT_TANH: return tanh2(x);
    case ACT_SIGMOID: return sigmoid2(x);
    default: return x;
  }
}

// d act(x) / dx evaluated at the pre-activation x
__device__ __forceinline__ f32x2 act_bwd2(f32x2 x, int act) {
  switch (act) {
    case ACT_GELU_ERF: {
      f32x2 q, e;
      erfc_parts(x, q, e);
      const f32x2 hq = q * 0.5f;
      return sel_nonneg(x, 1.0f - hq, hq) + x * e * 0.3989422804014327f;
    }
    case ACT_GELU_TANH: {
      const f32x2 x2 = x * x;
      const f32x2 s = sigmoid2_scaled(x * (x2 * GELU_T_K1 + GELU_T_K2));
      const f32x2 du2 = x2 * (6.0f * 0.7978845608028654f * 0.044715f) + 2.0f * 0.7978845608028654f;   // d(2u)/dx
      return s + x * s * (1.0f - s) * du2;
    }
    case ACT_RELU: return f32x2{x.x > 0.f ? 1.f : 0.f, x.y > 0.f ? 1.f : 0.f};
    case ACT_SILU: {
      const f32x2 s = sigmoid2(x);
      return s * (x * (1.0f - s) + 1.0f);
    }
    case ACT_QUICK_GELU: {
      const f32x2 s = sigmoid2(x * 1.702f);
      return s * (x * 1.702f * (1.0f - s) + 1.0f);
    }
    case ACT_TANH: { const f32x2 t = tanh2(x); return 1.0f - t * t; }
    case ACT_SIGMOID: { const f32x2 s = sigmoid2(x); return s * (1.0f - s); }
    default: return f32x2{1.0f, 1.0f};
  }
}
// Octet forms for the GEMM epilogue: ONE switch per eight values (the compiler does not hoist the switch of act_fwd2
// out of an unrolled loop by itself: it then costs a chain of scalar compares and branches per pair).
template <int ACT>
__device__ __forceinline__ void act_fwd8_c(float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 r = act_fwd2(f32x2{v[e], v[e + 1]}, ACT);
    v[e] = r.x; v[e + 1] = r.y;
  }
}
__device__ __forceinline__ void act_fwd8(float (&v)[8], int act) {
  switch (act) {
    case ACT_NONE: break;
    case ACT_GELU_ERF: act_fwd8_c<ACT_GELU_ERF>(v); break;
    case ACT_GELU_TANH: act_fwd8_c<ACT_GELU_TANH>(v); break;
    case ACT_RELU: act_fwd8_c<ACT_RELU>(v); break;
    case ACT_SILU: act_fwd8_c<ACT_SILU>(v); break;
    case ACT_QUICK_GELU: act_fwd8_c<ACT_QUICK_GELU>(v); break;
    case ACT_TANH: act_fwd8_c<ACT_TANH>(v); break;
    default: act_fwd8_c<ACT_SIGMOID>(v); break;
  }
}
// four values (one accumulator column group of the ring kernels' epilogue), one switch
template <int ACT>
__device__ __forceinline__ void act_fwd4_c(float (&v)[4]) {
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    const f32x2 r = act_fwd2(f32x2{v[e], v[e + 1]}, ACT);
    v[e] = r.x; v[e + 1] = r.y;
  }
}
__device__ __forceinline__ void act_fwd4(float (&v)[4], int act) {
  switch (act) {
    case ACT_NONE: break;
    case ACT_GELU_ERF: act_fwd4_c<ACT_GELU_ERF>(v); break;
    case ACT_GELU_TANH: act_fwd4_c<ACT_GELU_TANH>(v); break;
    case ACT_RELU: act_fwd4_c<ACT_RELU>(v); break;
    case ACT_SILU: act_fwd4_c<ACT_SILU>(v); break;
    case ACT_QUICK_GELU: act_fwd4_c<ACT_QUICK_GELU>(v); break;
    case ACT_TANH: act_fwd4_c<ACT_TANH>(v); break;
    default: act_fwd4_c<ACT_SIGMOID>(v); break;
  }
}
template <int ACT>
__device__ __forceinline__ void act_bwd8_mul_c(float (&v)[8], const float (&a)[8]) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 r = act_bwd2(f32x2{a[e], a[e + 1]}, ACT);
    v[e] *= r.x; v[e + 1] *= r.y;
  }
}
// v *= act'(a)
__device__ __forceinline__ void act_bwd8_mul(float (&v)[8], const float (&a)[8], int act) {
  switch (act) {
    case ACT_NONE: break;
    case ACT_GELU_ERF: act_bwd8_mul_c<ACT_GELU_ERF>(v, a); break;
    case ACT_GELU_TANH: act_bwd8_mul_c<ACT_GELU_TANH>(v, a); break;
    case ACT_RELU: act_bwd8_mul_c<ACT_RELU>(v, a); break;
    case ACT_SILU: act_bwd8_mul_c<ACT_SILU>(v, a); break;
    case ACT_QUICK_GELU: act_bwd8_mul_c<ACT_QUICK_GELU>(v, a); break;
    case ACT_TANH: act_bwd8_mul_c<ACT_TANH>(v, a); break;
    default: act_bwd8_mul_c<ACT_SIGMOID>(v, a); break;
  }
}
__device__ __forceinline__ float act_fwd(float x, int act) { return act_fwd2(f32x2{x, x}, act).x; }
__device__ __forceinline__ float act_bwd(float x, int act) { return act_bwd2(f32x2{x, x}, act).x; }

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
// Attention-probability dropout (round 4).  One hash per score ELEMENT (two 32-bit multiplies + three xor-shifts, ~10 VALU
// issues) made the dropout section 160 of the ~450 instructions of a 32 x 32 score tile in each of the three trunk kernels
// -- more than the softmax itself.  Now: one strong hash per (score row, 32-key tile),
//     tk = hash32(rowkey(row) + tile * 0x9E3779B9),
// and per element j = key % 32 of that tile ONE 24-bit multiply-add (v_mad_u32_u24, full rate):
//     v = (x[23:0] * DROP_C[(j & 3) + 4 (j >> 3)])[31:0] + tk,   x = ((j >> 2) & 1) ? rotr(tk, 12) : tk,   keep <=> v >= thr.
// ((j >> 2) & 1 is the lane half that owns key j in the transposed score layout, (j & 3) + 4 (j >> 3) its accumulator register:
// in the forward / dQ kernels the 16 constants are immediates and x is formed once per tile.)  Keep rate and pairwise / triple
// statistics: tests/test_attention_oracle.py; oracle/torch_ref.py::attn_drop_keep_mask restates it bit for bit.
__device__ __forceinline__ uint32_t drop_tilekey(uint32_t rowkey, uint32_t tile) { return hash32(rowkey + tile * 0x9E3779B9u); }
#define DVLA_DROP_C(r)                                                                                                          \
  ((r) == 0 ? 0xb60881u : (r) == 1 ? 0x554da5u : (r) == 2 ? 0x6dada9u : (r) == 3 ? 0x9e0fffu : (r) == 4 ? 0xd1517fu :            \
   (r) == 5 ? 0x966d65u : (r) == 6 ? 0x764223u : (r) == 7 ? 0xb59e1du : (r) == 8 ? 0x80cd71u : (r) == 9 ? 0x769d3bu :            \
   (r) == 10 ? 0xba1a8fu : (r) == 11 ? 0xf85869u : (r) == 12 ? 0xf94c5bu : (r) == 13 ? 0x905af7u : (r) == 14 ? 0xec5577u : 0xaa3cd1u)
__device__ __forceinline__ uint32_t drop_rot(uint32_t tk, uint32_t half) {      // half = (j >> 2) & 1
  return __builtin_amdgcn_alignbit(tk, tk, 12u * half);
}
__device__ __forceinline__ uint32_t drop_elem(uint32_t x, uint32_t tk, uint32_t c) {
  return (x & 0xffffffu) * c + tk;        // the compiler forms v_mad_u32_u24 (24-bit operands: c < 2^24)
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
// elementwise.hip: out[c] = sum of nrows fp32 partial rows (deterministic order); out bf16 or fp32
int dvla_reduce_partial_rows(const float* partial, int nrows, int64_t cols, int64_t stride, void* out, int out_bf16, hipStream_t stream);

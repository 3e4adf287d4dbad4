// gemm_skinny.h -- GEMM for FEW ROWS (M <= 128): the evaluation-time shapes of the hot path.
//
//   A closed-loop control step at one episode (utils/eval_utils_calvin.py:82-147 evaluates one episode per rank) runs the DiT
//   action head on 2 x 10 x 6 = 120 token rows, ten DDIM steps of twelve layers (models/action_model/models.py:128-160,
//   dreamvla_model.py:935-987), and the CLIP / state / projector GEMMs of the newest frame on 1-77 rows.  The tiled kernels
//   put such a problem on N / 128 workgroups that each walk the whole K: profiles/r04_rollout_step_summary_before.txt has 634
//   launches of the register-staged kernel per control step at 22 us each -- 14.1 of the step's 22.5 ms -- for 2 GFLOP.
//   These problems are weight-bandwidth problems: 120 x 2304 x 768 reads 3.5 MB of weights and 184 KB of activations.
//
//   One workgroup = one 32 x 32 tile of C, four or eight waves (eight from K = 512 on), each wave an equal share of K: the A and
//   B fragments of a k16-step come straight from global memory (16 bytes per lane: row l31 of the operand, 8 consecutive k) up to
//   twelve steps ahead of the MFMA that consumes them -- no LDS staging, no barrier in the K loop (a wave's share of K = 768 is in
//   flight at once: one memory round trip); (M / 32) x (N / 32) workgroups spread the weight stream over the whole chip (288 for
//   the qkv projection above, 96 for fc2 with K = 3072).  The partial tiles meet in LDS and go
//   through the same octet epilogue as the register-staged kernel (bias, activation, pre-activation, dropout, act', residual,
//   bf16 / fp32 store: gemm_impl.h epilogue_oct), so the two kernels differ in fp32 summation order only.
//   Layout: A(m, k) and B(n, k) k-contiguous (nn.Linear weights (out, in)); K % 16 == 0; 16-byte aligned rows.
#pragma once
#include "gemm_impl.h"

namespace dvla_gemm {

// SK_PREFETCH = k16-steps in flight per wave (template parameter P): 12 -> 24 x 16 B per lane; 6 when a wave's share of K is
// at most six steps (K <= 768 on eight waves) -- 48 instead of 96 staging registers, so that two workgroups share a CU and the
// 288 / 384 tiles of the DiT head's qkv / fc1 projections start in ONE round instead of 256 + a straggler round

// LN: the rows of A are layer-normalised (no affine) before they enter the product (dvla.h a_layernorm).  A wave then holds its
// WHOLE share of the row in the prefetch registers (K <= 16 * SK_PREFETCH * WAVES): row sums and sums of squares from the
// fragments, combined across the half-waves (which hold alternate octets of a row) and across the waves through LDS, then the
// fragments are normalised in place and rounded to bf16 -- the value a separate LayerNorm launch would have written.
template <int WAVES, bool LN, int SK_PREFETCH>
__global__ __launch_bounds__(64 * WAVES) void gemm_skinny_kernel(GemmKArgs p) {
  __shared__ float part[WAVES][32][33];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, g = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t m0 = (int64_t)blockIdx.y * 32, n0 = (int64_t)blockIdx.x * 32;
  const bool m_ok = m0 + l31 < p.M, n_ok = n0 + l31 < p.N;
  const bf16_t* arow = p.A + (m0 + l31) * p.lda + 8 * g;
  const bf16_t* brow = p.B + (n0 + l31) * p.ldb + 8 * g;
  const int nsteps = (int)(p.K / 16), per = (nsteps + WAVES - 1) / WAVES;
  const int s0 = wave * per, s1 = (s0 + per < nsteps) ? s0 + per : nsteps;

  auto ld = [&](const bf16_t* row, bool ok, int s) -> bf16x8 {
    uint4 u = make_uint4(0u, 0u, 0u, 0u);
    if (ok && s < s1) u = *reinterpret_cast<const uint4*>(row + 16 * s);
    return *reinterpret_cast<const bf16x8*>(&u);
  };
  bf16x8 fa[SK_PREFETCH], fb[SK_PREFETCH];
#pragma unroll
  for (int i = 0; i < SK_PREFETCH; ++i) { fa[i] = ld(arow, m_ok, s0 + i); fb[i] = ld(brow, n_ok, s0 + i); }
  // The epilogue reads its bias / residual octet only after the K loop and the barrier: a second full memory round trip on
  // a kernel that lives for two.  One dword of each is requested here, with the fragments, so that the lines are on their way
  // (the value itself is not used: the epilogue's own 16-byte loads then hit the cache).
  uint32_t touch = 0;
  if (t < 128) {
    const int64_t m = m0 + (t >> 2), n = n0 + (t & 3) * 8;
    if (m < p.M && n < p.N) {
      if (p.bias) touch ^= *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.bias) + n * (p.bias_f32 ? 4 : 2));
      if (p.residual) {
        const int64_t rr = p.res_rows > 0 ? (int64_t)((uint32_t)m % (uint32_t)p.res_rows) : m;
        touch ^= *reinterpret_cast<const uint32_t*>(p.residual + rr * p.ld_res + n);
      }
    }
  }
  if constexpr (LN) {
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < SK_PREFETCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float v = bf2f((bf16_t)fa[i][e]); sm += v; sq = fmaf(v, v, sq); }     // (steps past s1: zeros)
    sm += __shfl_xor(sm, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (g == 0) { part[wave][0][l31] = sm; part[wave][1][l31] = sq; }
    __syncthreads();
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { S += part[w][0][l31]; Q += part[w][1][l31]; }
    __syncthreads();               // (the partial-tile rows written below reuse this memory)
    const float inv_k = 1.0f / (float)p.K;
    const float mean = S * inv_k;
    const float var = fmaxf(Q * inv_k - mean * mean, 0.f);
    const float rstd = rsqrtf(var + p.a_ln_eps);
#pragma unroll
    for (int i = 0; i < SK_PREFETCH; ++i) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf2f((bf16_t)fa[i][e]) - mean) * rstd;
      union { uint32_t w[4]; bf16x8 f; } u;
      u.w[0] = pack2bf(v[0], v[1]); u.w[1] = pack2bf(v[2], v[3]); u.w[2] = pack2bf(v[4], v[5]); u.w[3] = pack2bf(v[6], v[7]);
      if (s0 + i < s1) fa[i] = u.f;         // (steps past this wave's share stay zero)
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int s = s0; s < s1; s += SK_PREFETCH) {
#pragma unroll
    for (int i = 0; i < SK_PREFETCH; ++i) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[i], acc, 0, 0, 0);       // (steps past s1 hold zeros)
      fa[i] = ld(arow, m_ok, s + SK_PREFETCH + i);
      fb[i] = ld(brow, n_ok, s + SK_PREFETCH + i);
    }
  }
  // acc[r] = C(m0 + (r & 3) + 8 (r >> 2) + 4 g, n0 + l31), this wave's quarter of K
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * g][l31] = acc[r];
  __syncthreads();
  if (t < 128) {
    const int mi = t >> 2, oc = t & 3;
    const int64_t m = m0 + mi, n = n0 + oc * 8;
    if (m < p.M && n < p.N) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += part[w][mi][oc * 8 + e];       // fixed order: deterministic
        v[e] = sum;
      }
      const bool full = n + 8 <= p.N && p.c_vec && p.aux_vec && p.epi_vec;
      if (full) epilogue_oct<true>(p, m, n, v, 0);
      else epilogue_oct<false>(p, m, n, v, 0);
    }
  }
  asm volatile("" ::"v"(touch));
}

// rows at or below which this kernel takes a k-contiguous problem: the DiT head at one episode (120 rows), the text tower (77),
// the newest frame's two views through the ViT (394) -- at 512 rows x N = 3072 x K = 768 the 1536 tiles re-read 150 MB from
// the L2s (~4 us of the chip's L2 bandwidth) where the 128 x 128 tiling keeps 96 workgroups busy for 24 us
constexpr int64_t SKINNY_MAX_M = 512;

inline bool skinny_ok(const GemmKArgs& a, int combo, int split_k) {
  return combo == 0 && split_k == 1 && a.M <= SKINNY_MAX_M && a.K >= 16 && a.K % 16 == 0 && a.a_vec && a.b_vec &&
         a.N <= (int64_t)65535 * 32;
}

// on-the-fly LayerNorm of A: the eight-wave variant with every wave's share of K resident in its prefetch registers
inline bool skinny_ln_ok(const GemmKArgs& a) { return a.K >= 512 && a.K <= (int64_t)16 * 12 * 8 && a.a_ln_eps > 0.f; }

// DVLA_SKINNY_CFG = 0 (default rule) | 1: 4 waves, 12 steps | 2: 8 waves, 12 steps | 3: 8 waves, 6 steps (K <= 768 only) | 4: 8 waves, 20
// steps (1536 < K <= 3072, at most 512 tiles) -- measurement
// (tests/gpu_skinny_perf.py); read per launch
inline int skinny_cfg() { const char* e = getenv("DVLA_SKINNY_CFG"); return e ? atoi(e) : 0; }

inline void launch_skinny(const GemmKArgs& a, hipStream_t stream) {
  dim3 grid((unsigned)((a.N + 31) / 32), (unsigned)((a.M + 31) / 32), 1);
  const bool short_k = a.K <= 16 * 6 * 8;       // a wave of the eight-wave variant holds its whole share in six steps
  int cfg = skinny_cfg();
  if (cfg == 3 && !short_k) cfg = 2;
  if (a.a_ln) {
    if (short_k && cfg != 2) hipLaunchKernelGGL((gemm_skinny_kernel<8, true, 6>), grid, dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((gemm_skinny_kernel<8, true, 12>), grid, dim3(512), 0, stream, a);
    return;
  }
  // long K on few tiles (the DiT head's fc2: K = 3072, 96 tiles): a workgroup alone on its CU is latency-bound -- 14.6 us with
  // twelve steps in flight and two refills (profiles/r04_skinny_perf.jsonl); twenty of the 24 steps of a wave's share in flight at once (24 would spill)
  const bool deep = a.K > 16 * 12 * 8 && a.K <= 16 * 24 * 8 && (int64_t)grid.x * grid.y <= 2 * 256;
  if (cfg == 0) cfg = a.K < 512 ? 1 : (short_k ? 3 : 2);      // (cfg 4 measured: 16.1 us against 14.6 for cfg 2 at K = 3072 -- not the default)
  if (cfg == 4 && !deep) cfg = 2;
  if (cfg == 1) hipLaunchKernelGGL((gemm_skinny_kernel<4, false, 12>), grid, dim3(256), 0, stream, a);
  else if (cfg == 3) hipLaunchKernelGGL((gemm_skinny_kernel<8, false, 6>), grid, dim3(512), 0, stream, a);
  else if (cfg == 4) hipLaunchKernelGGL((gemm_skinny_kernel<8, false, 20>), grid, dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((gemm_skinny_kernel<8, false, 12>), grid, dim3(512), 0, stream, a);
}

}  // namespace dvla_gemm

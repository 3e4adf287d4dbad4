// gemm_skinny.h -- GEMM for FEW ROWS (M <= 128): the evaluation-time shapes of the hot path.
//
//   A closed-loop control step at one episode (utils/eval_utils_calvin.py:82-147 evaluates one episode per rank) runs the DiT
//   action head on 2 x 10 x 6 = 120 token rows, ten DDIM steps of twelve layers (models/action_model/models.py:128-160,
//   dreamvla_model.py:935-987), and the CLIP / state / projector GEMMs of the newest frame on 1-77 rows.  The tiled kernels
//   put such a problem on N / 128 workgroups that each walk the whole K: profiles/r04_rollout_step_summary_before.txt has 634
//   launches of the register-staged kernel per control step at 22 us each -- 14.1 of the step's 22.5 ms -- for 2 GFLOP.
//   These problems are weight-bandwidth problems: 120 x 2304 x 768 reads 3.5 MB of weights and 184 KB of activations.
//
//   One workgroup = one 32 x 32 tile of C, four waves, each wave a quarter of K: the A and B fragments of a k16-step come
//   straight from global memory (16 bytes per lane: row l31 of the operand, 8 consecutive k) twelve steps ahead of the MFMA that
//   consumes them -- no LDS staging, no barrier in the K loop; (M / 32) x (N / 32) workgroups spread the weight stream over
//   the whole chip (288 for the qkv projection above, 96 for fc2 with K = 3072).  The four partial tiles meet in LDS and go
//   through the same octet epilogue as the register-staged kernel (bias, activation, pre-activation, dropout, act', residual,
//   bf16 / fp32 store: gemm_impl.h epilogue_oct), so the two kernels differ in fp32 summation order only.
//   Layout: A(m, k) and B(n, k) k-contiguous (nn.Linear weights (out, in)); K % 16 == 0; 16-byte aligned rows.
#pragma once
#include "gemm_impl.h"

namespace dvla_gemm {

constexpr int SK_PREFETCH = 12;   // k16-steps in flight per wave: 24 x 16 B per lane

__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmKArgs p) {
  __shared__ float part[4][32][33];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, g = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t m0 = (int64_t)blockIdx.y * 32, n0 = (int64_t)blockIdx.x * 32;
  const bool m_ok = m0 + l31 < p.M, n_ok = n0 + l31 < p.N;
  const bf16_t* arow = p.A + (m0 + l31) * p.lda + 8 * g;
  const bf16_t* brow = p.B + (n0 + l31) * p.ldb + 8 * g;
  const int nsteps = (int)(p.K / 16), per = (nsteps + 3) / 4;
  const int s0 = wave * per, s1 = (s0 + per < nsteps) ? s0 + per : nsteps;

  auto ld = [&](const bf16_t* row, bool ok, int s) -> bf16x8 {
    uint4 u = make_uint4(0u, 0u, 0u, 0u);
    if (ok && s < s1) u = *reinterpret_cast<const uint4*>(row + 16 * s);
    return *reinterpret_cast<const bf16x8*>(&u);
  };
  bf16x8 fa[SK_PREFETCH], fb[SK_PREFETCH];
#pragma unroll
  for (int i = 0; i < SK_PREFETCH; ++i) { fa[i] = ld(arow, m_ok, s0 + i); fb[i] = ld(brow, n_ok, s0 + i); }
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
      for (int e = 0; e < 8; ++e) v[e] = (part[0][mi][oc * 8 + e] + part[1][mi][oc * 8 + e]) + (part[2][mi][oc * 8 + e] + part[3][mi][oc * 8 + e]);
      const bool full = n + 8 <= p.N && p.c_vec && p.aux_vec && p.epi_vec;
      if (full) epilogue_oct<true>(p, m, n, v, 0);
      else epilogue_oct<false>(p, m, n, v, 0);
    }
  }
}

// rows at or below which the skinny kernel takes a k-contiguous problem
constexpr int64_t SKINNY_MAX_M = 128;

inline bool skinny_ok(const GemmKArgs& a, int combo, int split_k) {
  return combo == 0 && split_k == 1 && a.M <= SKINNY_MAX_M && a.K >= 16 && a.K % 16 == 0 && a.a_vec && a.b_vec &&
         a.N <= (int64_t)65535 * 32;
}

inline void launch_skinny(const GemmKArgs& a, hipStream_t stream) {
  dim3 grid((unsigned)((a.N + 31) / 32), (unsigned)((a.M + 31) / 32), 1);
  hipLaunchKernelGGL(gemm_skinny_kernel, grid, dim3(256), 0, stream, a);
}

}  // namespace dvla_gemm

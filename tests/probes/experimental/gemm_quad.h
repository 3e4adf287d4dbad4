// gemm_quad.h -- "quad" GEMM: 256 x 256 tile, K-tile 64, FOUR waves of 128 x 128 (one per SIMD, 512 registers each: the
// 256 accumulators live in AGPRs), operands global -> VGPR -> ds_write_b128 -> swizzled row-major LDS image (two 64-KiB
// buffers), fragments by ds_read_b128.  The structure of the vendor kernels that win this round's yardstick
// (profiles/r02_library_yardstick.txt): per K-tile and CU 128 KiB of fragment reads instead of the 192 of eight 128 x 64
// waves, and no LDS-DMA (whose per-piece issue cost is what holds the phase kernel at ~55 % matrix-pipe busy).
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T ), both operands k-contiguous (a_trans = b_trans = 0), K % 64 == 0, N % 64 == 0,
//   16-byte-vectorisable operands: the dispatcher checks (ring_ok<PCfg>), other layouts stay with the phase / ring kernels.
// One tile per workgroup (the hardware dispatcher balances; XCD-contiguous tile runs as in gemm_kernel).
//
// STATUS (round 2): NOT part of the build -- a starting point for the next round.  Correct (18 probe checks as variant 94).
//   first form (HIP uint4 staging arrays indexed from `#pragma unroll` loops): the arrays stay stack objects -- a scratch store
//     behind every global load inside the K loop: 272 TFLOP/s at 20832 x 4096 x 1024;
//   this form (ext-vector staging registers, static_for, half a K-tile of fragments read before its 32 MFMAs; no scratch in the
//     loop, 140 VGPRs + 256 AGPRs): 711 TFLOP/s there, 870 at K = 2048, 1042 at 8192^3 (phase kernel: 890 / 1088 / 1249);
//   with fragment double-buffering across the barrier and a whole K-tile of staging in flight: 663 / 807 / 977 -- slower.
// The compiler-scheduled loop leaves LDS-read latency, the ds_write burst and the barrier between the MFMA groups; the
// structure needs its loop written at the instruction level, as the vendor's kernels are.
// To try it: #include it from csrc/gemm.hip (path "../../tests/probes/experimental/gemm_quad.h"), instantiate
// launch_quad_one<EPI> and route a variant number to it.
#pragma once
#include "gemm_impl.h"

namespace dvla_gemm {

struct QCfg {
  static constexpr int BM = 256, BN = 256, NT = 256, TM = 4, TN = 4;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + B_BYTES;
  static constexpr int SMEM_BYTES = 2 * BUF_BYTES;   // 128 KiB
  static constexpr int NCH = 8;                      // 16-byte chunks per thread, operand and K-tile
  static constexpr int BKS = 64, GH = 4;             // (ring_item: K-tile, tile rows per raster group)
};

template <int EPI>
__global__ __launch_bounds__(QCfg::NT) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_quad_kernel(GemmKArgs p) {
  constexpr int BM = QCfg::BM, BN = QCfg::BN, NCH = QCfg::NCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l31 = lane & 31, g = lane >> 5;

  const int ntiles = p.tiles_m * p.tiles_n;
  int tile;
  {   // block b runs on XCD b % 8: every XCD gets a contiguous run of tile ids (GH-row raster inside: ring_item)
    const int b = blockIdx.x, q = ntiles / 8, r = ntiles % 8, xcd = b % 8, idx = b / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const RingItem w = ring_item<QCfg>(p, tile);     // same raster / K range arithmetic as the phase kernel (256 x 256, 64)
  const int64_t m0 = w.m0, n0 = w.n0;
  const int nk = w.ns;

  // staging map: thread t owns k-octet (t & 7) of rows (t >> 3) + 32 i, i = 0..7, of both operand tiles.  Addresses =
  // wave-uniform base (advances by one K-tile) + per-lane 32-bit byte offsets (rows past the end are clamped to the last
  // row: their products land in output rows / columns the epilogue never stores).
  const int r = t >> 3, oct = t & 7;
  uint32_t offa[NCH], offb[NCH];
  static_for<NCH>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    int64_t ra = m0 + r + 32 * i; ra = ra < p.M ? ra : p.M - 1;
    int64_t rb = n0 + r + 32 * i; rb = rb < p.N ? rb : p.N - 1;
    offa[i] = (uint32_t)(((ra - m0) * p.lda + oct * 8) * 2);
    offb[i] = (uint32_t)(((rb - n0) * p.ldb + oct * 8) * 2);
  });
  const char* baseA = reinterpret_cast<const char*>(p.A + m0 * p.lda + w.k_begin);
  const char* baseB = reinterpret_cast<const char*>(p.B + n0 * p.ldb + w.k_begin);
  int lds_st[NCH];
  static_for<NCH>([&](auto ic) { constexpr int i = decltype(ic)::value; lds_st[i] = rm_off(r + 32 * i, oct); });

  u32x4 st[NCH];     // one operand's 8 chunks at a time (ext-vector type + static_for: HIP's uint4 struct array indexed from a
                     // `#pragma unroll` loop stays a stack object -- scratch traffic inside the K loop, 272 TFLOP/s): A rides through the first half of the multiply, B through the second
  auto gloadA = [&]() {
    static_for<NCH>([&](auto ic) { constexpr int i = decltype(ic)::value; st[i] = *reinterpret_cast<const u32x4*>(baseA + offa[i]); });
    baseA += BK * 2;
  };
  auto gloadB = [&]() {
    static_for<NCH>([&](auto ic) { constexpr int i = decltype(ic)::value; st[i] = *reinterpret_cast<const u32x4*>(baseB + offb[i]); });
    baseB += BK * 2;
  };
  auto lstore = [&](char* img) {
    static_for<NCH>([&](auto ic) { constexpr int i = decltype(ic)::value; *reinterpret_cast<u32x4*>(img + lds_st[i]) = st[i]; });
  };

  f32x16 acc[4][4];   // [n-subtile i][m-subtile j]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // two k16-steps: all 16 fragments first (64 VGPRs), then 32 MFMAs back to back
  auto compute2 = [&](const char* cur, int ks0) {
    const char* ca = cur;
    const char* cb = cur + QCfg::A_BYTES;
    bf16x8 fa[2][4], fb[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[s2][j] = *reinterpret_cast<const bf16x8*>(ca + rm_off(wm * 128 + j * 32 + l31, (ks0 + s2) * 2 + g));
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[s2][i] = *reinterpret_cast<const bf16x8*>(cb + rm_off(wn * 128 + i * 32 + l31, (ks0 + s2) * 2 + g));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s2][i], fa[s2][j], acc[i][j], 0, 0, 0);
  };

  char* buf0 = smem;
  char* buf1 = smem + QCfg::BUF_BYTES;
  gloadA(); lstore(buf0);
  gloadB(); lstore(buf0 + QCfg::A_BYTES);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    char* cur = (kt & 1) ? buf1 : buf0;
    char* nxt = (kt & 1) ? buf0 : buf1;
    const bool more = kt + 1 < nk;
    if (more) gloadA();
    __builtin_amdgcn_sched_barrier(0);
    compute2(cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) { lstore(nxt); gloadB(); }
    __builtin_amdgcn_sched_barrier(0);
    compute2(cur, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (more) lstore(nxt + QCfg::A_BYTES);
    __syncthreads();
  }

#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
    reg_epilogue<4, EPI>(p, reinterpret_cast<f32x16 (&)[2][4]>(acc[2 * hf]), lane, m0 + wm * 128, n0 + wn * 128 + hf * 64, w.split);
}

template <int EPI>
void launch_quad_one(const GemmKArgs& a, int split_k, hipStream_t stream) {
  static bool attr_set = false;
  auto kern = &gemm_quad_kernel<EPI>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, QCfg::SMEM_BYTES);
    attr_set = true;
  }
  const int64_t items = (int64_t)a.tiles_m * a.tiles_n * split_k;
  hipLaunchKernelGGL(kern, dim3((unsigned)items), dim3(QCfg::NT), QCfg::SMEM_BYTES, stream, a);
}

}  // namespace dvla_gemm

// load_probe.cpp -- how fast can ONE CU pull GEMM operand panels out of L2 / Infinity Cache, and does the path matter?
// (measurement infrastructure).  256 persistent workgroups x 512 threads walk the work items of a 256 x 256-tile GEMM
// (same raster as the ring kernels) and, per 64-wide K-tile, fetch the 32-KiB A image + 32-KiB B image of the tile exactly
// as the GEMM kernels address them (k-contiguous operands, 128-byte rows, 8 rows x 128 B per wave instruction) -- and do
// nothing else.  Modes:
//   0  LDS-DMA (global_load_lds_dwordx4) into a 2 x 64-KiB LDS ring, <= 16 pieces in flight per wave
//   1  global_load_dwordx4 into VGPRs (two register sets of 8 x 16 B per wave), results discarded
//   2  A by LDS-DMA, B by VGPR loads (half each)
//   3  mode 0 with 64-byte rows (K-tile 32, the round-1 ring layout): half cache lines per request
//   4  mode 1 + ds_write_b128 of every loaded vector (the register-staged path complete)
// Reports microseconds and bytes / clock / CU (clock = 2.4 GHz nominal; the real clock is lower under load).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef unsigned short bf16_t;

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(512) void load_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int64_t M, int64_t N, int64_t K,
                                                   int tiles_m, int tiles_n, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nitems = tiles_m * tiles_n;
  const int grid = gridDim.x;
  const int perm = (grid & 7) == 0 ? (int)(blockIdx.x & 7) * (grid >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  constexpr int BK = (MODE == 3) ? 32 : 64;
  constexpr int LPR = BK / 8;                 // lanes per row
  constexpr int RPC = 64 / LPR;               // rows per 1-KiB piece
  constexpr int PIECES = 256 * BK * 2 / 1024; // pieces per operand image (32 or 16)
  constexpr int PPW = PIECES / 8;             // pieces per wave and operand
  uint4 acc = make_uint4(0, 0, 0, 0);
  int slot = 0;
  for (int it = 0;; ++it) {
    const int id = it * grid + perm;
    if (id >= nitems) break;
    const int per_panel = 4 * tiles_n;
    const int panel = id / per_panel, r = id - panel * per_panel;
    const int left = tiles_m - panel * 4;
    const int gh = left < 4 ? left : 4;
    const int tn = r / gh, tm = panel * 4 + (r - tn * gh);
    const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
    const bf16_t* pa[PPW]; const bf16_t* pb[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int c = wave * PPW + i;
      const int rl = c * RPC + lane / LPR, sl = lane % LPR;
      const int o = (BK == 32) ? (sl ^ ((rl >> 2) & 3)) : (sl ^ ((rl >> 1) & 7));
      int64_t ra = m0 + rl; ra = ra < M ? ra : M - 1;
      int64_t rb = n0 + rl; rb = rb < N ? rb : N - 1;
      pa[i] = A + ra * K + o * 8;
      pb[i] = B + rb * K + o * 8;
    }
    for (int64_t k = 0; k < K; k += BK) {
      const uint32_t base = smem_base + (uint32_t)((slot & 1) * 65536 + wave * (PPW * 1024));
      if (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) { glds16(pa[i], __builtin_amdgcn_readfirstlane(base + i * 1024)); pa[i] += BK; }
#pragma unroll
        for (int i = 0; i < PPW; ++i) { glds16(pb[i], __builtin_amdgcn_readfirstlane(base + 32768 + i * 1024)); pb[i] += BK; }
        wait_vmcnt<2 * PPW>();          // the previous K-tile's pieces have landed; this one's stay in flight
      } else if (MODE == 1 || MODE == 4) {
        uint4 v[2 * PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) { v[i] = *reinterpret_cast<const uint4*>(pa[i]); pa[i] += BK; }
#pragma unroll
        for (int i = 0; i < PPW; ++i) { v[PPW + i] = *reinterpret_cast<const uint4*>(pb[i]); pb[i] += BK; }
        if (MODE == 4) {
#pragma unroll
          for (int i = 0; i < 2 * PPW; ++i) *reinterpret_cast<uint4*>(smem + (slot & 1) * 65536 + (wave * 2 * PPW + i) * 1024 + lane * 16) = v[i];
        } else {
#pragma unroll
          for (int i = 0; i < 2 * PPW; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
        }
      } else {   // MODE 2
#pragma unroll
        for (int i = 0; i < PPW; ++i) { glds16(pa[i], __builtin_amdgcn_readfirstlane(base + i * 1024)); pa[i] += BK; }
        uint4 v[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) { v[i] = *reinterpret_cast<const uint4*>(pb[i]); pb[i] += BK; }
#pragma unroll
        for (int i = 0; i < PPW; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
      }
      ++slot;
    }
  }
  wait_vmcnt<0>();
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
static void run(const char* name, const bf16_t* A, const bf16_t* B, int64_t M, int64_t N, int64_t K, int cus, unsigned* sink) {
  const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)(N / 256);
  const int items = tiles_m * tiles_n;
  auto kern = &load_kernel<MODE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, sum = 0;
  const int R = 5;
  for (int r = 0; r < R + 1; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(items < cus ? items : cus), dim3(512), 131072, 0, A, B, M, N, K, tiles_m, tiles_n, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) { sum += ms; if (ms < best) best = ms; }
  }
  const double bytes = (double)items * 512.0 * K * 2;          // bytes moved into the CUs
  const int rounds = (items + cus - 1) / cus;
  const double per_cu = (double)rounds * 512.0 * K * 2;        // the busiest CU's bytes
  const double us = sum / R * 1e3;
  printf("{\"mode\": \"%s\", \"M\": %lld, \"N\": %lld, \"K\": %lld, \"avg_us\": %.1f, \"min_us\": %.1f, \"GBps_per_CU\": %.1f, \"B_per_clk_CU_at_2.4GHz\": %.1f, "
         "\"chip_TBps\": %.2f, \"equiv_TF_256sq\": %.0f}\n", name, (long long)M, (long long)N, (long long)K, us, best * 1e3, per_cu / us / 1e3,
         per_cu / us / 1e3 / 2.4, bytes / us / 1e6, 2.0 * M * N * K / us / 1e6);
  fflush(stdout);
}

__global__ void fill(bf16_t* p, int64_t n) { int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (bf16_t)(0x3c00 + (i * 2654435761u >> 20 & 0x3ff)); }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned* sink; CK(hipMalloc(&sink, 64));
  const int64_t shapes[][3] = {{20832, 4096, 1024}, {20832, 1024, 4096}, {8192, 8192, 8192}, {88256, 3072, 768}};
  for (auto& sh : shapes) {
    const int64_t M = sh[0], N = sh[1], K = sh[2];
    bf16_t *A, *B; CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2));
    hipLaunchKernelGGL(fill, dim3((unsigned)((M * K + 255) / 256)), dim3(256), 0, 0, A, M * K);
    hipLaunchKernelGGL(fill, dim3((unsigned)((N * K + 255) / 256)), dim3(256), 0, 0, B, N * K);
    run<0>("LDS-DMA, 128-B rows", A, B, M, N, K, cus, sink);
    run<3>("LDS-DMA, 64-B rows (BK 32)", A, B, M, N, K, cus, sink);
    run<1>("global_load_dwordx4 -> VGPR", A, B, M, N, K, cus, sink);
    run<4>("global_load_dwordx4 -> VGPR -> ds_write_b128", A, B, M, N, K, cus, sink);
    run<2>("A by LDS-DMA, B by VGPR", A, B, M, N, K, cus, sink);
    CK(hipFree(A)); CK(hipFree(B));
  }
  return 0;
}

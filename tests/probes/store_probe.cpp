// store_probe.cpp -- what does the C-tile write of a persistent GEMM cost by itself?  (measurement infrastructure)
// 256 workgroups x 512 threads (one per CU, like the ring GEMM) each write 256 x 256 bf16 tiles of an M x N matrix, tile
// list walked like the GEMM's work items, nothing else (no loads, no MFMA).  Store patterns per wave (64 x 128... a wave
// owns a 128-row x 64-column block of the tile, as in the 256x256 kernel):
//   0  accumulator-layout + half exchange: one instruction = 32 rows x 32 B        (reg_epilogue today)
//   1  transposed layout: one instruction = 8 rows x 128 B (whole lines)           (what the LDS patch produced)
//   2  8-byte stores from the raw accumulator layout: 32 rows x 16 B               (round-1 "direct store")
//   3  tile stored as a contiguous 128-KiB block (not a matrix tile): the streaming-write bound for this launch shape
//   4  pattern 1 with non-temporal stores
//   5  pattern 0 with non-temporal stores
// Build: hipcc --offload-arch=gfx950 -O3 tests/probes/store_probe.cpp -o build/store_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* C, int64_t M, int64_t N, int tiles_m, int tiles_n, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1;             // 2 x 4 waves of 128 x 64
  const bool one_xcd = (reps >> 30) & 1;     // --percu: only the workgroups of XCD 0 (blockIdx % 8 == 0) work
  if (one_xcd && (blockIdx.x & 7) != 0) return;
  const int nitems = ((reps >> 8) & 0x3fffff) ? ((reps >> 8) & 0x3fffff) : tiles_m * tiles_n;   // (--percu: item limit in the upper bits)
  reps &= 0xff;
  const int grid = one_xcd ? (int)gridDim.x / 8 : (int)gridDim.x;
  const int perm = one_xcd ? (int)(blockIdx.x >> 3)
                           : ((grid & 7) == 0 ? (int)(blockIdx.x & 7) * (grid >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  const uint4 v = make_uint4(0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + wave);
  const u32x4 vv = {v.x, v.y, v.z, v.w};
  for (int it = 0;; ++it) {
    const int id = it * grid + perm;
    if (id >= nitems) break;
    const int per_panel = 4 * tiles_n;
    const int panel = id / per_panel, r = id - panel * per_panel;
    const int left = tiles_m - panel * 4;
    const int gh = left < 4 ? left : 4;
    const int tn = r / gh, tm = panel * 4 + (r - tn * gh);
    const int64_t m0 = (int64_t)tm * 256 + wm * 128, n0 = (int64_t)tn * 256 + wn * 64;
    for (int rep = 0; rep < reps; ++rep) {
      if (PAT == 0 || PAT == 5) {
        const int l31 = lane & 31, g = lane >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t m = m0 + 32 * j + l31;
          if (m < M) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              uint4* dst = reinterpret_cast<uint4*>(C + m * N + n0 + 16 * q + 8 * g);
              if (PAT == 5) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(dst)); else *dst = v;
            }
          }
        }
      } else if (PAT == 1 || PAT == 4) {
        const int cg = lane & 7, r8 = lane >> 3;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int64_t m = m0 + 8 * s + r8;
          if (m < M) {
            uint4* dst = reinterpret_cast<uint4*>(C + m * N + n0 + 8 * cg);
            if (PAT == 4) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(dst)); else *dst = v;
          }
        }
      } else if (PAT == 6 || PAT == 7) {
        // 6: the whole-line mapping of reg_epilogue after the quad transposition (round 3): instruction c of slab j writes rows
        //    4a + c; a row's 8 pieces come from lanes 4a + b (pieces 2b) and 32 + 4a + b (pieces 2b + 1): whole lines per
        //    instruction, but a quad of lanes covers every OTHER 16 bytes of its line
        // 7: pieces in lane order: lane = (m0, m4, m3, piece[2:0]), instruction = (m1, m2): 8 consecutive lanes = one line
        const int l31 = lane & 31, g = lane >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            int row, piece;
            if (PAT == 6) { row = 4 * (l31 >> 2) + c; piece = 2 * (lane & 3) + g; }
            else { row = (lane >> 5) | ((c & 1) << 1) | ((c >> 1) << 2) | (((lane >> 3) & 1) << 3) | (((lane >> 4) & 1) << 4); piece = lane & 7; }
            const int64_t m = m0 + 32 * j + row;
            if (m < M) *reinterpret_cast<uint4*>(C + m * N + n0 + 8 * piece) = v;
          }
      } else if (PAT == 2) {
        const int l31 = lane & 31, g = lane >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t m = m0 + 32 * j + l31;
          if (m < M) {
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<uint2*>(C + m * N + n0 + 8 * q + 4 * g) = make_uint2(v.x, v.y);
          }
        }
      } else {
        // contiguous: item id owns bytes [id * 128 KiB, +128 KiB); the wave owns 16 KiB of it
        unsigned short* base = C + (int64_t)id * 65536 + wave * 8192;
#pragma unroll
        for (int s = 0; s < 16; ++s) *reinterpret_cast<uint4*>(base + s * 512 + lane * 8) = v;
      }
    }
  }
}

template <int PAT>
static void run(const char* name, unsigned short* C, int64_t M, int64_t N, int cus) {
  const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)(N / 256);
  const int items = tiles_m * tiles_n;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, sum = 0;
  const int R = 7;
  for (int r = 0; r < R + 1; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(store_kernel<PAT>, dim3(items < cus ? items : cus), dim3(512), 0, 0, C, M, N, tiles_m, tiles_n, 1);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) { sum += ms; if (ms < best) best = ms; }
  }
  const double bytes = (double)M * N * 2;
  printf("{\"pattern\": \"%s\", \"M\": %lld, \"N\": %lld, \"avg_us\": %.1f, \"min_us\": %.1f, \"TBps_avg\": %.2f}\n", name, (long long)M, (long long)N,
         sum / R * 1e3, best * 1e3, bytes / (sum / R * 1e-3) / 1e12);
  fflush(stdout);
}

// --percu: what ONE CU can push when the others are idle (is the epilogue's store tail bound by the CU's own store path or by
// the chip's write bandwidth?): G workgroups (G = 8 ... 256) each write 64 tiles of a 20832 x 4096 matrix; GB/s per workgroup.
template <int PAT>
static void run_percu(const char* name, unsigned short* C, int64_t M, int64_t N, int G, bool one_xcd = false) {
  const int tiles_m = (int)((M + 255) / 256), tiles_n = (int)(N / 256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  const int items = 64 * G < tiles_m * tiles_n ? 64 * G : tiles_m * tiles_n;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, 0));
    // tiles_m / tiles_n describe the whole matrix; only the first `items` tile ids are visited (grid = G)
    hipLaunchKernelGGL(store_kernel<PAT>, dim3(one_xcd ? G * 8 : G), dim3(512), 0, 0, C, M, N, tiles_m, tiles_n,
                       1 | (items << 8) | (one_xcd ? (1 << 30) : 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r && ms < best) best = ms;
  }
  const double bytes = (double)items * 256 * 256 * 2;
  printf("{\"percu\": \"%s%s\", \"workgroups\": %d, \"min_us\": %.1f, \"GBps_per_workgroup\": %.1f, \"TBps_total\": %.2f}\n", name, one_xcd ? " [all on ONE XCD]" : "", G, best * 1e3,
         bytes / G / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  if (argc > 1 && !strcmp(argv[1], "--percu")) {
    const int64_t M = 20832, N = 4096;
    unsigned short* C; CK(hipMalloc(&C, (size_t)(M + 256) * N * 2));
    for (int G : {8, 16, 32, 64, 128, 256}) {
      run_percu<3>("contiguous 128-KiB blocks", C, M, N, G);
      run_percu<1>("8 rows x 128 B per instruction", C, M, N, G);
      run_percu<6>("whole lines, quad-sparse lane order (reg_epilogue round 3)", C, M, N, G);
      run_percu<7>("whole lines, pieces in lane order", C, M, N, G);
      run_percu<0>("32 rows x 32 B per instruction (half exchange)", C, M, N, G);
      run_percu<2>("32 rows x 16 B per instruction (8-byte stores)", C, M, N, G);
    }
    for (int G : {4, 16, 32}) {      // the same number of workgroups packed onto one XCD: is the limit the XCD's path to the fabric?
      run_percu<1>("8 rows x 128 B per instruction", C, M, N, G, true);
      run_percu<0>("32 rows x 32 B per instruction (half exchange)", C, M, N, G, true);
    }
    CK(hipFree(C));
    return 0;
  }
  const int64_t shapes[][2] = {{20832, 4096}, {20832, 1024}, {88256, 3072}, {88256, 768}};
  for (auto& sh : shapes) {
    const int64_t M = sh[0], N = sh[1];
    unsigned short* C; CK(hipMalloc(&C, (size_t)(M + 256) * N * 2));
    run<3>("contiguous 128-KiB blocks", C, M, N, cus);
    run<1>("8 rows x 128 B per instruction", C, M, N, cus);
    run<0>("32 rows x 32 B per instruction (half exchange)", C, M, N, cus);
    run<2>("32 rows x 16 B per instruction (8-byte stores)", C, M, N, cus);
    run<4>("8 rows x 128 B, non-temporal", C, M, N, cus);
    run<5>("32 rows x 32 B, non-temporal", C, M, N, cus);
    CK(hipFree(C));
  }
  return 0;
}

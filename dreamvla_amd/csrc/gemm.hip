// gemm.hip -- host-side dispatcher of the bf16 GEMM kernels (kernel templates: gemm_impl.h; their instantiations are
// compiled one (configuration, operand layout) per translation unit in gemm_inst_*.hip so that the build runs in parallel).
#include <stdlib.h>

#include <mutex>
#include "gemm_phase.h"
#include "gemm_skinny.h"

namespace dvla_gemm {
#define DVLA_EXTERN_REG(CF, AT, BT, DBG) extern template void launch_one<CF, AT, BT, DBG>(const GemmKArgs&, int, hipStream_t);
#define DVLA_EXTERN_RING(RC, AT, BT, EPI) extern template void launch_ring_one<RC, AT, BT, EPI>(const GemmKArgs&, int, hipStream_t);
#define DVLA_EXTERN_PHASE(AT, BT, EPI, DBG) extern template void launch_phase_one<AT, BT, EPI, DBG>(const GemmKArgs&, int, hipStream_t);
#include "gemm_inst_list.h"
#undef DVLA_EXTERN_REG
#undef DVLA_EXTERN_RING
#undef DVLA_EXTERN_PHASE
}  // namespace dvla_gemm

using namespace dvla_gemm;
namespace {
// split-K reduction: C[m,n] (+)= sum_s ws[s][m][n].  HBM-bound (splits x 4 B read + 2 or 4 B written per element): a thread
// owns 8 consecutive n of one row -- two 16-byte loads per slice, one 16-byte bf16 store (or two fp32 stores); the scalar
// fallback takes ragged / unaligned outputs.  (Round 1's one-element-per-thread version ran at ~1.3 TB/s: 2.5 ms per step.)
// The k-sum partials of the same launch (dvla.h ksum_*) are reduced by extra workgroups of the same kernel (blockIdx >= main):
// no second reduction launch per weight gradient.
struct KsumJob { const float* partial; void* out; int64_t len; int rows; int out_bf16; unsigned main_blocks; };
// a tail workgroup = 16 columns x 16 row parts (a thread adds rows / 16 partial rows: one batch of loads in flight, not a chain
// of `rows` dependent ones), combined through LDS in a fixed order: deterministic
__device__ __forceinline__ bool ksum_tail(const KsumJob& kj) {
  if (blockIdx.x < kj.main_blocks) return false;
  __shared__ float red[16][16];
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int64_t c = (int64_t)(blockIdx.x - kj.main_blocks) * 16 + cl;
  float s = 0.f;
  if (c < kj.len)
#pragma unroll 4
    for (int r = part; r < kj.rows; r += 16) s += kj.partial[(int64_t)r * kj.len + c];
  red[part][cl] = s;
  __syncthreads();
  if (part == 0 && c < kj.len) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][cl];
    if (kj.out_bf16) reinterpret_cast<bf16_t*>(kj.out)[c] = f2bf(tot); else reinterpret_cast<float*>(kj.out)[c] = tot;
  }
  return true;
}
__global__ void splitk_reduce_vec_kernel(const float* __restrict__ ws, void* C, int64_t ldc, int c_f32, int accumulate,
                                         int64_t M, int64_t N, int splits, KsumJob kj) {
  if (ksum_tail(kj)) return;
  const int64_t octs = N >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * octs) return;
  const int64_t m = idx / octs, n = (idx - m * octs) << 3;
  const int64_t MN = M * N;
  const float* src = ws + m * N + n;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  for (int k = 0; k < splits; ++k) {          // fixed order: deterministic
    const float4 x = *reinterpret_cast<const float4*>(src + (int64_t)k * MN);
    const float4 y = *reinterpret_cast<const float4*>(src + (int64_t)k * MN + 4);
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
  }
  if (c_f32) {
    float* c = reinterpret_cast<float*>(C) + m * ldc + n;
    if (accumulate) {
      const float4 o0 = *reinterpret_cast<const float4*>(c), o1 = *reinterpret_cast<const float4*>(c + 4);
      a.x += o0.x; a.y += o0.y; a.z += o0.z; a.w += o0.w; b.x += o1.x; b.y += o1.y; b.z += o1.z; b.w += o1.w;
    }
    *reinterpret_cast<float4*>(c) = a;
    *reinterpret_cast<float4*>(c + 4) = b;
  } else {
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(C) + m * ldc + n) =
        make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
  }
}
// the same reduction with the GEMM's epilogue on the sums: bias -> activation -> (+ residual, joined to the ROUNDED branch value
// for bf16 outputs, like epilogue_oct) -> store.  8 columns per thread, N % 8 == 0, 16-byte aligned operands.
struct SplitEpi {
  const void* bias; int bias_f32; int act;
  const bf16_t* residual; int64_t ld_res; int64_t res_rows;
};
__global__ void splitk_reduce_epi_kernel(const float* __restrict__ ws, void* C, int64_t ldc, int c_f32, int64_t M, int64_t N,
                                         int splits, SplitEpi e) {
  const int64_t octs = N >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * octs) return;
  const int64_t m = idx / octs, n = (idx - m * octs) << 3;
  const int64_t MN = M * N;
  const float* src = ws + m * N + n;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < splits; ++k) {          // fixed order: deterministic
    const float4 x = *reinterpret_cast<const float4*>(src + (int64_t)k * MN);
    const float4 y = *reinterpret_cast<const float4*>(src + (int64_t)k * MN + 4);
    v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w; v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
  }
  if (e.bias) {
    if (e.bias_f32) {
      const float4 b0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(e.bias) + n);
      const float4 b1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(e.bias) + n + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    } else {
      const uint4 b = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(e.bias) + n);
      const uint32_t w[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[2 * i] += __uint_as_float(w[i] << 16); v[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
    }
  }
  act_fwd8(v, e.act);
  if (e.residual) {
    if (!c_f32) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = bf2f(f2bf(v[i]));
    }
    const int64_t rr = e.res_rows > 0 ? (int64_t)((uint32_t)m % (uint32_t)e.res_rows) : m;
    const uint4 r = *reinterpret_cast<const uint4*>(e.residual + rr * e.ld_res + n);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] += __uint_as_float(w[i] << 16); v[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
  }
  if (c_f32) {
    float* c = reinterpret_cast<float*>(C) + m * ldc + n;
    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(C) + m * ldc + n) =
        make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
  }
}
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, void* C, int64_t ldc, int c_f32, int accumulate,
                                     int64_t M, int64_t N, int splits, KsumJob kj) {
  if (ksum_tail(kj)) return;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * N) return;
  const int64_t m = idx / N, n = idx % N;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += ws[(int64_t)k * M * N + idx];
  if (c_f32) {
    float* c = reinterpret_cast<float*>(C) + m * ldc + n;
    *c = accumulate ? (*c + s) : s;
  } else {
    reinterpret_cast<bf16_t*>(C)[m * ldc + n] = f2bf(s);
  }
}

int g_gemm_variant = -1;   // 0 auto; force: 2 register-staged S, 4 / 5 / 6 / 7 ring 256^2 / 256x128 / 128^2 / 256x128 BK 64; 
inline int gemm_variant() {
  if (g_gemm_variant < 0) {
    const char* e = getenv("DVLA_GEMM_VARIANT");
    g_gemm_variant = e ? atoi(e) : 0;
  }
  return g_gemm_variant;
}


inline bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }


// tile counts of a configuration + the reciprocals ring_item decodes with (d == 1: 2^32 - 1, the correction makes it exact)
template <int BM, int BN, int GH>
void set_tiles(GemmKArgs& a) {
  a.tiles_m = (int)((a.M + BM - 1) / BM);
  a.tiles_n = (int)((a.N + BN - 1) / BN);
  auto inv = [](int64_t d) { return d <= 1 ? 0xffffffffu : (uint32_t)((1ull << 32) / (uint64_t)d); };
  a.inv_ntiles = inv((int64_t)a.tiles_m * a.tiles_n);
  a.inv_per_panel = inv((int64_t)GH * a.tiles_n);
}

template <class CF>
void launch_cfg(GemmKArgs& a, int combo, int split_k, hipStream_t stream) {
  a.tiles_m = (int)((a.M + CF::BM - 1) / CF::BM);
  a.tiles_n = (int)((a.N + CF::BN - 1) / CF::BN);
  switch (combo) {
    case 0: launch_one<CF, false, false>(a, split_k, stream); break;
    case 1: launch_one<CF, false, true>(a, split_k, stream); break;
    case 2: launch_one<CF, true, false>(a, split_k, stream); break;
    default: launch_one<CF, true, true>(a, split_k, stream); break;
  }
}

template <class RC, bool AT, bool BT>
void launch_ring_epi(const GemmKArgs& a, int split_k, hipStream_t stream) {
  switch (epi_class(a)) {
    case EPI_P0: launch_ring_one<RC, AT, BT, EPI_P0>(a, split_k, stream); break;
    case EPI_P_ERF: launch_ring_one<RC, AT, BT, EPI_P_ERF>(a, split_k, stream); break;
    case EPI_P_TANH: launch_ring_one<RC, AT, BT, EPI_P_TANH>(a, split_k, stream); break;
    case EPI_A0: launch_ring_one<RC, AT, BT, EPI_A0>(a, split_k, stream); break;
    case EPI_A_ERF: launch_ring_one<RC, AT, BT, EPI_A_ERF>(a, split_k, stream); break;
    case EPI_A_TANH: launch_ring_one<RC, AT, BT, EPI_A_TANH>(a, split_k, stream); break;
    case EPI_F32: launch_ring_one<RC, AT, BT, EPI_F32>(a, split_k, stream); break;
    default: launch_ring_one<RC, AT, BT, EPI_GEN>(a, split_k, stream); break;
  }
}
template <class RC>
void launch_ring(GemmKArgs& a, int combo, int split_k, hipStream_t stream) {
  set_tiles<RC::BM, RC::BN, RC::GH>(a);
  switch (combo) {
    case 0: launch_ring_epi<RC, false, false>(a, split_k, stream); break;
    case 1: launch_ring_epi<RC, false, true>(a, split_k, stream); break;
    case 2: launch_ring_epi<RC, true, false>(a, split_k, stream); break;
    default: launch_ring_epi<RC, true, true>(a, split_k, stream); break;
  }
}
template <bool AT, bool BT>
void launch_phase_epi(const GemmKArgs& a, int split_k, hipStream_t stream) {
  switch (epi_class(a)) {
    case EPI_P0: launch_phase_one<AT, BT, EPI_P0, 0>(a, split_k, stream); break;
    case EPI_P_ERF: launch_phase_one<AT, BT, EPI_P_ERF, 0>(a, split_k, stream); break;
    case EPI_P_TANH: launch_phase_one<AT, BT, EPI_P_TANH, 0>(a, split_k, stream); break;
    case EPI_A0: launch_phase_one<AT, BT, EPI_A0, 0>(a, split_k, stream); break;
    case EPI_A_ERF: launch_phase_one<AT, BT, EPI_A_ERF, 0>(a, split_k, stream); break;
    case EPI_A_TANH: launch_phase_one<AT, BT, EPI_A_TANH, 0>(a, split_k, stream); break;
    case EPI_F32: launch_phase_one<AT, BT, EPI_F32, 0>(a, split_k, stream); break;
    default: launch_phase_one<AT, BT, EPI_GEN, 0>(a, split_k, stream); break;
  }
}
// ---- stream-K scratch: one 256-KiB fp32 slab and one flag per workgroup, per (device, stream); allocated on first use
// (never while the stream is being captured: such launches take the plain schedule), kept for the life of the process.
// Flags are zero between launches (the owner of a shared tile lowers the flag it waited for).
struct SkScratch { int dev; hipStream_t stream; float* slabs; unsigned* flags; int groups; };
static SkScratch g_sk[16];
static int g_sk_n = 0;
static std::mutex g_sk_mu;
static int g_streamk = -1;   // -1: read DVLA_GEMM_STREAMK at first use; dvla_set_gemm_schedule overrides
static bool streamk_enabled() {
  if (g_streamk < 0) { const char* e = getenv("DVLA_GEMM_STREAMK"); g_streamk = (e && e[0] == '0') ? 0 : 1; }
  return g_streamk == 1;
}
static bool sk_scratch(hipStream_t stream, int groups, float** slabs, unsigned** flags) {
  if (!streamk_enabled()) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lk(g_sk_mu);
  for (int i = 0; i < g_sk_n; ++i)
    if (g_sk[i].dev == dev && g_sk[i].stream == stream && g_sk[i].groups >= groups) { *slabs = g_sk[i].slabs; *flags = g_sk[i].flags; return true; }
  if (g_sk_n == 16) return false;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (st != hipStreamCaptureStatusNone) return false;
  SkScratch e{dev, stream, nullptr, nullptr, groups};
  if (hipMalloc(reinterpret_cast<void**>(&e.slabs), (size_t)groups * PCfg::BM * PCfg::BN * 4) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMalloc(reinterpret_cast<void**>(&e.flags), (size_t)(groups + 1) * 4) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(e.slabs); return false; }
  if (hipMemsetAsync(e.flags, 0, (size_t)(groups + 1) * 4, stream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(e.slabs); (void)hipFree(e.flags); return false; }
  g_sk[g_sk_n++] = e;
  *slabs = e.slabs; *flags = e.flags;
  return true;
}
// SUPER-tiles (16 consecutive tile ids each) the stream-K part of the hybrid schedule covers: what is left beyond whole
// rounds, plus one round (so that a group's range is at least one super-tile long and a tile is shared by at most two
// workgroups); 0 = the plain schedule is already even / the problem is smaller than one round / odd CU count
// full = true ("full stream-K", variant 10): EVERY super-tile is part of the K-iteration ranges.  Same number of shared tiles
// (one per group boundary), but the ranges of different groups now cross tile boundaries at different times for the whole
// launch: the epilogues (an HBM write burst of 128-256 KiB per CU that the plain rounds issue from all 256 CUs at once)
// spread out under the other groups' main loops.
inline int streamk_tiles(int64_t tiles, int cus, bool full = false) {
  if (cus % 16 != 0) return 0;
  const int64_t groups = cus / 16, st = (tiles + 15) / 16;
  if (st < groups || st % groups == 0) return 0;
  if (full) return (int)st;
  const int64_t dp_rounds = st / groups - 1;
  return (int)(st - dp_rounds * groups);
}
// stream_k: 0 = plain schedule, 1 = hybrid (partial rounds only), 2 = full
void launch_phase(GemmKArgs& a, int combo, int split_k, hipStream_t stream, int stream_k = 0) {
  set_tiles<PCfg::BM, PCfg::BN, PCfg::GH>(a);
  a.sk_tiles = 0;
  if (a.K % PCfg::BKS != 0) {       // partial last K-tile (phase_tail): its own instantiation, plain schedule
    if (a.ksum_op != 0) launch_phase_one<true, true, EPI_F32, 128 | 8192>(a, split_k, stream);
    else launch_phase_one<true, true, EPI_F32, 128>(a, split_k, stream);
    return;
  }
  if (a.ksum_op != 0) {             // k-sums: the TT layout's fp32 class with the summing code (phase_ok_ksum admits nothing else)
    launch_phase_one<true, true, EPI_F32, 8192>(a, split_k, stream);
    return;
  }
  if (stream_k && split_k == 1 && a.ksum_op == 0) {     // (k-sums: one workgroup per (tile, K slice) owns a partial row)
    const int groups = num_cus();
    const int r = streamk_tiles((int64_t)a.tiles_m * a.tiles_n, groups, stream_k == 2);
    if (r > 0 && sk_scratch(stream, groups, &a.sk_slabs, &a.sk_flags)) a.sk_tiles = r;
  }
  switch (combo) {
    case 0: launch_phase_epi<false, false>(a, split_k, stream); break;
    case 1: launch_phase_epi<false, true>(a, split_k, stream); break;
    case 2: launch_phase_epi<true, false>(a, split_k, stream); break;
    default: launch_phase_epi<true, true>(a, split_k, stream); break;
  }
}
template <class RC>
bool ring_ok(const GemmKArgs& a, int combo) {
  if (!a.a_vec || !a.b_vec || !a.c_vec || !a.aux_vec || !a.epi_vec) return false;
  if (a.K < RC::BKS || a.K % RC::BKS != 0 || a.k_per_split % RC::BKS != 0 || (a.N & 63) != 0) return false;
  if ((int64_t)((a.M + RC::BM - 1) / RC::BM) * ((a.N + RC::BN - 1) / RC::BN) * a.split_k >= (1ll << 30)) return false;
  if (a.M < RC::BM || a.N < RC::BN) return false;
  if ((combo & 2) && (a.M % RC::BM != 0)) return false;   // r-contiguous A: whole row panels only
  if ((combo & 1) && (a.N % RC::BN != 0)) return false;
  // the ring kernels address a DMA piece as a 32-bit byte offset from the tile's origin (round 5): the farthest piece of a tile -- row
  // BM - 1 of a k-contiguous operand, k row BKS - 1 of an r-contiguous one -- must stay below 4 GiB whatever the leading dimension
  // (a view into a wider buffer may have a row stride of millions of elements); otherwise the register-staged kernel takes it
  // (round-5 ADVICE)
  const int64_t a_far = (int64_t)(((combo & 2) ? RC::BKS : RC::BM) - 1) * a.lda * 2 + 2 * (int64_t)((combo & 2) ? RC::BM : RC::BKS);
  const int64_t b_far = (int64_t)(((combo & 1) ? RC::BKS : RC::BN) - 1) * a.ldb * 2 + 2 * (int64_t)((combo & 1) ? RC::BN : RC::BKS);
  if (a_far >= (1ll << 32) || b_far >= (1ll << 32)) return false;
  return true;
}

// the phase kernel addresses its operands as base + 32-bit byte offset: both must span less than 4 GiB
// a contraction length that is not a multiple of the K-tile: the TT layout's fp32 class only (gemm_phase.h DBG & 128: the weight
// gradients over 20 832 tokens), whole k16-steps, at least two K-tiles
bool phase_tail(const GemmKArgs& a, int combo) {
  return a.K % PCfg::BKS != 0 && combo == 3 && epi_class(a) == EPI_F32 && a.K % 16 == 0 && a.K >= 2 * PCfg::BKS;
}
bool phase_ok(const GemmKArgs& a, int combo) {
  if (phase_tail(a, combo)) {
    GemmKArgs b = a;
    b.K = a.K - a.K % PCfg::BKS;       // every other requirement as for a whole number of K-tiles
    if (!ring_ok<PCfg>(b, combo)) return false;
  } else if (!ring_ok<PCfg>(a, combo)) return false;
  const int64_t a_rows = (combo & 2) ? a.K : a.M, b_rows = (combo & 1) ? a.K : a.N;
  return a_rows * a.lda * 2 < (1ll << 32) && b_rows * a.ldb * 2 < (1ll << 32);
}

// fraction of workgroup slots kept busy when `tiles` workgroups run `slots` at a time
inline double fill(int64_t tiles, int64_t slots) {
  const int64_t rounds = (tiles + slots - 1) / slots;
  return (double)tiles / (double)(rounds * slots);
}

}  // namespace

// what the last dvla_gemm_bf16 call of this process actually launched (tests assert that a forced configuration ran and did
// not fall back): 2 register-staged, 4 / 6 / 7 ring 256^2 / 128^2 / 256x128, 8 phase (plain schedule), 9 / 10 phase with the
// stream-K hybrid / full schedule ENGAGED (a stream-K request that does not apply reports 8); 0 = nothing launched yet
// measurement variants of the phase kernel (tests/probes/gemm_probe.cpp; NN layout, plain bf16 epilogue only): 40 = the round-4
// main loop (piece placement 13 + its issue code), 41 = its s_memtime build, 42 = placement 13 with the round-5 issue code,
// 43 = every piece behind the fragment reads' wait
static bool launch_phase_experiment(int idx, GemmKArgs& a, int split_k, hipStream_t stream) {
  switch (idx) {
    case 0: launch_phase_one<false, false, 0, 7424>(a, split_k, stream); return true;
    case 1: launch_phase_one<false, false, 0, 7488>(a, split_k, stream); return true;
    case 2: launch_phase_one<false, false, 0, 3328>(a, split_k, stream); return true;
    case 3: launch_phase_one<false, false, 0, 256>(a, split_k, stream); return true;
    case 4: launch_phase_one<false, false, 0, 32768>(a, split_k, stream); return true;            // 44: round-5 boundary, buffer-descriptor DMA
    case 5: launch_phase_one<false, false, 0, 16384 | 65536>(a, split_k, stream); return true;    // 45: deferred epilogue, fragment addresses hoisted
    case 6: launch_phase_one<false, false, 0, 16384 | 65536 | 131072>(a, split_k, stream); return true;    // 46: 45 + the in-loop epilogue at priority 2
    case 7: launch_phase_one<false, false, 0, 16384>(a, split_k, stream); return true;                      // 47: deferred epilogue as first measured (fragment addresses re-derived per K-tile)
    default: return false;
  }
}

static int g_last_variant = 0;
extern "C" int dvla_last_gemm_variant(void) { return g_last_variant; }
extern "C" void dvla_set_gemm_variant(int v) { g_gemm_variant = v; }
extern "C" void dvla_set_gemm_schedule(int oversubscribe, int stream_k) {
  if (oversubscribe >= 1) dvla_gemm::gemm_oversubscribe() = oversubscribe > 16 ? 16 : oversubscribe;
  if (stream_k >= 0) g_streamk = stream_k ? 1 : 0;
}

extern "C" void dvla_get_gemm_schedule(int* oversubscribe, int* stream_k) {
  if (oversubscribe) *oversubscribe = dvla_gemm::gemm_oversubscribe();
  if (stream_k) *stream_k = streamk_enabled() ? 1 : 0;
}

extern "C" int64_t dvla_gemm_ksum_partial_rows(int32_t split_k) {
  // (x 4: the phase kernel spreads a row's sum over up to four tiles, each with its own KSUM_PARTS partial rows)
  const int64_t fused = (int64_t)(split_k > 1 ? split_k : 1) * dvla_gemm::KSUM_PARTS * 4, fallback = dvla_colsum_partial_rows();
  return fused > fallback ? fused : fallback;
}

extern "C" int dvla_gemm_bf16(const dvla_gemm_params* q, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!q || !q->A || !q->B || !q->C) return DVLA_ERR_ARG;
  if (q->M < 0 || q->N < 0 || q->K < 0) return DVLA_ERR_ARG;
  if (q->M == 0 || q->N == 0) return DVLA_OK;
  if (q->accumulate && q->c_dtype != DVLA_DT_F32) return DVLA_ERR_ARG;
  if (q->dropout_p < 0.f || q->dropout_p >= 1.f) return DVLA_ERR_ARG;
  const int split_k = q->split_k > 1 ? q->split_k : 1;
  // split-K: the partial sums go to the workspace raw; bias / activation / residual are applied by the reduction pass
  // (splitk_reduce_epi_kernel: the forward GEMMs of the evaluation engine with few hundred rows and a long K)
  const bool split_epi = split_k > 1 && (q->bias || q->act || q->residual);
  if (split_k > 1) {
    if (!q->workspace || q->preact || q->dact_aux || q->dropout_p > 0.f) return DVLA_ERR_ARG;
    if (split_epi) {
      if (q->accumulate || q->ksum_operand != 0) return DVLA_ERR_ARG;
      const bool c_ok = q->c_dtype == DVLA_DT_F32 ? ((q->ldc % 4 == 0) && aligned(q->C, 16)) : ((q->ldc % 8 == 0) && aligned(q->C, 16));
      if ((q->N & 7) != 0 || !aligned(q->workspace, 16) || !c_ok || (q->bias && !aligned(q->bias, 16)) ||
          (q->residual && !((q->ld_res % 8 == 0) && aligned(q->residual, 16))))
        return DVLA_ERR_UNSUPPORTED;
    }
  }
  if (q->ksum_operand != 0) {
    if ((q->ksum_operand != 1 && q->ksum_operand != 2) || !q->ksum || !q->ksum_workspace) return DVLA_ERR_ARG;
    if (q->ksum_dtype != DVLA_DT_F32 && q->ksum_dtype != DVLA_DT_BF16) return DVLA_ERR_ARG;
  }
  GemmKArgs a;
  a.sk_tiles = 0; a.sk_slabs = nullptr; a.sk_flags = nullptr;
  a.ksum_ws = q->ksum_workspace; a.ksum_op = 0;   // set below, once the configuration is known to carry the summing code
  a.ksum_parts = KSUM_PARTS;
  a.a_ln = q->a_layernorm != 0; a.a_ln_eps = q->a_ln_eps;
  a.A = reinterpret_cast<const bf16_t*>(q->A); a.lda = q->lda;
  a.B = reinterpret_cast<const bf16_t*>(q->B); a.ldb = q->ldb;
  a.C = q->C; a.ldc = q->ldc; a.c_f32 = (q->c_dtype == DVLA_DT_F32);
  a.M = q->M; a.N = q->N; a.K = q->K;
  a.bias = q->bias; a.bias_f32 = (q->bias_dtype == DVLA_DT_F32);
  a.act = q->act;
  a.preact = reinterpret_cast<bf16_t*>(q->preact); a.ld_preact = q->ld_preact;
  a.dact_aux = reinterpret_cast<const bf16_t*>(q->dact_aux); a.ld_dact = q->ld_dact; a.dact = q->dact;
  a.has_drop = q->dropout_p > 0.f;
  a.drop_scale = a.has_drop ? 1.0f / (1.0f - q->dropout_p) : 1.0f;
  {
    double thr = (double)q->dropout_p * 4294967296.0;
    a.drop_thr = thr >= 4294967295.0 ? 4294967295u : (uint32_t)thr;
  }
  a.seed_lo = q->seed_lo; a.seed_hi = q->seed_hi;
  a.residual = reinterpret_cast<const bf16_t*>(q->residual); a.ld_res = q->ld_res; a.res_rows = q->res_rows;
  a.accumulate = q->accumulate;
  a.split_k = split_k; a.workspace = reinterpret_cast<float*>(q->workspace);
  {
    int64_t nkt = (q->K + BK - 1) / BK;
    int64_t per = (nkt + split_k - 1) / split_k;
    a.k_per_split = per * BK;
  }
  a.a_vec = (q->lda % 8 == 0) && aligned(q->A, 16);
  a.b_vec = (q->ldb % 8 == 0) && aligned(q->B, 16);
  a.c_vec = a.c_f32 ? ((q->ldc % 4 == 0) && aligned(q->C, 16)) : ((q->ldc % 8 == 0) && aligned(q->C, 16));
  a.aux_vec = 1;
  if (q->preact && !((q->ld_preact % 8 == 0) && aligned(q->preact, 16))) a.aux_vec = 0;
  if (q->dact_aux && !((q->ld_dact % 8 == 0) && aligned(q->dact_aux, 16))) a.aux_vec = 0;
  if (q->residual && !((q->ld_res % 8 == 0) && aligned(q->residual, 16))) a.aux_vec = 0;
  if (split_epi) { a.bias = nullptr; a.act = 0; a.residual = nullptr; }      // (the main kernel writes raw partial sums)
  a.epi_vec = 1;  // bias / split-K workspace vector access
  if (q->bias && !aligned(q->bias, 16)) a.epi_vec = 0;
  if (split_k > 1 && ((q->N & 3) != 0 || !aligned(q->workspace, 16))) a.epi_vec = 0;

  const int combo = (q->a_trans ? 2 : 0) | (q->b_trans ? 1 : 0);
  int variant = gemm_variant();
  if (variant >= 100) variant %= 100;   // (hundreds digit: reserved for A/B knobs of measurement builds)
  {
    // Configuration choice: cheapest by a two-constant model per configuration, T = rounds * (fixed + stage * ns),
    // rounds = ceil(work items / workgroup slots), ns = 32-wide K stages per item.  Constants in microseconds per round,
    // measured on MI355X with tests/probes/gemm_probe.cpp (K sweep at 20832 x 4096, plain epilogue, round 2:
    // profiles/r02_gemm_probe_final.txt): with the register-only epilogue the fixed cost per round is 2.9 / 3.6 / 5.8 / 6.6 us
    // for 128^2 / 256x128 BK64 / 256^2 / phase (it was 5.7 / 5.9 / 18.7 with the LDS-patch epilogue of round 1), the slope per
    // stage 0.58 / 0.49 / 0.85 / 0.775.  The online tuner (dreamvla_amd.ops.GemmTuner) refines this per problem key in-model.
    // The register-staged kernel is the fallback for shapes the DMA kernels do not take (ragged N, unaligned operands, tiny).
    int choice = 0;   // 0 = register-staged S (always valid), 1 = ring L, 3 = ring S, 4 = ring M64, 5 = phase, 11 = skinny
    if (a.a_ln && !(skinny_ok(a, combo, split_k) && skinny_ln_ok(a))) return DVLA_ERR_UNSUPPORTED;   // only the few-rows kernel normalises
    if ((variant == 0 || variant == 11 || a.a_ln) && skinny_ok(a, combo, split_k)) {
      choice = 11;      // few rows (evaluation-time shapes): one 32 x 32 tile per workgroup, K split over its four waves
    } else if (variant == 0) {
      const double ns = (double)(a.k_per_split < a.K ? a.k_per_split : a.K) / 32.0;
      const int slots = num_cus();
      double best = 1e30;
      auto consider = [&](int c, int64_t bm, int64_t bn, int wg_per_cu, double fixed_us, double stage_us) {
        const int64_t items = ((q->M + bm - 1) / bm) * ((q->N + bn - 1) / bn) * split_k;
        const int64_t rounds = (items + (int64_t)slots * wg_per_cu - 1) / ((int64_t)slots * wg_per_cu);
        const double t = (double)rounds * (fixed_us + stage_us * ns);
        if (t < best) { best = t; choice = c; }
      };
      if (ring_ok<RCfgS>(a, combo)) consider(3, 128, 128, 2, 2.9, 0.58);
      if (ring_ok<RCfgM64>(a, combo)) consider(4, 256, 128, 1, 3.6, 0.49);
      if (ring_ok<RCfgL>(a, combo)) consider(1, 256, 256, 1, 5.8, 0.85);
      if (phase_ok(a, combo)) {
        consider(5, 256, 256, 1, 6.6, 0.775);
        // stream-K hybrid of the same kernel: fractional rounds, plus one slab write + one slab read per workgroup and the
        // less regular operand reuse of the shared round (~18 us measured over the plain schedule at equal round counts)
        const int64_t tiles = ((q->M + 255) / 256) * ((q->N + 255) / 256);
        if (split_k == 1 && streamk_enabled() && streamk_tiles(tiles, slots) > 0) {
          const double t = (double)tiles / slots * (6.6 + 0.775 * ns) + 18.0;
          if (t < best) { best = t; choice = 6; }
        }
      }
    } else if (variant == 4 && ring_ok<RCfgL>(a, combo)) choice = 1;
    else if (variant == 6 && ring_ok<RCfgS>(a, combo)) choice = 3;
    else if (variant == 7 && ring_ok<RCfgM64>(a, combo)) choice = 4;
    else if (variant == 8 && phase_ok(a, combo)) choice = 5;
    else if (variant == 9 && phase_ok(a, combo)) choice = 6;
    else if (variant == 10 && phase_ok(a, combo)) choice = 7;
    else if (variant > 80 && variant < 90 && combo == 0 && phase_ok(a, combo)) choice = 80 + (variant - 80);
    else if (variant >= 40 && variant < 48 && combo == 0 && phase_ok(a, combo) && epi_class(a) == EPI_P0) choice = variant;
    // k-sums ride on the ring kernels and (round 5) on the phase kernel of the fp32-output class (split-K partial sums or fp32 C:
    // the weight gradients); the other configurations get the column-sum kernel below
    const bool phase_choice = choice == 5 || choice == 6 || choice == 7;
    const bool ksum_fused = q->ksum_operand != 0 && epi_class(a) == EPI_F32 &&
                            (choice == 1 || choice == 3 || choice == 4 || (phase_choice && combo == 3));   // phase: the TT layout (every weight gradient)
    if (ksum_fused) {
      a.ksum_op = q->ksum_operand;
      if (phase_choice) {          // tiles of a tile row (A sums) / tile column (B sums) that share the sum: up to four
        const int64_t other = q->ksum_operand == 1 ? (q->N + PCfg::BN - 1) / PCfg::BN : (q->M + PCfg::BM - 1) / PCfg::BM;
        a.ksum_parts = KSUM_PARTS * (int)(other < 4 ? other : 4);
      }
    }
    switch (choice) {
      case 1: launch_ring<RCfgL>(a, combo, split_k, stream); break;
      case 3: launch_ring<RCfgS>(a, combo, split_k, stream); break;
      case 4: launch_ring<RCfgM64>(a, combo, split_k, stream); break;
      case 5: launch_phase(a, combo, split_k, stream); break;
      case 6: launch_phase(a, combo, split_k, stream, 1); break;   // falls back to the plain schedule when stream-K does not apply
      case 7: launch_phase(a, combo, split_k, stream, 2); break;
      case 11: launch_skinny(a, stream); break;
      case 81: case 83: case 84: case 85: case 86: case 88: case 89:   // ablations (plain bf16 epilogue only): 81 no MFMA, 83 no MFMA + no reads, 84 no DMA, 85 no epilogue, 86 raw stores only
        set_tiles<PCfg::BM, PCfg::BN, PCfg::GH>(a);
        if (epi_class(a) != EPI_P0) return DVLA_ERR_UNSUPPORTED;
        if (choice == 81) launch_phase_one<false, false, 0, 1>(a, split_k, stream);
        else if (choice == 83) launch_phase_one<false, false, 0, 3>(a, split_k, stream);
        else if (choice == 84) launch_phase_one<false, false, 0, 4>(a, split_k, stream);
        else if (choice == 85) launch_phase_one<false, false, 0, 16>(a, split_k, stream);
        else if (choice == 86) launch_phase_one<false, false, 0, 32>(a, split_k, stream);
        else if (choice == 88) launch_phase_one<false, false, 0, 16448>(a, split_k, stream);   // 88: the stamps build of the deferred epilogue
        else launch_phase_one<false, false, 0, 64>(a, split_k, stream);   // 89: s_memtime stamps into p.workspace
        break;
      default:
        if (choice >= 40 && choice < 48) {
          set_tiles<PCfg::BM, PCfg::BN, PCfg::GH>(a);
          a.sk_tiles = 0;
          if (launch_phase_experiment(choice - 40, a, split_k, stream)) break;
        }
        launch_cfg<CfgS>(a, combo, split_k, stream); break;
    }
    g_last_variant = choice == 11 ? 11 : choice == 1 ? 4 : choice == 3 ? 6 : choice == 4 ? 7 : choice == 5 ? 8
                   : (choice == 6 || choice == 7) ? (a.sk_tiles > 0 ? (choice == 6 ? 9 : 10) : 8) : choice >= 80 ? choice : 2;
  }
  int rc = dvla_check_launch();
  if (rc != DVLA_OK) return rc;
  KsumJob kj{nullptr, nullptr, 0, 0, 0, 0xffffffffu};
  if (q->ksum_operand != 0) {
    const int64_t len = q->ksum_operand == 1 ? q->M : q->N;
    if (a.ksum_op != 0) {
      if (split_k > 1) kj = KsumJob{a.ksum_ws, q->ksum, len, split_k * a.ksum_parts, q->ksum_dtype == DVLA_DT_BF16, 0u};   // rides on the split-K reduction below
      else rc = dvla_reduce_partial_rows(a.ksum_ws, split_k * a.ksum_parts, len, len, q->ksum, q->ksum_dtype == DVLA_DT_BF16, stream);
    } else {
      // this configuration does not sum: the column-sum kernel over the operand as it lies in memory (k-major layouts only)
      const bool kmajor = q->ksum_operand == 1 ? q->a_trans != 0 : q->b_trans != 0;
      if (!kmajor) return DVLA_ERR_UNSUPPORTED;
      rc = dvla_colsum_dt(q->ksum_operand == 1 ? q->A : q->B, q->ksum_operand == 1 ? q->lda : q->ldb, q->K, len, q->ksum,
                          q->ksum_dtype, q->ksum_workspace, stream_);
    }
    if (rc != DVLA_OK) return rc;
  }
  if (split_k > 1) {
    const int64_t total = q->M * q->N;
    const bool vec = (q->N & 7) == 0 && aligned(a.workspace, 16) && aligned(q->C, 16) &&
                     (a.c_f32 ? (q->ldc & 3) == 0 : (q->ldc & 7) == 0);
    const unsigned main_blocks = (unsigned)(((vec ? total / 8 : total) + 255) / 256);
    const unsigned tail_blocks = kj.partial ? (unsigned)((kj.len + 15) / 16) : 0u;
    if (kj.partial) kj.main_blocks = main_blocks;
    if (split_epi) {
      SplitEpi e;
      e.bias = q->bias; e.bias_f32 = q->bias_dtype == DVLA_DT_F32; e.act = q->act;
      e.residual = reinterpret_cast<const bf16_t*>(q->residual); e.ld_res = q->ld_res; e.res_rows = q->res_rows;
      hipLaunchKernelGGL(splitk_reduce_epi_kernel, dim3(main_blocks), dim3(256), 0, stream, a.workspace, q->C, q->ldc, a.c_f32, q->M,
                         q->N, split_k, e);
    } else if (vec)
      hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3(main_blocks + tail_blocks), dim3(256), 0, stream,
                         a.workspace, q->C, q->ldc, a.c_f32, q->accumulate, q->M, q->N, split_k, kj);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(main_blocks + tail_blocks), dim3(256), 0, stream,
                         a.workspace, q->C, q->ldc, a.c_f32, q->accumulate, q->M, q->N, split_k, kj);
    rc = dvla_check_launch();
  }
  return rc;
}

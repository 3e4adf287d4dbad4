// gemm.hip -- bf16 MFMA GEMM with fused epilogue for gfx950 (MI355X).
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T ),   fp32 accumulate on v_mfma_f32_32x32x16_bf16
//
// Two tile configurations of ONE kernel template (K-tile = 64 in both):
//   S : 128x128 block tile, 4 waves (2x2), 64x64 per wave (2x2 MFMA tiles),  64 KiB LDS, 2 workgroups / CU   [default]
//   L : 256x256 block tile, 8 waves (2x4), 128x64 per wave (4x2 MFMA tiles), 128 KiB LDS, 1 workgroup / CU
// Ablation on MI355X (profiles/r01_gemm_ablation.txt): the S kernel without MFMAs takes 96 % of the full kernel's time
// and without global loads / LDS writes 58 %, i.e. it is bound by the operand staging path (~22 B/clk/CU from L2), not by
// the matrix pipe.  The L tile halves that traffic per flop but currently loses more to its single staging set and
// 1-workgroup occupancy than it gains (805 vs 880 TFLOP/s at 8192^3), so it is opt-in (DVLA_GEMM_VARIANT=3).
//
// Operands are staged global -> registers -> LDS.  Two register sets per operand: while tile kt is multiplied out of
// LDS, tile kt+1 waits in one set (written to the other LDS buffer after the MFMAs) and the 16-byte global loads of
// tile kt+2 are already in flight into the other set (the steady-state loop is branch-free so the compiler can wait
// with a counted s_waitcnt vmcnt(8) instead of draining the queue).  One barrier per K-tile.
//
// Both operands may have either memory order (the autograd backward GEMMs need every combination):
//   "k-contiguous" : element (r,k) at P[r*ld + k] -> swizzled row-major LDS image (rm_off), fragments by one
//                    conflict-free ds_read_b128 per lane.
//   "r-contiguous" : element (r,k) at P[k*ld + r] -> LDS "quad-interleaved" 8-byte units [k/4][rows]
//                    (unit = {k..k+3} of one row): each thread transposes a 4(k) x 8(rows) block in
//                    registers between its four coalesced 16-B global loads and four ds_write_b128;
//                    fragments by two conflict-free ds_read_b64 per lane.
// The MFMA k-slot <-> k mapping is the same for both layouts (slot (g,j) <-> k = 16*ks + 8*g + j).
//
// The MFMA is issued as mfma(a = B-operand fragment (n), b = A-operand fragment (m)) so that a lane
// owns ONE output row m and 4 consecutive n per accumulator quad; the epilogue transposes each 32-row
// slab through a wave-private fp32 LDS patch so a lane ends up with 8 consecutive n of one row and
// reads bias / residual / aux and writes C with 16-byte accesses.
//
// Measured and rejected (kept out of the tree): LDS-DMA staging (global_load_lds_dwordx4) of the k-contiguous operand
// with a 2-buffer ring was 8-12 % SLOWER than the 2-deep register prefetch (the barrier drains vmcnt(0)).
#include <stdlib.h>

#pragma once
#include "common.h"
#include "../../include/dvla.h"
#include <type_traits>

namespace dvla_gemm {

constexpr int BK = 64;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).  `#pragma unroll` is only a request
// (the optimizer declines it for the large epilogue body) and a rolled loop would index the accumulator array
// dynamically, i.e. put 128 accumulator registers into scratch.
template <int V> struct IntC { static constexpr int value = V; };
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(IntC<N - 1>{});
  }
}

template <int WM_, int WN_, int TM_, int TN_, bool DEEP_>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
  static constexpr bool DEEP = DEEP_;   // two register sets (prefetch distance 2) vs one (distance 1: fewer VGPRs)
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + B_BYTES;
  static constexpr int SMEM_BYTES = 2 * BUF_BYTES;
  static_assert(NT == 2 * BM && NT == 2 * BN, "staging maps assume 4 x 16 B per thread and operand");
  static_assert(TN == 2, "epilogue patch is 64 columns wide");
};
using CfgS = Cfg<2, 2, 2, 2, true>;   // 128 x 128, 256 threads

struct GemmKArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* B; int64_t ldb;
  void* C; int64_t ldc; int c_f32;
  int64_t M, N, K;
  const void* bias; int bias_f32;
  int act;
  bf16_t* preact; int64_t ld_preact;
  const bf16_t* dact_aux; int64_t ld_dact; int dact;
  float drop_scale; uint32_t drop_thr; uint32_t seed_lo, seed_hi; int has_drop;
  const bf16_t* residual; int64_t ld_res; int64_t res_rows;
  int accumulate;
  int split_k; int64_t k_per_split; float* workspace;
  int a_vec, b_vec, c_vec, aux_vec, epi_vec;
  int tiles_m, tiles_n;
  // floor(2^32 / (tiles_m * tiles_n)) and floor(2^32 / (GH * tiles_n)) of the launching configuration (set_tiles): the
  // ring / phase kernels decode a work item with a multiply-high and one correction instead of integer divisions
  uint32_t inv_ntiles, inv_per_panel;
  // stream-K hybrid schedule of the phase kernel (gemm_phase.h): the first sk_tiles tiles are cut into gridDim.x equal
  // K-iteration ranges, the others run one tile per workgroup and round; 0 = plain schedule
  int sk_tiles; float* sk_slabs; unsigned* sk_flags;
  // k-sums of an operand (dvla.h ksum_*): fp32 partials [split][KSUM_PARTS][len], len = M (op 1) or N (op 2); EPI_F32 kernels only
  float* ksum_ws; int ksum_op;
  int ksum_parts;     // partial rows per K slice: KSUM_PARTS (ring kernels), KSUM_PARTS x (tiles that share a row's sum) (phase kernel)
  // rows of A layer-normalised on the fly (dvla.h a_layernorm): few-rows kernel only
  int a_ln; float a_ln_eps;
};
constexpr int KSUM_PARTS = 8;   // (up to 4 waves that share the operand rows, each taking every 4th k16-step) x (2 half-waves)

// acc += sel . (the eight bf16 of a fragment), sel = two bf16 ones or 0 (adds nothing): four v_dot2c_f32_bf16 as pure builtins, no
// branch -- in the compiler-scheduled ring loop a volatile asm or a branch between the k16-steps keeps the scheduler from
// overlapping the next step's fragment reads with this step's MFMAs (measured: +10 % per launch).  (Element pairs spelled out:
// indexing a bit-cast uint32x4 copy of f inside an unrolled loop made hipcc use dword 0 four times.)
__device__ __forceinline__ float ksum_add_sel(bf16x8 f, uint32_t sel, float acc) {
  typedef __attribute__((ext_vector_type(2))) short s2;
  const hw_bf16x2 o = __builtin_bit_cast(hw_bf16x2, sel);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_bf16x2, s2{f[0], f[1]}), o, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_bf16x2, s2{f[2], f[3]}), o, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_bf16x2, s2{f[4], f[5]}), o, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_bf16x2, s2{f[6], f[7]}), o, acc, false);
  return acc;
}
// a wave's k-sums of 32-row blocks -> the partial buffer.  pgroup < nshare: which of the nshare waves that hold the same operand
// rows this is; the pgroups nobody owns (nshare .. 3) are zero-filled by pgroup 0.
// part0 (phase kernel): first of this tile's KSUM_PARTS partial rows inside the K slice's p.ksum_parts.
template <int NB, int NS>
__device__ __forceinline__ void ksum_store(const GemmKArgs& p, const float (&sum)[NS], int lane, int split, int pgroup, int nshare,
                                           int64_t idx0, int64_t len, int part0 = 0) {
  static_assert(NB <= NS, "block count");
  const int l31 = lane & 31, g = lane >> 5;
  const int64_t row0 = (int64_t)split * p.ksum_parts + part0;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int64_t idx = idx0 + b * 32 + l31;
    if (idx < len) {
      p.ksum_ws[(row0 + pgroup * 2 + g) * len + idx] = sum[b];
      if (pgroup == 0)
        for (int z = nshare; z < KSUM_PARTS / 2; ++z) p.ksum_ws[(row0 + z * 2 + g) * len + idx] = 0.f;
    }
  }
}

// Row-major image of a k-contiguous operand: row r = 128 B = 8 slots of 16 B; k-octet o of row r lives in slot
// o ^ ((r >> 1) & 7).  Unpadded and conflict-free for ds_read_b128: in every 16-lane read group the rows of equal
// parity have distinct (r >> 1) & 7.
__device__ __forceinline__ int rm_off(int row, int oct) { return row * 128 + ((oct ^ ((row >> 1) & 7)) << 4); }

// ---- staging: global -> registers (4 x 16 B per thread and operand; ROWS = tile rows, NT = 2 * ROWS threads) ----
// k-contiguous operand: thread t loads rows (t>>3) + (NT/8)*i (i = 0..3), k-octet (t&7): 8 lanes = one 128-B row segment
// r-contiguous operand: thread t loads k = 4*(t / (ROWS/8)) + kk (kk = 0..3), rows (t % (ROWS/8))*8 .. +7
template <bool TRANS, int ROWS>
__device__ __forceinline__ void stage_load(uint4 (&reg)[4], const bf16_t* __restrict__ P, int64_t ld, int64_t row0,
                                           int64_t rows, int64_t k0, int64_t k_end, bool vec_ok, int t) {
  if (!TRANS) {
    const int r = t >> 3, kc = (t & 7) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) reg[i] = load8_guard(P, ld, row0 + r + (ROWS / 4) * i, k0 + kc, rows, k_end, vec_ok);
  } else {
    const int kq = t / (ROWS / 8), r0 = (t % (ROWS / 8)) * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) reg[kk] = load8_guard(P, ld, k0 + 4 * kq + kk, row0 + r0, k_end, rows, vec_ok);
  }
}

// Fast path (interior K-tile of a 16-B-vectorisable operand): four unconditional 16-B loads from per-thread pointers that
// simply advance by one K-tile per iteration.  Rows past the end of a k-contiguous operand are clamped to the last row
// (their products land in output rows the epilogue never stores); an r-contiguous operand takes the fast path only when
// the whole row panel is in range.  Everything else (K tail, ragged / unaligned operands) goes through stage_load.
template <bool TRANS, int ROWS>
__device__ __forceinline__ void fast_ptrs(const bf16_t* (&ptr)[4], const bf16_t* __restrict__ P, int64_t ld, int64_t row0,
                                          int64_t rows, int64_t k0, int t) {
  if (!TRANS) {
    const int r = t >> 3, kc = (t & 7) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t row = row0 + r + (ROWS / 4) * i;
      row = row < rows ? row : rows - 1;
      ptr[i] = P + row * ld + k0 + kc;
    }
  } else {
    const int kq = t / (ROWS / 8), r0 = (t % (ROWS / 8)) * 8;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ptr[kk] = P + (k0 + 4 * kq + kk) * ld + row0 + r0;
  }
}
template <bool TRANS>
__device__ __forceinline__ void fast_load(uint4 (&reg)[4], const bf16_t* (&ptr)[4], int64_t ld) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    reg[i] = *reinterpret_cast<const uint4*>(ptr[i]);
    ptr[i] += TRANS ? (int64_t)BK * ld : (int64_t)BK;
  }
}

// ---- staging: registers -> LDS ---------------------------------------------------------------------
// k-contiguous: swizzled row-major image (rm_off).
// r-contiguous: "quad-interleaved" 8-byte units [k/4][ROWS] = {k, k+1, k+2, k+3} of one row; the 4(k) x 8(rows)
// register block is transposed in registers (two 16-bit merges per output dword) and leaves as four ds_write_b128.
template <bool TRANS, int ROWS>
__device__ __forceinline__ void stage_store(const uint4 (&reg)[4], char* lds, int t) {
  if (!TRANS) {
    const int r = t >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(lds + rm_off(r + (ROWS / 4) * i, t & 7)) = reg[i];
  } else {
    const int kq = t / (ROWS / 8), r0 = (t % (ROWS / 8)) * 8;
    uint32_t in[4][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { in[kk][0] = reg[kk].x; in[kk][1] = reg[kk].y; in[kk][2] = reg[kk].z; in[kk][3] = reg[kk].w; }
    uint32_t o[16];  // row r0+i -> dwords o[2i] = {k0,k1}, o[2i+1] = {k2,k3}
#pragma unroll
    for (int w = 0; w < 4; ++w) {  // dword w of every k-vector holds rows r0+2w (low half) and r0+2w+1 (high half)
      o[4 * w + 0] = (in[0][w] & 0xffffu) | (in[1][w] << 16);
      o[4 * w + 1] = (in[2][w] & 0xffffu) | (in[3][w] << 16);
      o[4 * w + 2] = (in[0][w] >> 16) | (in[1][w] & 0xffff0000u);
      o[4 * w + 3] = (in[2][w] >> 16) | (in[3][w] & 0xffff0000u);
    }
    uint4* dst = reinterpret_cast<uint4*>(lds + ((size_t)(kq * ROWS + r0)) * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
  }
}

// ---- LDS -> MFMA fragment: row `row` of the tile, k16-step ks (0..3), lane group g; slot (g,j) <-> k = 16ks+8g+j --
template <bool TRANS, int ROWS>
__device__ __forceinline__ bf16x8 frag_load(const char* lds, int row, int ks, int g) {
  if (!TRANS) {
    return *reinterpret_cast<const bf16x8*>(lds + rm_off(row, ks * 2 + g));
  } else {
    const int q0 = ks * 4 + g * 2;
    union { uint2 h[2]; bf16x8 v; } u;
    u.h[0] = *reinterpret_cast<const uint2*>(lds + ((size_t)(q0 * ROWS + row)) * 8);
    u.h[1] = *reinterpret_cast<const uint2*>(lds + ((size_t)((q0 + 1) * ROWS + row)) * 8);
    return u.v;
  }
}

__device__ __forceinline__ void unpack8f(const uint4& u, float (&f)[8]) {
  f[0] = bf2f((bf16_t)(u.x & 0xffff)); f[1] = bf2f((bf16_t)(u.x >> 16));
  f[2] = bf2f((bf16_t)(u.y & 0xffff)); f[3] = bf2f((bf16_t)(u.y >> 16));
  f[4] = bf2f((bf16_t)(u.z & 0xffff)); f[5] = bf2f((bf16_t)(u.z >> 16));
  f[6] = bf2f((bf16_t)(u.w & 0xffff)); f[7] = bf2f((bf16_t)(u.w >> 16));
}
__device__ __forceinline__ void load8_aux(const bf16_t* q, int64_t n, int64_t N, bool vec, float (&f)[8]) {
  if (vec) {
    unpack8f(*reinterpret_cast<const uint4*>(q), f);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (n + e < N) ? bf2f(q[e]) : 0.f;
  }
}
__device__ __forceinline__ void store8_bf16(bf16_t* q, int64_t n, int64_t N, bool vec, const float (&v)[8]) {
  if (vec) {
    *reinterpret_cast<uint4*>(q) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (n + e < N) q[e] = f2bf(v[e]);
  }
}

// eight consecutive outputs of row m: columns n .. n+7 (values arrive in fp32 from the LDS transpose).
// FULL = the whole octet is in range and every pointer involved is 16-B vectorisable: no per-element guards.
template <bool FULL>
__device__ __forceinline__ void epilogue_oct(const GemmKArgs& p, int64_t m, int64_t n, float (&v)[8], int split) {
  if (p.split_k > 1) {  // raw partial sums -> workspace[split][m][n]
    float* w = p.workspace + ((int64_t)split * p.M + m) * p.N + n;
    if (FULL) {
      *reinterpret_cast<float4*>(w) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(w + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) w[e] = v[e];
    }
    return;
  }
  if (p.bias) {
    if (FULL && !p.bias_f32) {
      float bv[8];
      unpack8f(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.bias) + n), bv);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bv[e];
    } else if (FULL) {
      const float4 b0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.bias) + n);
      const float4 b1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.bias) + n + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N)
          v[e] += p.bias_f32 ? reinterpret_cast<const float*>(p.bias)[n + e]
                             : bf2f(reinterpret_cast<const bf16_t*>(p.bias)[n + e]);
    }
  }
  if (p.preact) {
    store8_bf16(p.preact + m * p.ld_preact + n, n, p.N, FULL, v);
    // the activation sees the value that was stored (bf16), exactly like act(preact_tensor)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
  }
  act_fwd8(v, p.act);
  if (p.has_drop) {
    const uint32_t rowkey = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)m);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t h = drop_hash_rk(rowkey, (uint32_t)(n + e));
      v[e] = (h >= p.drop_thr) ? v[e] * p.drop_scale : 0.f;
    }
  }
  // bf16 outputs: act' / residual are applied to the ROUNDED branch value -- the reference materialises
  // `dropout(act(linear(x)))` and `dY @ W` as bf16 tensors first (same rule in the ring kernels' epilogue)
  if (!p.c_f32 && (p.dact_aux || p.residual)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
  }
  if (p.dact_aux) {
    float a[8];
    load8_aux(p.dact_aux + m * p.ld_dact + n, n, p.N, FULL, a);
    act_bwd8_mul(v, a, p.dact);
  }
  if (p.residual) {
    float a[8];
    const int64_t rr = p.res_rows > 0 ? (int64_t)((uint32_t)m % (uint32_t)p.res_rows) : m;
    load8_aux(p.residual + rr * p.ld_res + n, n, p.N, FULL, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += a[e];
  }
  if (p.c_f32) {
    float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
    if (p.accumulate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (FULL || n + e < p.N) c[e] += v[e];
    } else if (FULL) {
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) c[e] = v[e];
    }
  } else {
    store8_bf16(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n, n, p.N, FULL, v);
  }
}

// Tile epilogue shared by all kernel families.  acc[i][j][r] holds (m = 32j + l31, n = 32i + 8*(r>>2) + 4*g + (r&3)) of
// the wave's (TM*32) x 64 patch.  Each 32-row slab is transposed through a wave-private fp32 LDS patch [32][64+4] so
// that a lane then owns 8 consecutive n of one row (16-byte aux loads / C stores).  A lane's column octet is the same
// for all its rows, so its bias values are loaded once; the residual / act'-aux octets of a slab are all requested
// BEFORE the slab is transposed so their latency overlaps the LDS traffic instead of serialising 4 loads per slab.
template <int TM>
__device__ __forceinline__ void tile_epilogue(const GemmKArgs& p, f32x16 (&acc)[2][TM], char* smem, int wave, int lane,
                                              int64_t m_base, int64_t n_base, int split) {
  const bool has_dact = p.dact_aux != nullptr;
  constexpr int PATCH_LD = 68;  // floats per patch row (272 B: 16-B aligned, 4-bank skew per row)
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PATCH_LD);
  const int l31 = lane & 31, g = lane >> 5;
  const int cg = lane & 7, r8 = lane >> 3;            // this lane's column octet and row-in-group
  const int64_t n = n_base + cg * 8;
  // whole-tile fast path: every octet of this wave's columns is in range and all pointers are 16-B vectorisable
  const bool tile_full = (n_base + 64 <= p.N) && p.c_vec && p.aux_vec && p.epi_vec;
  float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (tile_full && p.bias && p.split_k <= 1) {
    if (!p.bias_f32) {
      unpack8f(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.bias) + n), bias8);
    } else {
      const float4 b0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.bias) + n);
      const float4 b1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.bias) + n + 4);
      bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
      bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
    }
  }
  static_for<TM>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int64_t mrow0 = m_base + j * 32 + r8;       // rows mrow0 + 8*it, it = 0..3
    // one prefetch array: the residual octets if there is a residual, else the act' operand (a launch with both
    // loads the act' operand late; none of the model's GEMMs has both)
    uint4 pre[4];
    if (tile_full && p.split_k <= 1) {                 // request the slab's aux octets now
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int64_t m = mrow0 + 8 * it;
        const int64_t mc = m < p.M ? m : p.M - 1;
        if (p.residual) {
          const int64_t rr = p.res_rows > 0 ? (int64_t)((uint32_t)mc % (uint32_t)p.res_rows) : mc;
          pre[it] = *reinterpret_cast<const uint4*>(p.residual + rr * p.ld_res + n);
        } else if (has_dact) {
          pre[it] = *reinterpret_cast<const uint4*>(p.dact_aux + mc * p.ld_dact + n);
        }
      }
    }
    __syncthreads();  // operand buffers (j = 0) / previous slab (j > 0) no longer read
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<float4*>(patch + l31 * PATCH_LD + 32 * i + 8 * rq + 4 * g) =
            make_float4(acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]);
    __syncthreads();
    if (tile_full && p.split_k <= 1) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + r8;
        const int64_t m = mrow0 + 8 * it;
        const float4 lo = *reinterpret_cast<const float4*>(patch + row * PATCH_LD + cg * 8);
        const float4 hi = *reinterpret_cast<const float4*>(patch + row * PATCH_LD + cg * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (m < p.M) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bias8[e];
          if (p.preact) {
            store8_bf16(p.preact + m * p.ld_preact + n, n, p.N, true, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));   // the activation sees the stored (bf16) value
          }
          act_fwd8(v, p.act);
          if (p.has_drop) {
            const uint32_t rowkey = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)m);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint32_t h = drop_hash_rk(rowkey, (uint32_t)(n + e));
              v[e] = (h >= p.drop_thr) ? v[e] * p.drop_scale : 0.f;
            }
          }
          if (!p.c_f32 && (has_dact || p.residual)) {   // act' / residual see the rounded branch value (see epilogue_oct)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
          }
          if (has_dact) {
            float a[8];
            uint4 au = pre[it];
            if (p.residual) au = *reinterpret_cast<const uint4*>(p.dact_aux + m * p.ld_dact + n);
            unpack8f(au, a);
            act_bwd8_mul(v, a, p.dact);
          }
          if (p.residual) {
            float a[8];
            unpack8f(pre[it], a);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a[e];
          }
          if (p.c_f32) {
            float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
            if (p.accumulate) {
#pragma unroll
              for (int e = 0; e < 8; ++e) c[e] += v[e];
            } else {
              *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
              *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
          } else {
            store8_bf16(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n, n, p.N, true, v);
          }
        }
      }
    } else {                                            // ragged / unaligned tiles and split-K slabs: guarded path
#pragma unroll 2
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + r8;
        const int64_t m = mrow0 + 8 * it;
        if (m < p.M && n < p.N) {
          const float4 lo = *reinterpret_cast<const float4*>(patch + row * PATCH_LD + cg * 8);
          const float4 hi = *reinterpret_cast<const float4*>(patch + row * PATCH_LD + cg * 8 + 4);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          if (tile_full) epilogue_oct<true>(p, m, n, v, split);
          else epilogue_oct<false>(p, m, n, v, split);
        }
      }
    }
  });
}

template <class CF, bool A_T, bool B_T, int DBG = 0>
__global__ __launch_bounds__(CF::NT) void gemm_kernel(GemmKArgs p) {
  constexpr int BM = CF::BM, BN = CF::BN, TM = CF::TM, TN = CF::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [buf][A|B]
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave % CF::WM, wn = wave / CF::WM;
  const int l31 = lane & 31, g = lane >> 5;

  // XCD-aware tile order: block b runs on XCD b % 8; give every XCD a contiguous run of tile ids so that
  // the n-tiles sharing one A row-panel hit the same L2 (bijective form, cdna guide section 5 T1).
  const int ntiles = p.tiles_m * p.tiles_n;
  int tile;
  {
    const int b = blockIdx.x, q = ntiles / 8, r = ntiles % 8, xcd = b % 8, idx = b / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int split = blockIdx.y;
  const int64_t k_begin = (int64_t)split * p.k_per_split;
  const int64_t k_end = (k_begin + p.k_per_split < p.K) ? (k_begin + p.k_per_split) : p.K;

  f32x16 acc[TN][TM];  // [n-subtile i][m-subtile j]
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (int)((k_end - k_begin + BK - 1) / BK);
  // number of leading K-tiles that are complete (fast path eligible); the ragged tail tile takes the guarded loader
  const int nk_full = (int)((k_end - k_begin) / BK);
  const bool a_fast = p.a_vec && (!A_T || m0 + BM <= p.M);
  const bool b_fast = p.b_vec && (!B_T || n0 + BN <= p.N);
  const bf16_t* pa[4];
  const bf16_t* pb[4];
  fast_ptrs<A_T, BM>(pa, p.A, p.lda, m0, p.M, k_begin, t);
  fast_ptrs<B_T, BN>(pb, p.B, p.ldb, n0, p.N, k_begin, t);
  uint4 ra0[4], rb0[4], ra1[CF::DEEP ? 4 : 1], rb1[CF::DEEP ? 4 : 1];
  auto load_tile = [&](int kt, uint4 (&ra)[4], uint4 (&rb)[4]) {
    const int64_t k0 = k_begin + (int64_t)kt * BK;
    if (a_fast && kt < nk_full) fast_load<A_T>(ra, pa, p.lda);
    else stage_load<A_T, BM>(ra, p.A, p.lda, m0, p.M, k0, k_end, p.a_vec, t);
    if (b_fast && kt < nk_full) fast_load<B_T>(rb, pb, p.ldb);
    else stage_load<B_T, BN>(rb, p.B, p.ldb, n0, p.N, k0, k_end, p.b_vec, t);
  };
  auto store_tile = [&](const uint4 (&ra)[4], const uint4 (&rb)[4], char* buf) {
    stage_store<A_T, BM>(ra, buf, t);
    stage_store<B_T, BN>(rb, buf + CF::A_BYTES, t);
  };
  auto compute = [&](const char* cur) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[j] = frag_load<A_T, BM>(cur, wm * (TM * 32) + j * 32 + l31, ks, g);
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[i] = frag_load<B_T, BN>(cur + CF::A_BYTES, wn * (TN * 32) + i * 32 + l31, ks, g);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
    }
  };
  char* buf0 = smem;
  char* buf1 = smem + CF::BUF_BYTES;
  if constexpr (CF::DEEP) {
    if (nk > 0) {
      load_tile(0, ra0, rb0);
      store_tile(ra0, rb0, buf0);
      if (nk > 1) load_tile(1, ra1, rb1);
    }
    __syncthreads();
    int kt = 0;
    if (a_fast && b_fast) {
      // steady state, no conditionals inside: the compiler can then count the 8 newer loads and wait with vmcnt(8)
      // (a conditional prefetch forces s_waitcnt vmcnt(0) at the join and serialises load latency with the MFMAs)
      while (kt + 3 < nk_full) {
        // DBG (ablation builds only): 1 = no global loads / LDS writes, 2 = no MFMAs, 4 = no barriers
        if (!(DBG & 1)) { fast_load<A_T>(ra0, pa, p.lda); fast_load<B_T>(rb0, pb, p.ldb); }
        __builtin_amdgcn_sched_barrier(0);  // issue the prefetch BEFORE the MFMAs (the scheduler would sink it)
        if (!(DBG & 2)) compute(buf0);
        __builtin_amdgcn_sched_barrier(0);  // keep the ds_writes (and their vmcnt wait) BEHIND the MFMAs
        if (!(DBG & 1)) store_tile(ra1, rb1, buf1);
        if (!(DBG & 4)) __syncthreads();
        if (!(DBG & 1)) { fast_load<A_T>(ra1, pa, p.lda); fast_load<B_T>(rb1, pb, p.ldb); }
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & 2)) compute(buf1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & 1)) store_tile(ra0, rb0, buf0);
        if (!(DBG & 4)) __syncthreads();
        kt += 2;
      }
    }
    for (; kt < nk; kt += 2) {   // remaining (<= 3 full tiles + ragged tail) and the generic / guarded path
      if (kt + 2 < nk) load_tile(kt + 2, ra0, rb0);
      compute(buf0);
      if (kt + 1 < nk) store_tile(ra1, rb1, buf1);
      __syncthreads();
      if (kt + 1 >= nk) break;
      if (kt + 3 < nk) load_tile(kt + 3, ra1, rb1);
      compute(buf1);
      if (kt + 2 < nk) store_tile(ra0, rb0, buf0);
      __syncthreads();
    }
  } else {
    // one register set: tile kt+1 is loaded while tile kt is multiplied (128 accumulator + 24 fragment registers per
    // lane leave no room for a second set under the 256-register / 2-waves-per-SIMD budget of a 512-thread workgroup)
    if (nk > 0) { load_tile(0, ra0, rb0); store_tile(ra0, rb0, buf0); }
    __syncthreads();
    int kt = 0;
    if (a_fast && b_fast) {
      while (kt + 2 < nk_full) {
        fast_load<A_T>(ra0, pa, p.lda); fast_load<B_T>(rb0, pb, p.ldb);
        __builtin_amdgcn_sched_barrier(0);
        compute(buf0);
        __builtin_amdgcn_sched_barrier(0);
        store_tile(ra0, rb0, buf1);
        __syncthreads();
        fast_load<A_T>(ra0, pa, p.lda); fast_load<B_T>(rb0, pb, p.ldb);
        __builtin_amdgcn_sched_barrier(0);
        compute(buf1);
        __builtin_amdgcn_sched_barrier(0);
        store_tile(ra0, rb0, buf0);
        __syncthreads();
        kt += 2;
      }
    }
    for (; kt < nk; ++kt) {
      char* cur = (kt & 1) ? buf1 : buf0;
      char* nxt = (kt & 1) ? buf0 : buf1;
      const bool more = kt + 1 < nk;
      if (more) load_tile(kt + 1, ra0, rb0);
      compute(cur);
      if (more) store_tile(ra0, rb0, nxt);
      __syncthreads();
    }
  }

  // (the register-only epilogue of the ring kernels was measured here too: 264 VGPRs -> one workgroup per CU, 30-60 % slower)
  tile_epilogue<TM>(p, acc, smem, wave, lane, m0 + wm * (TM * 32), n0 + wn * 64, split);
}

// ====================================================================================================
// Ring kernel: PERSISTENT workgroups, LDS-DMA operand staging (global_load_lds_dwordx4) through a 4-stage LDS ring
// that never drains between output tiles, and a barrier-free, wave-private epilogue.
//
//   work item = (output tile, K split).  grid = min(items, workgroup slots of the chip); workgroup b runs items
//   it * grid + perm(b), it = 0, 1, ...  (perm puts the workgroups of one XCD on consecutive items, and consecutive
//   items form compact GH x (32 / GH) blocks of tiles: the 32 workgroups of an XCD share few operand panels in its L2).
//
//   stage = K-slice of 32: (BM + BN) x 32 bf16.  The DMA cursor runs PD = 3 stages ahead of the multiply THROUGH item
//   boundaries: while the last stages of a tile are multiplied and while its epilogue runs, the first stages of the
//   next tile are already in flight (measured: with one launch per tile the ring fill + C burst of every tile cost
//   ~19 us per 256-tile round at 20832 x 4096 -- 40 % of a K = 1024 GEMM -- because all CUs do them at the same time).
//   A wave waits for ITS pieces of a stage with a counted s_waitcnt vmcnt(stages ahead * pieces-per-wave), then ONE
//   raw s_barrier makes every wave's pieces visible and proves everybody is done with the previous stage, whose
//   buffer is immediately refilled.  No VGPR staging, no ds_write, no vmcnt(0) in the steady state.  (The counted
//   waits ignore the epilogue's own loads / stores, which are YOUNGER than the pieces waited for: VMEM operations of
//   a wave complete in order, so a smaller count only waits longer, never too little.)
//   k-contiguous operand : LDS rows of 64 B (4 slots of 16 B), k-octet o of row r in slot o ^ ((r>>2)&3)
//                          (the swizzle is applied on the per-lane GLOBAL address; the LDS image is lane-linear as
//                          LDS-DMA requires), fragments by conflict-free ds_read_b128.
//   r-contiguous operand : copied as it is, [32 k][ROWS] (16-B piece (k, ro) in slot ro ^ 4*(k&3) of its k-row), and read
//                          with the hardware transpose ds_read_b64_tr_b16 (lane i of a 16-lane group receives 4 consecutive k
//                          of row i; semantics probed on hardware: tests/probes/tr_probe.hip) -- so Conv1D weights and
//                          the weight-gradient GEMMs need no register transpose either.
//   epilogue             : entirely in registers (reg_epilogue below): bias / activation / dropout / act' / residual in the
//                          accumulator layout, a half-wave exchange (v_permlane32_swap) turns two column quads into 16
//                          contiguous bytes per lane, stores and store-layout loads are 16 B wide.  No LDS, no waits, no
//                          workgroup barrier: the ring keeps streaming underneath.  (Round 1 transposed every 32 x 64 slab
//                          through a wave-private LDS patch: ds_write -> wait -> ds_read -> wait -> store chains that cost
//                          120 us of a 335-us fused fc1 launch; profiles/r02_gemm_probe_*.txt.)
// Requirements (host falls back to the register-staged kernel otherwise): 16-B-vectorisable operands / outputs,
// K-range % 32 == 0, r-contiguous operands with rows % tile == 0.
// ====================================================================================================
template <int WM_, int WN_, int TM_, int TN_, int NS_, int WPE_, int GH_, int BKS_>
struct RCfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
  static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NWAVES = WM * WN, NT = NWAVES * 64;
  static constexpr int BKS = BKS_;           // K-slice per ring stage: 32 (64-B LDS rows) or 64 (128-B rows = whole cache lines
                                             // of a k-contiguous operand per DMA request)
  static constexpr int NS = NS_;             // NS ring stages: NS - 1 stages are in flight ahead of the one computed
  static constexpr int WPE = WPE_;           // waves per SIMD the register budget must allow
  static constexpr int GH = GH_;             // tile rows per raster group
  static constexpr int A_BYTES = BM * BKS * 2, B_BYTES = BN * BKS * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int RING_BYTES = NS * STAGE_BYTES;
  static constexpr int SMEM_BYTES = RING_BYTES;    // the epilogue runs in registers: the whole allocation is ring
  static constexpr int WG_PER_CU = (2 * SMEM_BYTES <= 160 * 1024) ? 2 : 1;
  static constexpr int A_CHUNKS = A_BYTES / 1024, B_CHUNKS = B_BYTES / 1024;
  static constexpr int CPW = (A_CHUNKS + B_CHUNKS) / NWAVES;   // DMA instructions per wave and stage
  static_assert((A_CHUNKS + B_CHUNKS) % NWAVES == 0, "chunks must divide evenly over the waves");
  static_assert(TN == 2, "epilogue slabs are 64 columns wide");
  static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");
  static_assert(NS >= 2 && NS <= 5 && (BKS == 32 || BKS == 64), "supported shapes");
  static_assert((NS - 1) * CPW <= 63, "vmcnt is a 6-bit counter");
};
using RCfgL = RCfg<2, 4, 4, 2, 4, 2, 4, 32>;    // 256 x 256, BK 32, 8 waves of 128 x 64, 128 KiB ring, 1 workgroup / CU
using RCfgS = RCfg<2, 2, 2, 2, 4, 2, 8, 32>;    // 128 x 128, BK 32, 4 waves of  64 x 64,  64 KiB ring, 2 workgroups / CU
using RCfgM64 = RCfg<4, 2, 2, 2, 3, 2, 4, 64>;  // 256 x 128, BK 64 (whole 128-B lines of a k-contiguous operand), 144 KiB ring
// (round 4: a 128 x 128 configuration with BK 64 and four stages -- 96 KiB in flight per workgroup -- was built for the evaluation
// engine's 930-row trunk GEMMs, which take 15 us on every configuration: 14.4 / 15.4 / 16.6 us against RCfgS's 14.4 / 15.1 / 16.1
// (profiles/r04_midrows_perf.jsonl, columns v12_*): those launches are not bound by operand latency; removed again.)

// per-lane global source address of chunk c of an operand tile at k0 (the LDS destination of lane l is chunk base + 16 l)
template <bool TRANS, int ROWS, int BKS>
__device__ __forceinline__ const bf16_t* dma_src(const bf16_t* __restrict__ P, int64_t ld, int64_t row0, int64_t rows,
                                                 int64_t k0, int c, int lane) {
  if (!TRANS) {   // k-contiguous: LDS rows of 2 BKS bytes; chunk c = 1 KiB = 16 (BK 32) or 8 (BK 64) consecutive rows
    constexpr int LPR = BKS / 8;             // 16-B slots (= lanes) per row
    const int rl = c * (64 / LPR) + lane / LPR;
    const int sl = lane % LPR;
    const int o = (BKS == 32) ? (sl ^ ((rl >> 2) & 3)) : (sl ^ ((rl >> 1) & 7));
    int64_t row = row0 + rl;
    row = row < rows ? row : rows - 1;
    return P + row * ld + k0 + o * 8;
  } else {        // chunk c = 64 consecutive 16-B pieces of the [k][ROWS] image
    constexpr int PPR = ROWS / 8;   // pieces per k-row
    const int piece = c * 64 + lane;
    const int k = piece / PPR, slot = piece % PPR;
    const int ro = slot ^ (4 * (k & 3));
    return P + (k0 + k) * ld + row0 + ro * 8;
  }
}

// One LDS-DMA instruction, issued from inline asm so that hipcc does not count it (it would otherwise put an
// s_waitcnt vmcnt(0) in front of the next ds_read and serialise the ring).  M0 (LDS base of the transfer) is saved and
// restored inside the statement (cdna guide section 5.7).  lds_dst must be wave-uniform.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// the same from a wave-uniform base (SGPR pair) + a per-lane 32-bit byte offset
__device__ __forceinline__ void glds16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// Round 5, the phase kernel's piece issue: M0 = <LDS address of the wave's piece 0 of the image> + IMM in ONE scalar add, the hazard
// nop, the DMA -- M0 is NOT saved / restored.  Legal only in a kernel none of whose other instructions reads M0:
// tests/test_phase_isa.py disassembles every gemm_phase_kernel of the built library and asserts exactly that.
// the ring kernels' form: the LDS address is a scalar the loop already holds (stage base + the wave's piece offset)
__device__ __forceinline__ void glds16s_m0(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// PAD (round 6): wait states between the M0 write and the DMA.  0 is the M0 hazard alone.  hipcc may reload a spilled SGPR (the scalar
// base) with v_readlane_b32 right in front of the statement, and an SGPR written by the VALU needs FIVE wait states before a VMEM
// instruction reads it -- padding hipcc does for its own instructions, not for these.  tests/test_phase_isa.py audits every built kernel
// for that pattern; the one instantiation it found (the NN layout's generic epilogue class, whose scalar pressure is the highest) gets PAD = 3.
template <int IMM, int PAD = 0>
__device__ __forceinline__ void glds16s_lean(const void* sbase, uint32_t voff, uint32_t lds_base) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop %4\n\tglobal_load_lds_dwordx4 %0, %1"
               : : "v"(voff), "s"(sbase), "s"(lds_base), "n"(IMM), "n"(PAD) : "memory", "scc");
}

// Round 6, the deferred-epilogue builds: the same piece through a BUFFER descriptor over the whole operand -- address = base + per-lane
// offset + scalar offset, rows past the operand's end are refused by the descriptor's bounds check (no per-lane clamp), so a wave's
// four pieces of an image need two per-lane offsets (even / odd piece: the swizzle alternates) and scalar offsets instead of four
// per-lane offsets and the hoisted row / column parts they were built from (gemm_phase.h BUFDMA).  soff must have been written well
// ahead of the statement (an SGPR written by the SALU needs five wait states in front of a VMEM instruction that reads it).
typedef __attribute__((ext_vector_type(4))) uint32_t rsrc4;
__device__ __forceinline__ rsrc4 operand_rsrc(const void* base, uint64_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return rsrc4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes, 0x00020000u};
}
template <int IMM>
__device__ __forceinline__ void glds16b_lean(rsrc4 rs, uint32_t voff, uint32_t soff, uint32_t lds_base) {
  // (s_nop 3: with the s_add in front, five wait states between a compiler reload of rs / soff -- v_readlane_b32 of a spilled SGPR,
  //  which hipcc may place right in front of the statement and does not pad for instructions it cannot see -- and the VMEM read)
  asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               : : "v"(voff), "s"(rs), "s"(soff), "s"(lds_base), "n"(IMM) : "memory", "scc");
}

typedef __attribute__((ext_vector_type(4))) short s16x4;

// fragment of the 32-row sub-tile starting at tile row `rbase`, k16-step ks (0 .. BKS/16 - 1) of the stage
template <bool TRANS, int ROWS, int BKS>
__device__ __forceinline__ bf16x8 ring_frag(const char* lds_oper, int rbase, int ks, int lane) {
  if (!TRANS) {
    const int row = rbase + (lane & 31), o = 2 * ks + (lane >> 5);
    if (BKS == 32) return *reinterpret_cast<const bf16x8*>(lds_oper + row * 64 + ((o ^ ((row >> 2) & 3)) << 4));
    return *reinterpret_cast<const bf16x8*>(lds_oper + row * 128 + ((o ^ ((row >> 1) & 7)) << 4));
  } else {
    const int gi = lane >> 4, c = lane & 15;
    const int r = rbase + 16 * (gi & 1) + 4 * (c & 3);                 // first of the 4 rows this lane FETCHES
    union { s16x4 h[2]; bf16x8 v; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 16 * ks + 8 * (gi >> 1) + 4 * h + (c >> 2);       // k-row this lane fetches from
      const int byte = k * (ROWS * 2) + ((((r >> 3) ^ (4 * (c >> 2)))) << 4) + (r & 7) * 2;
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_oper + byte));
    }
    return u.v;
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// work item -> (tile origin, K range).  Consecutive ids sweep groups of GH tile rows column by column.
struct RingItem { int64_t m0, n0, k_begin; int ns, split; };
// n / d for 0 <= n < 2^31, d >= 2, inv = floor(2^32 / d): the multiply-high is the quotient or one short of it
// (n * (2^32/d - inv) / 2^32 < 1), so one compare settles it.  Uniform operands: ~6 scalar instructions instead of the
// ~35 of an integer division, and a work item needs three of them four times per tile (two cursors, loop top, epilogue).
__device__ __forceinline__ int div_by(int n, int d, uint32_t inv) {
  int q = (int)__umulhi((uint32_t)n, inv);
  return (n - q * d >= d) ? q + 1 : q;
}
template <class RC>
__device__ __forceinline__ RingItem ring_item(const GemmKArgs& p, int id) {
  static_assert((RC::GH & (RC::GH - 1)) == 0, "GH is a power of two");
  const int ntiles = p.tiles_m * p.tiles_n;
  const int split = p.split_k > 1 ? div_by(id, ntiles, p.inv_ntiles) : 0, tile = id - split * ntiles;
  const int per_panel = RC::GH * p.tiles_n;
  const int panel = div_by(tile, per_panel, p.inv_per_panel), r = tile - panel * per_panel;
  const int left = p.tiles_m - panel * RC::GH;
  const int gh = left < RC::GH ? left : RC::GH;
  const int tn = gh == RC::GH ? (int)((unsigned)r / (unsigned)RC::GH) : r / gh, tm = panel * RC::GH + (r - tn * gh);   // only the ragged last panel divides
  RingItem it;
  it.m0 = (int64_t)tm * RC::BM; it.n0 = (int64_t)tn * RC::BN; it.split = split;
  it.k_begin = (int64_t)split * p.k_per_split;
  const int64_t k_end = (it.k_begin + p.k_per_split < p.K) ? (it.k_begin + p.k_per_split) : p.K;
  it.ns = (int)((k_end - it.k_begin + RC::BKS - 1) / RC::BKS);   // K range % BKS == 0 (ring_ok) except the phase kernel's partial last tile
  return it;
}

// ---- register-only epilogue (no LDS, no waits): the accumulator layout already holds 4 consecutive n per lane and
// quad; one v_permlane32_swap per packed dword exchanges the quads of column groups (2q, 2q+1) between the two half-waves
// so that a lane ends up with 8 consecutive bf16 (16 B) of its row:
//     lanes  0..31 : [own quad 2q   | upper half's quad 2q  ]  -> columns 32i + 16q + 0..7
//     lanes 32..63 : [lower's 2q+1  | own quad 2q+1         ]  -> columns 32i + 16q + 8..15
// (cdna guide T21).  The same exchange maps a 16-byte load of a store-layout operand (residual, act' operand) back to
// the accumulator layout, so bias / activation / dropout / act' / residual are all evaluated in registers.  fp32
// outputs need no exchange at all (a quad is already 16 B).  One store instruction covers 32 rows x 32 B; the four
// stores of a (32-row, 64-column) slab complete its 128-byte lines in the L2.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__device__ __forceinline__ void swap_halves(uint32_t& lo_grp, uint32_t& hi_grp) {
  const auto r = __builtin_amdgcn_permlane32_swap(lo_grp, hi_grp, false, false);
  lo_grp = r[0]; hi_grp = r[1];
}
__device__ __forceinline__ void unpack2(uint32_t u, float& a, float& b) { a = bf2f((bf16_t)(u & 0xffff)); b = bf2f((bf16_t)(u >> 16)); }

// The epilogue class is a COMPILE-TIME parameter of the kernels (the dispatcher picks the instantiation):
//   EPI_P0 / P_ERF / P_TANH : bf16 output, no load after the first store; activation none / erf-GELU / tanh-GELU (+ bias,
//                             pre-activation store, dropout)
//   EPI_A0 / A_ERF / A_TANH : bf16 output with ONE store-layout operand: a residual (A0: no activation) or the act' operand
//                             of a GELU (no forward activation); its vectors are requested per batch of 2 row slabs before
//                             that batch's first store
//   EPI_F32                 : fp32 C / split-K partial sums / accumulation (16-byte stores straight from the accumulator quads)
//   EPI_GEN                 : everything else for bf16 outputs (ReLU / SiLU / QuickGELU heads, activation + residual, both
//                             operands): activation codes are runtime switches
// Why compile time and not `if (p.residual)` / `switch (p.act)` inside one body (both measured, profiles/r02_gemm_phase_ablation.txt):
//  * hipcc's wait-count pass cannot know which side of a runtime branch ran, so in front of every store whose address / data
//    temporaries were allocated to registers that ANOTHER path loads into it emits s_waitcnt vmcnt(0) -- on the hardware that
//    drains every outstanding store and the LDS-DMA queue of the next tile, once per store: 85 us of a 275-us plain
//    20832 x 4096 x 1024 launch (raw stores alone: 30 us);
//  * a load issued after stores can only be waited for together with those stores (one in-order counter): load latency and
//    store latency serialise per unit; with the loads in front of the stores the compiler's counted waits are exact;
//  * an eight-way activation switch per 4 values, fully unrolled over 2 x TM x 4 groups, is ~25 000 instructions of
//    epilogue per kernel: the fused fc1 epilogue spent more time fetching code than storing (+110 us at 20832 x 4096).
enum { EPI_P0 = 0, EPI_P_ERF = 1, EPI_P_TANH = 2, EPI_A0 = 3, EPI_A_ERF = 4, EPI_A_TANH = 5, EPI_F32 = 6, EPI_GEN = 7, EPI_COUNT = 8 };
constexpr bool epi_aux(int e) { return e == EPI_A0 || e == EPI_A_ERF || e == EPI_A_TANH || e == EPI_GEN; }
constexpr int epi_act(int e) {    // forward activation: compile-time code, or -1 = read p.act
  return (e == EPI_P0 || e == EPI_A0 || e == EPI_A_ERF || e == EPI_A_TANH) ? ACT_NONE : e == EPI_P_ERF ? ACT_GELU_ERF : e == EPI_P_TANH ? ACT_GELU_TANH : -1;
}
constexpr int epi_dact(int e) {   // act' operand: 0 = none, compile-time code, or -1 = runtime (p.dact_aux / p.dact)
  return e == EPI_A_ERF ? ACT_GELU_ERF : e == EPI_A_TANH ? ACT_GELU_TANH : (e == EPI_GEN || e == EPI_F32) ? -1 : 0;
}
constexpr int epi_res(int e) {    // residual: 0 = none, 1 = present, -1 = runtime
  return e == EPI_A0 ? 1 : (e == EPI_GEN || e == EPI_F32) ? -1 : 0;
}
inline int epi_class(const GemmKArgs& a) {
  if (a.split_k > 1 || a.c_f32) return EPI_F32;
  const bool dact = a.dact_aux != nullptr, res = a.residual != nullptr;
  // dropout is compiled into the residual class only (the model drops in front of residual additions); elsewhere -> generic
  if (a.has_drop && !(res && !dact)) return EPI_GEN;
  if (!dact && !res) return a.act == ACT_NONE ? EPI_P0 : a.act == ACT_GELU_ERF ? EPI_P_ERF : a.act == ACT_GELU_TANH ? EPI_P_TANH : EPI_GEN;
  if (a.act != ACT_NONE || (dact && res) || a.preact != nullptr) return EPI_GEN;
  if (res) return EPI_A0;
  return a.dact == ACT_GELU_ERF ? EPI_A_ERF : a.dact == ACT_GELU_TANH ? EPI_A_TANH : EPI_GEN;
}

template <int ACT>
__device__ __forceinline__ void act_fwd4_sel(float (&z)[4], int runtime_act) {
  if constexpr (ACT >= 0) { if constexpr (ACT != ACT_NONE) act_fwd4_c<ACT>(z); } else act_fwd4(z, runtime_act);
}
template <int ACT>
__device__ __forceinline__ void act_bwd8_mul_sel(float (&v)[8], const float (&a)[8], int runtime_act) {
  if constexpr (ACT > 0) act_bwd8_mul_c<ACT>(v, a); else act_bwd8_mul(v, a, runtime_act);
}

// ---- whole-line stores.  After the half-wave exchange a lane holds, per 32-row slab, four 16-byte PIECES of its row: piece c
// (c = 2 i + q: column group i, quad pair q) covers bytes 16 (2c + g) .. + 15 of the row's 128-byte line (a wave's 64 bf16
// columns ARE one cache line per row).  Stored as they are, one instruction writes 32 rows x 32 B: 32 partial lines.  Measured
// with tests/probes/store_probe.cpp --percu (round 3): that pattern sustains 17.7 B/clk per CU however idle the rest of the chip
// is, against 55 B/clk for instructions that write 8 whole lines -- the store tail of a 256 x 256 tile (128 KiB, 256 KiB with a
// pre-activation) is bound by the CU's OWN store path, not by HBM, and the main loop's in-order vmcnt waits expose all of it
// (a wave cannot wait for a DMA piece issued after its stores without waiting for the stores: gemm_phase.h).
// quad_transpose exchanges the piece index c with the two low lane bits inside every quad of lanes (two rounds of DPP
// quad_perm + select: 32 VALU, no LDS): afterwards lane 4a + b holds in P[c] piece 2b + g of ROW 4a + c, i.e. instruction c
// writes rows {4a + c} x all 8 pieces = 8 whole lines.  The same exchange (an involution) turns whole-line LOADS of a
// store-layout operand (residual, act' operand) into the per-row pieces the accumulator layout needs.
__device__ __forceinline__ uint32_t dpp_quad_xor1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false); }
__device__ __forceinline__ uint32_t dpp_quad_xor2(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false); }
// One exchange stage for two of the four pieces = 8 dwords: O = lane in MASK ? OWN : quad_perm(OTHER), as ONE
// v_cndmask_b32_dpp per dword.  (The compiler keeps select conditions in arbitrary SGPR pairs, i.e. the VOP3 form of
// v_cndmask, which has no DPP encoding on gfx9: it emits v_mov_b32_dpp + v_cndmask_b32, twice the VALU slots; the epilogue
// is VALU-bound -- the matrix pipe idles while it runs.)  The two s_mov_b32 are also the two wait states a DPP read
// needs after a VALU write of its source.
#define DVLA_QT_STAGE(QP, MLO, O, OTH, OWN)                                                                          \
  asm("s_mov_b32 vcc_lo, " MLO "\n\ts_mov_b32 vcc_hi, " MLO "\n\t"                                                    \
      "v_cndmask_b32_dpp %0, %8, %16, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                                      \
      "v_cndmask_b32_dpp %1, %9, %17, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                                      \
      "v_cndmask_b32_dpp %2, %10, %18, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                                     \
      "v_cndmask_b32_dpp %3, %11, %19, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                                     \
      "v_cndmask_b32_dpp %4, %12, %20, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                                     \
      "v_cndmask_b32_dpp %5, %13, %21, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                                     \
      "v_cndmask_b32_dpp %6, %14, %22, vcc " QP " row_mask:0xf bank_mask:0xf\n\t"                                     \
      "v_cndmask_b32_dpp %7, %15, %23, vcc " QP " row_mask:0xf bank_mask:0xf"                                          \
      : "=&v"(O[0]), "=&v"(O[1]), "=&v"(O[2]), "=&v"(O[3]), "=&v"(O[4]), "=&v"(O[5]), "=&v"(O[6]), "=&v"(O[7])         \
      : "v"(OTH[0]), "v"(OTH[1]), "v"(OTH[2]), "v"(OTH[3]), "v"(OTH[4]), "v"(OTH[5]), "v"(OTH[6]), "v"(OTH[7]),        \
        "v"(OWN[0]), "v"(OWN[1]), "v"(OWN[2]), "v"(OWN[3]), "v"(OWN[4]), "v"(OWN[5]), "v"(OWN[6]), "v"(OWN[7])         \
      : "vcc")
__device__ __forceinline__ void quad_transpose(u32x4 (&P)[4], int /*lane*/) {
  // flat views: pieces {a, b} -> 8 dwords
  auto gather = [&](uint32_t (&v)[8], const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int d = 0; d < 4; ++d) { v[d] = a[d]; v[4 + d] = b[d]; }
  };
  uint32_t ev[8], od[8], te[8], to[8];
  // stage 1 (lane xor 1): T[c] = odd(c) != odd(lane) ? xor1(P[c ^ 1]) : P[c]; v_cndmask takes OWN where the vcc bit is set
  gather(ev, P[0], P[2]);                     // even pieces, their partners are the odd ones
  gather(od, P[1], P[3]);
  DVLA_QT_STAGE("quad_perm:[1,0,3,2]", "0x55555555", te, od, ev);   // even c keeps its own on even lanes
  DVLA_QT_STAGE("quad_perm:[1,0,3,2]", "0xaaaaaaaa", to, ev, od);   // odd c keeps its own on odd lanes
  // te = {T0, T2}, to = {T1, T3}.  stage 2 (lane xor 2): P[c] = (c & 2) != (lane & 2) ? xor2(T[c ^ 2]) : T[c]
  uint32_t lo[8], hi[8], plo[8], phi[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) { lo[d] = te[d]; lo[4 + d] = to[d]; hi[d] = te[4 + d]; hi[4 + d] = to[4 + d]; }   // {T0, T1}, {T2, T3}
  DVLA_QT_STAGE("quad_perm:[2,3,0,1]", "0x33333333", plo, hi, lo);  // c = 0, 1 keep their own where lane & 2 == 0
  DVLA_QT_STAGE("quad_perm:[2,3,0,1]", "0xcccccccc", phi, lo, hi);
#pragma unroll
  for (int d = 0; d < 4; ++d) { P[0][d] = plo[d]; P[1][d] = plo[4 + d]; P[2][d] = phi[d]; P[3][d] = phi[4 + d]; }
}
#undef DVLA_QT_STAGE

// Row-block window onto a row-major matrix for the whole-line stores / loads of the epilogue (round 5): a buffer descriptor whose
// base is element (m_base, n_base) and whose size ends with row M - 1, so that
//   * an access is ONE instruction with a per-lane 32-bit offset computed once per tile and a scalar row offset -- the flat form
//     cost ~9 VALU instructions of 64-bit address arithmetic + a compare / saveexec / branch per 16-byte store, a third of the
//     VALU work of a plain slab, in an epilogue that is VALU-bound with the matrix pipe idle (profiles/r05_gemm_boundary.txt);
//   * rows at or past M fall outside the descriptor: stores are dropped, loads return zero -- no row test, no clamp.
// (offsets inside a wave's 128 rows x 64 columns stay below 4 GiB for any leading dimension the dispatcher admits; the size field
// saturates at 4 GiB - 1, which can only happen when all 128 rows are inside the matrix anyway.)
struct RowWindow {
  __amdgpu_buffer_rsrc_t rs;
  uint32_t lane_off;      // this lane's byte offset: row 4 qa of the window, its 16 bytes of the line
  uint32_t row_bytes;     // leading dimension in bytes (uniform)
};
template <int ELEM>
__device__ __forceinline__ RowWindow row_window(const void* base, int64_t ld, int64_t m_base, int64_t n_base, int64_t M, int qa, int col_elem) {
  RowWindow w;
  const int64_t left = M - m_base;
  const uint64_t span = left > 0 ? (uint64_t)left * (uint64_t)ld * ELEM : 0;
  char* b = const_cast<char*>(reinterpret_cast<const char*>(base)) + (m_base * ld + n_base) * ELEM;
  w.rs = __builtin_amdgcn_make_buffer_rsrc(b, 0, span > 0xffffffffull ? 0xffffffffu : (uint32_t)span, 0x00020000);
  w.row_bytes = (uint32_t)(ld * ELEM);
  w.lane_off = (uint32_t)(4 * qa) * w.row_bytes + (uint32_t)(col_elem * ELEM);
  return w;
}
__device__ __forceinline__ void window_store(const RowWindow& w, int row, u32x4 v) {     // row: uniform (32 j + c)
  __builtin_amdgcn_raw_buffer_store_b128(v, w.rs, w.lane_off, (uint32_t)row * w.row_bytes, 0);
}
__device__ __forceinline__ u32x4 window_load(const RowWindow& w, int row) {
  return __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.lane_off, (uint32_t)row * w.row_bytes, 0);
}

struct NoStamp { __device__ __forceinline__ void operator()(int) const {} };   // timeline builds pass a recorder instead
// NO_BIAS: the bias is already inside the accumulators (the phase kernel's deferred-epilogue builds fold it in with one MFMA per
// accumulator block, gemm_phase.h)
template <int TM, int EPI, class ST = NoStamp, bool NO_BIAS = false>
__device__ __forceinline__ void reg_epilogue(const GemmKArgs& p, f32x16 (&acc)[2][TM], int lane, int64_t m_base,
                                             int64_t n_base, int split, ST st = ST()) {
  if (n_base >= p.N) return;   // N % 64 == 0: a wave's 64 columns are all inside or all outside
  constexpr bool AUXV = epi_aux(EPI);            // store-layout operand vectors are prefetched (bf16 outputs)
  constexpr bool AUX = AUXV || EPI == EPI_F32;   // the class may read an act' operand / a residual / old C values at all
  const int l31 = lane & 31, g = lane >> 5;
  const bool split_out = EPI == EPI_F32 && p.split_k > 1;
  constexpr bool f32_out = EPI == EPI_F32;
  const bool has_dact = epi_dact(EPI) > 0 || (epi_dact(EPI) < 0 && p.dact_aux != nullptr);
  const bool has_res = epi_res(EPI) > 0 || (epi_res(EPI) < 0 && p.residual != nullptr);
  const bool has_bias = !NO_BIAS && !split_out && p.bias != nullptr;
  const bool two_aux = has_dact && has_res;   // none of the model's GEMMs has both: the act' operand is then read late

  // ================= split-K partial sums (the weight gradients): whole 128-byte lines of fp32 =================
  // A lane's accumulator quads (i, rq) are the 16-byte pieces 2 rq + g of its row's 32-float line of column group i: the same
  // 4 x 4 exchange as below makes every store instruction write 8 whole lines instead of 32 rows x 32 B (the partial-line pattern
  // sustains 17.7 B/clk per CU: 256 KiB of partials per tile = ~15 k cycles of store issue, ~5 k with whole lines).
  if constexpr (EPI == EPI_F32) {
    if (split_out && (p.N & 31) == 0 && (reinterpret_cast<uintptr_t>(p.workspace) & 127) == 0) {
      const int qb = lane & 3, qa = l31 >> 2;
      float* ws = p.workspace + (int64_t)split * p.M * p.N + n_base + 4 * (2 * qb + g);
      static_for<TM>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        static_for<2>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          u32x4 P[4];
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            P[rq] = u32x4{__float_as_uint(acc[i][j][4 * rq]), __float_as_uint(acc[i][j][4 * rq + 1]),
                          __float_as_uint(acc[i][j][4 * rq + 2]), __float_as_uint(acc[i][j][4 * rq + 3])};
          quad_transpose(P, lane);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int64_t mr = m_base + j * 32 + 4 * qa + c;
            if (mr < p.M) *reinterpret_cast<u32x4*>(ws + mr * p.N + 32 * i) = P[c];
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      return;
    }
  }

  // ================= whole-line path: every bf16 class except EPI_GEN (see quad_transpose above) =================
  if constexpr (EPI != EPI_F32 && EPI != EPI_GEN) {
    const int qb = lane & 3, qa = l31 >> 2;
    const int64_t ncol_line = n_base + 8 * (2 * qb + g);      // this lane's 16 bytes of a line in the store / load pattern
    const int col_line = 8 * (2 * qb + g);
    const RowWindow wC = row_window<2>(p.C, p.ldc, m_base, n_base, p.M, qa, col_line);
    // the bias goes into the accumulators once, up front (no live bias registers, no load behind a store)
    if (has_bias) {
      static_for<2>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        float bf[4][4];
        if (p.bias_f32) {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.bias) + n_base + 32 * i + 8 * rq + 4 * g);
            bf[rq][0] = b.x; bf[rq][1] = b.y; bf[rq][2] = b.z; bf[rq][3] = b.w;
          }
        } else {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const uint2 b = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.bias) + n_base + 32 * i + 8 * rq + 4 * g);
            unpack2(b.x, bf[rq][0], bf[rq][1]); unpack2(b.y, bf[rq][2], bf[rq][3]);
          }
        }
        static_for<TM>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * rq + e] += bf[rq][e];
        });
      });
    }
    // whole lines of the store-layout operand of row slab j: instruction c reads rows {4a + c}, this lane's piece 2b + g
    const bool aux_window = AUXV && !(has_res && p.res_rows > 0);      // (a periodic residual -- the position table -- keeps the flat form)
    RowWindow wX = wC;
    if constexpr (AUXV) {
      if (aux_window)
        wX = has_res ? row_window<2>(p.residual, p.ld_res, m_base, n_base, p.M, qa, col_line)
                     : row_window<2>(p.dact_aux, p.ld_dact, m_base, n_base, p.M, qa, col_line);
    }
    auto load_aux_lines = [&](int j, u32x4 (&L)[4]) {
      if (aux_window) {
#pragma unroll
        for (int c = 0; c < 4; ++c) L[c] = window_load(wX, j * 32 + c);
        return;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t m = m_base + j * 32 + 4 * qa + c;
        const int64_t mc = m < p.M ? m : p.M - 1;
        const int64_t rr = (has_res && p.res_rows > 0) ? (int64_t)((uint32_t)mc % (uint32_t)p.res_rows) : mc;
        const bf16_t* src = has_res ? p.residual + rr * p.ld_res : p.dact_aux + mc * p.ld_dact;
        L[c] = *reinterpret_cast<const u32x4*>(src + ncol_line);
      }
    };
    st(0);
    // dropout and the pre-activation store are COMPILE-TIME inside the slab loop: as run-time flags they were tested once per
    // column group (32 uniform branches per tile, each re-deriving its condition from a spilled SGPR: a slab's conversion took
    // 720-1700 cycles by the s_memtime slab stamps of the timeline build, 410-510 without them); a class holds at most two
    // copies of the loop (P classes: with / without pre-activation; residual class: with / without dropout)
    auto slabs = [&](auto dropc, auto prec) {
      constexpr bool DROP = decltype(dropc)::value, PRE = decltype(prec)::value;
      RowWindow wP = wC;
      if constexpr (PRE) wP = row_window<2>(p.preact, p.ld_preact, m_base, n_base, p.M, qa, col_line);
      u32x4 nx[4];      // (unused, and removed by the compiler, in the classes without a store-layout operand) raw lines of the NEXT row slab: requested one slab ahead, in front of this slab's stores
      if constexpr (AUXV) load_aux_lines(0, nx);
      static_for<TM>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        u32x4 ax[4];
        if constexpr (AUXV) {
#pragma unroll
          for (int c = 0; c < 4; ++c) ax[c] = nx[c];
          quad_transpose(ax, lane);            // -> ax[c] = bytes 16 (2c + g) .. of THIS lane's row (the old store-layout vector)
        }
        const int64_t m = m_base + j * 32 + l31;
        uint32_t rowkey = 0;
        if constexpr (DROP) rowkey = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)m);
        u32x4 PC[4], PP[4];
        static_for<4>([&](auto cc) {
          constexpr int c = decltype(cc)::value, i = c >> 1, q = c & 1;
          float z0[4], z1[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { z0[e] = acc[i][j][8 * q + e]; z1[e] = acc[i][j][8 * q + 4 + e]; }
          if constexpr (PRE) {
            uint32_t a0 = pack2bf(z0[0], z0[1]), a1 = pack2bf(z0[2], z0[3]), b0 = pack2bf(z1[0], z1[1]), b1 = pack2bf(z1[2], z1[3]);
            // the activation sees the stored (bf16) pre-activation
            unpack2(a0, z0[0], z0[1]); unpack2(a1, z0[2], z0[3]); unpack2(b0, z1[0], z1[1]); unpack2(b1, z1[2], z1[3]);
            swap_halves(a0, b0); swap_halves(a1, b1);
            PP[c] = u32x4{a0, a1, b0, b1};
          }
          act_fwd4_sel<epi_act(EPI)>(z0, p.act);
          act_fwd4_sel<epi_act(EPI)>(z1, p.act);
          if constexpr (DROP) {
            const uint32_t n = (uint32_t)(n_base + 32 * i + 16 * q + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t h0 = drop_hash_rk(rowkey, n + e), h1 = drop_hash_rk(rowkey, n + 8 + e);
              z0[e] = (h0 >= p.drop_thr) ? z0[e] * p.drop_scale : 0.f;
              z1[e] = (h1 >= p.drop_thr) ? z1[e] * p.drop_scale : 0.f;
            }
          }
          uint32_t a0 = pack2bf(z0[0], z0[1]), a1 = pack2bf(z0[2], z0[3]), b0 = pack2bf(z1[0], z1[1]), b1 = pack2bf(z1[2], z1[3]);
          if constexpr (AUXV) {
            // act' / residual are applied to the ROUNDED branch value (the reference materialises it as a bf16 tensor)
            unpack2(a0, z0[0], z0[1]); unpack2(a1, z0[2], z0[3]); unpack2(b0, z1[0], z1[1]); unpack2(b1, z1[2], z1[3]);
            uint32_t ux = ax[c][0], uy = ax[c][1], uz = ax[c][2], uw = ax[c][3];
            swap_halves(ux, uz); swap_halves(uy, uw);   // -> accumulator layout: (x, y) = quad 2q, (z, w) = quad 2q+1
            if (has_dact) {
              float a8[8];
              unpack2(ux, a8[0], a8[1]); unpack2(uy, a8[2], a8[3]); unpack2(uz, a8[4], a8[5]); unpack2(uw, a8[6], a8[7]);
              float v8[8] = {z0[0], z0[1], z0[2], z0[3], z1[0], z1[1], z1[2], z1[3]};
              act_bwd8_mul_sel<epi_dact(EPI)>(v8, a8, p.dact);
#pragma unroll
              for (int e = 0; e < 4; ++e) { z0[e] = v8[e]; z1[e] = v8[4 + e]; }
            } else {
              float r0, r1;
              unpack2(ux, r0, r1); z0[0] += r0; z0[1] += r1;
              unpack2(uy, r0, r1); z0[2] += r0; z0[3] += r1;
              unpack2(uz, r0, r1); z1[0] += r0; z1[1] += r1;
              unpack2(uw, r0, r1); z1[2] += r0; z1[3] += r1;
            }
            a0 = pack2bf(z0[0], z0[1]); a1 = pack2bf(z0[2], z0[3]); b0 = pack2bf(z1[0], z1[1]); b1 = pack2bf(z1[2], z1[3]);
          }
          swap_halves(a0, b0); swap_halves(a1, b1);
          PC[c] = u32x4{a0, a1, b0, b1};
          // the next slab's lines are requested HALF WAY through this one: column group 0's accumulators and operand pieces are
          // dead by now (a full slab of look-ahead next to 128 accumulators spilled 15-22 registers), and the request still sits
          // in front of this slab's stores, so waiting for it never waits for them
          if constexpr (AUXV && c == 1 && j + 1 < TM) load_aux_lines(j + 1, nx);
        });
        if constexpr (PRE) {
          quad_transpose(PP, lane);
#pragma unroll
          for (int c = 0; c < 4; ++c) window_store(wP, j * 32 + c, PP[c]);
        }
        quad_transpose(PC, lane);
        st(1 + 2 * j);
#pragma unroll
        for (int c = 0; c < 4; ++c) window_store(wC, j * 32 + c, PC[c]);
        __builtin_amdgcn_sched_barrier(0);
        st(2 + 2 * j);
      });
    };
    if constexpr (EPI == EPI_A0) {
      if (p.has_drop) slabs(std::true_type{}, std::false_type{}); else slabs(std::false_type{}, std::false_type{});
    } else if constexpr (AUXV) {
      slabs(std::false_type{}, std::false_type{});
    } else {
      if (p.preact) slabs(std::false_type{}, std::true_type{}); else slabs(std::false_type{}, std::false_type{});
    }
    return;
  }

  // Store-layout operand vectors (EPI_AUX): requested per BATCH of JB = 2 row slabs (8 vectors = 32 VGPRs per lane) before
  // that batch's first store.  A 128-row wave block (TM = 4) therefore has ONE point per tile where loads follow stores
  // (all 16 vectors at once = 64 VGPRs next to the 128 accumulators made hipcc spill ~100 registers).
  // The bias of the wave's 64 columns is loaded ONCE per tile, before the first store: a load issued after stores can only be
  // waited for together with them (one in-order counter), and re-loading it per (row batch, column group) made three of
  // the four s_waitcnt per tile drain every store issued so far.
  float bias_f[2][4][4];
  auto load_bias = [&](auto ic) {     // column group i of the wave's 64 columns -> fp32, settled at once (the compiler would
    constexpr int i = decltype(ic)::value;   // otherwise wait at every use site, in the middle of the store sequence)
    if (p.bias_f32) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.bias) + n_base + 32 * i + 8 * rq + 4 * g);
        bias_f[i][rq][0] = b.x; bias_f[i][rq][1] = b.y; bias_f[i][rq][2] = b.z; bias_f[i][rq][3] = b.w;
      }
    } else {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const uint2 b = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.bias) + n_base + 32 * i + 8 * rq + 4 * g);
        unpack2(b.x, bias_f[i][rq][0], bias_f[i][rq][1]); unpack2(b.y, bias_f[i][rq][2], bias_f[i][rq][3]);
      }
    }
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(bias_f[i][rq][e]));
  };
  // classes without store-layout operands have the registers to hold both column groups: ONE load point per tile, before the
  // first store (a load issued after stores can only be waited for together with them: re-loading per (row batch, column
  // group) made three of four waits per tile drain every store issued so far).  The others load per column group.
  constexpr bool BIAS_ONCE = EPI == EPI_P0 || EPI == EPI_P_ERF || EPI == EPI_P_TANH || EPI == EPI_F32;
  if constexpr (BIAS_ONCE) {
    if (has_bias) { static_for<2>([&](auto ic) { load_bias(ic); }); }
  }
  constexpr int JB = TM < 2 ? TM : 2, NB = TM / JB;
  static_for<NB>([&](auto bc) {
  constexpr int jb0 = decltype(bc)::value * JB;
  u32x4 aux[AUXV ? 4 * JB : 1];   // [(2 i + q) * JB + (j - jb0)], compile-time indices only (plain vector type: registers)
  if constexpr (AUXV) {
    if (has_res || has_dact)          // (EPI_GEN also serves activation-only epilogues: nothing to prefetch then)
    static_for<JB>([&](auto jc) {
      constexpr int j = jb0 + decltype(jc)::value;
      const int64_t m = m_base + j * 32 + l31;
      const int64_t mc = m < p.M ? m : p.M - 1;
      const int64_t rr = (has_res && p.res_rows > 0) ? (int64_t)((uint32_t)mc % (uint32_t)p.res_rows) : mc;
      const bf16_t* src = has_res ? p.residual + rr * p.ld_res : p.dact_aux + mc * p.ld_dact;
      static_for<4>([&](auto cc) {
        constexpr int i = decltype(cc)::value >> 1, q = decltype(cc)::value & 1;
        aux[(2 * i + q) * JB + (j - jb0)] = *reinterpret_cast<const u32x4*>(src + n_base + 32 * i + 16 * q + 8 * g);
      });
    });
  }

  // Work unit = one 32 x 32 accumulator tile (i, j); i (column group) is the OUTER loop so that only the 8 packed bias
  // dwords of one column group are live; a sched_barrier closes every unit (keeps the scheduler from hoisting all the
  // conversion work of the block to the top and spilling next to the 128 accumulators).
  static_for<2>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (!BIAS_ONCE) {
      if (has_bias) load_bias(ic);
    }
    static_for<JB>([&](auto jc) {
      constexpr int j = jb0 + decltype(jc)::value;
      const int64_t m = m_base + j * 32 + l31;
      const bool row_ok = m < p.M;
      const int64_t mc = row_ok ? m : p.M - 1;
      uint32_t rowkey = 0;
      if (p.has_drop) rowkey = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)m);

      auto biased = [&](int rq, float (&z)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = acc[i][j][4 * rq + e];
        if (has_bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) z[e] += bias_f[i][rq][e];
        }
      };
      auto act_drop = [&](int rq, float (&z)[4]) {
        act_fwd4_sel<epi_act(EPI)>(z, p.act);
        if (p.has_drop) {
          const uint32_t n = (uint32_t)(n_base + 32 * i + 8 * rq + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t h = drop_hash_rk(rowkey, n + e);
            z[e] = (h >= p.drop_thr) ? z[e] * p.drop_scale : 0.f;
          }
        }
      };

      if constexpr (f32_out) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          float z[4];
          biased(rq, z);
          const int64_t n = n_base + 32 * i + 8 * rq + 4 * g;
          if (split_out) {
            if (row_ok) *reinterpret_cast<float4*>(p.workspace + ((int64_t)split * p.M + m) * p.N + n) = make_float4(z[0], z[1], z[2], z[3]);
            continue;
          }
          if (p.preact) {
            if (row_ok) *reinterpret_cast<uint2*>(p.preact + m * p.ld_preact + n) = make_uint2(pack2bf(z[0], z[1]), pack2bf(z[2], z[3]));
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = bf2f(f2bf(z[e]));
          }
          act_drop(rq, z);
          if constexpr (AUX) {
            if (has_dact) {
              const uint2 a = *reinterpret_cast<const uint2*>(p.dact_aux + mc * p.ld_dact + n);
              float a0, a1, a2, a3;
              unpack2(a.x, a0, a1); unpack2(a.y, a2, a3);
              z[0] *= act_bwd(a0, p.dact); z[1] *= act_bwd(a1, p.dact); z[2] *= act_bwd(a2, p.dact); z[3] *= act_bwd(a3, p.dact);
            }
            if (has_res) {
              const int64_t rr = p.res_rows > 0 ? (int64_t)((uint32_t)mc % (uint32_t)p.res_rows) : mc;
              const uint2 a = *reinterpret_cast<const uint2*>(p.residual + rr * p.ld_res + n);
              float a0, a1, a2, a3;
              unpack2(a.x, a0, a1); unpack2(a.y, a2, a3);
              z[0] += a0; z[1] += a1; z[2] += a2; z[3] += a3;
            }
          }
          if (row_ok) {
            float4* c = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + m * p.ldc + n);
            float4 o = make_float4(z[0], z[1], z[2], z[3]);
            if constexpr (AUX) {
              if (p.accumulate) { const float4 old = *c; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
            }
            *c = o;
          }
        }
      } else {
        static_for<2>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const int64_t ncol = n_base + 32 * i + 16 * q + 8 * g;   // this lane's 16 bytes in the store layout
          float z0[4], z1[4];
          biased(2 * q, z0);
          biased(2 * q + 1, z1);
          if (p.preact) {
            uint32_t a0 = pack2bf(z0[0], z0[1]), a1 = pack2bf(z0[2], z0[3]), b0 = pack2bf(z1[0], z1[1]), b1 = pack2bf(z1[2], z1[3]);
            // the activation sees the stored (bf16) pre-activation
            unpack2(a0, z0[0], z0[1]); unpack2(a1, z0[2], z0[3]); unpack2(b0, z1[0], z1[1]); unpack2(b1, z1[2], z1[3]);
            swap_halves(a0, b0); swap_halves(a1, b1);
            if (row_ok) *reinterpret_cast<uint4*>(p.preact + m * p.ld_preact + ncol) = make_uint4(a0, a1, b0, b1);
          }
          act_drop(2 * q, z0);
          act_drop(2 * q + 1, z1);
          uint32_t a0 = pack2bf(z0[0], z0[1]), a1 = pack2bf(z0[2], z0[3]), b0 = pack2bf(z1[0], z1[1]), b1 = pack2bf(z1[2], z1[3]);
          if constexpr (AUXV) {
            // act' / residual are applied to the ROUNDED branch value (the reference materialises it as a bf16 tensor)
            unpack2(a0, z0[0], z0[1]); unpack2(a1, z0[2], z0[3]); unpack2(b0, z1[0], z1[1]); unpack2(b1, z1[2], z1[3]);
            if (has_dact) {
              u32x4 u = aux[(2 * i + q) * JB + (j - jb0)];
              if (two_aux) u = *reinterpret_cast<const u32x4*>(p.dact_aux + mc * p.ld_dact + ncol);
              uint32_t ux = u.x, uy = u.y, uz = u.z, uw = u.w;
              swap_halves(ux, uz); swap_halves(uy, uw);   // -> accumulator layout: (x, y) = quad 2q, (z, w) = quad 2q+1
              float a8[8];
              unpack2(ux, a8[0], a8[1]); unpack2(uy, a8[2], a8[3]); unpack2(uz, a8[4], a8[5]); unpack2(uw, a8[6], a8[7]);
              float v8[8] = {z0[0], z0[1], z0[2], z0[3], z1[0], z1[1], z1[2], z1[3]};
              act_bwd8_mul_sel<epi_dact(EPI)>(v8, a8, p.dact);
#pragma unroll
              for (int e = 0; e < 4; ++e) { z0[e] = v8[e]; z1[e] = v8[4 + e]; }
            }
            if (has_res) {
              const u32x4 u = aux[(2 * i + q) * JB + (j - jb0)];
              uint32_t ux = u.x, uy = u.y, uz = u.z, uw = u.w;
              swap_halves(ux, uz); swap_halves(uy, uw);
              float r0, r1;
              unpack2(ux, r0, r1); z0[0] += r0; z0[1] += r1;
              unpack2(uy, r0, r1); z0[2] += r0; z0[3] += r1;
              unpack2(uz, r0, r1); z1[0] += r0; z1[1] += r1;
              unpack2(uw, r0, r1); z1[2] += r0; z1[3] += r1;
            }
            a0 = pack2bf(z0[0], z0[1]); a1 = pack2bf(z0[2], z0[3]); b0 = pack2bf(z1[0], z1[1]); b1 = pack2bf(z1[2], z1[3]);
          }
          swap_halves(a0, b0); swap_halves(a1, b1);
          if (row_ok) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + ncol) = make_uint4(a0, a1, b0, b1);
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  });
}

// ================= pieces of the phase kernel's DEFERRED epilogue (round 6, gemm_phase.h) =================
// The phase kernel's tile boundary was 16-22 % of a K = 1024 launch with the matrix pipe idle (profiles/r05_gemm_epilogue_decomposition.txt).
// Its deferred builds run the epilogue of tile t at the head of the next tile's first load segment, one wave group at a time, while
// the partner wave of the SIMD multiplies (gemm_phase.h DEFER).  The pieces below are what that path needs.
//
// Loads the compiler must not count (as the LDS-DMA pieces: its own s_waitcnt in front of the first use would be computed without
// the DMA pieces issued in between and drain them): issued from inline asm, waited for by wait_vmcnt_dyn + settle.
// (s_nop 4: an SGPR written by a VALU instruction -- hipcc reloads spilled SGPRs with v_readlane_b32, possibly right in front of the
// statement -- needs five wait states before a VMEM instruction reads it; hipcc pads that only for instructions it can see.  The
// first build without the nops read the bias through a stale base register: memory faults in every erf-GELU launch.
// tests/probes/asm_hazard_audit.py checks the compiler's assembly for this pattern.)
__device__ __forceinline__ uint32_t aload_b32(const void* sbase, uint32_t voff) {
  uint32_t r;
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
__device__ __forceinline__ uint32_t aload_u16(const void* sbase, uint32_t voff) {
  uint32_t r;
  asm volatile("s_nop 4\n\tglobal_load_ushort %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
__device__ __forceinline__ void settle(uint32_t& a, uint32_t& b) { asm volatile("" : "+v"(a), "+v"(b) :: "memory"); }
// s_waitcnt vmcnt(n) for a wave-uniform n = 0, 4, 8, ... 60 (the immediate is six bits); other values round DOWN (waits for more)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n >> 2) {
    case 0: wait_vmcnt<0>(); break;   case 1: wait_vmcnt<4>(); break;   case 2: wait_vmcnt<8>(); break;   case 3: wait_vmcnt<12>(); break;
    case 4: wait_vmcnt<16>(); break;  case 5: wait_vmcnt<20>(); break;  case 6: wait_vmcnt<24>(); break;  case 7: wait_vmcnt<28>(); break;
    case 8: wait_vmcnt<32>(); break;  case 9: wait_vmcnt<36>(); break;  case 10: wait_vmcnt<40>(); break; case 11: wait_vmcnt<44>(); break;
    case 12: wait_vmcnt<48>(); break; case 13: wait_vmcnt<52>(); break; case 14: wait_vmcnt<56>(); break; default: wait_vmcnt<60>(); break;
  }
}
// The bias as ONE extra multiply per accumulator block: acc[n][m] += sum_k F[n][k] . O[m][k] with F[n][0..2] = a three-way bf16 split of
// bias[n] (hi + mid + lo == the fp32 value: 3 x 8 significant bits; a bf16 bias is its own `hi`) and O[m][0..2] = 1 -- k = 0..3 live in
// the lower half-wave of a v_mfma_f32_32x32x8_bf16_1k operand, the upper half-wave holds zeros.  128 v_add per tile and wave (and the
// bias registers they read) become 8 short MFMAs.
typedef __attribute__((ext_vector_type(4))) short bf16x4v;
__device__ __forceinline__ bf16x4v bias_split(float b, bool lower_half) {
  const float b0 = lower_half ? b : 0.f;
  const bf16_t hi = f2bf(b0);
  const float r1 = b0 - bf2f(hi);
  const bf16_t mid = f2bf(r1);
  const bf16_t lo = f2bf(r1 - bf2f(mid));
  return bf16x4v{(short)hi, (short)mid, (short)lo, 0};
}
__device__ __forceinline__ bf16x4v bias_ones(bool lower_half) {
  const short one = lower_half ? (short)0x3f80 : (short)0;
  return bf16x4v{one, one, one, 0};
}

template <class RC, bool A_T, bool B_T, int EPI>
__global__ __launch_bounds__(RC::NT) __attribute__((amdgpu_waves_per_eu(RC::WPE, RC::WPE)))
void gemm_ring_kernel(GemmKArgs p) {
  constexpr int BM = RC::BM, BN = RC::BN, TM = RC::TM, TN = RC::TN, CPW = RC::CPW, NS = RC::NS, PD = RC::NS - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave % RC::WM, wn = wave / RC::WM;

  const int nitems = p.tiles_m * p.tiles_n * p.split_k;
  const int grid = gridDim.x;
  // workgroup b sits on XCD b % 8: give each XCD a run of consecutive items
  // (round 6: also for grids that are not a multiple of 8 -- XCD x then hosts grid / 8 (+ 1 for x < grid % 8) workgroups.  The
  // identity map such grids used to get spread consecutive items -- the K slices of one tile, neighbouring tiles of one operand
  // panel -- over all eight L2s: 252-item weight gradients ran 20-26 % slower than their 216-item siblings,
  // profiles/r06_small_dw_sweep.txt.)
  const int perm = [&] {
    const int q8 = grid >> 3, r8 = grid & 7, xcd = (int)(blockIdx.x & 7), idx = (int)(blockIdx.x >> 3);
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }();
  auto item_of = [&](int it) -> int {   // item of iteration `it`, or -1
    const int id = it * grid + perm;
    return id < nitems ? id : -1;
  };

  // ---- DMA cursor: per-wave plan of CPW pieces per stage (fixed operand / chunk per slot i); destinations are wave-uniform
  // byte offsets inside a stage buffer.  Round 5 (as in the phase kernel, gemm_phase.h LEAN): the K position of the cursor lives
  // in two SCALAR bases (operand + origin of the item's tile + k), advanced once per stage; a piece's per-lane source is a
  // 32-bit offset inside the tile, computed when the cursor opens an item and constant until the next one; M0 is written, not
  // saved / restored (tests/test_phase_isa.py covers these kernels too).  A piece was ~14 instructions (64-bit per-lane pointer
  // advance, readfirstlane of the LDS address, M0 save / set / restore), now 4. ----
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t src[CPW];
  bool is_a[CPW];
  uint32_t dst[CPW];
  const int64_t step_a = 2 * (A_T ? (int64_t)RC::BKS * p.lda : (int64_t)RC::BKS);
  const int64_t step_b = 2 * (B_T ? (int64_t)RC::BKS * p.ldb : (int64_t)RC::BKS);
  const char* kb_a = reinterpret_cast<const char*>(p.A);
  const char* kb_b = reinterpret_cast<const char*>(p.B);
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int q = wave * CPW + i;   // wave-uniform
    is_a[i] = q < RC::A_CHUNKS;
    dst[i] = is_a[i] ? (uint32_t)(q * 1024) : (uint32_t)(RC::A_BYTES + (q - RC::A_CHUNKS) * 1024);
  }
  int cur_it = 0, cur_s = 0, cur_ns = 0;   // cursor: item iteration, stage inside it, stages of it
  bool cur_live = false;
  auto cursor_open = [&](int it) {
    const int id = item_of(it);
    cur_live = id >= 0;
    if (!cur_live) return;
    const RingItem w = ring_item<RC>(p, id);
    cur_ns = w.ns; cur_s = 0;
    // origin of the item's operand tiles (element (m0 | n0, k_begin)): every piece's source lies at or behind it
    kb_a = reinterpret_cast<const char*>(A_T ? p.A + w.k_begin * p.lda + w.m0 : p.A + w.m0 * p.lda + w.k_begin);
    kb_b = reinterpret_cast<const char*>(B_T ? p.B + w.k_begin * p.ldb + w.n0 : p.B + w.n0 * p.ldb + w.k_begin);
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int q = wave * CPW + i;
      src[i] = is_a[i] ? (uint32_t)(reinterpret_cast<const char*>(dma_src<A_T, BM, RC::BKS>(p.A, p.lda, w.m0, p.M, w.k_begin, q, lane)) - kb_a)
                       : (uint32_t)(reinterpret_cast<const char*>(dma_src<B_T, BN, RC::BKS>(p.B, p.ldb, w.n0, p.N, w.k_begin, q - RC::A_CHUNKS, lane)) - kb_b);
    }
  };
  int islot = 0, inflight = 0;             // ring slot of the next issue; stages issued and not yet consumed
  auto issue_next = [&]() {
    if (!cur_live) return;
    const uint32_t st = smem_base + (uint32_t)(islot * RC::STAGE_BYTES);
#pragma unroll
    for (int i = 0; i < CPW; ++i) glds16s_m0(is_a[i] ? kb_a : kb_b, src[i], st + dst[i]);
    kb_a += step_a;
    kb_b += step_b;
    islot = (islot + 1 == NS) ? 0 : islot + 1;
    ++inflight;
    if (++cur_s == cur_ns) cursor_open(++cur_it);
  };
  cursor_open(0);
#pragma unroll
  for (int d = 0; d < PD; ++d) issue_next();

  // k-sums (dvla.h ksum_*): of the waves that hold the same operand rows (same wm: WN of them; same wn: WM), wave x sums the
  // k16-steps ks with ks % W == x (ks_wn / ks_wm = -1: this launch does not sum that operand)
  constexpr bool KSUM = EPI == EPI_F32 && RC::WM <= 4 && RC::WN <= 4;
  const int ks_wn = (KSUM && p.ksum_op == 1) ? wn : -1, ks_wm = (KSUM && p.ksum_op == 2) ? wm : -1;

  int cslot = 0;
  for (int it = 0;; ++it) {
    const int id = item_of(it);
    if (id < 0) break;
    const RingItem w = ring_item<RC>(p, id);

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ksa[TM], ksb[TN];          // k-sums of this wave's operand rows (EPI_F32 kernels; dead code elsewhere)
#pragma unroll
    for (int j = 0; j < TM; ++j) ksa[j] = 0.f;
#pragma unroll
    for (int i = 0; i < TN; ++i) ksb[i] = 0.f;

    for (int s = 0; s < w.ns; ++s) {
      // this wave's pieces of the oldest stage in flight have landed; the younger ones stay in flight
      if (PD >= 3 && inflight >= 3) wait_vmcnt<(PD >= 3 ? 2 : 0) * CPW>();
      else if (PD >= 2 && inflight == 2) wait_vmcnt<(PD >= 2 ? 1 : 0) * CPW>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // everybody's pieces are visible; everybody is done with the previous stage
      __builtin_amdgcn_sched_barrier(0);
      --inflight;
      issue_next();                    // refills the previous stage's buffer (possibly with the NEXT item's data)
      __builtin_amdgcn_sched_barrier(0);
      const char* st = smem + cslot * RC::STAGE_BYTES;
#pragma unroll
      for (int ks = 0; ks < RC::BKS / 16; ++ks) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[j] = ring_frag<A_T, BM, RC::BKS>(st, wm * (TM * 32) + j * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < TN; ++i) fb[i] = ring_frag<B_T, BN, RC::BKS>(st + RC::A_BYTES, wn * (TN * 32) + i * 32, ks, lane);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
        if constexpr (KSUM) {   // selector 0 in the waves whose k16-step this is not (and in launches that do not sum)
          const uint32_t sa = ks_wn == ks % RC::WN ? 0x3f803f80u : 0u, sb = ks_wm == ks % RC::WM ? 0x3f803f80u : 0u;
#pragma unroll
          for (int j = 0; j < TM; ++j) ksa[j] = ksum_add_sel(fa[j], sa, ksa[j]);
#pragma unroll
          for (int i = 0; i < TN; ++i) ksb[i] = ksum_add_sel(fb[i], sb, ksb[i]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      cslot = (cslot + 1 == NS) ? 0 : cslot + 1;
    }
    if constexpr (KSUM) {
      if (p.ksum_op == 1 && w.n0 == 0) ksum_store<TM, TM>(p, ksa, lane, w.split, wn, RC::WN < 4 ? RC::WN : 4, w.m0 + wm * (TM * 32), p.M);
      if (p.ksum_op == 2 && w.m0 == 0) ksum_store<TN, TN>(p, ksb, lane, w.split, wm, RC::WM < 4 ? RC::WM : 4, w.n0 + wn * 64, p.N);
    }
    reg_epilogue<TM, EPI>(p, acc, lane, w.m0 + wm * (TM * 32), w.n0 + wn * 64, w.split);
  }
}

// Workgroups per CU slot of the persistent kernels (plain schedule).  1 = one workgroup per slot walks its share of the
// tiles (best on an otherwise idle GPU: every prologue but the first hides under the previous tile's epilogue).  k > 1 =
// k times as many, shorter-lived workgroups handed out by the hardware dispatcher as slots free up: when a kernel on another
// stream (an RCCL collective during the backward pass) holds some CUs, a static share per CU makes the whole launch wait for
// the workgroups that could not start -- measured 1.7x for 16 of 256 CUs taken (profiles/r02_gemm_cu_contention.txt).
inline int& gemm_oversubscribe() {
  static int k = [] { const char* e = getenv("DVLA_GEMM_OVERSUBSCRIBE"); int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
  return k;
}

// workgroups of a persistent launch over `items` work items on `slots` co-resident workgroup slots: one per slot, or (over-
// subscription k > 1) up to k per slot with EQUAL item counts (ceil(items / (k slots)) items each, no ragged second wave)
inline int64_t persistent_grid(int64_t items, int64_t slots) {
  const int k = gemm_oversubscribe();
  if (k <= 1 || items <= slots) return items < slots ? items : slots;
  const int64_t per = (items + k * slots - 1) / (k * slots);
  return (items + per - 1) / per;
}

inline int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <class CF, bool A_T, bool B_T, int DBG = 0>
void launch_one(const GemmKArgs& a, int split_k, hipStream_t stream) {
  static bool attr_set = false;
  auto kern = &gemm_kernel<CF, A_T, B_T, DBG>;
  if (!attr_set) {  // > 64 KiB of LDS per workgroup needs the opt-in (160 KiB per CU on gfx950)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM_BYTES);
    attr_set = true;
  }
  dim3 grid((unsigned)(a.tiles_m * a.tiles_n), (unsigned)split_k, 1), block(CF::NT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, CF::SMEM_BYTES, stream, a);
}

template <class RC, bool A_T, bool B_T, int EPI>
void launch_ring_one(const GemmKArgs& a, int split_k, hipStream_t stream) {
  static bool attr_set = false;
  auto kern = &gemm_ring_kernel<RC, A_T, B_T, EPI>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RC::SMEM_BYTES);
    attr_set = true;
  }
  const int64_t items = (int64_t)a.tiles_m * a.tiles_n * split_k;
  dim3 grid((unsigned)persistent_grid(items, (int64_t)num_cus() * RC::WG_PER_CU), 1, 1), block(RC::NT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, RC::SMEM_BYTES, stream, a);
}

}  // namespace dvla_gemm

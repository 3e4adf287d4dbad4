// attention.hip -- fused multi-head attention (head_dim 64) forward + backward for gfx950.
//
// "Transposed flash attention": every score tile is computed as S^T = K.Q^T with
// v_mfma_f32_32x32x16_bf16(a = K fragment, b = Q fragment), so a lane owns ONE query column
// (q = lane & 31) and 16 of the 32 keys of the tile in its accumulator registers
// (key = (r&3) + 8*(r>>2) + 4*(lane>>5)).  Consequences:
//   * the softmax row max / row sum are in-lane reductions plus ONE cross-half exchange;
//   * the running max m, sum l and the O rescale factor are per-lane scalars;
//   * P (bf16, hardware v_cvt_pk_bf16_f32) is already in MFMA B-operand layout for O^T = V^T.P^T: no
//     cross-lane movement, no LDS round trip for P.  The k-slot <-> key permutation this implies is
//     applied to the V^T fragment, read from a "pair-interleaved" LDS image [key/2][d]
//     (dword = {key even, key odd}).
// K/V tiles (32 keys) are staged global -> registers -> LDS once per 128-query block (4 waves share
// them), double-buffered with the next tile's global loads in flight during the MFMAs; one barrier per
// tile.  Masks are 0/-inf only (models/dreamvla_model.py:25-66) and reach the kernel as bit tables plus
// a per-32x32-tile map: fully masked tiles are skipped, fully visible tiles never touch the table, mixed
// tiles read ONE 32-bit word per lane.  `key_index` lets the caller drop keys nobody can see (the
// prediction-query columns of the trunk mask) without copying K/V.
//
// Backward = delta kernel (rowsum dO.O) + dQ kernel (same structure as forward) + dK/dV kernel (a wave
// owns 32 keys, loops over query tiles; S = Q.K^T orientation so a lane owns one key and dK^T/dV^T
// accumulate in registers).  No atomics; S is recomputed in each (7 MFMA passes instead of 5).
#include "common.h"
#include "../../include/dvla.h"

namespace {

constexpr int AT_THREADS = 256;
constexpr int RM72 = 72;                 // row-major LDS row stride in bf16 (144 B, 16-B aligned, conflict-free)
constexpr int RM_BYTES = 32 * RM72 * 2;  // 4608
constexpr int PI_BYTES = 16 * 64 * 4;    // 4096
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnKArgs {
  const bf16_t *q, *k, *v; bf16_t* o;
  int64_t qsb, qst, qsh, ksb, kst, ksh, vsb, vst, vsh, osb, ost, osh;
  int B, H, Lq, Lk;
  float scale;
  const int32_t* key_index;
  const uint32_t *bits_q, *bits_k;
  const uint8_t* tile_map; int nqt, nkt;
  int has_drop; uint32_t drop_thr; float inv_keep; uint32_t seed_lo, seed_hi;
  float c1;   // scale * inv_keep, formed on the host: a kernel argument is an SGPR operand, a product formed in the kernel a VGPR
              // (gfx950 has no scalar float multiply) -- one register the 128-VGPR dQ kernel does not have
  float* lse;
  const bf16_t* dout; int64_t dsb, dst, dsh;
  float* delta;
  bf16_t *dq, *dk, *dv;
  int64_t dqsb, dqst, dqsh, dksb, dkst, dksh, dvsb, dvst, dvsh;
  int st16;   // bit 0 / 1 / 2 / 3: o / dq / dk / dv rows are 16-byte aligned (16-byte output stores)
  int fuse_delta;   // the dQ ring kernel computes and writes delta itself (no attn_delta_kernel launch)
};

// accumulator register r of lane group g  <->  row index inside the 32-row MFMA tile
__device__ __forceinline__ int acc_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

__device__ __forceinline__ uint4 load16(const bf16_t* p, bool ok) {
  return ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0u, 0u, 0u, 0u);
}

// thread `u` (0..127) of a 128-thread staging group owns tile rows (2*pr, 2*pr+1), columns oct*8..oct*8+7
__device__ __forceinline__ void stage_rows_load(uint4 (&reg)[2], const bf16_t* base, int64_t stride, int row0, int nrows,
                                                const int32_t* index, int u) {
  const int pr = u >> 3, oct = u & 7;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = row0 + 2 * pr + e;
    const bool ok = row < nrows;
    const int src = (ok && index) ? index[row] : row;
    reg[e] = load16(base + (int64_t)src * stride + oct * 8, ok);
  }
}
__device__ __forceinline__ void stage_store_rm(const uint4 (&reg)[2], char* lds, int u) {
  const int pr = u >> 3, oct = u & 7;
  *reinterpret_cast<uint4*>(lds + ((2 * pr) * RM72 + oct * 8) * 2) = reg[0];
  *reinterpret_cast<uint4*>(lds + ((2 * pr + 1) * RM72 + oct * 8) * 2) = reg[1];
}
__device__ __forceinline__ void stage_store_pi(const uint4 (&reg)[2], char* lds, int u) {
  const int pr = u >> 3, oct = u & 7;
  const uint32_t a[4] = {reg[0].x, reg[0].y, reg[0].z, reg[0].w};  // even row, cols oct*8 + {0,1},{2,3},...
  const uint32_t b[4] = {reg[1].x, reg[1].y, reg[1].z, reg[1].w};  // odd row
  uint32_t o[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = (a[i] & 0xffffu) | (b[i] << 16);
    o[2 * i + 1] = (a[i] >> 16) | (b[i] & 0xffff0000u);
  }
  uint32_t* dst = reinterpret_cast<uint32_t*>(lds) + pr * 64 + oct * 8;
  *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<uint4*>(dst + 4) = make_uint4(o[4], o[5], o[6], o[7]);
}
// row-major fragment: row (lane&31) of the tile, d-slots 16*s + 8*g .. +7
__device__ __forceinline__ bf16x8 frag_rm(const char* lds, int row, int s, int g) {
  return *reinterpret_cast<const bf16x8*>(lds + (row * RM72 + s * 16 + g * 8) * 2);
}
// transposed fragment from the pair-interleaved image: "row" index = column c (a d value), k-slots =
// tile rows acc_row(8*mm + jj, g), jj = 0..7   (pairs 8mm+2g, 8mm+2g+1, 8mm+4+2g, 8mm+4+2g+1)
__device__ __forceinline__ bf16x8 frag_pi(const char* lds, int c, int mm, int g) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(lds) + (8 * mm + 2 * g) * 64 + c;
  union { uint32_t w[4]; bf16x8 v; } u;
  u.w[0] = src[0]; u.w[1] = src[64]; u.w[2] = src[4 * 64]; u.w[3] = src[5 * 64];
  return u.v;
}
__device__ __forceinline__ bf16x8 pack_frag(const float* p) {
  union { uint32_t w[4]; bf16x8 v; } u;
  u.w[0] = pack2bf(p[0], p[1]); u.w[1] = pack2bf(p[2], p[3]);
  u.w[2] = pack2bf(p[4], p[5]); u.w[3] = pack2bf(p[6], p[7]);
  return u.v;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// store the transposed accumulators (lane = token row, registers = d) of one token.
//   vec16: the two half-waves hold alternating 4-value groups of the same row (d = 32 db + 8 rq + 4 g + e); one
//   v_permlane32_swap per packed dword turns two groups into 8 consecutive bf16 per lane -- 4 stores of 16 B per lane
//   instead of 16 of 8 B.  (The output tail of these kernels is store-ISSUE bound: with 8-byte pieces the "loop skeleton"
//   build of the forward kernel -- no arithmetic, no K / V traffic -- took 62 % of the full kernel's time at L = 205,
//   profiles/r02_attn_ablation.txt.)  Needs 16-byte aligned rows; otherwise the 8-byte form.
__device__ __forceinline__ void fa_swap_halves(uint32_t& lo_grp, uint32_t& hi_grp) {
  const auto r = __builtin_amdgcn_permlane32_swap(lo_grp, hi_grp, false, false);
  lo_grp = r[0]; hi_grp = r[1];
}
__device__ __forceinline__ void store_token(bf16_t* dst, const f32x16 (&acc)[2], float mul, int g, bool vec16) {
  if (vec16) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t a0 = pack2bf(acc[db][8 * q] * mul, acc[db][8 * q + 1] * mul), a1 = pack2bf(acc[db][8 * q + 2] * mul, acc[db][8 * q + 3] * mul);
        uint32_t b0 = pack2bf(acc[db][8 * q + 4] * mul, acc[db][8 * q + 5] * mul), b1 = pack2bf(acc[db][8 * q + 6] * mul, acc[db][8 * q + 7] * mul);
        fa_swap_halves(a0, b0); fa_swap_halves(a1, b1);
        *reinterpret_cast<uint4*>(dst + 32 * db + 16 * q + 8 * g) = make_uint4(a0, a1, b0, b1);
      }
    return;
  }
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int d = 32 * db + 8 * rq + 4 * g;
      *reinterpret_cast<uint2*>(dst + d) = make_uint2(pack2bf(acc[db][4 * rq] * mul, acc[db][4 * rq + 1] * mul),
                                                      pack2bf(acc[db][4 * rq + 2] * mul, acc[db][4 * rq + 3] * mul));
    }
}
// bit position of accumulator register r inside a visibility word already shifted right by 4*g
__device__ __forceinline__ bool vis_bit(uint32_t vg, int r) { return (vg >> ((r & 3) + 8 * (r >> 2))) & 1u; }
__device__ __forceinline__ uint32_t low_mask(int n) { return n >= 32 ? 0xffffffffu : ((1u << n) - 1u); }
// x where the visibility bit of register r is set, `masked_bits` (a float's bit pattern: 0 or -inf) where it is not.  The bit is
// sign-extended to a lane mask (v_bfe_i32) and merged with one v_bfi_b32 / v_and_b32: two VALU slots per element, where the
// compare + conditional move needs an and, a compare, the move and a hazard slot between the last two (round 5; same values)
// (round 6: the extract is an OPAQUE instruction.  With the builtin, hipcc recognised `x & sext(bit)` in the MASKED_BITS == 0 form -- the
// three backward kernels -- as a select and put back and + compare + hazard nop + conditional move: 64 issue slots per mixed tile
// where this function means 32; the forward kernel's -inf form kept the extract + bit-select.)
template <uint32_t MASKED_BITS>
__device__ __forceinline__ float vis_select(uint32_t vg, int r, float x) {
  const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)vg, (uint32_t)((r & 3) + 8 * (r >> 2)), 1u);     // 0 / ~0
  const uint32_t xi = __builtin_bit_cast(uint32_t, x);
  return __builtin_bit_cast(float, MASKED_BITS ? ((xi & m) | (~m & MASKED_BITS)) : (xi & m));
}
// all sixteen registers of a score tile (vg: the lane's visibility word already shifted right by 4 g).
// The zeroing form (the three backward kernels) is ONE asm statement that updates the sixteen values in place through two scratch
// registers: hipcc recognised `x & sext(bit)` of the builtin form as a select and put back and + compare + hazard nop + conditional
// move (64 issue slots per mixed tile where 32 are meant); sixteen separate asm statements cost a guard nop each and, with results
// apart from their operands, sixteen registers that the 128-register dQ kernel does not have (it spilled Q / dO fragments and
// reloaded them inside the tile loop, which drains the DMA ring) -- round 6, all three read off the assembly.
#define DVLA_VIS2(a, b, pa, pb)                                                                           \
  "v_bfe_i32 %16, %18, " #pa ", 1\n\tv_bfe_i32 %17, %18, " #pb ", 1\n\tv_and_b32 %" #a ", %" #a ", %16\n\tv_and_b32 %" #b ", %" #b ", %17\n\t"
template <uint32_t MASKED_BITS>
__device__ __forceinline__ void vis_select16(uint32_t vg, float (&x)[16]) {
  if (MASKED_BITS == 0u) {
    uint32_t t0, t1;
    asm(DVLA_VIS2(0, 1, 0, 1) DVLA_VIS2(2, 3, 2, 3) DVLA_VIS2(4, 5, 8, 9) DVLA_VIS2(6, 7, 10, 11)
        DVLA_VIS2(8, 9, 16, 17) DVLA_VIS2(10, 11, 18, 19) DVLA_VIS2(12, 13, 24, 25) DVLA_VIS2(14, 15, 26, 27)
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
          "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]),
          "=&v"(t0), "=&v"(t1)
        : "v"(vg));
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = vis_select<MASKED_BITS>(vg, r, x[r]);
  }
}
#undef DVLA_VIS2
constexpr uint32_t NEG_INF_BITS = 0xff800000u;

// The tile flags of a ring kernel's walk, for walks of at most 64 tiles (L <= 2048), as three wave-uniform 64-bit words taken by
// ballot from the LDS table: `need` -- some wave of the workgroup has work in the tile (the ring's walk), `any` / `mixed` -- this
// wave's own two flag bits.  Finding the next tile and decoding this wave's flag are then a shift and a find-first-set on SGPRs;
// the table walk they replace is an LDS byte read with its lgkmcnt(0) round trip per probed tile, twice per tile iteration
// (once for the DMA cursor three tiles ahead, once for the tile being multiplied).  Longer walks keep the table walk.
struct TileWords {
  uint64_t need, any, mixed;
  int n;            // tiles in the walk; 0: the words are not in use (n > 64)
  __device__ __forceinline__ int next(int t) const {
    if (t >= n) return -1;
    const uint64_t m = need >> t;
    return m ? t + __builtin_ctzll(m) : -1;
  }
  __device__ __forceinline__ int flag(int t) const { return (int)((any >> t) & 1u) + (int)((mixed >> t) & 1u); }   // 0 / 1 / 2
};
__device__ __forceinline__ TileWords tile_words(const uint8_t* flags, int n, int wave, int lane) {
  TileWords w{0, 0, 0, 0};
  if (n <= 64) {
    const int f = lane < n ? (int)flags[lane] : 0;
    const int fw = (f >> (2 * wave)) & 3;
    w.need = __ballot(f != 0);
    w.any = __ballot(fw != 0);
    w.mixed = __ballot(fw == 2);
    w.n = n;
  }
  return w;
}

__device__ __forceinline__ int tile_flag(const AttnKArgs& p, int qt, int kt) {
  if (qt >= p.nqt || kt >= p.nkt) return 0;
  return p.tile_map ? (int)p.tile_map[qt * p.nkt + kt] : 1;
}

// ====================================================================================================
// forward
// ====================================================================================================
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(AttnKArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (RM_BYTES + PI_BYTES)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, g = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qt0 = blockIdx.x * 4;
  const int qt = qt0 + wave;
  const int q = qt * 32 + l31;
  const bool q_ok = q < p.Lq;
  const float scale_log2 = p.scale * LOG2E;

  const bf16_t* qb = p.q + (int64_t)b * p.qsb + (int64_t)h * p.qsh;
  const bf16_t* kb = p.k + (int64_t)b * p.ksb + (int64_t)h * p.ksh;
  const bf16_t* vb = p.v + (int64_t)b * p.vsb + (int64_t)h * p.vsh;

  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 u = load16(qb + (int64_t)q * p.qst + 16 * s + 8 * g, q_ok);
    qf[s] = *reinterpret_cast<const bf16x8*>(&u);
  }
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 oacc[2] = {zero16(), zero16()};
  const int rowid = (b * p.H + h) * p.Lq + q;
  uint32_t rowkey = 0;
  if (p.has_drop) rowkey = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)rowid);

  auto blk_need = [&](int kt) -> bool {
    return (tile_flag(p, qt0, kt) | tile_flag(p, qt0 + 1, kt) | tile_flag(p, qt0 + 2, kt) | tile_flag(p, qt0 + 3, kt)) != 0;
  };
  auto next_needed = [&](int kt) -> int {
    while (kt < p.nkt && !blk_need(kt)) ++kt;
    return kt < p.nkt ? kt : -1;
  };
  const bool is_v_loader = t < 128;
  const int u = is_v_loader ? t : t - 128;
  uint4 reg[2];
  auto g_load = [&](int kt) {
    if (is_v_loader) stage_rows_load(reg, vb, p.vst, kt * 32, p.Lk, p.key_index, u);
    else stage_rows_load(reg, kb, p.kst, kt * 32, p.Lk, p.key_index, u);
  };
  auto l_store = [&](int buf) {
    char* base = smem + buf * (RM_BYTES + PI_BYTES);
    if (is_v_loader) stage_store_pi(reg, base + RM_BYTES, u);
    else stage_store_rm(reg, base, u);
  };

  int kt = next_needed(0);
  if (kt >= 0) { g_load(kt); l_store(0); }
  __syncthreads();
  int cur = 0;
  while (kt >= 0) {
    const int ktn = next_needed(kt + 1);
    if (ktn >= 0) g_load(ktn);
    const int flag = tile_flag(p, qt, kt);
    if (flag != 0) {
      const char* ks = smem + cur * (RM_BYTES + PI_BYTES);
      const char* vs = ks + RM_BYTES;
      f32x16 sacc = zero16();
#pragma unroll
      for (int s = 0; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(ks, l31, s, g), qf[s], sacc, 0, 0, 0);
      const int k0 = kt * 32;
      uint32_t vis = 0xffffffffu;
      if (flag == 2 && q_ok) vis = p.bits_q[q * p.nkt + kt];
      if (k0 + 32 > p.Lk) vis &= low_mask(p.Lk - k0);
      float sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r];
      if (__any(vis != 0xffffffffu)) {
        const uint32_t vg = vis >> (4 * g);
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = vis_bit(vg, r) ? sv[r] : -INFINITY;
      }
      float mt = sv[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sv[r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      // the running maximum is kept as an INTEGER in the log2 domain (rounded up): every rescale factor alpha is then an exact
      // power of two, so the bf16 rounding of P commutes with it -- O = sum_j bf16(2^(s_j - M)) v_j / sum_j 2^(s_j - M) whatever
      // the tile order or the intermediate maxima, which is what lets oracle/torch_ref.py::attention_bf16 restate the kernel's
      // arithmetic exactly (same two rounding points: P and dS) instead of waving a 3e-3 tolerance through
      const float m_new = fmaxf(m_run, ceilf(mt * scale_log2));
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2(m_run - m_safe);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sv[r] = fast_exp2(fmaf(sv[r], scale_log2, -m_safe)); rs += sv[r]; }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run * alpha + rs;
      m_run = m_new;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
      }
      if (p.has_drop) {        // one hash per (row, tile), one 24-bit multiply-add per element (common.h)
        const uint32_t tk = drop_tilekey(rowkey, (uint32_t)kt);
        const uint32_t dx = drop_rot(tk, (uint32_t)g);
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = (drop_elem(dx, tk, DVLA_DROP_C(r)) >= p.drop_thr) ? sv[r] : 0.f;   // 1 / keep: on the output (inv_l)
      }
      const bf16x8 pf0 = pack_frag(sv), pf1 = pack_frag(sv + 8);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_pi(vs, 32 * db + l31, 0, g), pf0, oacc[db], 0, 0, 0);
        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_pi(vs, 32 * db + l31, 1, g), pf1, oacc[db], 0, 0, 0);
      }
    }
    if (ktn >= 0) l_store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    kt = ktn;
  }
  if (q_ok) {
    const float inv_l = l_run > 0.f ? p.inv_keep / l_run : 0.f;     // 1 / keep of the dropout rides on 1 / l (1 without dropout)
    store_token(p.o + (int64_t)b * p.osb + (int64_t)q * p.ost + (int64_t)h * p.osh, oacc, inv_l, g, (p.st16 & 1) != 0);
    if (p.lse && g == 0) p.lse[rowid] = l_run > 0.f ? (m_run + log2f(l_run)) * LN2 : INFINITY;
  }
}

// ====================================================================================================
// forward, LDS-DMA ring version (the default; the kernel above is the fallback for masks / key lists too large for LDS)
//
//   K and V tiles (32 keys x 64 d, 4 KiB each, row-major as they lie in HBM) are copied global -> LDS by
//   global_load_lds_dwordx4 through a 4-stage ring, three tiles ahead of the one being multiplied: no staging
//   registers, no ds_write pass, and the HBM / L2 latency of a tile is hidden behind the two tiles before it (the
//   register-staged kernel above looks ONE 32-key tile ahead -- ~300 MFMA clocks -- and stalls on every tile).
//     K image : rows of 128 B, d-octet o of key r in slot o ^ ((r >> 1) & 7) (swizzle applied on the per-lane SOURCE
//               address, LDS image lane-linear as LDS-DMA requires): conflict-free ds_read_b128 fragments.
//     V image : the same rows with slot o ^ (4 * ((r >> 1) & 1)); V^T fragments come straight out of it with the
//               hardware transpose ds_read_b64_tr_b16 (lane = d, 4 consecutive keys per instruction), conflict-free.
//   The 2-bit tile flags of the workgroup's four query tiles, the key gather list and the per-query visibility words
//   are copied to LDS once, so the tile loop contains no global load except the DMA (an ordinary load inside the loop
//   would make hipcc drain the DMA ring with s_waitcnt vmcnt(0) at its first use).
// ====================================================================================================
constexpr int FA_NS = 4, FA_TILE = 4096, FA_STAGE = 2 * FA_TILE, FA_RING = FA_NS * FA_STAGE;

// one LDS-DMA request: 16 B per lane from sbase (wave-uniform: SGPR pair) + voff (per-lane BYTE offset, 32 bit) to LDS
// lds_dst + 16 lane.  The scalar-base form keeps the per-lane address state at one VGPR per operand (64-bit per-lane pointers
// cost the dQ kernel 4 spilled VGPRs at 128 -- and a scratch reload inside the loop is a VMEM load: s_waitcnt vmcnt(0),
// i.e. the ring drained every tile).  The host takes the staged kernels when an operand spans 4 GiB or more.
// M0 (the LDS base of the transfer) is set and NOT restored: nothing else in these kernels reads M0 (tests/test_phase_isa.py checks
// the built code object), so the save / restore pair of rounds 2-4 was two scalar instructions per request for nothing.
__device__ __forceinline__ void fa_glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// four fragments read from LDS are all on their way before the first of them is used (an empty asm that "uses" the four)
__device__ __forceinline__ void fa_group4(bf16x8 (&f)[4]) { asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); }
template <int N>
__device__ __forceinline__ void fa_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((ext_vector_type(4))) short fa_s16x4;
// K fragment: key row (lane & 31), d-slots 16 s + 8 g .. + 7
__device__ __forceinline__ bf16x8 fa_frag_k(const char* kt, int row, int s, int g) {
  return *reinterpret_cast<const bf16x8*>(kt + row * 128 + ((((2 * s + g) ^ ((row >> 1) & 7))) << 4));
}
// V^T fragment for output half db (d = 32 db + (lane & 31)) and key half mm: k-slot j of lane group g is key
// 16 mm + 8 (j >> 2) + 4 g + (j & 3) -- the key of P register 8 mm + j (acc_row) -- fetched as two transposing reads.
__device__ __forceinline__ bf16x8 fa_frag_vt(const char* vt, int db, int mm, int lane) {
  const int gi = lane >> 4, c = lane & 15;
  const int d = 32 * db + 16 * (gi & 1) + 4 * (c & 3);            // first of the 4 d values this lane FETCHES
  union { fa_s16x4 h[2]; bf16x8 v; } u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int key = 16 * mm + 8 * h + 4 * (gi >> 1) + (c >> 2);   // key row this lane fetches from
    const int byte = key * 128 + ((((d >> 3) ^ (4 * ((key >> 1) & 1)))) << 4) + (d & 7) * 2;
    u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fa_s16x4*)(vt + byte));
  }
  return u.v;
}

__host__ __device__ inline int fa_pad16(int n) { return (n + 15) & ~15; }
// Registers filled by ordinary global loads BEFORE a DMA-ring loop must be waited for before the loop: hipcc places the wait at
// the first USE, i.e. inside the loop, as s_waitcnt vmcnt(0) -- which on the hardware also drains every LDS-DMA transfer the
// ring has in flight (one in-order counter): each tile iteration then pays the full L2 / HBM latency of the tile issued a
// few instructions earlier (cdna guide section 6, trap 4b; found in the ISA of all three ring kernels in round 2).  An empty
// asm that "modifies" the register makes the compiler wait right here.
template <class T>
__device__ __forceinline__ void fa_settle(T& v) { asm volatile("" : "+v"(v)); }

// dynamic LDS of the ring kernels: ring | tile flags [nkt] | key gather list [Lk] | visibility words [128][nkt]
__host__ __device__ inline size_t fa_smem_bytes(int nkt, int Lk, bool has_index, bool has_bits) {
  return (size_t)FA_RING + fa_pad16(nkt) + (has_index ? (size_t)fa_pad16(Lk * 4) : 0) + (has_bits ? (size_t)128 * nkt * 4 : 0);
}

// DBG (ablation builds selected by env DVLA_ATTN_DBG, results garbage by design, timing only -- tests/gpu_attn_ablate.py):
//   1 no barrier / DMA wait in the tile loop, 2 no softmax arithmetic, 4 no P.V (fragment reads + MFMAs), 8 no K.Q^T,
//   16 no DMA after the prologue, 32 no Q load / O store, 64 return at once (launch + dispatch)
template <int DBG>
__global__ __launch_bounds__(AT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd_ring_kernel(AttnKArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (DBG & 64) return;                       // launch + workgroup dispatch only
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, g = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int qt0 = blockIdx.x * 4;
  const int qt = qt0 + wave;
  const int q = qt * 32 + l31;
  const bool q_ok = (DBG & 32) ? false : q < p.Lq;   // 32: no Q load, no O store
  const float scale_log2 = p.scale * LOG2E;
  const bool has_bits = p.tile_map != nullptr && p.bits_q != nullptr;

  uint8_t* flags = reinterpret_cast<uint8_t*>(smem + FA_RING);
  int32_t* kidx = reinterpret_cast<int32_t*>(smem + FA_RING + fa_pad16(p.nkt));
  uint32_t* bits = reinterpret_cast<uint32_t*>(smem + FA_RING + fa_pad16(p.nkt) + (p.key_index ? fa_pad16(p.Lk * 4) : 0));

  // The tables depend on the query block only: a workgroup loads them ONCE and then walks (batch, head) items blockIdx.y,
  // blockIdx.y + gridDim.y, ... (round 2's ablation: the skeleton of this kernel -- tables, flag scan, no DMA, no arithmetic --
  // was 44 of the trunk's 89 us with one item per workgroup; the host sizes gridDim.y for a few items per workgroup when there
  // are tables and one item otherwise).
  for (int kt = t; kt < p.nkt; kt += AT_THREADS) {
    int f = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) f |= tile_flag(p, qt0 + w, kt) << (2 * w);
    flags[kt] = (uint8_t)f;
  }
  if (p.key_index)
    for (int i = t; i < p.Lk; i += AT_THREADS) kidx[i] = p.key_index[i];
  if (has_bits)
    for (int i = t; i < 128 * p.nkt; i += AT_THREADS) {
      const int ql = i / p.nkt, kt = i - ql * p.nkt;
      const int qq = qt0 * 32 + ql;
      bits[i] = qq < p.Lq ? p.bits_q[(int64_t)qq * p.nkt + kt] : 0u;
    }
  __syncthreads();
  const TileWords tw = tile_words(flags, p.nkt, wave, lane);

  const int n_items = p.B * p.H;
  for (int bh = blockIdx.y; bh < n_items; bh += gridDim.y) {
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16_t* qb = p.q + (int64_t)b * p.qsb + (int64_t)h * p.qsh;
  const bf16_t* kb = p.k + (int64_t)b * p.ksb + (int64_t)h * p.ksh;
  const bf16_t* vb = p.v + (int64_t)b * p.vsb + (int64_t)h * p.vsh;

  // the Q fragments are requested FIRST and waited for LAST (after the first K / V tiles are on their way): the item's memory
  // round trips -- Q, first tiles -- overlap instead of queueing (short sequences are latency-bound: L = 205 has 7 tiles)
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 u = load16(qb + (int64_t)q * p.qst + 16 * s + 8 * g, q_ok);
    qf[s] = *reinterpret_cast<const bf16x8*>(&u);
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 oacc[2] = {zero16(), zero16()};
  const int rowid = (b * p.H + h) * p.Lq + q;
  uint32_t rowkey = 0;
  if (p.has_drop) rowkey = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)rowid);

  auto next_needed = [&](int kt) -> int {
    if (tw.n) return tw.next(kt);
    while (kt < p.nkt && __builtin_amdgcn_readfirstlane((int)flags[kt]) == 0) ++kt;
    return kt < p.nkt ? kt : -1;
  };
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t kst2 = (uint32_t)p.kst * 2u, vst2 = (uint32_t)p.vst * 2u;   // row strides in bytes (operand < 4 GiB: host check)
  // wave w copies rows 8w .. 8w+7 of the K tile and of the V tile: lane -> (row 8w + lane/8, LDS slot lane % 8)
  const int rl = 8 * wave + (lane >> 3);
  const int oct_k = (lane & 7) ^ ((rl >> 1) & 7);
  const int oct_v = (lane & 7) ^ (4 * ((rl >> 1) & 1));
  auto issue = [&](int kt, int slot) {
    int row = kt * 32 + rl;
    row = row < p.Lk ? row : p.Lk - 1;
    const uint32_t src = p.key_index ? (uint32_t)kidx[row] : (uint32_t)row;
    const uint32_t dst = smem_base + (uint32_t)(slot * FA_STAGE + wave * 1024);
    fa_glds16(kb, src * kst2 + oct_k * 16, __builtin_amdgcn_readfirstlane(dst));
    fa_glds16(vb, src * vst2 + oct_v * 16, __builtin_amdgcn_readfirstlane(dst + FA_TILE));
  };

  int kt = next_needed(0);
  int kti = kt, islot = 0, cslot = 0, inflight = 0;
#pragma unroll
  for (int d = 0; d < FA_NS - 1; ++d)
    if (kti >= 0) {
      issue(kti, islot);
      islot = (islot + 1) & (FA_NS - 1); ++inflight;
      kti = next_needed(kti + 1);
    }
#pragma unroll
  for (int s = 0; s < 4; ++s) fa_settle(qf[s]);
  while (kt >= 0) {
    if (!(DBG & 1)) {
      if (inflight >= 3) fa_wait_vmcnt<4>();
      else if (inflight == 2) fa_wait_vmcnt<2>();
      else fa_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    --inflight;
    if (kti >= 0) {
      if (!(DBG & 16)) issue(kti, islot);
      islot = (islot + 1) & (FA_NS - 1); ++inflight;
      kti = next_needed(kti + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int flag = tw.n ? tw.flag(kt) : (__builtin_amdgcn_readfirstlane((int)flags[kt]) >> (2 * wave)) & 3;
    if (flag != 0) {
      const char* ks = smem + cslot * FA_STAGE;
      const char* vs = ks + FA_TILE;
      f32x16 sacc = zero16();
      if (!(DBG & 8)) {
        // all four K fragments are requested before the first multiply (fa_group4): hipcc otherwise recycles ONE fragment register
        // quad -- read, lgkmcnt(0), multiply, four times over -- and the wave sits through four LDS round trips per tile
        bf16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = fa_frag_k(ks, l31, s, g);
        fa_group4(kf);
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], sacc, 0, 0, 0);
      }
      const int k0 = kt * 32;
      uint32_t vis = 0xffffffffu;
      if (flag == 2) vis = bits[(wave * 32 + l31) * p.nkt + kt];
      if (k0 + 32 > p.Lk) vis &= low_mask(p.Lk - k0);
      float sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r];
      if (__any(vis != 0xffffffffu)) {
        const uint32_t vg = vis >> (4 * g);
        vis_select16<NEG_INF_BITS>(vg, sv);
      }
      if (!(DBG & 2)) {
      float mt = sv[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sv[r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      // the running maximum is kept as an INTEGER in the log2 domain (rounded up): every rescale factor alpha is then an exact
      // power of two, so the bf16 rounding of P commutes with it -- O = sum_j bf16(2^(s_j - M)) v_j / sum_j 2^(s_j - M) whatever
      // the tile order or the intermediate maxima, which is what lets oracle/torch_ref.py::attention_bf16 restate the kernel's
      // arithmetic exactly (same two rounding points: P and dS) instead of waving a 3e-3 tolerance through
      const float m_new = fmaxf(m_run, ceilf(mt * scale_log2));
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2(m_run - m_safe);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sv[r] = fast_exp2(fmaf(sv[r], scale_log2, -m_safe)); rs += sv[r]; }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run * alpha + rs;
      m_run = m_new;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
      }
      }
      if (p.has_drop) {        // one hash per (row, tile), one 24-bit multiply-add per element (common.h)
        const uint32_t tk = drop_tilekey(rowkey, (uint32_t)kt);
        const uint32_t dx = drop_rot(tk, (uint32_t)g);
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = (drop_elem(dx, tk, DVLA_DROP_C(r)) >= p.drop_thr) ? sv[r] : 0.f;   // 1 / keep: on the output (inv_l)
      }
      if (DBG & 4) {
        const bf16x8 pf0 = pack_frag(sv), pf1 = pack_frag(sv + 8);
        asm volatile("" :: "v"(pf0), "v"(pf1));
      } else {
        // the four V^T fragments (eight transposing reads) are requested first, P is packed under their latency, and the two
        // output halves alternate so that consecutive multiplies do not wait on each other's accumulator
        bf16x8 vf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vf[i] = fa_frag_vt(vs, i >> 1, i & 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 pf0 = pack_frag(sv), pf1 = pack_frag(sv + 8);
        fa_group4(vf);
        oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pf0, oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], pf0, oacc[1], 0, 0, 0);
        oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pf1, oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3], pf1, oacc[1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    cslot = (cslot + 1) & (FA_NS - 1);
    kt = next_needed(kt + 1);
  }
  if (q_ok) {
    const float inv_l = l_run > 0.f ? p.inv_keep / l_run : 0.f;     // 1 / keep of the dropout rides on 1 / l (1 without dropout)
    store_token(p.o + (int64_t)b * p.osb + (int64_t)q * p.ost + (int64_t)h * p.osh, oacc, inv_l, g, (p.st16 & 1) != 0);
    if (p.lse && g == 0) p.lse[rowid] = l_run > 0.f ? (m_run + log2f(l_run)) * LN2 : INFINITY;
  }
  __syncthreads();   // every wave is done with the ring before the next item's first DMA lands in it
  }
}

// ====================================================================================================
// backward: delta[b,h,q] = sum_d dO * O   (8 lanes per row of 64)
// ====================================================================================================
__global__ void attn_delta_kernel(AttnKArgs p) {
  const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gt >> 3;
  const int part = (int)(gt & 7);
  const int64_t nrows = (int64_t)p.B * p.H * p.Lq;
  float s = 0.f;
  if (row < nrows) {
    const int64_t q = row % p.Lq, bh = row / p.Lq, h = bh % p.H, b = bh / p.H;
    const uint4 a = *reinterpret_cast<const uint4*>(p.dout + b * p.dsb + q * p.dst + h * p.dsh + part * 8);
    const uint4 c = *reinterpret_cast<const uint4*>(p.o + b * p.osb + q * p.ost + h * p.osh + part * 8);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s += bf2f((bf16_t)(aw[i] & 0xffff)) * bf2f((bf16_t)(cw[i] & 0xffff));
      s += bf2f((bf16_t)(aw[i] >> 16)) * bf2f((bf16_t)(cw[i] >> 16));
    }
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (row < nrows && part == 0) p.delta[row] = s;
}

// ====================================================================================================
// backward: dQ  (one wave = 32 queries, loop over key tiles; same orientation as forward)
//   LDS per buffer: K row-major | K pair-interleaved | V row-major
// ====================================================================================================
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_kernel(AttnKArgs p) {
  constexpr int BUF = 2 * RM_BYTES + PI_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, g = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int qt0 = blockIdx.x * 4;
  const int qt = qt0 + wave;
  const int q = qt * 32 + l31;
  const bool q_ok = q < p.Lq;
  const float scale_log2 = p.scale * LOG2E;
  const bf16_t* qb = p.q + (int64_t)b * p.qsb + (int64_t)h * p.qsh;
  const bf16_t* kb = p.k + (int64_t)b * p.ksb + (int64_t)h * p.ksh;
  const bf16_t* vb = p.v + (int64_t)b * p.vsb + (int64_t)h * p.vsh;
  const bf16_t* dob = p.dout + (int64_t)b * p.dsb + (int64_t)h * p.dsh;

  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 a = load16(qb + (int64_t)q * p.qst + 16 * s + 8 * g, q_ok);
    const uint4 c = load16(dob + (int64_t)q * p.dst + 16 * s + 8 * g, q_ok);
    qf[s] = *reinterpret_cast<const bf16x8*>(&a);
    dof[s] = *reinterpret_cast<const bf16x8*>(&c);
  }
  const int rowid = (b * p.H + h) * p.Lq + q;
  const float lse2 = q_ok ? p.lse[rowid] * LOG2E : INFINITY;
  const float dlt = q_ok ? p.delta[rowid] * p.scale : 0.f;       // delta and dP enter dS already scaled: dS = P (c1 dP' - scale delta),
  const float c1 = p.c1;                                          // c1 = scale / keep, dP' = the kept dP
  uint32_t rowkey = 0;
  if (p.has_drop) rowkey = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)rowid);
  f32x16 dqacc[2] = {zero16(), zero16()};

  auto blk_need = [&](int kt) -> bool {
    return (tile_flag(p, qt0, kt) | tile_flag(p, qt0 + 1, kt) | tile_flag(p, qt0 + 2, kt) | tile_flag(p, qt0 + 3, kt)) != 0;
  };
  auto next_needed = [&](int kt) -> int {
    while (kt < p.nkt && !blk_need(kt)) ++kt;
    return kt < p.nkt ? kt : -1;
  };
  const bool is_k_loader = t < 128;
  const int u = is_k_loader ? t : t - 128;
  uint4 reg[2];
  auto g_load = [&](int kt) {
    if (is_k_loader) stage_rows_load(reg, kb, p.kst, kt * 32, p.Lk, p.key_index, u);
    else stage_rows_load(reg, vb, p.vst, kt * 32, p.Lk, p.key_index, u);
  };
  auto l_store = [&](int buf) {
    char* base = smem + buf * BUF;
    if (is_k_loader) { stage_store_rm(reg, base, u); stage_store_pi(reg, base + RM_BYTES, u); }
    else stage_store_rm(reg, base + RM_BYTES + PI_BYTES, u);
  };

  int kt = next_needed(0);
  if (kt >= 0) { g_load(kt); l_store(0); }
  __syncthreads();
  int cur = 0;
  while (kt >= 0) {
    const int ktn = next_needed(kt + 1);
    if (ktn >= 0) g_load(ktn);
    const int flag = tile_flag(p, qt, kt);
    if (flag != 0) {
      const char* k_rm = smem + cur * BUF;
      const char* k_pi = k_rm + RM_BYTES;
      const char* v_rm = k_pi + PI_BYTES;
      f32x16 sacc = zero16(), dpacc = zero16();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(k_rm, l31, s, g), qf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(v_rm, l31, s, g), dof[s], dpacc, 0, 0, 0);
      }
      const int k0 = kt * 32;
      uint32_t vis = 0xffffffffu;
      if (flag == 2 && q_ok) vis = p.bits_q[q * p.nkt + kt];
      if (k0 + 32 > p.Lk) vis &= low_mask(p.Lk - k0);
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(fmaf(sacc[r], scale_log2, -lse2));
      if (__any(vis != 0xffffffffu)) {
        const uint32_t vg = vis >> (4 * g);
        vis_select16<0u>(vg, ds);
      }
      // dS = P (c1 dP' - dlt), dlt = scale * delta, c1 = scale / keep, dP' = the kept dP: scale and 1 / keep cost nothing here.
      // (The dropout test stays INSIDE the element loop in this kernel: hipcc then keeps sixteen small basic blocks -- a scalar test
      // and a branch per element, ~3 issue slots each -- but hoisted around the loop, as in the dK/dV kernel, the straight-line
      // schedule needs more than the 128 registers of four waves per SIMD and reloads Q / dO fragments from scratch inside the
      // tile loop, which drains the DMA ring (round 6, measured in the assembly: 6 reloads per tile).)
      uint32_t tk = 0u, dx = 0u;   // one hash per (row, tile), one 24-bit multiply-add per element (common.h)
      if (p.has_drop) { tk = drop_tilekey(rowkey, (uint32_t)kt); dx = drop_rot(tk, (uint32_t)g); }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float dp = dpacc[r];
        if (p.has_drop) dp = (drop_elem(dx, tk, DVLA_DROP_C(r)) >= p.drop_thr) ? dp : 0.f;
        ds[r] = ds[r] * fmaf(dp, c1, -dlt);
      }
      const bf16x8 f0 = pack_frag(ds), f1 = pack_frag(ds + 8);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_pi(k_pi, 32 * db + l31, 0, g), f0, dqacc[db], 0, 0, 0);
        dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_pi(k_pi, 32 * db + l31, 1, g), f1, dqacc[db], 0, 0, 0);
      }
    }
    if (ktn >= 0) l_store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    kt = ktn;
  }
  if (q_ok) store_token(p.dq + (int64_t)b * p.dqsb + (int64_t)q * p.dqst + (int64_t)h * p.dqsh, dqacc, 1.0f, g, (p.st16 & 2) != 0);
}

// ====================================================================================================
// backward: dK, dV  (one wave = 32 keys, loop over query tiles)
//   S = Q.K^T orientation: mfma(a = Q fragment (row = query), b = K fragment) -> lane owns one key,
//   registers = 16 queries.  dV^T[d][key] += dO^T[d][q] . Pdrop[q][key],  dK^T[d][key] += Q^T[d][q] . dS[q][key]
//   LDS per buffer: Q row-major | Q pair-interleaved | dO row-major | dO pair-interleaved | lse2[32] | delta[32]
// ====================================================================================================
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkv_kernel(AttnKArgs p) {
  constexpr int STAT = 2 * 32 * 4;
  constexpr int BUF = 2 * RM_BYTES + 2 * PI_BYTES + STAT;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, g = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kt0 = blockIdx.x * 4;
  const int ktw = kt0 + wave;
  const int key = ktw * 32 + l31;
  const bool key_ok = key < p.Lk;
  const int key_row = (key_ok && p.key_index) ? p.key_index[key] : key;
  const float scale_log2 = p.scale * LOG2E;
  const float c1 = p.c1;                          // dS = P (c1 dP' - scale delta): scale and 1 / keep ride on one fma
  const bf16_t* qb = p.q + (int64_t)b * p.qsb + (int64_t)h * p.qsh;
  const bf16_t* kb = p.k + (int64_t)b * p.ksb + (int64_t)h * p.ksh;
  const bf16_t* vb = p.v + (int64_t)b * p.vsb + (int64_t)h * p.vsh;
  const bf16_t* dob = p.dout + (int64_t)b * p.dsb + (int64_t)h * p.dsh;

  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 a = load16(kb + (int64_t)key_row * p.kst + 16 * s + 8 * g, key_ok);
    const uint4 c = load16(vb + (int64_t)key_row * p.vst + 16 * s + 8 * g, key_ok);
    kf[s] = *reinterpret_cast<const bf16x8*>(&a);
    vf[s] = *reinterpret_cast<const bf16x8*>(&c);
  }
  f32x16 dkacc[2] = {zero16(), zero16()}, dvacc[2] = {zero16(), zero16()};
  const int bh_row0 = (b * p.H + h) * p.Lq;

  auto blk_need = [&](int qt) -> bool {
    return (tile_flag(p, qt, kt0) | tile_flag(p, qt, kt0 + 1) | tile_flag(p, qt, kt0 + 2) | tile_flag(p, qt, kt0 + 3)) != 0;
  };
  auto next_needed = [&](int qt) -> int {
    while (qt < p.nqt && !blk_need(qt)) ++qt;
    return qt < p.nqt ? qt : -1;
  };
  const bool is_q_loader = t < 128;
  const int u = is_q_loader ? t : t - 128;
  uint4 reg[2];
  float stat = 0.f;
  auto g_load = [&](int qt) {
    if (is_q_loader) stage_rows_load(reg, qb, p.qst, qt * 32, p.Lq, nullptr, u);
    else stage_rows_load(reg, dob, p.dst, qt * 32, p.Lq, nullptr, u);
    if (t < 64) {
      const int qq = qt * 32 + (t & 31);
      if (t < 32) stat = qq < p.Lq ? p.lse[bh_row0 + qq] * LOG2E : INFINITY;
      else stat = qq < p.Lq ? p.delta[bh_row0 + qq] * p.scale : 0.f;    // scaled once here: dS = P (c1 dP' - scale delta)
    }
  };
  auto l_store = [&](int buf) {
    char* base = smem + buf * BUF + (is_q_loader ? 0 : (RM_BYTES + PI_BYTES));
    stage_store_rm(reg, base, u);
    stage_store_pi(reg, base + RM_BYTES, u);
    if (t < 64) reinterpret_cast<float*>(smem + buf * BUF + 2 * RM_BYTES + 2 * PI_BYTES)[t] = stat;
  };

  int qt = next_needed(0);
  if (qt >= 0) { g_load(qt); l_store(0); }
  __syncthreads();
  int cur = 0;
  while (qt >= 0) {
    const int qtn = next_needed(qt + 1);
    if (qtn >= 0) g_load(qtn);
    const int flag = tile_flag(p, qt, ktw);
    if (flag != 0) {
      const char* q_rm = smem + cur * BUF;
      const char* q_pi = q_rm + RM_BYTES;
      const char* do_rm = q_pi + PI_BYTES;
      const char* do_pi = do_rm + RM_BYTES;
      const float* st = reinterpret_cast<const float*>(do_pi + PI_BYTES);
      f32x16 sacc = zero16(), dpacc = zero16();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(q_rm, l31, s, g), kf[s], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(do_rm, l31, s, g), vf[s], dpacc, 0, 0, 0);
      }
      const int q0 = qt * 32;
      float lse2[16], dlt[16];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 a = *reinterpret_cast<const float4*>(st + 8 * rq + 4 * g);
        const float4 c = *reinterpret_cast<const float4*>(st + 32 + 8 * rq + 4 * g);
        lse2[4 * rq] = a.x; lse2[4 * rq + 1] = a.y; lse2[4 * rq + 2] = a.z; lse2[4 * rq + 3] = a.w;
        dlt[4 * rq] = c.x; dlt[4 * rq + 1] = c.y; dlt[4 * rq + 2] = c.z; dlt[4 * rq + 3] = c.w;
      }
      uint32_t vis = key_ok ? 0xffffffffu : 0u;  // bit i: query q0+i sees this lane's key
      if (flag == 2 && key_ok) vis = p.bits_k[key * p.nqt + qt];
      float pr[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[r] = fast_exp2(fmaf(sacc[r], scale_log2, -lse2[r]));  // q >= Lq: lse2 = +inf -> 0
      if (__any(vis != 0xffffffffu)) {
        const uint32_t vg = vis >> (4 * g);
        vis_select16<0u>(vg, pr);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float dp = dpacc[r];
        float pdrop = pr[r];
        if (p.has_drop) {      // (fallback kernel: the tile key is hashed per element here; the ring kernel keeps a table)
          const uint32_t rk = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)(bh_row0 + q0 + acc_row(r, g)));
          const uint32_t tk = drop_tilekey(rk, (uint32_t)ktw);
          const bool keep = drop_elem(drop_rot(tk, (uint32_t)((l31 >> 2) & 1)), tk, DVLA_DROP_C((l31 & 3) + 4 * (l31 >> 3))) >= p.drop_thr;
          dp = keep ? dp : 0.f;                // 1 / keep: in c1 for dS, on the stored dV for P
          pdrop = keep ? pdrop : 0.f;
        }
        ds[r] = pr[r] * fmaf(dp, c1, -dlt[r]);     // the staged statistics hold scale * delta; c1 = scale / keep
        pr[r] = pdrop;
      }
      const bf16x8 pf0 = pack_frag(pr), pf1 = pack_frag(pr + 8);
      const bf16x8 sf0 = pack_frag(ds), sf1 = pack_frag(ds + 8);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        dvacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_pi(do_pi, 32 * db + l31, 0, g), pf0, dvacc[db], 0, 0, 0);
        dvacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_pi(do_pi, 32 * db + l31, 1, g), pf1, dvacc[db], 0, 0, 0);
        dkacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_pi(q_pi, 32 * db + l31, 0, g), sf0, dkacc[db], 0, 0, 0);
        dkacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_pi(q_pi, 32 * db + l31, 1, g), sf1, dkacc[db], 0, 0, 0);
      }
    }
    if (qtn >= 0) l_store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    qt = qtn;
  }
  if (key_ok) {
    store_token(p.dk + (int64_t)b * p.dksb + (int64_t)key_row * p.dkst + (int64_t)h * p.dksh, dkacc, 1.0f, g, (p.st16 & 4) != 0);
    store_token(p.dv + (int64_t)b * p.dvsb + (int64_t)key_row * p.dvst + (int64_t)h * p.dvsh, dvacc, p.inv_keep, g, (p.st16 & 8) != 0);
  }
}

inline bool ok16(const void* ptr, int64_t s0, int64_t s1, int64_t s2) {
  return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && (s0 % 8 == 0) && (s1 % 8 == 0) && (s2 % 8 == 0);
}

int fill_args(const dvla_attn_params* q, AttnKArgs& a) {
  if (!q || !q->q || !q->k || !q->v || !q->o) return DVLA_ERR_ARG;
  if (q->B <= 0 || q->H <= 0 || q->Lq <= 0 || q->Lk <= 0) return DVLA_ERR_ARG;
  if (q->dropout_p < 0.f || q->dropout_p >= 1.f || !(q->scale > 0.f)) return DVLA_ERR_ARG;
  if ((int64_t)q->B * q->H * q->Lq >= (1LL << 31)) return DVLA_ERR_UNSUPPORTED;
  if (!ok16(q->q, q->q_stride_b, q->q_stride_t, q->q_stride_h) || !ok16(q->k, q->k_stride_b, q->k_stride_t, q->k_stride_h) ||
      !ok16(q->v, q->v_stride_b, q->v_stride_t, q->v_stride_h) ||
      !(reinterpret_cast<uintptr_t>(q->o) % 8 == 0 && q->o_stride_b % 4 == 0 && q->o_stride_t % 4 == 0 && q->o_stride_h % 4 == 0))
    return DVLA_ERR_UNSUPPORTED;
  if (q->tile_map && !q->mask_bits_q) return DVLA_ERR_ARG;
  a.q = (const bf16_t*)q->q; a.k = (const bf16_t*)q->k; a.v = (const bf16_t*)q->v; a.o = (bf16_t*)q->o;
  a.qsb = q->q_stride_b; a.qst = q->q_stride_t; a.qsh = q->q_stride_h;
  a.ksb = q->k_stride_b; a.kst = q->k_stride_t; a.ksh = q->k_stride_h;
  a.vsb = q->v_stride_b; a.vst = q->v_stride_t; a.vsh = q->v_stride_h;
  a.osb = q->o_stride_b; a.ost = q->o_stride_t; a.osh = q->o_stride_h;
  a.B = q->B; a.H = q->H; a.Lq = q->Lq; a.Lk = q->Lk;
  a.scale = q->scale;
  a.key_index = q->key_index;
  a.bits_q = q->mask_bits_q; a.bits_k = q->mask_bits_k;
  a.tile_map = q->tile_map;
  a.nqt = (q->Lq + 31) / 32; a.nkt = (q->Lk + 31) / 32;
  a.has_drop = q->dropout_p > 0.f;
  a.inv_keep = a.has_drop ? 1.0f / (1.0f - q->dropout_p) : 1.0f;
  a.c1 = q->scale * a.inv_keep;
  {
    double thr = (double)q->dropout_p * 4294967296.0;
    a.drop_thr = thr >= 4294967295.0 ? 4294967295u : (uint32_t)thr;
  }
  a.seed_lo = q->seed_lo; a.seed_hi = q->seed_hi;
  a.lse = q->lse;
  a.dout = (const bf16_t*)q->dout; a.dsb = q->do_stride_b; a.dst = q->do_stride_t; a.dsh = q->do_stride_h;
  a.delta = q->delta;
  a.dq = (bf16_t*)q->dq; a.dk = (bf16_t*)q->dk; a.dv = (bf16_t*)q->dv;
  a.dqsb = q->dq_stride_b; a.dqst = q->dq_stride_t; a.dqsh = q->dq_stride_h;
  a.dksb = q->dk_stride_b; a.dkst = q->dk_stride_t; a.dksh = q->dk_stride_h;
  a.dvsb = q->dv_stride_b; a.dvst = q->dv_stride_t; a.dvsh = q->dv_stride_h;
  a.fuse_delta = 0;
  a.st16 = (ok16(q->o, q->o_stride_b, q->o_stride_t, q->o_stride_h) ? 1 : 0) |
           ((q->dq && ok16(q->dq, q->dq_stride_b, q->dq_stride_t, q->dq_stride_h)) ? 2 : 0) |
           ((q->dk && ok16(q->dk, q->dk_stride_b, q->dk_stride_t, q->dk_stride_h)) ? 4 : 0) |
           ((q->dv && ok16(q->dv, q->dv_stride_b, q->dv_stride_t, q->dv_stride_h)) ? 8 : 0);
  return DVLA_OK;
}

}  // namespace

namespace {

// ====================================================================================================
// backward, LDS-DMA ring versions.  Same ring as the forward kernel; every 32 x 64 tile is stored ONCE, row-major with
// 16-byte slot o of row r at o ^ fa_sw(r), fa_sw(r) = ((r>>1)&1)<<2 | (r>>2)&3.  That one image serves both fragment
// kinds conflict-free: row fragments by ds_read_b128 (the 8 rows of equal parity in a 16-lane read group get 8
// different slots) and transposed fragments by ds_read_b64_tr_b16 (keys k and k+2, which share the bank row, land in
// different 64-byte halves) -- so the pair-interleaved second copies of the register-staged kernels are gone.
// ====================================================================================================
__device__ __forceinline__ int fa_sw(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ bf16x8 fa_frag_rm(const char* tile, int row, int s, int g) {
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((((2 * s + g) ^ fa_sw(row))) << 4));
}
// transposed fragment: lane = column d = 32 db + (lane & 31) of the tile, k-slot j of lane group g = tile row
// 16 mm + 8 (j >> 2) + 4 g + (j & 3)  (= acc_row(8 mm + j, g), the row of P / dS register 8 mm + j)
__device__ __forceinline__ bf16x8 fa_frag_tr(const char* tile, int db, int mm, int lane) {
  const int gi = lane >> 4, c = lane & 15;
  const int d = 32 * db + 16 * (gi & 1) + 4 * (c & 3);
  union { fa_s16x4 h[2]; bf16x8 v; } u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 16 * mm + 8 * h + 4 * (gi >> 1) + (c >> 2);
    const int byte = row * 128 + ((((d >> 3) ^ fa_sw(row))) << 4) + (d & 7) * 2;
    u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fa_s16x4*)(tile + byte));
  }
  return u.v;
}

// Three waves per SIMD (round 6; rounds 2-5: four).  At 128 registers the kernel could neither request the eight S / dP fragments of a
// tile as a group (9 of its 12 multiplies waited for their own LDS read) nor take its dropout selects out of sixteen per-element
// basic blocks (the straight-line schedule spilled Q / dO fragments and reloaded them inside the tile loop, which drains the DMA
// ring).  At 168 it does both with 154 registers and no spill; trunk backward with dropout 266.6 -> 260.2 us, without 245.0 -> 236.1
// (same-box A/B, profiles/r06_attn_dq3_ab.txt).
__global__ __launch_bounds__(AT_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_bwd_dq_ring_kernel(AttnKArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, g = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = blockIdx.z, h = blockIdx.y;
  const int qt0 = blockIdx.x * 4;
  const int qt = qt0 + wave;
  const int q = qt * 32 + l31;
  const bool q_ok = q < p.Lq;
  const float scale_log2 = p.scale * LOG2E;
  const bool has_bits = p.tile_map != nullptr && p.bits_q != nullptr;

  uint8_t* flags = reinterpret_cast<uint8_t*>(smem + FA_RING);
  int32_t* kidx = reinterpret_cast<int32_t*>(smem + FA_RING + fa_pad16(p.nkt));
  uint32_t* bits = reinterpret_cast<uint32_t*>(smem + FA_RING + fa_pad16(p.nkt) + (p.key_index ? fa_pad16(p.Lk * 4) : 0));

  const bf16_t* qb = p.q + (int64_t)b * p.qsb + (int64_t)h * p.qsh;
  const bf16_t* kb = p.k + (int64_t)b * p.ksb + (int64_t)h * p.ksh;
  const bf16_t* vb = p.v + (int64_t)b * p.vsb + (int64_t)h * p.vsh;
  const bf16_t* dob = p.dout + (int64_t)b * p.dsb + (int64_t)h * p.dsh;

  // requested first, waited for after the first K / V tiles are on their way (see the forward kernel)
  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 a = load16(qb + (int64_t)q * p.qst + 16 * s + 8 * g, q_ok);
    const uint4 c = load16(dob + (int64_t)q * p.dst + 16 * s + 8 * g, q_ok);
    qf[s] = *reinterpret_cast<const bf16x8*>(&a);
    dof[s] = *reinterpret_cast<const bf16x8*>(&c);
  }
  const int rowid = (b * p.H + h) * p.Lq + q;
  float lse2 = q_ok ? p.lse[rowid] * LOG2E : INFINITY;
  float dlt;
  if (p.fuse_delta) {
    // delta = rowsum(dO * O) computed here from the dO fragment already in registers and the matching O fragment (the two
    // half-waves hold the two halves of the row), written out for the dK/dV kernel: the separate delta launch is gone
    const bf16_t* ob = p.o + (int64_t)b * p.osb + (int64_t)h * p.osh;
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4 ov = load16(ob + (int64_t)q * p.ost + 16 * s + 8 * g, q_ok);
      const uint4 dv = *reinterpret_cast<const uint4*>(&dof[s]);
      const uint32_t ow[4] = {ov.x, ov.y, ov.z, ov.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc += bf2f((bf16_t)(dw[i] & 0xffff)) * bf2f((bf16_t)(ow[i] & 0xffff));
        acc += bf2f((bf16_t)(dw[i] >> 16)) * bf2f((bf16_t)(ow[i] >> 16));
      }
    }
    acc += __shfl_xor(acc, 32, 64);
    dlt = q_ok ? acc : 0.f;
    if (q_ok && g == 0) p.delta[rowid] = acc;
  } else {
    dlt = q_ok ? p.delta[rowid] : 0.f;
  }
  dlt *= p.scale;                              // delta and dP enter dS already scaled: dS = P (c1 dP' - scale delta),
  const float c1 = p.c1;                        // c1 = scale / keep, dP' = the kept dP -- one fma per element, no multiply by scale
  for (int kt = t; kt < p.nkt; kt += AT_THREADS) {
    int f = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) f |= tile_flag(p, qt0 + w, kt) << (2 * w);
    flags[kt] = (uint8_t)f;
  }
  if (p.key_index)
    for (int i = t; i < p.Lk; i += AT_THREADS) kidx[i] = p.key_index[i];
  if (has_bits)
    for (int i = t; i < 128 * p.nkt; i += AT_THREADS) {
      const int ql = i / p.nkt, kt = i - ql * p.nkt;
      const int qq = qt0 * 32 + ql;
      bits[i] = qq < p.Lq ? p.bits_q[(int64_t)qq * p.nkt + kt] : 0u;
    }
  uint32_t rowkey = 0;
  if (p.has_drop) rowkey = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)rowid);
  f32x16 dqacc[2] = {zero16(), zero16()};
  __syncthreads();
  const TileWords tw = tile_words(flags, p.nkt, wave, lane);

  auto next_needed = [&](int kt) -> int {
    if (tw.n) return tw.next(kt);
    while (kt < p.nkt && __builtin_amdgcn_readfirstlane((int)flags[kt]) == 0) ++kt;
    return kt < p.nkt ? kt : -1;
  };
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int rl = 8 * wave + (lane >> 3);
  const int oct = (lane & 7) ^ fa_sw(rl);
  const uint32_t kst2 = (uint32_t)p.kst * 2u, vst2 = (uint32_t)p.vst * 2u;
  auto issue = [&](int kt, int slot) {
    int row = kt * 32 + rl;
    row = row < p.Lk ? row : p.Lk - 1;
    const uint32_t src = p.key_index ? (uint32_t)kidx[row] : (uint32_t)row;
    const uint32_t dst = smem_base + (uint32_t)(slot * FA_STAGE + wave * 1024);
    fa_glds16(kb, src * kst2 + oct * 16, __builtin_amdgcn_readfirstlane(dst));
    fa_glds16(vb, src * vst2 + oct * 16, __builtin_amdgcn_readfirstlane(dst + FA_TILE));
  };

  int kt = next_needed(0);
  int kti = kt, islot = 0, cslot = 0, inflight = 0;
#pragma unroll
  for (int d = 0; d < FA_NS - 1; ++d)
    if (kti >= 0) {
      issue(kti, islot);
      islot = (islot + 1) & (FA_NS - 1); ++inflight;
      kti = next_needed(kti + 1);
    }
#pragma unroll
  for (int s = 0; s < 4; ++s) { fa_settle(qf[s]); fa_settle(dof[s]); }
  fa_settle(lse2); fa_settle(dlt);
  while (kt >= 0) {
    if (inflight >= 3) fa_wait_vmcnt<4>();
    else if (inflight == 2) fa_wait_vmcnt<2>();
    else fa_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    --inflight;
    if (kti >= 0) {
      issue(kti, islot);
      islot = (islot + 1) & (FA_NS - 1); ++inflight;
      kti = next_needed(kti + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int flag = tw.n ? tw.flag(kt) : (__builtin_amdgcn_readfirstlane((int)flags[kt]) >> (2 * wave)) & 3;
    if (flag != 0) {
      const char* ks = smem + cslot * FA_STAGE;
      const char* vs = ks + FA_TILE;
      f32x16 sacc = zero16(), dpacc = zero16();
      {
        bf16x8 fk[4], fv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { fk[s] = fa_frag_rm(ks, l31, s, g); fv[s] = fa_frag_rm(vs, l31, s, g); }
        fa_group4(fk); fa_group4(fv);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[s], qf[s], sacc, 0, 0, 0);
          dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[s], dof[s], dpacc, 0, 0, 0);
        }
      }
      const int k0 = kt * 32;
      uint32_t vis = 0xffffffffu;
      if (flag == 2) vis = bits[(wave * 32 + l31) * p.nkt + kt];
      if (k0 + 32 > p.Lk) vis &= low_mask(p.Lk - k0);
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(fmaf(sacc[r], scale_log2, -lse2));
      if (__any(vis != 0xffffffffu)) {
        const uint32_t vg = vis >> (4 * g);
        vis_select16<0u>(vg, ds);
      }
      // dS = P (c1 dP' - dlt), dlt = scale * delta, c1 = scale / keep, dP' = the kept dP: scale and 1 / keep cost nothing here.
      // ONE branch around two complete element loops (no scalar test and branch per element; see the kernel's header).
      if (p.has_drop) {
        const uint32_t tk = drop_tilekey(rowkey, (uint32_t)kt);
        const uint32_t dx = drop_rot(tk, (uint32_t)g);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dp = (drop_elem(dx, tk, DVLA_DROP_C(r)) >= p.drop_thr) ? dpacc[r] : 0.f;
          ds[r] = ds[r] * fmaf(dp, c1, -dlt);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = ds[r] * fmaf(dpacc[r], c1, -dlt);
      }
      const bf16x8 f0 = pack_frag(ds), f1 = pack_frag(ds + 8);
      bf16x8 kt4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) kt4[i] = fa_frag_tr(ks, i >> 1, i & 1, lane);
      fa_group4(kt4);
      dqacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt4[0], f0, dqacc[0], 0, 0, 0);
      dqacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt4[2], f0, dqacc[1], 0, 0, 0);
      dqacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt4[1], f1, dqacc[0], 0, 0, 0);
      dqacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt4[3], f1, dqacc[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    cslot = (cslot + 1) & (FA_NS - 1);
    kt = next_needed(kt + 1);
  }
  if (q_ok) store_token(p.dq + (int64_t)b * p.dqsb + (int64_t)q * p.dqst + (int64_t)h * p.dqsh, dqacc, 1.0f, g, (p.st16 & 2) != 0);
}

// dynamic LDS of the dK/dV ring kernel: ring | tile flags [nqt] | visibility words [128 keys][nqt] | lse2 [32 nqt] | delta [32 nqt]
// | dropout tile keys [4 key tiles of the workgroup][32 nqt] (common.h: hash32(rowkey(query) + key tile * golden), hashed once per
// workgroup; an element then costs a rotate + a 24-bit multiply-add instead of a hash)
__host__ __device__ inline size_t fa_dkv_smem_bytes(int nqt, bool has_bits, bool has_drop) {
  return (size_t)FA_RING + fa_pad16(nqt) + (has_bits ? (size_t)128 * nqt * 4 : 0) + (size_t)2 * 32 * nqt * 4 +
         (has_drop ? (size_t)4 * 32 * nqt * 4 : 0);
}

__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkv_ring_kernel(AttnKArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, g = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = blockIdx.z, h = blockIdx.y;
  const int kt0 = blockIdx.x * 4;
  const int ktw = kt0 + wave;
  const int key = ktw * 32 + l31;
  const bool key_ok = key < p.Lk;
  const int key_row = (key_ok && p.key_index) ? p.key_index[key] : key;
  const float scale_log2 = p.scale * LOG2E;
  const float c1 = p.c1;                          // dS = P (c1 dP' - scale delta): scale and 1 / keep ride on one fma
  const bool has_bits = p.tile_map != nullptr && p.bits_k != nullptr;
  const bf16_t* qb = p.q + (int64_t)b * p.qsb + (int64_t)h * p.qsh;
  const bf16_t* kb = p.k + (int64_t)b * p.ksb + (int64_t)h * p.ksh;
  const bf16_t* vb = p.v + (int64_t)b * p.vsb + (int64_t)h * p.vsh;
  const bf16_t* dob = p.dout + (int64_t)b * p.dsb + (int64_t)h * p.dsh;
  const int bh_row0 = (b * p.H + h) * p.Lq;

  uint8_t* flags = reinterpret_cast<uint8_t*>(smem + FA_RING);
  uint32_t* bits = reinterpret_cast<uint32_t*>(smem + FA_RING + fa_pad16(p.nqt));
  float* lse2s = reinterpret_cast<float*>(smem + FA_RING + fa_pad16(p.nqt) + (has_bits ? (size_t)128 * p.nqt * 4 : 0));
  float* dlts = lse2s + 32 * p.nqt;
  uint32_t* tks = reinterpret_cast<uint32_t*>(dlts + 32 * p.nqt);      // [4][32 nqt], only with dropout

  // requested first, waited for after the first Q / dO tiles are on their way (see the forward kernel)
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 a = load16(kb + (int64_t)key_row * p.kst + 16 * s + 8 * g, key_ok);
    const uint4 c = load16(vb + (int64_t)key_row * p.vst + 16 * s + 8 * g, key_ok);
    kf[s] = *reinterpret_cast<const bf16x8*>(&a);
    vf[s] = *reinterpret_cast<const bf16x8*>(&c);
  }
  for (int qt = t; qt < p.nqt; qt += AT_THREADS) {
    int f = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) f |= tile_flag(p, qt, kt0 + w) << (2 * w);
    flags[qt] = (uint8_t)f;
  }
  if (has_bits)
    for (int i = t; i < 128 * p.nqt; i += AT_THREADS) {
      const int kl = i / p.nqt, qt = i - kl * p.nqt;
      const int kk = kt0 * 32 + kl;
      bits[i] = kk < p.Lk ? p.bits_k[(int64_t)kk * p.nqt + qt] : 0u;
    }
  for (int i = t; i < 32 * p.nqt; i += AT_THREADS) {
    lse2s[i] = i < p.Lq ? p.lse[bh_row0 + i] * LOG2E : INFINITY;   // q >= Lq: exp2(s - inf) = 0
    dlts[i] = i < p.Lq ? p.delta[bh_row0 + i] * p.scale : 0.f;    // scaled once here: dS = P (c1 dP' - scale delta)
    if (p.has_drop) {
      const uint32_t rk = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)(bh_row0 + i));
#pragma unroll
      for (int w = 0; w < 4; ++w) tks[w * 32 * p.nqt + i] = drop_tilekey(rk, (uint32_t)(kt0 + w));
    }
  }
  // this lane's key j = l31 of its tile: multiplier and operand rotation of the element generator (common.h)
  const uint32_t drop_cj = DVLA_DROP_C((l31 & 3) + 4 * (l31 >> 3));
  const uint32_t drop_half = (uint32_t)((l31 >> 2) & 1);
  const uint32_t* tkw = tks + wave * 32 * p.nqt;
  f32x16 dkacc[2] = {zero16(), zero16()}, dvacc[2] = {zero16(), zero16()};
  __syncthreads();
  const TileWords tw = tile_words(flags, p.nqt, wave, lane);

  auto next_needed = [&](int qt) -> int {
    if (tw.n) return tw.next(qt);
    while (qt < p.nqt && __builtin_amdgcn_readfirstlane((int)flags[qt]) == 0) ++qt;
    return qt < p.nqt ? qt : -1;
  };
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int rl = 8 * wave + (lane >> 3);
  const int oct = (lane & 7) ^ fa_sw(rl);
  const uint32_t qst2 = (uint32_t)p.qst * 2u, dst2 = (uint32_t)p.dst * 2u;
  auto issue = [&](int qt, int slot) {
    int row = qt * 32 + rl;
    row = row < p.Lq ? row : p.Lq - 1;
    const uint32_t dst = smem_base + (uint32_t)(slot * FA_STAGE + wave * 1024);
    fa_glds16(qb, (uint32_t)row * qst2 + oct * 16, __builtin_amdgcn_readfirstlane(dst));
    fa_glds16(dob, (uint32_t)row * dst2 + oct * 16, __builtin_amdgcn_readfirstlane(dst + FA_TILE));
  };

  int qt = next_needed(0);
  int qti = qt, islot = 0, cslot = 0, inflight = 0;
#pragma unroll
  for (int d = 0; d < FA_NS - 1; ++d)
    if (qti >= 0) {
      issue(qti, islot);
      islot = (islot + 1) & (FA_NS - 1); ++inflight;
      qti = next_needed(qti + 1);
    }
#pragma unroll
  for (int s = 0; s < 4; ++s) { fa_settle(kf[s]); fa_settle(vf[s]); }
  while (qt >= 0) {
    // dK / dV accumulators pinned to the accumulation registers at the loop top (round 6): hipcc kept them in AGPRs around the
    // multiplies but carried them through the loop's join in VGPRs -- 64 v_accvgpr_read at the top of every tile and 64
    // v_accvgpr_write in front of its multiplies, 128 of the tile's 352 vector instructions, copies that move nothing
    asm volatile("" : "+a"(dkacc[0]), "+a"(dkacc[1]), "+a"(dvacc[0]), "+a"(dvacc[1]));
    if (inflight >= 3) fa_wait_vmcnt<4>();
    else if (inflight == 2) fa_wait_vmcnt<2>();
    else fa_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    --inflight;
    if (qti >= 0) {
      issue(qti, islot);
      islot = (islot + 1) & (FA_NS - 1); ++inflight;
      qti = next_needed(qti + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int flag = tw.n ? tw.flag(qt) : (__builtin_amdgcn_readfirstlane((int)flags[qt]) >> (2 * wave)) & 3;
    if (flag != 0) {
      const char* qs = smem + cslot * FA_STAGE;
      const char* dos = qs + FA_TILE;
      f32x16 sacc = zero16(), dpacc = zero16();
      {   // all eight Q / dO fragments requested before the first multiply (see the forward kernel)
        bf16x8 fq[4], fd[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { fq[s] = fa_frag_rm(qs, l31, s, g); fd[s] = fa_frag_rm(dos, l31, s, g); }
        fa_group4(fq); fa_group4(fd);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[s], kf[s], sacc, 0, 0, 0);
          dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[s], vf[s], dpacc, 0, 0, 0);
        }
      }
      asm volatile("" : "+v"(sacc), "+v"(dpacc));   // scores and dP are consumed by the VALU: results in VGPRs, not read back from AGPRs
      const int q0 = qt * 32;
      float lse2[16], dlt[16];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float4 a = *reinterpret_cast<const float4*>(lse2s + q0 + 8 * rq + 4 * g);
        const float4 c = *reinterpret_cast<const float4*>(dlts + q0 + 8 * rq + 4 * g);
        lse2[4 * rq] = a.x; lse2[4 * rq + 1] = a.y; lse2[4 * rq + 2] = a.z; lse2[4 * rq + 3] = a.w;
        dlt[4 * rq] = c.x; dlt[4 * rq + 1] = c.y; dlt[4 * rq + 2] = c.z; dlt[4 * rq + 3] = c.w;
      }
      uint32_t vis = key_ok ? 0xffffffffu : 0u;  // bit i: query q0+i sees this lane's key
      if (flag == 2) vis = bits[(wave * 32 + l31) * p.nqt + qt];
      float pr[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[r] = fast_exp2(fmaf(sacc[r], scale_log2, -lse2[r]));
      if (__any(vis != 0xffffffffu)) {
        const uint32_t vg = vis >> (4 * g);
        vis_select16<0u>(vg, pr);
      }
      float dp[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = dpacc[r];
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[r] = pr[r];          // the UNDROPPED probabilities form dS; `pr` goes on to dV
      if (p.has_drop) {        // ONE branch around all sixteen elements (inside the element loop hipcc kept a scalar test and a
        // branch per element -- sixteen basic blocks, no select scheduled into the hazard slots of the one before it; round 6)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          // tile keys of queries q0 + 8 rq + 4 g .. + 3 (= acc_row(4 rq + e, g)) for this wave's key tile
          const uint4 a = *reinterpret_cast<const uint4*>(tkw + q0 + 8 * rq + 4 * g);
          const uint32_t rk[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const bool keep = drop_elem(drop_rot(rk[e], drop_half), rk[e], drop_cj) >= p.drop_thr;
            dp[r] = keep ? dp[r] : 0.f;            // 1 / keep: in c1 for dS, on the stored dV for P
            pr[r] = keep ? pr[r] : 0.f;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) ds[r] = ds[r] * fmaf(dp[r], c1, -dlt[r]);   // the table holds scale * delta; c1 = scale / keep
      const bf16x8 pf0 = pack_frag(pr), pf1 = pack_frag(pr + 8);
      const bf16x8 sf0 = pack_frag(ds), sf1 = pack_frag(ds + 8);
      {   // the eight transposed fragments first, then eight multiplies that walk the four accumulators round-robin
        bf16x8 td[4], tq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { td[i] = fa_frag_tr(dos, i >> 1, i & 1, lane); tq[i] = fa_frag_tr(qs, i >> 1, i & 1, lane); }
        fa_group4(td); fa_group4(tq);
        dvacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[0], pf0, dvacc[0], 0, 0, 0);
        dkacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[0], sf0, dkacc[0], 0, 0, 0);
        dvacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[2], pf0, dvacc[1], 0, 0, 0);
        dkacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[2], sf0, dkacc[1], 0, 0, 0);
        dvacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[1], pf1, dvacc[0], 0, 0, 0);
        dkacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[1], sf1, dkacc[0], 0, 0, 0);
        dvacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[3], pf1, dvacc[1], 0, 0, 0);
        dkacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[3], sf1, dkacc[1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    cslot = (cslot + 1) & (FA_NS - 1);
    qt = next_needed(qt + 1);
  }
  if (key_ok) {
    store_token(p.dk + (int64_t)b * p.dksb + (int64_t)key_row * p.dkst + (int64_t)h * p.dksh, dkacc, 1.0f, g, (p.st16 & 4) != 0);
    store_token(p.dv + (int64_t)b * p.dvsb + (int64_t)key_row * p.dvst + (int64_t)h * p.dvsh, dvacc, p.inv_keep, g, (p.st16 & 8) != 0);
  }
}


// ====================================================================================================
// backward for SHORT dense sequences, one pass per (batch, head)  (round 4)
//
//   The dream-head decoders run B = 448 x 16 heads of L = 205 / 265 tokens (dreamvla_model.py:806-904): 7-9 key tiles per
//   sequence.  Under the two ring kernels every 128-token workgroup pays its prologue (fragment loads, tables, ring fill) and its
//   store tail for 7 tile iterations of work, K / V / Q / dO are each read twice from HBM (2.0 GB at B = 448 where 1.5 GB is
//   needed) and the matrix pipe is 10-15 % busy (profiles/r03_attn_perf.jsonl: 829 us for L = 205).  Here a PERSISTENT workgroup
//   of 8 waves takes whole (batch, head) items:
//     * Q, dO (region A) and K, V (region B) of the item arrive by LDS-DMA ONCE, as 32-row tiles in the swizzled image of the
//       ring kernels (one image serves row fragments and hardware-transposed fragments);
//     * phase X (the dQ kernel's tile loop): wave j owns query tile j -- Q / dO fragments from region A, delta = rowsum(dO * O)
//       on its way (written with log2-domain lse to an LDS table), K / V tiles streamed from region B, dQ^T in registers;
//     * phase Y (the dK/dV kernel's tile loop): wave j owns key tile j -- K / V fragments from region B, Q / dO tiles streamed
//       from region A, dK^T / dV^T in registers;
//     * no barrier inside a phase (everything a wave reads is resident and read-only): the two waves of a SIMD overlap each
//       other's MFMA and softmax VALU freely;
//     * region B is re-filled with the NEXT item's K / V as soon as every wave has taken its K / V fragments (phase Y streams
//       only region A); region A tile by tile behind phase Y's walk (one barrier per query tile in that phase keeps the
//       waves in step): the next item's loads run under this item's second phase.
//   Same arithmetic, same tile order and the same two rounding points as the ring kernels (oracle/torch_ref.py::attention_bf16).
//   (The same idea for the MASKED trunk -- K / V of the 378 compacted keys resident, the 21 query tiles handed out heaviest-first
//   from a ticket counter, no barrier inside an item -- was built, passed every parity case and was NOT faster than the ring kernels:
//   forward 81.2 vs 77.1 us, backward 284 vs 271 (profiles/r04_attn_resident_experiment.jsonl); it is not in the tree.  With two
//   items per CU and 6 visible key tiles per query tile these kernels sit at the per-score instruction cost, not at barriers.)
//   LDS: 4 operands x nt tiles x 4 KiB + 2 x 32 nt floats = 16 640 nt bytes: 116 KiB at L = 205 (nt = 7), 150 KiB at L = 265.
//   Sequences of more than 8 tiles give wave j the tiles j and j + 8.
// ====================================================================================================
constexpr int SB_WAVES = 8, SB_THREADS = 64 * SB_WAVES, SB_MAX_TILES = 9;
__host__ __device__ inline size_t sb_smem_bytes(int nt) { return (size_t)nt * (4 * FA_TILE + 2 * 32 * 4); }

__global__ __launch_bounds__(SB_THREADS) void attn_bwd_short_kernel(AttnKArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, g = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int L = p.Lq;
  const int nt = p.nqt;                       // Lq == Lk: query tiles == key tiles
  const int rounds = (nt + SB_WAVES - 1) / SB_WAVES;
  const float scale_log2 = p.scale * LOG2E;

  char* const regA = smem;                                   // Q tiles [nt] | dO tiles [nt]
  char* const regB = smem + (size_t)2 * nt * FA_TILE;        // K tiles [nt] | V tiles [nt]
  float* const lse2s = reinterpret_cast<float*>(smem + (size_t)4 * nt * FA_TILE);
  float* const dlts = lse2s + 32 * nt;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t offA = 0u, offB = (uint32_t)(2 * nt * FA_TILE);

  // LDS-DMA: wave w copies rows 8 (w & 3) .. + 7 of the tiles t = (w >> 2), (w >> 2) + 2, ...: lane -> (row, LDS slot lane % 8)
  const int rl = 8 * (wave & 3) + (lane >> 3);
  const int oct = (lane & 7) ^ fa_sw(rl);
  auto issue_op = [&](const bf16_t* base, uint32_t stride_bytes, uint32_t region) {
    for (int tt = wave >> 2; tt < nt; tt += 2) {
      int row = tt * 32 + rl;
      row = row < L ? row : L - 1;
      const uint32_t dst = smem_base + region + (uint32_t)(tt * FA_TILE + (wave & 3) * 1024);
      fa_glds16(base, (uint32_t)row * stride_bytes + (uint32_t)(oct * 16), __builtin_amdgcn_readfirstlane(dst));
    }
  };
  const uint32_t qst2 = (uint32_t)p.qst * 2u, kst2 = (uint32_t)p.kst * 2u, vst2 = (uint32_t)p.vst * 2u, dst2 = (uint32_t)p.dst * 2u;
  auto issue_A = [&](int item) {
    const int b = item / p.H, h = item - b * p.H;
    issue_op(p.q + (int64_t)b * p.qsb + (int64_t)h * p.qsh, qst2, offA);
    issue_op(p.dout + (int64_t)b * p.dsb + (int64_t)h * p.dsh, dst2, offA + (uint32_t)(nt * FA_TILE));
  };
  // one tile of region A (Q and dO) for `item`: issued by the four waves whose tile parity it is
  auto issue_A_tile = [&](int item, int tt) {
    const int b = item / p.H, h = item - b * p.H;
    int row = tt * 32 + rl;
    row = row < L ? row : L - 1;
    const uint32_t dst = smem_base + offA + (uint32_t)(tt * FA_TILE + (wave & 3) * 1024);
    fa_glds16(p.q + (int64_t)b * p.qsb + (int64_t)h * p.qsh, (uint32_t)row * qst2 + (uint32_t)(oct * 16), __builtin_amdgcn_readfirstlane(dst));
    fa_glds16(p.dout + (int64_t)b * p.dsb + (int64_t)h * p.dsh, (uint32_t)row * dst2 + (uint32_t)(oct * 16),
              __builtin_amdgcn_readfirstlane(dst + (uint32_t)(nt * FA_TILE)));
  };
  auto issue_B = [&](int item) {
    const int b = item / p.H, h = item - b * p.H;
    issue_op(p.k + (int64_t)b * p.ksb + (int64_t)h * p.ksh, kst2, offB);
    issue_op(p.v + (int64_t)b * p.vsb + (int64_t)h * p.vsh, vst2, offB + (uint32_t)(nt * FA_TILE));
  };

  const int n_items = p.B * p.H;
  int item = blockIdx.x;
  if (item >= n_items) return;
  issue_A(item);
  issue_B(item);

  // rows of O (for delta) and lse of this wave's first query tile: requested one item AHEAD, at the start of the previous item's
  // phase Y -- at the top of the item loop the compiler waits for the previous stores before it may reuse their registers
  // (vmcnt(0): the DMA included), so a request placed there would start its round trip only after the DMA has landed
  uint4 ofr[4];
  float lse_in = 0.f;
  auto prefetch_o = [&](int it_) {
    const int b_ = it_ / p.H, h_ = it_ - b_ * p.H;
    const bf16_t* ob_ = p.o + (int64_t)b_ * p.osb + (int64_t)h_ * p.osh;
    const int q = wave * 32 + l31;
    const bool ok = wave < nt && q < L;
#pragma unroll
    for (int s = 0; s < 4; ++s) ofr[s] = load16(ob_ + (int64_t)q * p.ost + 16 * s + 8 * g, ok);
    lse_in = ok ? p.lse[(b_ * p.H + h_) * L + q] : 0.f;
  };
  prefetch_o(item);

  for (; item < n_items; item += gridDim.x) {
    const int b = item / p.H, h = item - b * p.H;
    const int next = item + (int)gridDim.x;
    const int row0 = (b * p.H + h) * L;
    const bf16_t* ob = p.o + (int64_t)b * p.osb + (int64_t)h * p.osh;

    fa_wait_vmcnt<0>();            // this wave's share of regions A and B has landed (and its O / lse rows)
    // (tell the compiler so on EVERY path: a wave without a tile never uses them, and registers with a load the compiler
    // believes pending cost a vmcnt(0) -- which would also drain the DMA just issued -- where the next prefetch overwrites them)
#pragma unroll
    for (int s = 0; s < 4; ++s) { fa_settle(ofr[s].x); fa_settle(ofr[s].y); fa_settle(ofr[s].z); fa_settle(ofr[s].w); }
    fa_settle(lse_in);
    __syncthreads();               // ... and everybody else's

    // ---------------- phase X: dQ of query tile j, delta and lse (log2 domain) of its rows into the LDS table ----------------
    for (int r = 0; r < rounds; ++r) {
      const int j = wave + SB_WAVES * r;
      if (j >= nt) break;
      const int q = j * 32 + l31;
      const bool q_ok = q < L;
      if (r > 0) {                 // a second tile of this wave (sequences of more than 8 tiles): its O / lse rows now
#pragma unroll
        for (int s = 0; s < 4; ++s) ofr[s] = load16(ob + (int64_t)q * p.ost + 16 * s + 8 * g, q_ok);
        lse_in = q_ok ? p.lse[row0 + q] : 0.f;
      }
      const char* qs = regA + (size_t)j * FA_TILE;
      const char* dos = regA + (size_t)(nt + j) * FA_TILE;
      bf16x8 qf[4], dof[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { qf[s] = fa_frag_rm(qs, l31, s, g); dof[s] = fa_frag_rm(dos, l31, s, g); }
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 dv = *reinterpret_cast<const uint4*>(&dof[s]);
        const uint32_t ow[4] = {ofr[s].x, ofr[s].y, ofr[s].z, ofr[s].w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc += bf2f((bf16_t)(dw[i] & 0xffff)) * bf2f((bf16_t)(ow[i] & 0xffff));
          acc += bf2f((bf16_t)(dw[i] >> 16)) * bf2f((bf16_t)(ow[i] >> 16));
        }
      }
      acc += __shfl_xor(acc, 32, 64);
      const float lse2 = q_ok ? lse_in * LOG2E : INFINITY;       // q >= L: exp2(s - inf) = 0
      const float dlt = q_ok ? acc * p.scale : 0.f;              // scale * delta (here and in the table): dS = P fma(dP, scale, -dlt)
      if (g == 0) { lse2s[j * 32 + l31] = lse2; dlts[j * 32 + l31] = dlt; }
      f32x16 dqacc[2] = {zero16(), zero16()};
      for (int kt = 0; kt < nt; ++kt) {
        const char* ks = regB + (size_t)kt * FA_TILE;
        const char* vs = regB + (size_t)(nt + kt) * FA_TILE;
        f32x16 sacc = zero16(), dpacc = zero16();
        {   // all eight K / V fragments requested before the first multiply (attn_fwd_ring_kernel: fa_group4)
          bf16x8 fk[4], fv[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) { fk[s] = fa_frag_rm(ks, l31, s, g); fv[s] = fa_frag_rm(vs, l31, s, g); }
          fa_group4(fk); fa_group4(fv);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[s], qf[s], sacc, 0, 0, 0);
            dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv[s], dof[s], dpacc, 0, 0, 0);
          }
        }
        const int k0 = kt * 32;
        float ds[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) ds[rr] = fast_exp2(fmaf(sacc[rr], scale_log2, -lse2));
        if (k0 + 32 > L) {           // the ragged last key tile (wave-uniform)
          asm volatile("");          // (keeps this a BRANCH: hipcc if-converted it into sixteen selects on every tile)
          const uint32_t vg = low_mask(L - k0) >> (4 * g);
          vis_select16<0u>(vg, ds);
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) ds[rr] = ds[rr] * fmaf(dpacc[rr], p.scale, -dlt);
        const bf16x8 f0 = pack_frag(ds), f1 = pack_frag(ds + 8);
        bf16x8 kt4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kt4[i] = fa_frag_tr(ks, i >> 1, i & 1, lane);
        fa_group4(kt4);
        dqacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt4[0], f0, dqacc[0], 0, 0, 0);
        dqacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt4[2], f0, dqacc[1], 0, 0, 0);
        dqacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt4[1], f1, dqacc[0], 0, 0, 0);
        dqacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt4[3], f1, dqacc[1], 0, 0, 0);
      }
      if (q_ok) store_token(p.dq + (int64_t)b * p.dqsb + (int64_t)q * p.dqst + (int64_t)h * p.dqsh, dqacc, 1.0f, g, (p.st16 & 2) != 0);
    }
    __syncthreads();               // the lse / delta table is complete

    // ---------------- phase Y: dK, dV of key tile j ----------------
    for (int r = 0; r < rounds; ++r) {
      const int j = wave + SB_WAVES * r;
      const bool active = j < nt;
      const int key = j * 32 + l31;
      const bool key_ok = active && key < L;
      bf16x8 kf[4], vf[4];
      if (active) {
        const char* ks = regB + (size_t)j * FA_TILE;
        const char* vs = regB + (size_t)(nt + j) * FA_TILE;
#pragma unroll
        for (int s = 0; s < 4; ++s) { kf[s] = fa_frag_rm(ks, l31, s, g); vf[s] = fa_frag_rm(vs, l31, s, g); }
      }
      if (r == rounds - 1) {
        __syncthreads();           // every wave holds the K / V fragments of its last tile: region B is free
        // (the prefetch first: the compiler guards the registers it overwrites against the dQ stores that last read them with
        // a vmcnt wait, which must not sit behind the DMA issue -- it would drain it)
        if (next < n_items) { prefetch_o(next); issue_B(next); }
      }
      // In the LAST round every wave (with or without a tile) walks the query tiles in step: after the barrier that ends
      // tile qt nobody reads it again, and the next item's Q / dO tile qt is requested into its place -- region A is re-filled
      // under this phase instead of after it (the load used to be exposed at the top of every item: ~5 k of ~50 k cycles).
      const bool last_round = r == rounds - 1;
      if (!active && !last_round) continue;
      f32x16 dkacc[2] = {zero16(), zero16()}, dvacc[2] = {zero16(), zero16()};
      for (int qt = 0; qt < nt; ++qt) {
        const char* qs = regA + (size_t)qt * FA_TILE;
        const char* dos = regA + (size_t)(nt + qt) * FA_TILE;
        if (qt > 0 && last_round) {
          __syncthreads();         // every wave is done with tile qt - 1
          if (next < n_items && (wave >> 2) == ((qt - 1) & 1)) issue_A_tile(next, qt - 1);
        }
        if (!active) continue;
        f32x16 sacc = zero16(), dpacc = zero16();
        {
          bf16x8 fq[4], fd[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) { fq[s] = fa_frag_rm(qs, l31, s, g); fd[s] = fa_frag_rm(dos, l31, s, g); }
          fa_group4(fq); fa_group4(fd);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[s], kf[s], sacc, 0, 0, 0);
            dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[s], vf[s], dpacc, 0, 0, 0);
          }
        }
        const int q0 = qt * 32;
        float pr[16], ds[16];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4 a = *reinterpret_cast<const float4*>(lse2s + q0 + 8 * rq + 4 * g);
          const float4 c = *reinterpret_cast<const float4*>(dlts + q0 + 8 * rq + 4 * g);
          const float l4[4] = {a.x, a.y, a.z, a.w}, d4[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int rr = 4 * rq + e;
            float pv = fast_exp2(fmaf(sacc[rr], scale_log2, -l4[e]));
            pv = key_ok ? pv : 0.f;
            pr[rr] = pv;
            ds[rr] = pv * fmaf(dpacc[rr], p.scale, -d4[e]);
          }
        }
        const bf16x8 pf0 = pack_frag(pr), pf1 = pack_frag(pr + 8);
        const bf16x8 sf0 = pack_frag(ds), sf1 = pack_frag(ds + 8);
        {
          bf16x8 td[4], tq[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { td[i] = fa_frag_tr(dos, i >> 1, i & 1, lane); tq[i] = fa_frag_tr(qs, i >> 1, i & 1, lane); }
          fa_group4(td); fa_group4(tq);
          dvacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[0], pf0, dvacc[0], 0, 0, 0);
          dkacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[0], sf0, dkacc[0], 0, 0, 0);
          dvacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[2], pf0, dvacc[1], 0, 0, 0);
          dkacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[2], sf0, dkacc[1], 0, 0, 0);
          dvacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[1], pf1, dvacc[0], 0, 0, 0);
          dkacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[1], sf1, dkacc[0], 0, 0, 0);
          dvacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[3], pf1, dvacc[1], 0, 0, 0);
          dkacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[3], sf1, dkacc[1], 0, 0, 0);
        }
      }
      if (key_ok) {
        store_token(p.dk + (int64_t)b * p.dksb + (int64_t)key * p.dkst + (int64_t)h * p.dksh, dkacc, 1.0f, g, (p.st16 & 4) != 0);
        store_token(p.dv + (int64_t)b * p.dvsb + (int64_t)key * p.dvst + (int64_t)h * p.dvsh, dvacc, 1.0f, g, (p.st16 & 8) != 0);
      }
    }
    __syncthreads();               // every wave is done with the last query tile and with the table
    if (next < n_items && (wave >> 2) == ((nt - 1) & 1)) issue_A_tile(next, nt - 1);
  }
}


// ====================================================================================================
// forward for SHORT dense sequences (same shapes as attn_bwd_short_kernel: the ViT's L = 197, the decoders' 205 / 265): a persistent
// workgroup of 8 waves takes whole (batch, head) items -- K and V arrive ONCE by LDS-DMA (the ring kernel reads them once per
// 128-query block: twice at L = 205, three times at 265), wave j owns query tile j, no barrier inside the tile loop.  K / V are
// single-buffered (8 KiB per tile: 56-72 KiB) so that TWO workgroups share a CU and fill each other's load phases.
// Same arithmetic and tile order as attn_fwd_ring_kernel (integer running maximum in the log2 domain).
// ====================================================================================================
__host__ __device__ inline size_t sf_smem_bytes(int nt) { return (size_t)nt * 2 * FA_TILE; }

__global__ __launch_bounds__(SB_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd_short_kernel(AttnKArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, g = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int L = p.Lq;
  const int nt = p.nqt;
  const int rounds = (nt + SB_WAVES - 1) / SB_WAVES;
  const float scale_log2 = p.scale * LOG2E;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int rl = 8 * (wave & 3) + (lane >> 3);
  const int oct = (lane & 7) ^ fa_sw(rl);
  const uint32_t kst2 = (uint32_t)p.kst * 2u, vst2 = (uint32_t)p.vst * 2u;
  auto issue_op = [&](const bf16_t* base, uint32_t stride_bytes, uint32_t region) {
    for (int tt = wave >> 2; tt < nt; tt += 2) {
      int row = tt * 32 + rl;
      row = row < L ? row : L - 1;
      const uint32_t dst = smem_base + region + (uint32_t)(tt * FA_TILE + (wave & 3) * 1024);
      fa_glds16(base, (uint32_t)row * stride_bytes + (uint32_t)(oct * 16), __builtin_amdgcn_readfirstlane(dst));
    }
  };
  const int n_items = p.B * p.H;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / p.H, h = item - b * p.H;
    const bf16_t* qb = p.q + (int64_t)b * p.qsb + (int64_t)h * p.qsh;
    issue_op(p.k + (int64_t)b * p.ksb + (int64_t)h * p.ksh, kst2, 0u);
    issue_op(p.v + (int64_t)b * p.vsb + (int64_t)h * p.vsh, vst2, (uint32_t)(nt * FA_TILE));
    for (int r = 0; r < rounds; ++r) {
      const int j = wave + SB_WAVES * r;
      const bool active = j < nt;
      const int q = j * 32 + l31;
      const bool q_ok = active && q < L;
      bf16x8 qf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 u = load16(qb + (int64_t)q * p.qst + 16 * s + 8 * g, q_ok);
        qf[s] = *reinterpret_cast<const bf16x8*>(&u);
      }
      if (r == 0) {
        fa_wait_vmcnt<0>();        // this wave's share of K / V (and its Q rows)
        __syncthreads();
      }
      if (!active) continue;
      float m_run = -INFINITY, l_run = 0.f;
      f32x16 oacc[2] = {zero16(), zero16()};
      for (int kt = 0; kt < nt; ++kt) {
        const char* ks = smem + (size_t)kt * FA_TILE;
        const char* vs = smem + (size_t)(nt + kt) * FA_TILE;
        f32x16 sacc = zero16();
        {
          bf16x8 fk[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) fk[s] = fa_frag_rm(ks, l31, s, g);
          fa_group4(fk);
#pragma unroll
          for (int s = 0; s < 4; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[s], qf[s], sacc, 0, 0, 0);
        }
        const int k0 = kt * 32;
        float sv[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) sv[rr] = sacc[rr];
        if (k0 + 32 > L) {
          asm volatile("");          // (keeps this a BRANCH: hipcc if-converted it into sixteen selects on every tile)
          const uint32_t vg = low_mask(L - k0) >> (4 * g);
          vis_select16<NEG_INF_BITS>(vg, sv);
        }
        float mt = sv[0];
#pragma unroll
        for (int rr = 1; rr < 16; ++rr) mt = fmaxf(mt, sv[rr]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, ceilf(mt * scale_log2));     // integer running maximum: see attn_fwd_ring_kernel
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2(m_run - m_safe);
        float rs = 0.f;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) { sv[rr] = fast_exp2(fmaf(sv[rr], scale_log2, -m_safe)); rs += sv[rr]; }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        if (__any(alpha != 1.0f)) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) { oacc[0][rr] *= alpha; oacc[1][rr] *= alpha; }
        }
        bf16x8 vt4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vt4[i] = fa_frag_tr(vs, i >> 1, i & 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 pf0 = pack_frag(sv), pf1 = pack_frag(sv + 8);
        fa_group4(vt4);
        oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt4[0], pf0, oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt4[2], pf0, oacc[1], 0, 0, 0);
        oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt4[1], pf1, oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt4[3], pf1, oacc[1], 0, 0, 0);
      }
      if (q_ok) {
        const float inv_l = l_run > 0.f ? 1.0f / l_run : 0.f;
        store_token(p.o + (int64_t)b * p.osb + (int64_t)q * p.ost + (int64_t)h * p.osh, oacc, inv_l, g, (p.st16 & 1) != 0);
        if (p.lse && g == 0) p.lse[(b * p.H + h) * L + q] = l_run > 0.f ? (m_run + log2f(l_run)) * LN2 : INFINITY;
      }
    }
    __syncthreads();               // every wave is done with K / V before the next item's DMA lands
  }
}


}  // namespace

// The ring kernels address K / V / Q / dO rows as a 32-bit byte offset from the (batch, head) base: every row they can touch
// must lie below 4 GiB.  Rows named through key_index are bounded by the contract of include/dvla.h (entries < 2^18).
static bool attn_span32(int64_t rows, int64_t stride_elems, bool gathered) {
  const int64_t r = gathered ? (rows > (1 << 18) ? rows : (1 << 18)) : rows;
  return stride_elems >= 0 && r * stride_elems * 2 + 128 < (1LL << 32);
}

// DVLA_ATTN_STAGED=1 selects the register-staged kernels (A/B measurements, tests of the fallback path)
static bool attn_force_staged() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("DVLA_ATTN_STAGED"); v = (e && e[0] == '1') ? 1 : 0; }
  return v == 1;
}

// DVLA_ATTN_SHORT=0 keeps short dense sequences on the two ring kernels (A/B measurements; read per launch)
static bool attn_short_enabled() { const char* e = getenv("DVLA_ATTN_SHORT"); return !(e && e[0] == '0'); }
static bool attn_short_fwd_enabled() { const char* e = getenv("DVLA_ATTN_SHORT_FWD"); return !(e && e[0] == '0'); }
static int attn_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return cus;
}

// DVLA_ATTN_DBG=<bits>: ablation builds of the forward ring kernel (timing only, results garbage by design) -- compiled only
// with -DDVLA_ATTN_ABLATION (DVLA_ABLATIONS=1 python -c 'import __graft_entry__ as g; g.build(force=True)'); the product library
// holds the real kernel alone and never looks at the variable.
#ifdef DVLA_ATTN_ABLATION
static int attn_dbg() { const char* e = getenv("DVLA_ATTN_DBG"); return e ? atoi(e) : 0; }   // (re-read per launch: the sweep changes it)
#endif

// grid of the forward ring kernel: x = 128-query blocks, y = workgroups that share the (batch, head) items of a query block.
// (Forward only: the same loop in the dQ and dK/dV ring kernels made them 40-55 % SLOWER -- 16 bytes of scratch in dQ at 128 VGPRs,
// 186 -> 196 VGPRs in dK/dV, even at one item per workgroup -- and was taken out again; tests/gpu_attn_perf.py.)
// With mask tables (loaded once per workgroup) a workgroup takes up to four items as long as at least three workgroups per CU (the
// resident count) remain; without tables one item per workgroup as before.  Measured at the trunk's shape (B 32, H 16, L 651,
// tests/gpu_attn_items.py): 1 item 91.3 us, 2 items 84.7, 4 items 80.4, 8 items (384 workgroups: the chip half empty) 129.
// DVLA_ATTN_ITEMS=<n> forces n items per workgroup (measurement).
static dim3 ring_items_grid(int nblk, const AttnKArgs& a) {
  static int forced = -1, cus = 0;
  if (forced < 0) {
    const char* e = getenv("DVLA_ATTN_ITEMS"); forced = e ? atoi(e) : 0;
    int dev = 0; hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int nqb = nblk, items = a.B * a.H;
  int y = items;
  const bool tables = a.tile_map != nullptr || a.key_index != nullptr;
  if (forced > 0) y = (items + forced - 1) / forced;
  else if (tables) { const int by4 = (items + 3) / 4, fill = (3 * cus + nqb - 1) / nqb; y = by4 > fill ? by4 : fill; if (y > items) y = items; }
  if (y < 1) y = 1;
  if (y > 65535) y = 65535;
  return dim3((unsigned)nqb, (unsigned)y, 1u);
}
static dim3 fwd_ring_grid(const AttnKArgs& a) { return ring_items_grid((a.nqt + 3) / 4, a); }

extern "C" int dvla_attn_fwd(const dvla_attn_params* q, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  AttnKArgs a;
  int rc = fill_args(q, a);
  if (rc != DVLA_OK) return rc;
  dim3 grid((unsigned)((a.nqt + 3) / 4), (unsigned)a.H, (unsigned)a.B), block(AT_THREADS);
  const dim3 ring_grid = fwd_ring_grid(a);
  const size_t smem = fa_smem_bytes(a.nkt, a.Lk, a.key_index != nullptr, a.tile_map != nullptr && a.bits_q != nullptr);
  const bool span_kv = attn_span32(a.Lk, a.kst, a.key_index != nullptr) && attn_span32(a.Lk, a.vst, a.key_index != nullptr);
  if (attn_short_fwd_enabled() && !a.tile_map && !a.key_index && !a.has_drop && a.Lq == a.Lk && a.nqt > 2 && a.nqt <= SB_MAX_TILES &&
      span_kv && !attn_force_staged()) {
    // short dense sequences (ViT L = 197, decoders 205 / 265): whole (batch, head) items per persistent workgroup, K / V read once
    const size_t sm = sf_smem_bytes(a.nqt);
    static uint64_t attr_done = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!(attr_done & (1ull << dev))) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_short_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attr_done |= 1ull << dev;
    }
    const int64_t items = (int64_t)a.B * a.H;
    int64_t wgs = (int64_t)attn_num_cus() * 2;
    if (wgs > items) wgs = items;
    hipLaunchKernelGGL(attn_fwd_short_kernel, dim3((unsigned)wgs), dim3(SB_THREADS), sm, stream, a);
    return dvla_check_launch();
  }
  if (smem <= 64 * 1024 && span_kv && !attn_force_staged())
#ifndef DVLA_ATTN_ABLATION
    hipLaunchKernelGGL(attn_fwd_ring_kernel<0>, ring_grid, block, smem, stream, a);
#else
    switch (attn_dbg()) {
      case 1: hipLaunchKernelGGL(attn_fwd_ring_kernel<1>, ring_grid, block, smem, stream, a); break;
      case 2: hipLaunchKernelGGL(attn_fwd_ring_kernel<2>, ring_grid, block, smem, stream, a); break;
      case 4: hipLaunchKernelGGL(attn_fwd_ring_kernel<4>, ring_grid, block, smem, stream, a); break;
      case 8: hipLaunchKernelGGL(attn_fwd_ring_kernel<8>, ring_grid, block, smem, stream, a); break;
      case 16: hipLaunchKernelGGL(attn_fwd_ring_kernel<16>, ring_grid, block, smem, stream, a); break;
      case 14: hipLaunchKernelGGL(attn_fwd_ring_kernel<14>, ring_grid, block, smem, stream, a); break;
      case 15: hipLaunchKernelGGL(attn_fwd_ring_kernel<15>, ring_grid, block, smem, stream, a); break;
      case 31: hipLaunchKernelGGL(attn_fwd_ring_kernel<31>, ring_grid, block, smem, stream, a); break;
      case 63: hipLaunchKernelGGL(attn_fwd_ring_kernel<63>, ring_grid, block, smem, stream, a); break;
      case 64: hipLaunchKernelGGL(attn_fwd_ring_kernel<64>, ring_grid, block, smem, stream, a); break;
      default: hipLaunchKernelGGL(attn_fwd_ring_kernel<0>, ring_grid, block, smem, stream, a); break;
    }
#endif
  else   // mask tables / key list too large for LDS
    hipLaunchKernelGGL(attn_fwd_kernel, grid, block, 0, stream, a);
  return dvla_check_launch();
}

extern "C" int dvla_attn_bwd(const dvla_attn_params* q, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  AttnKArgs a;
  int rc = fill_args(q, a);
  if (rc != DVLA_OK) return rc;
  if (!q->dout || !q->lse || !q->delta || !q->dq || !q->dk || !q->dv) return DVLA_ERR_ARG;
  if (q->tile_map && !q->mask_bits_k) return DVLA_ERR_ARG;
  if (!ok16(q->dout, q->do_stride_b, q->do_stride_t, q->do_stride_h)) return DVLA_ERR_UNSUPPORTED;
  auto ok8 = [](const void* ptr, int64_t s0, int64_t s1, int64_t s2) {
    return (reinterpret_cast<uintptr_t>(ptr) % 8 == 0) && (s0 % 4 == 0) && (s1 % 4 == 0) && (s2 % 4 == 0);
  };
  if (!ok8(q->dq, q->dq_stride_b, q->dq_stride_t, q->dq_stride_h) || !ok8(q->dk, q->dk_stride_b, q->dk_stride_t, q->dk_stride_h) ||
      !ok8(q->dv, q->dv_stride_b, q->dv_stride_t, q->dv_stride_h) || !ok16(q->o, q->o_stride_b, q->o_stride_t, q->o_stride_h))
    return DVLA_ERR_UNSUPPORTED;
  const int64_t nrows = (int64_t)a.B * a.H * a.Lq;
  dim3 block(AT_THREADS);
  const bool span_all = attn_span32(a.Lk, a.kst, false) && attn_span32(a.Lk, a.vst, false) && attn_span32(a.Lq, a.qst, false) &&
                        attn_span32(a.Lq, a.dst, false);
  if (attn_short_enabled() && !a.tile_map && !a.key_index && !a.has_drop && a.Lq == a.Lk && a.nqt <= SB_MAX_TILES && span_all &&
      !attn_force_staged()) {
    // short dense sequences (the dream-head decoders, L = 205 / 265): one pass per (batch, head), operands read once
    const size_t smem = sb_smem_bytes(a.nqt);
    static uint64_t attr_done = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!(attr_done & (1ull << dev))) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_short_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done |= 1ull << dev;
    }
    int per_cu = (int)((size_t)(160 * 1024) / smem);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2048 / SB_THREADS) per_cu = 2048 / SB_THREADS;
    const int64_t items = (int64_t)a.B * a.H;
    int64_t wgs = (int64_t)attn_num_cus() * per_cu;
    if (wgs > items) wgs = items;
    hipLaunchKernelGGL(attn_bwd_short_kernel, dim3((unsigned)wgs), dim3(SB_THREADS), smem, stream, a);
    return dvla_check_launch();
  }
  const size_t smem_dq = fa_smem_bytes(a.nkt, a.Lk, a.key_index != nullptr, a.tile_map != nullptr && a.bits_q != nullptr);
  const size_t smem_dkv = fa_dkv_smem_bytes(a.nqt, a.tile_map != nullptr && a.bits_k != nullptr, a.has_drop != 0);
  const dim3 grid_dq((unsigned)((a.nqt + 3) / 4), (unsigned)a.H, (unsigned)a.B);
  const dim3 grid_dkv((unsigned)((a.nkt + 3) / 4), (unsigned)a.H, (unsigned)a.B);
  const bool span_kv = attn_span32(a.Lk, a.kst, a.key_index != nullptr) && attn_span32(a.Lk, a.vst, a.key_index != nullptr);
  const bool span_q = attn_span32(a.Lq, a.qst, false) && attn_span32(a.Lq, a.dst, false);
  const bool dq_ring = smem_dq <= 64 * 1024 && span_kv && !attn_force_staged();
  a.fuse_delta = dq_ring ? 1 : 0;     // the dQ ring kernel covers every query row: it computes delta on its way
  if (!dq_ring) {
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((nrows * 8 + 255) / 256)), dim3(256), 0, stream, a);
    rc = dvla_check_launch();
    if (rc != DVLA_OK) return rc;
  }
  if (dq_ring)
    hipLaunchKernelGGL(attn_bwd_dq_ring_kernel, grid_dq, block, smem_dq, stream, a);
  else
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid_dq, block, 0, stream, a);
  rc = dvla_check_launch();
  if (rc != DVLA_OK) return rc;
  // (two workgroups of this kernel share a CU -- 186 VGPRs -- so a workgroup may take 80 KiB of its 160: the dropout tile keys
  // of a 930-token window are 15 KiB on top of 56)
  if (smem_dkv <= 80 * 1024 && span_q && !attn_force_staged()) {
    if (smem_dkv > 64 * 1024) {
      static uint64_t attr_done = 0;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
      if (!(attr_done & (1ull << dev))) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_done |= 1ull << dev;
      }
    }
    hipLaunchKernelGGL(attn_bwd_dkv_ring_kernel, grid_dkv, block, smem_dkv, stream, a);
  }
  else
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid_dkv, block, 0, stream, a);
  return dvla_check_launch();
}

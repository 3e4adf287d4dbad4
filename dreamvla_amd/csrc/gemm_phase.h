// gemm_phase.h -- "phase" GEMM: 256 x 256 tile, K-tile 64, two wave groups in ping-pong on every SIMD.
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T ), same parameter block / operand layouts / epilogue as the ring kernels.
//
// Why another structure (profiles/r02_gemm_probe_*.txt): every ring configuration ends up at the same 520-860 TFLOP/s on
// the model's shapes -- one barrier per 32-wide stage with all eight waves reading fragments, then all eight multiplying,
// leaves the matrix pipe idle while LDS is read and LDS idle while the pipe runs; 64-byte LDS rows fetch half cache lines;
// and a K-tile of 64 does not fit a 3-deep stage ring at 256 x 256 (3 x 64 KiB).  This kernel keeps 32-KiB operand IMAGES
// (three of A, two of B) and gets its look-ahead from refilling each image as soon as ITS last fragment read has retired:
//
//   waves 0-3 (group 0: rows 0-127 of the tile) and waves 4-7 (group 1: rows 128-255) sit pairwise on the four SIMDs and
//   run the same segment sequence one barrier apart, so that a SIMD always has one wave multiplying and one loading:
//
//     per K-tile u (buffer u & 1):    LOAD1 | MFMA1 | LOAD2 | MFMA2        (| = s_barrier; group 1 is one segment behind)
//       LOAD1: fragments a-lo (rows 0-63 of the wave's 128) x 4 k-steps and all B fragments (64 columns x 4 k-steps)
//       MFMA1: 16 x v_mfma_f32_32x32x16_bf16 (a-lo x B)
//       LOAD2: fragments a-hi into the a-lo registers
//       MFMA2: 16 MFMAs (a-hi x B, B fragments still in registers)
//     and this wave's 4 DMA pieces of the A image of tile u+2 (slot (u+2) % 3: last read in LOAD2(u-1)) and 4 of the B image
//     of tile u+2 (buffer u&1: its B fragments were all read in LOAD1(u), by both groups -- hence from LOAD2 on).  Rounds 2-4
//     spread them over the four segments (LOAD1: A0 | MFMA1: A1 A2 | LOAD2: A3 B0 | MFMA2: B1 B2 B3); round 5 issues them at
//     the head of the two load segments, in three instructions each (piece_place / LEAN below: +15 % at 8192^3).
//     every ds_read of a segment is waited for (lgkmcnt(0)) BEFORE the barrier that ends the segment -- the wait is hidden
//     under the partner's MFMA segment -- so "read in segment s" means "retired by the end of s", which is what the refill
//     rule above needs.  RAW: at the end of MFMA2(u) every wave waits vmcnt(8): all of its pieces except the eight it issued
//     during tile u (A(u+2), B(u+2)) have landed, i.e. A(u+1) and B(u+1); group 1 (whose MFMA2(u) ends one slot after group 0
//     starts reading tile u+1) waits for the same pieces at the end of its LOAD2(u) (vmcnt(n): the n pieces of tile u issued so far are
//     younger -- pieces_up_to() of the placement table).  The barriers publish.  Both operands run two tiles ahead: three A images (tile u in slot u % 3) and two B
//     images (whose fragments are all read in LOAD1 and which are therefore free again from LOAD2 on): 160 KiB of LDS.
//   The DMA stream never stops at an output-tile boundary (two cursors walk the work-item list ahead of the multiply); at a
//   boundary the groups re-align with one extra barrier, run the register-only epilogue at the same time and re-stagger.
//
// Requirements (dispatcher falls back otherwise): as ring_ok with BKS = 64: 16-B-vectorisable operands / outputs,
// K-range % 64 == 0, N % 64 == 0, M >= 256, N >= 256, r-contiguous operands with rows % 256 == 0.
// DBG & 128 (round 4, the TT layout's fp32 class = the weight gradients): K % 64 may be 16 / 32 / 48 -- the trunk's weight
// gradients contract over 20 832 = 651 x 32 tokens and ran on the BK-32 ring kernel at 775-850 TFLOP/s where this kernel runs the
// same problem class at 1 200-1 300 (profiles/r03_gemm_breakdown.json).  The last K-tile of the last K slice is then partial:
// its DMA pieces that would read k rows past the operand are pointed at the tile's first row instead (same count of pieces:
// the counted waits do not change), and the A fragments of its dead k16-steps are zeroed after the fragment reads have
// retired, so whatever the B fragments of those steps hold is multiplied by zero.
#pragma once
#include "gemm_impl.h"

namespace dvla_gemm {

struct PCfg {
  static constexpr int BM = 256, BN = 256, BKS = 64, GH = 4, NT = 512, TM = 4, TN = 2;
  static constexpr int A_BYTES = BM * BKS * 2, B_BYTES = BN * BKS * 2;
  static constexpr int NA = 3, NB = 2;               // A images: tile u in slot u % 3 (two tiles of look-ahead); B images: u % 2
  static constexpr int B_BASE = NA * A_BYTES;
  static constexpr int SMEM_BYTES = NA * A_BYTES + NB * B_BYTES;   // 160 KiB: the whole LDS of a CU
  static constexpr int WG_PER_CU = 1;
  static constexpr int CPW = 4;                      // DMA pieces (1 KiB each) per wave and operand image
};

// Where in a K-tile a wave issues its eight LDS-DMA pieces (A0..A3 of image A(u+2): legal anywhere in tile u; B0..B3 of image
// B(u+2): from LOAD2 on -- file header).  Slots: 0 / 1 / 2 = LOAD1 before the fragment reads / behind them / behind their
// lgkmcnt(0); 3..6 = MFMA1 behind k16-step 0..3; 7 / 8 / 9 = LOAD2 likewise; 10..13 = MFMA2 behind k16-step 0..3.  Pieces of an
// operand must be in non-decreasing slot order (the cursor advances behind the fourth).  PL = (DBG >> 8) & 15 selects the table.
// Round 5 (profiles/r05_gemm_placement_probe*.txt, 12 placements x {round-4 issue code, three-instruction issue}, NN plain,
// same-process A/B): with every piece in a LOAD segment IN FRONT of the fragment reads the multiply segments are 568-588 cycles
// (512 of MFMA) instead of 700-880 with two or three pieces between their MFMAs, and the load segments still fit beside them --
// the four waves of a group issue their pieces at the same point of the program, the CU's one address path takes them one after
// the other (16 cycles per 1-KiB piece), and what a piece costs the ISSUING wave is its place in that queue: harmless in a wave
// that is about to wait for LDS anyway, ~100 cycles of idle matrix pipe in a wave that is multiplying.  8192^3: 1 272 -> 1 405
// TFLOP/s from the placement alone, -> 1 490 together with the three-instruction issue (below); K = 1024: 992 -> 1 192.
struct PiecePlace { int slot[8]; };
__device__ __forceinline__ constexpr std::true_type live_or(std::true_type, bool) { return {}; }
__device__ __forceinline__ constexpr bool live_or(std::false_type, bool live) { return live; }
__host__ __device__ constexpr PiecePlace piece_place(int pl) {
  switch (pl) {
    case 1:  return {{2, 2, 2, 2, 9, 9, 9, 9}};        // everything behind the fragment reads' wait, in front of the barrier (-1 ... -3 %)
    case 13: return {{1, 4, 6, 8, 8, 11, 12, 13}};     // rounds 2-4: LOAD1: A0 | MFMA1: A1 A2 | LOAD2: A3 B0 | MFMA2: B1 B2 B3 (-7 %)
    default: return {{0, 0, 0, 0, 7, 7, 7, 7}};        // production: in front of the fragment reads of the two load segments
  }
}
__host__ __device__ constexpr int pieces_up_to(int pl, int slot) {
  int n = 0;
  for (int i = 0; i < 8; ++i) n += piece_place(pl).slot[i] <= slot ? 1 : 0;
  return n;
}

// 16-byte agent-coherent accesses to a stream-K slab (256 KiB, register layout) through a buffer descriptor with the sc1 bit:
// write-through stores / loads served at the device coherence point -- the two workgroups that share a tile sit on different
// XCDs, whose L2s do not snoop each other (cdna guide section 6, "in-launch combine").
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc(float* slab) {
  return __builtin_amdgcn_make_buffer_rsrc(slab, 0, PCfg::BM * PCfg::BN * 4, 0x00020000);
}
// DBG (ablation builds, results are garbage by design; variants 81..86 of the NT layout): 1 = no MFMAs, 2 = no fragment
// reads, 4 = no DMA, 8 = no barriers inside the K loop
template <bool A_T, bool B_T, int EPI, int DBG = 0>
__global__ __launch_bounds__(PCfg::NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_phase_kernel(GemmKArgs p) {
  constexpr int BM = PCfg::BM, BN = PCfg::BN, BKS = PCfg::BKS, CPW = PCfg::CPW;
  constexpr bool TAIL = (DBG & 128) != 0;
  constexpr int PL = (DBG >> 8) & 15;
  // DEFER (DBG & 16384, round 6; the P classes): the epilogue of a whole tile (kind 0) that has a successor runs at the head of LOAD1 of
  // the NEXT tile's first K-tile -- every fragment register is dead there, the accumulators are complete, and the partner group is in a
  // multiply segment (group 0's epilogue beside group 1's MFMA2 of the old tile, group 1's beside group 0's MFMA1 of the new one: the
  // VALU and the CU's store path serve ONE group at a time instead of both with the matrix pipe idle) -- and that K-tile's MFMA1 / MFMA2
  // overwrite the accumulator halves through a zero C operand (no zeroing).  The groups keep their one-segment skew across such a
  // boundary: no re-align / re-stagger barriers, no drain of the DMA queue.  The bias enters the accumulators as one short MFMA per
  // block in the first K-tile (no bias registers, no adds in the epilogue).
  constexpr bool DEFER = (DBG & 16384) != 0;
  // ZC (the deferred builds): the first k16-step of a segment's first K-tile multiplies into a literal-zero C operand instead of 128
  // accumulator registers zeroed by the VALU at every tile top.  (On its own -- the product loop with its two copies, per-tile
  // accumulators left undefined -- hipcc spills 691 registers around the first-K-tile branches: not a separable piece.)
  constexpr bool ZC = DEFER;
  static_assert(!(DBG & 32768) || (!TAIL && (DBG & 4096) == 0), "buffer DMA path: lean issue code, whole K-tiles");
  static_assert(!DEFER || (!TAIL && (DBG & (4096 | 8192)) == 0 && PL == 0 && (EPI == EPI_P0 || EPI == EPI_P_ERF || EPI == EPI_P_TANH)),
                "deferred epilogue: production loop, P classes");
  static_assert(!TAIL || (A_T && B_T), "partial K-tiles: k-major operands only");
  static_assert(piece_place(PL).slot[4] >= 7, "B pieces: from LOAD2 on");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wave >> 2, wc = wave & 3;    // group = row half of the tile, wc = 64-column quarter

  const int nitems = p.tiles_m * p.tiles_n * p.split_k;
  const int grid = gridDim.x;
  // (round 6: also for grids that are not a multiple of 8 -- XCD x then hosts grid / 8 (+ 1 for x < grid % 8) workgroups.  The
  // identity map such grids used to get spread consecutive items -- the K slices of one tile, neighbouring tiles of one operand
  // panel -- over all eight L2s: 252-item weight gradients ran 20-26 % slower than their 216-item siblings,
  // profiles/r06_small_dw_sweep.txt.)
  const int perm = [&] {
    const int q8 = grid >> 3, r8 = grid & 7, xcd = (int)(blockIdx.x & 7), idx = (int)(blockIdx.x >> 3);
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }();
  // ---- work segments.  Plain schedule: segment `it` = item it * grid + perm (a whole tile, or one K slice of it).
  // Stream-K hybrid (p.sk_tiles = R > 0 SUPER-tiles, split_k == 1, one workgroup per CU):
  //   a super-tile = SG = 16 consecutive tile ids (the raster makes that a 4 x 4 block of tiles), worked on by a GROUP of
  //   16 workgroups of adjacent rank (one XCD) in K lockstep, workgroup i of the group on tile i of the super-tile -- so the
  //   4 x 4 block's operand slices are fetched once per group and K-tile, as in the plain schedule (ranges per single
  //   workgroup, each at its own K offset, share nothing: 1.3 GB of L2 misses per 20832 x 1024 x 4096 launch, HBM-bound);
  //   the K-iterations of the first R super-tiles (R * nsT, super-tile-major) are cut into grid / 16 equal contiguous ranges,
  //   one per group; then one whole tile per workgroup and round from the rest (16 R + j * grid + perm).
  //   R >= number of groups, so a range is at least nsT long and a tile is shared by at most two workgroups: its HEAD
  //   (from k = 0) is the LAST stream-K segment of workgroup `perm`, its TAIL (to the end) the FIRST segment of workgroup
  //   perm + 16.  The tail's accumulators go to slab perm + 16 (fp32, register layout, write-through stores) and flag
  //   perm + 16 is raised; the head's workgroup -- the tile's owner -- waits for that flag, adds the slab and runs the
  //   epilogue.  A tail is the first thing its workgroup does and waits for nothing, so every wait ends as soon as that
  //   workgroup has been scheduled and has run its first segment (all workgroups are co-resident: one per CU; a late one only
  //   delays one epilogue of its predecessor).  The owner lowers the flag again: no per-launch memset.
  constexpr int SG = 16;
  const int nsT = (int)(p.K / BKS);
  const int sk_R = p.sk_tiles;
  const int sk_gi = perm & (SG - 1);
  unsigned sk_lo = 0, sk_hi = 0;
  int sk_t0 = 0, sk_n = 0;
  if (sk_R > 0) {
    const unsigned groups = (unsigned)grid / SG, g = (unsigned)perm / SG;
    const unsigned long long I = (unsigned long long)sk_R * (unsigned)nsT;
    sk_lo = (unsigned)(I * g / groups);
    sk_hi = (unsigned)(I * (g + 1) / groups);
    if (sk_hi > sk_lo) { sk_t0 = (int)(sk_lo / (unsigned)nsT); sk_n = (int)((sk_hi - 1) / (unsigned)nsT) - sk_t0 + 1; }
    while (sk_n > 0 && (sk_t0 + sk_n - 1) * SG + sk_gi >= nitems) --sk_n;   // the last super-tile may be ragged
  }
  struct Seg { int m0, n0, k_begin, ns, split, kind; };   // kind: 0 whole item, 1 head (owner), 2 tail
  // valid k16-steps of a segment's last K-tile (4 = whole); only a build with TAIL can see less
  auto tail_steps = [&](const Seg& sg) -> int {
    if constexpr (!TAIL) return 4;
    const int64_t k_last = (int64_t)sg.k_begin + (int64_t)(sg.ns - 1) * BKS;
    const int64_t left = p.K - k_last;
    return left >= BKS ? 4 : (int)(left / 16);
  };
  auto seg_of = [&](int it, Seg& sg) -> bool {
    int id;
    sg.kind = 0;
    if (sk_R == 0) {
      id = it * grid + perm;
      if (id >= nitems) return false;
    } else if (it < sk_n) {
      const int st = sk_t0 + it;
      id = st * SG + sk_gi;
      const unsigned t_begin = (unsigned)st * (unsigned)nsT, t_end = t_begin + (unsigned)nsT;
      const unsigned b = sk_lo > t_begin ? sk_lo : t_begin, e = sk_hi < t_end ? sk_hi : t_end;
      const RingItem w = ring_item<PCfg>(p, id);
      sg.m0 = (int)w.m0; sg.n0 = (int)w.n0; sg.split = 0;
      sg.k_begin = (int)(b - t_begin) * BKS; sg.ns = (int)(e - b);
      sg.kind = b != t_begin ? 2 : (e != t_end ? 1 : 0);
      return true;
    } else {
      id = sk_R * SG + (it - sk_n) * grid + perm;
      if (id >= nitems) return false;
    }
    const RingItem w = ring_item<PCfg>(p, id);
    sg.m0 = (int)w.m0; sg.n0 = (int)w.n0; sg.k_begin = (int)w.k_begin; sg.ns = w.ns; sg.split = w.split;
    return true;
  };
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // CACHE (off in the DBG & 4096 build): one decode of a work segment per tile instead of four.  A segment index is decoded by the multiply loop
  // (its own tile), by the look-ahead test of the LEAN build (the next one) and by both DMA cursors when they cross into the next
  // segment -- ~150 dependent scalar instructions each (ring_item: three divisions by reciprocal, one real division by the ragged
  // raster height), all of them on the tile boundary's critical path.  The tile top decodes segment it + 1 once and keeps it.
  // (not in the partial-K-tile build that also carries k-sums: at the register limit, and its segments are 80+ K-tiles long)
  constexpr bool CACHE = (DBG & 4096) == 0 && !((DBG & 128) != 0 && (DBG & 8192) != 0);
  Seg c_seg = {0, 0, 0, 0, 0, 0};
  int c_it = -1;
  bool c_ok = false;
  auto seg_cached = [&](int it, Seg& sg) -> bool {
    if constexpr (CACHE) {
      if (it != c_it) { c_ok = seg_of(it, c_seg); c_it = it; }
      sg = c_seg;
      return c_ok;
    } else {
      return seg_of(it, sg);
    }
  };

  // ---- DMA cursors: cursor X walks the K-tiles of the work-item list for operand X; this wave owns pieces
  // 4*wave .. 4*wave+3 of every 32-piece operand image.  nA / nB = number of tiles issued so far (buffer = n & 1). ----
  // per-lane source of piece i as a 32-bit BYTE offset from the operand base (a scalar pair): one VGPR and one 32-bit add per
  // piece instead of a 64-bit pointer -- 8 VGPRs of a loop that sits at the register limit (launch_phase takes operands that
  // span less than 4 GiB; the others go to the ring kernels)
  uint32_t srcA[CPW];
  uint32_t srcB[CPW];
  int itA = 0, ktA = 0, nkA = 0, nA = 0; bool liveA = false;   // nA = slot (mod 3) of the next A image to issue
  int itB = 0, ktB = 0, nkB = 0, nB = 0; bool liveB = false;
  int tvA = 4, tvB = 4;                                         // tail_steps of the segments the cursors are in
  const uint32_t stepA = (uint32_t)(2 * (A_T ? (int64_t)BKS * p.lda : (int64_t)BKS));
  const uint32_t stepB = (uint32_t)(2 * (B_T ? (int64_t)BKS * p.ldb : (int64_t)BKS));
  // LEAN (every build but DBG & 4096): a piece is THREE instructions -- `s_add_u32 m0, <LDS address of this wave's piece 0 of the image>, 1024 i`,
  // the hazard nop, the DMA -- instead of ~12 (liveness branch, shift + add for the LDS address, M0 saved / set / restored, the
  // per-lane source advanced by a v_add + v_mov).  The K position lives in the SCALAR base of the instruction (one s_add_u32 +
  // s_addc_u32 per operand and K-tile) and the per-lane offsets stay what openA / openB computed; M0 is not restored (no
  // instruction of these kernels reads it: tests/test_phase_isa.py); the liveness test is made once per TILE (two copies of the K
  // loop, see kloop below -- two copies of the K-tile BODY inside one loop made the register allocator spill the accumulators).
  constexpr bool LEAN = (DBG & 4096) == 0;        // DBG & 4096: the round-4 issue code (measurement variant 40 of gemm.hip)
  const char* kbA = reinterpret_cast<const char*>(p.A);
  const char* kbB = reinterpret_cast<const char*>(p.B);
  uint32_t m0A = smem_base + (uint32_t)(wave * (CPW * 1024));
  uint32_t m0B = m0A + (uint32_t)PCfg::B_BASE;
  // BUFDMA (the deferred-epilogue builds): pieces through a buffer descriptor over the whole operand (gemm_impl.h glds16b_lean).
  // Piece i of this wave = voX[i & 1] (per lane, relative to the segment's origin) + soX / soX2 (scalar: origin + K position, for
  // pieces 0-1 / 2-3).  k-contiguous operand: piece c = 4 wave + i holds rows 8 c + lane / 8, slot lane % 8 swizzled by
  // (row >> 1) & 7 = (lane >> 4) + 4 (i & 1); r-contiguous: k rows 2 c + lane / 32, slot lane % 32 swizzled by 4 (k & 3) =
  // 4 (2 (i & 1) + lane / 32) -- dma_src's maps, with the piece index split into parity (per lane) and pair (scalar).
  constexpr bool BUFDMA = DEFER || (DBG & 32768) != 0;     // (DBG & 32768 alone: measurement build, the round-5 boundary on the buffer path)
  uint32_t voA[2] = {0, 0}, voB[2] = {0, 0};
  uint32_t soA = 0, soA2 = 0, soB = 0, soB2 = 0;
  const rsrc4 rsA = operand_rsrc(p.A, (uint64_t)(A_T ? p.K : p.M) * (uint64_t)p.lda * 2);
  const rsrc4 rsB = operand_rsrc(p.B, (uint64_t)(B_T ? p.K : p.N) * (uint64_t)p.ldb * 2);
  auto buf_offsets = [&](auto trans_c, int64_t ld, int64_t row0, int64_t k0, uint32_t (&vo)[2], uint32_t& so, uint32_t& so2) {
    constexpr bool TR = decltype(trans_c)::value;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));          // (per segment, not hoisted: these are ~10 VALU)
    const uint32_t ldb2 = (uint32_t)(ld * 2);
    if constexpr (!TR) {
      const uint32_t r = (uint32_t)(wave * 32 + (lane_o >> 3)), sl = (uint32_t)(lane_o & 7), sw = (uint32_t)(lane_o >> 4);
      vo[0] = r * ldb2 + ((sl ^ sw) << 4);
      vo[1] = (r + 8) * ldb2 + ((sl ^ (sw + 4)) << 4);
      so = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((row0 * ld + k0) * 2));
      so2 = so + 16 * ldb2;
    } else {
      const uint32_t k = (uint32_t)(wave * 8 + (lane_o >> 5)), sl = (uint32_t)(lane_o & 31), h = (uint32_t)(lane_o >> 5);
      vo[0] = k * ldb2 + ((sl ^ (4 * h)) << 4);
      vo[1] = (k + 2) * ldb2 + ((sl ^ (4 * (2 + h))) << 4);
      so = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((k0 * ld + row0) * 2));
      so2 = so + 4 * ldb2;
    }
  };
  auto openA = [&](int it) {
    Seg w;
    liveA = seg_cached(it, w);
    if (!liveA) return;
    nkA = w.ns; ktA = 0; tvA = tail_steps(w);
    if constexpr (BUFDMA) { buf_offsets(std::integral_constant<bool, A_T>{}, p.lda, w.m0, w.k_begin, voA, soA, soA2); return; }
    if constexpr (LEAN) kbA = reinterpret_cast<const char*>(p.A);
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      srcA[i] = (uint32_t)(reinterpret_cast<const char*>(dma_src<A_T, BM, BKS>(p.A, p.lda, w.m0, p.M, w.k_begin, wave * CPW + i, lane)) -
                           reinterpret_cast<const char*>(p.A));
  };
  auto openB = [&](int it) {
    Seg w;
    liveB = seg_cached(it, w);
    if (!liveB) return;
    nkB = w.ns; ktB = 0; tvB = tail_steps(w);
    if constexpr (BUFDMA) { buf_offsets(std::integral_constant<bool, B_T>{}, p.ldb, w.n0, w.k_begin, voB, soB, soB2); return; }
    if constexpr (LEAN) kbB = reinterpret_cast<const char*>(p.B);
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      srcB[i] = (uint32_t)(reinterpret_cast<const char*>(dma_src<B_T, BN, BKS>(p.B, p.ldb, w.n0, p.N, w.k_begin, wave * CPW + i, lane)) -
                           reinterpret_cast<const char*>(p.B));
  };
  // one DMA piece (i = 0..3) of the next A / B image; *_done advances the cursor after the 4th (the placement table decides
  // where in the K-tile each piece goes).
  // partial last K-tile (TAIL): piece c = 4 wave + i of a k-major image holds k rows 2 c and 2 c + 1 of the tile (lanes 0-31 /
  // 32-63, dma_src); a row at or past 16 tail_steps lies outside the operand: read the tile's row 0 in its place
  auto tail_back = [&](int i, int tv, int64_t ld) -> uint32_t {
    const int k_local = 2 * (wave * CPW + i) + (lane >> 5);
    return k_local >= 16 * tv ? (uint32_t)(2 * (int64_t)k_local * ld) : 0u;
  };
  auto pieceA = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (BUFDMA) { glds16b_lean<i * 1024>(rsA, voA[i & 1], (i >> 1) ? soA2 : soA, m0A); return; }
    uint32_t off = srcA[i];
    if constexpr (TAIL) { if (tvA < 4 && ktA == nkA - 1) off -= tail_back(i, tvA, p.lda); }
    if constexpr (LEAN) {
      glds16s_lean<i * 1024, EPI == EPI_GEN ? 3 : 0>(kbA, off, m0A);
    } else {
      const uint32_t dst = smem_base + (uint32_t)(nA * PCfg::A_BYTES + wave * (CPW * 1024) + i * 1024);
      glds16s(p.A, off, __builtin_amdgcn_readfirstlane(dst));
      srcA[i] += stepA;
    }
  };
  auto doneA = [&]() {
    nA = nA == PCfg::NA - 1 ? 0 : nA + 1;
    if constexpr (BUFDMA) { soA += stepA; soA2 += stepA; }
    if constexpr (LEAN) { kbA += stepA; m0A = nA == 0 ? m0A - (uint32_t)((PCfg::NA - 1) * PCfg::A_BYTES) : m0A + (uint32_t)PCfg::A_BYTES; }
    if (++ktA == nkA) openA(++itA);
  };
  auto pieceB = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (BUFDMA) { glds16b_lean<i * 1024>(rsB, voB[i & 1], (i >> 1) ? soB2 : soB, m0B); return; }
    uint32_t off = srcB[i];
    if constexpr (TAIL) { if (tvB < 4 && ktB == nkB - 1) off -= tail_back(i, tvB, p.ldb); }
    if constexpr (LEAN) {
      glds16s_lean<i * 1024, EPI == EPI_GEN ? 3 : 0>(kbB, off, m0B);
    } else {
      const uint32_t dst = smem_base + (uint32_t)(PCfg::B_BASE + (nB & 1) * PCfg::B_BYTES + wave * (CPW * 1024) + i * 1024);
      glds16s(p.B, off, __builtin_amdgcn_readfirstlane(dst));
      srcB[i] += stepB;
    }
  };
  auto doneB = [&]() {
    ++nB;
    if constexpr (BUFDMA) { soB += stepB; soB2 += stepB; }
    if constexpr (LEAN) { kbB += stepB; m0B = (nB & 1) ? m0B + (uint32_t)PCfg::B_BYTES : m0B - (uint32_t)PCfg::B_BYTES; }
    if (++ktB == nkB) openB(++itB);
  };
  auto issueA = [&]() -> bool {
    if (!liveA) return false;
    static_for<CPW>([&](auto ic) { pieceA(ic); });
    doneA();
    return true;
  };
  auto issueB = [&]() -> bool {
    if (!liveB) return false;
    static_for<CPW>([&](auto ic) { pieceB(ic); });
    doneB();
    return true;
  };

  // the pieces the placement table puts into slot SLOT (ia / ib: the cursors were live when the segment that owns them began)
  auto emit = [&](auto slot_c, auto ia, auto ib) {      // ia / ib: bool, or std::true_type in the all-live loop body
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr PiecePlace pp = piece_place(PL);
    if (DBG & 4) return;
    static_for<4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (pp.slot[i] == SLOT) { if (ia) { pieceA(ic); if (i == 3) doneA(); } }
    });
    static_for<4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (pp.slot[4 + i] == SLOT) { if (ib) { pieceB(ic); if (i == 3) doneB(); } }
    });
  };
#define DVLA_SLOT(N, IA, IB) do { __builtin_amdgcn_sched_barrier(0); emit(std::integral_constant<int, N>{}, IA, IB); __builtin_amdgcn_sched_barrier(0); } while (0)

  // ---- prologue: A(0), B(0), A(1), B(1) in this order; the first two must have landed before the first fragment read ----
  openA(0); openB(0);
  issueA();
  issueB();
  const bool a1 = issueA();
  const bool b1 = issueB();
  if (a1 && b1) wait_vmcnt<2 * CPW>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // DBG & 64: waves 0 and 4 of workgroup 0 record s_memtime at the eight segment edges of their first 32 K-tiles into
  // p.workspace (as uint64[2][32][8]) -- the timeline the schedule is tuned against (tests/probes/gemm_probe.cpp --stamps)
  uint64_t* stamps = nullptr;
  if constexpr ((DBG & 64) != 0) {
    if (blockIdx.x == 0 && (wave == 0 || wave == 4) && p.workspace) stamps = reinterpret_cast<uint64_t*>(p.workspace) + (wave >> 2) * 256;
  }
  auto stamp = [&](int kt_global, int e) {
    if constexpr ((DBG & 64) != 0) {
      if (stamps && kt_global < 32 && lane == 0) stamps[kt_global * 8 + e] = __builtin_amdgcn_s_memtime();
    }
  };
  // tile-boundary stamps (uint64[2][16][4] behind the K-tile stamps): 0 = K loop left, 1 = groups re-aligned, 2 = epilogue
  // code done (all stores issued), 3 = the next tile's K loop entered
  auto stamp_tile = [&](int tile, int e) {
    if constexpr ((DBG & 64) != 0) {
      if (stamps && tile < 16 && lane == 0)
        (reinterpret_cast<uint64_t*>(p.workspace) + 512 + (wave >> 2) * 64)[tile * 4 + e] = __builtin_amdgcn_s_memtime();
    }
  };
  // k-sums (dvla.h ksum_*; fp32 class only), round 5.  Rounds 3-4 could not carry them here: dots interleaved with the MFMAs competed
  // with the matrix pipe (+5 ... +26 % per launch, profiles/r03_gemm_ksum_probe_phase_dots.txt) and the weight gradients that need
  // a bias gradient went to the BK-32 ring kernel or paid a column-sum pass.  With the DMA pieces out of the multiply segments the
  // LOAD segments end in 240-370 cycles of waiting for the partner's multiply (profiles/r05_gemm_stamps_k1024.txt), so the dots go
  // THERE: behind the segment's lgkmcnt(0), in front of its barrier, on the fragments that segment has just read -- LOAD1: the two
  // a-lo fragments (and, for the B operand, the two B fragments) of ONE k16-step, LOAD2: the two a-hi fragments of that step.
  // Which step: the waves that hold the same rows share the four k16-steps of a K-tile -- the four waves of a group hold the
  // same A rows (wave wc takes step wc: pgroup wc of 4), the two groups hold the same B columns (group g takes steps g and g + 2:
  // pgroup g of 2).  16 v_dot2c_f32_bf16 per wave and K-tile, none in a multiply segment.  A launch waits for its slowest workgroup,
  // so the sum of a tile row (column) is SPREAD over up to four of its tiles: tile column c < S = min(4, tiles_n) sums the K-tiles
  // with kt % S == c into its own KSUM_PARTS partial rows (p.ksum_parts = 8 S rows per K slice; the reduction adds them up) -- with
  // only the first tile column summing, the dots cost a launch 7-13 % (profiles/r05_gemm_ksum_probe.txt).
  // The summing code exists only in the builds the dispatcher launches when a k-sum is asked for (DBG & 8192; the TT layout = the
  // weight gradients): a launch without k-sums runs the identical kernel without the four accumulators and their flags.
  constexpr bool KSUM = EPI == EPI_F32 && (DBG & 8192) != 0;
  // (Round 5, measured and not kept: no s_setprio at all -0.5 ... +1 %, static priority 1 for group 1 without per-segment flips
  // -1 ... -3 %; the groups keeping their one-segment skew through the tile boundary -- group 0's epilogue beside group 1's last
  // multiply segment -- -1 ... -2 %: profiles/r05_gemm_placement_probe*.txt, variants 55 / 56 / 71 / 72.)
  int u = 0, ua = 0;   // global K-tile counter of the multiply, and u % 3
  f32x16 acc_carried[DEFER ? 2 : 1][DEFER ? 4 : 1];      // DEFER: the accumulators live across tile boundaries
  if constexpr (DEFER) {     // (defined once: the first K-tile of every segment overwrites them)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_carried[i][j][r] = 0.f;
  }
  bool pend = false;          // DEFER: slabs 1-3 of the previous tile (origin pm0, pn0) are still in the accumulators
  int pm0 = 0, pn0 = 0;
  bool aligned_now = true;    // the two groups are at the same program point (false: group 1 runs one segment behind)
  for (int it = 0;; ++it) {
    // The segment's tile origin / split / kind stay live across the K loop as SCALARS (6 SGPRs, spilled to VGPR lanes when
    // the allocator runs out: one v_readlane each) -- re-deriving them for the epilogue was ~150 dependent scalar instructions
    // = ~800 cycles per tile with the matrix pipe idle (`entry` of the slab stamps).  Everything per-lane is still derived
    // from lane_e after the loop (the loop sits at the 256-VGPR limit).
    Seg w;
    if (!seg_cached(it, w)) break;
    w.m0 = __builtin_amdgcn_readfirstlane(w.m0); w.n0 = __builtin_amdgcn_readfirstlane(w.n0);
    w.split = __builtin_amdgcn_readfirstlane(w.split); w.kind = __builtin_amdgcn_readfirstlane(w.kind);
    const int seg_ns = w.ns;
    const int seg_tv = __builtin_amdgcn_readfirstlane(tail_steps(w));

    // per tile (dead at the tile top -- with two copies of the K loop anything else costs a register shuffle through scratch); zeroed
    // here, or (ZC) left undefined: the first K-tile's multiplies take a literal-zero C operand
    f32x16 acc_tile[DEFER ? 1 : 2][DEFER ? 1 : 4];
    f32x16 (&acc)[2][4] = *[&]() { if constexpr (DEFER) return &acc_carried; else return &acc_tile; }();
    if constexpr (!ZC) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    // DEFER: is this tile's epilogue deferred (a whole tile with a successor); the bias of the wave's 64 columns (lane l: column
    // n0 + 64 wc + l) is requested in LOAD2 of the tile's LAST K-tile and added behind the K loop -- a register that crossed the
    // in-loop epilogue of the previous tile tipped hipcc's allocator into spilling the DMA offsets (171 spills for ONE register)
    Seg nx;
    bool has_next = false, defer_this = false, tile_bias = false;
    uint32_t braw = 0;
    int bias_young = 0;       // DMA pieces issued behind the bias load
    if constexpr (DEFER) {
      has_next = seg_cached(it + 1, nx);
      defer_this = w.kind == 0 && has_next;
      tile_bias = p.bias != nullptr && w.kind != 2;      // (a stream-K tail's partial sums are added to the owner's, which carry it)
    }
    const bool fresh = DEFER ? aligned_now : true;      // this tile starts behind an old-style boundary (or is the first)

    float ksa[4] = {0.f, 0.f, 0.f, 0.f};     // k-sums of this wave's A row blocks (a-lo 0, 1; a-hi 2, 3) -- or, [0] and [1], of its B column blocks (a launch sums one operand)
    // S = p.ksum_parts / KSUM_PARTS tiles of a row (column) share its sum; this tile's column (row) index c takes the K-tiles with
    // kt % S == c: ks_cnt counts down to its next K-tile (-1: this tile does not sum).  Two live scalars.
    const int ks_c = !KSUM ? 0 : p.ksum_op == 1 ? w.n0 / BN : w.m0 / BM;
    int ks_cnt = (KSUM && ks_c * KSUM_PARTS < p.ksum_parts) ? ks_c : -1;
    if (grp == 1 && fresh) __builtin_amdgcn_s_barrier();   // stagger: group 1 runs one segment behind group 0
    aligned_now = false;
    __builtin_amdgcn_sched_barrier(0);
    if (it > 0) stamp_tile(it - 1, 3);

    // the K loop of one output tile; live_c = std::true_type: both DMA cursors stay live for the whole tile, every liveness test
    // folds away.  The cursors run two K-tiles ahead of the multiply, so that holds whenever a next work segment of at least two
    // K-tiles exists: the LEAN build runs this copy for every tile but the workgroup's last.
    auto kloop = [&](auto live_c) {
    for (int kt = 0; kt < seg_ns; ++kt, ++u, ua = (ua == PCfg::NA - 1 ? 0 : ua + 1)) {
      const char* bufA = smem + ua * PCfg::A_BYTES;                         // ua = u % 3
      const char* bufB = smem + PCfg::B_BASE + (u & 1) * PCfg::B_BYTES;
      bf16x8 fa[2][4], fb[2][4];   // [sub-tile][k16-step]
      if (DBG & 2) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) { fa[x][y] = bf16x8{1, 2, 3, 4, 5, 6, 7, (short)lane}; fb[x][y] = bf16x8{1, 2, 3, 4, 5, 6, 7, (short)kt}; }
      }

      // (DMA piece placement: piece_place() at the top of the file.  Both operands run TWO tiles ahead -- A(u+2) and B(u+2) are
      // issued during tile u; three A images make that possible -- so the counted waits below name pieces issued a whole tile
      // earlier and do not stall.)
      // The first K-tile behind an epilogue waits for NOTHING (round 5).  vmcnt is one in-order counter for loads and stores: a
      // counted wait behind the epilogue's 16-32 stores would wait for every one of them -- 2 200 cycles in the first K-tile of
      // a tile by the s_memtime stamps (profiles/r05_gemm_boundary.txt) -- although the pieces it is there for (A(u+1), B(u+1),
      // issued a whole K-tile before the epilogue) landed long ago.  So every DMA piece is waited for IN FRONT of the epilogue
      // (vmcnt(0) below: the youngest is half a K-tile old) and K-tile 0 of the next tile skips both of its waits; K-tile 1's
      // wait then meets stores that have had a whole K-tile to drain.
      // (Not in the partial-K-tile build: its K ranges are 80+ K-tiles long, and one more live scalar tips its loop into spilling.)
      const bool after_epi = LEAN && !TAIL && kt == 0 && it > 0 && fresh;
      const bool first = kt == 0;
      // (DEFER: the lane-dependent parts of the fragment addresses are re-derived per K-tile -- ~10 VALU in a load segment --
      //  instead of living in four registers across the in-loop epilogue, where hipcc spilled them and reloaded them behind vmcnt(0))
      int lane_k = lane;
      if constexpr (DEFER && (DBG & 65536) == 0) asm volatile("" : "+v"(lane_k));
      int nst = 0;      // DEFER: epilogue stores this wave has issued in this K-tile so far (the counted waits skip them)
      // ---------------- LOAD1 ----------------
      stamp(u, 0);
      const auto ia = live_or(live_c, liveA);
      DVLA_SLOT(0, ia, false);
      if constexpr (DEFER) {
        // the previous tile's epilogue: its accumulators are complete, every fragment register is dead, the partner group multiplies
        if (first && pend && pn0 + wc * 64 < p.N) {
          int lane_e = lane;
          asm volatile("" : "+v"(lane_e));
          if constexpr ((DBG & 131072) != 0) __builtin_amdgcn_s_setprio(2);      // (measurement: the epilogue above the partner's multiply segment)
          if constexpr ((DBG & 64) != 0) {      // slab stamps of the in-loop epilogue (same slots as the boundary epilogue's)
            auto st = [&](int idx) {
              __builtin_amdgcn_sched_barrier(0);
              if (stamps && it == 2 && lane == 0)
                (reinterpret_cast<uint64_t*>(p.workspace) + 640 + (wave >> 2) * 16)[idx] = __builtin_amdgcn_s_memtime();
              __builtin_amdgcn_sched_barrier(0);
            };
            reg_epilogue<4, EPI, decltype(st), true>(p, acc, lane_e, (int64_t)pm0 + grp * 128, (int64_t)pn0 + wc * 64, 0, st);
          } else {
            reg_epilogue<4, EPI, NoStamp, true>(p, acc, lane_e, (int64_t)pm0 + grp * 128, (int64_t)pn0 + wc * 64, 0);
          }
          nst = p.preact ? 32 : 16;
          if constexpr ((DBG & 131072) != 0) __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) if (!(DBG & 2)) fb[i][ks] = ring_frag<B_T, BN, BKS>(bufB, wc * 64 + i * 32, ks, lane_k);
#pragma unroll
        for (int j = 0; j < 2; ++j) if (!(DBG & 2)) fa[j][ks] = ring_frag<A_T, BM, BKS>(bufA, grp * 128 + j * 32, ks, lane_k);
      }
      DVLA_SLOT(1, ia, false);
      wait_lds();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TAIL) {      // the dead k16-steps of a partial last K-tile contribute nothing (fragment reads have retired)
        if (kt == seg_ns - 1 && seg_tv < 4) {
#pragma unroll
          for (int ks = 1; ks < 4; ++ks)
            if (ks >= seg_tv) { fa[0][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; fa[1][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (KSUM) {      // k-sums of this wave's k16-step (see the top of the kernel): a-lo now, a-hi in LOAD2
        const int tv_now = (TAIL && kt == seg_ns - 1) ? seg_tv : 4;       // dead k16-steps of a partial last K-tile hold no data
        if (ks_cnt == 0 && p.ksum_op == 1) {
          static_for<4>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if (wc == ks) { ksa[0] = ksum_add_sel(fa[0][ks], 0x3f803f80u, ksa[0]); ksa[1] = ksum_add_sel(fa[1][ks], 0x3f803f80u, ksa[1]); }
          });
        }
        if (ks_cnt == 0 && p.ksum_op == 2) {
          static_for<4>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if ((ks & 1) == grp && ks < tv_now) { ksa[0] = ksum_add_sel(fb[0][ks], 0x3f803f80u, ksa[0]); ksa[1] = ksum_add_sel(fb[1][ks], 0x3f803f80u, ksa[1]); }
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      DVLA_SLOT(2, ia, false);
      stamp(u, 1);
      if (!(DBG & 8)) __builtin_amdgcn_s_barrier();
      stamp(u, 2);
      // ---------------- MFMA1 ----------------
      __builtin_amdgcn_s_setprio(1);
      static_for<4>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if constexpr (ZC && ks == 0) {
          if (first) {        // a segment's first K-tile: C = 0 (nothing to zero)
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i][0], fa[j][0], zero, 0, 0, 0);
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i][0], fa[j][0], acc[i][j], 0, 0, 0);
          }
        } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (!(DBG & 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i][ks], fa[j][ks], acc[i][j], 0, 0, 0);
            else asm volatile("" :: "v"(fb[i][ks]), "v"(fa[j][ks]));
        }
        if constexpr (pieces_up_to(PL, 3 + ks) != pieces_up_to(PL, 2 + ks)) DVLA_SLOT(3 + ks, ia, false);
      });
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      stamp(u, 3);
      if (!(DBG & 8)) __builtin_amdgcn_s_barrier();
      stamp(u, 4);
      // ---------------- LOAD2 ----------------
      const auto ib = live_or(live_c, liveB);
      if constexpr (DEFER) {
        if (kt == seg_ns - 1 && tile_bias) {      // the tile's bias: in flight during its last multiply segment (older than this K-tile's B pieces)
          const int nb = w.n0 + wc * 64;
          const uint32_t col = (uint32_t)((nb < p.N ? nb : 0) + lane_k);
          braw = p.bias_f32 ? aload_b32(p.bias, 4 * col) : aload_u16(p.bias, 2 * col);
          bias_young = ib ? CPW : 0;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      DVLA_SLOT(7, ia, ib);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) if (!(DBG & 2)) fa[j][ks] = ring_frag<A_T, BM, BKS>(bufA, grp * 128 + 64 + j * 32, ks, lane_k);
      DVLA_SLOT(8, ia, ib);
      // group 1 sits one segment behind group 0, which reads tile u+1 in the next slot: group 1's pieces of A(u+1) / B(u+1)
      // (issued during ITS tile u-1) must have landed by now.  Younger than those: the pieces of this tile issued so far
      // (and, DEFER, the slab stores of this K-tile).
      if (grp == 1 && !after_epi) {
        if (DEFER && nst != 0) { if (ia && ib) wait_vmcnt_dyn(pieces_up_to(PL, 8) + nst); else wait_vmcnt<0>(); }
        else { if (ia && ib) wait_vmcnt<pieces_up_to(PL, 8)>(); else wait_vmcnt<0>(); }
      }
      wait_lds();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TAIL) {
        if (kt == seg_ns - 1 && seg_tv < 4) {
#pragma unroll
          for (int ks = 1; ks < 4; ++ks)
            if (ks >= seg_tv) { fa[0][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; fa[1][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (KSUM) {
        if (ks_cnt == 0 && p.ksum_op == 1) {
          static_for<4>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if (wc == ks) { ksa[2] = ksum_add_sel(fa[0][ks], 0x3f803f80u, ksa[2]); ksa[3] = ksum_add_sel(fa[1][ks], 0x3f803f80u, ksa[3]); }
          });
        }
        ks_cnt = ks_cnt > 0 ? ks_cnt - 1 : (ks_cnt == 0 ? p.ksum_parts / KSUM_PARTS - 1 : ks_cnt);
        __builtin_amdgcn_sched_barrier(0);
      }
      DVLA_SLOT(9, ia, ib);
      stamp(u, 5);
      if (!(DBG & 8)) __builtin_amdgcn_s_barrier();
      stamp(u, 6);
      // ---------------- MFMA2 ----------------
      __builtin_amdgcn_s_setprio(1);
      static_for<4>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        if constexpr (ZC && ks == 0) {
          if (first) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i][0], fa[j][0], zero, 0, 0, 0);
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i][0], fa[j][0], acc[i][2 + j], 0, 0, 0);
          }
        } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (!(DBG & 1)) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i][ks], fa[j][ks], acc[i][2 + j], 0, 0, 0);
            else asm volatile("" :: "v"(fb[i][ks]), "v"(fa[j][ks]));
        }
        if constexpr (pieces_up_to(PL, 10 + ks) != pieces_up_to(PL, 9 + ks)) DVLA_SLOT(10 + ks, ia, ib);
      });
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // everything but this tile's eight pieces (A(u+2), B(u+2)) -- and, DEFER, its slab stores -- has landed: A(u+1), B(u+1)
      if (!after_epi) {
        if (DEFER && nst != 0) { if (ia && ib) wait_vmcnt_dyn(2 * CPW + nst); else wait_vmcnt<0>(); }
        else { if (ia && ib) wait_vmcnt<2 * CPW>(); else wait_vmcnt<0>(); }
      }
      if constexpr (DEFER) { if (first) pend = false; }
      stamp(u, 7);
      if (!(DBG & 8)) __builtin_amdgcn_s_barrier();
        }
    };
    if constexpr (LEAN && !(TAIL && KSUM)) {      // (the partial-K-tile + k-sum build keeps one copy of its loop: registers)
      if constexpr (DEFER) {
        // ONE copy of the loop: the accumulators now live across tile boundaries, and two loops that hold them in different
        // registers cost a shuffle through scratch at every tile top (hipcc spilled 491 VGPRs)
        kloop(std::false_type{});
      } else {
        const bool all_live = seg_cached(it + 1, nx) && nx.ns >= 2;
        if (all_live) kloop(std::true_type{}); else kloop(std::false_type{});
      }
    } else {
      kloop(std::false_type{});
    }
    if constexpr (KSUM) {
      // rows of the four A blocks: a-lo j at grp * 128 + 32 j, a-hi j at grp * 128 + 64 + 32 j = ksum_store's consecutive 32-row blocks
      const int tc = p.ksum_op == 1 ? w.n0 / BN : w.m0 / BM;       // (re-derived: not kept live across the K loop)
      if (ks_cnt >= 0 && p.ksum_op == 1) ksum_store<4, 4>(p, ksa, lane, w.split, wc, 4, (int64_t)w.m0 + grp * 128, p.M, tc * KSUM_PARTS);
      if (ks_cnt >= 0 && p.ksum_op == 2) ksum_store<2, 4>(p, ksa, lane, w.split, grp, 2, (int64_t)w.n0 + wc * 64, p.N, tc * KSUM_PARTS);
    }
    if constexpr (DEFER) {
      if (tile_bias) {
        // the bias as one short MFMA per accumulator block (gemm_impl.h bias_split): group 0 issues them beside group 1's last
        // multiply segment, group 1 beside group 0's first of the next tile -- in front of an epilogue that outlasts either anyway
        wait_vmcnt_dyn(bias_young);
        uint32_t b_lo = braw, b_hi = braw;
        settle(b_lo, b_hi);
        if (!p.bias_f32) { b_lo <<= 16; b_hi <<= 16; }
        swap_halves(b_lo, b_hi);            // b_hi: lanes 0-31 now hold what lanes 32-63 loaded (columns 32-63)
        const bool lower = lane < 32;
        const bf16x4v bfr0 = bias_split(__uint_as_float(b_lo), lower), bfr1 = bias_split(__uint_as_float(b_hi), lower);
        const bf16x4v ones = bias_ones(lower);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(bfr0, ones, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(bfr1, ones, acc[1][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (defer_this) { pend = true; pm0 = w.m0; pn0 = w.n0; continue; }   // slabs 1-3 ride in the next tile's first K-tile; the groups stay skewed
    }
    stamp_tile(it, 0);
    if constexpr (LEAN && !TAIL) wait_vmcnt<0>();   // every DMA piece has landed before the first store is issued (see after_epi)
    if (grp == 0) __builtin_amdgcn_s_barrier();   // re-align: both groups run the epilogue together
    aligned_now = true;
    __builtin_amdgcn_sched_barrier(0);
    stamp_tile(it, 1);
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));              // (the per-lane epilogue addresses the compiler would hoist out of the segment
                                                  //  loop and spill are re-derived per tile from lane_e)
    if (DBG & 16) {        // no epilogue at all (keep the accumulators alive)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else if (DBG & 32) { // stores only: 16 x 16 B per lane straight from accumulator registers (no conversion, no exchange)
      const int l31 = lane_e & 31, g = lane_e >> 5;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t m = w.m0 + grp * 128 + j * 32 + l31;
        if (m < p.M) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + w.n0 + wc * 64 + 16 * q + 8 * g) =
                make_uint4(__float_as_uint(acc[q >> 1][j][4 * (q & 1)]), __float_as_uint(acc[q >> 1][j][4 * (q & 1) + 1]),
                           __float_as_uint(acc[q >> 1][j][4 * (q & 1) + 2]), __float_as_uint(acc[q >> 1][j][4 * (q & 1) + 3]));
        }
      }
    } else if (w.kind == 2) {
      // stream-K tail: the 128 accumulators of every lane -> slab `perm`, quad (i, j, rq) of wave `wave` at
      // [(wave * 32 + quad) * 64 + lane] (16 B per lane, 1 KiB per instruction), write-through (sc1: the owner sits on
      // another XCD); then every wave drains its stores, the workgroup meets, and one lane raises the flag.
      const __amdgpu_buffer_rsrc_t rs = sk_rsrc(p.sk_slabs + (size_t)perm * (BM * BN));
      const int voff = (wave * 32 * 64 + lane_e) * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const u32x4 v = {__float_as_uint(acc[i][j][4 * rq]), __float_as_uint(acc[i][j][4 * rq + 1]),
                             __float_as_uint(acc[i][j][4 * rq + 2]), __float_as_uint(acc[i][j][4 * rq + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + ((i * 4 + j) * 4 + rq) * 1024, 0, 16);
          }
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (t == 0) __hip_atomic_store(p.sk_flags + perm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (w.kind == 1) {
        // stream-K head = owner: wait for the tail of this tile (workgroup perm + 16), add its slab, lower the flag
        if (t == 0) {
          while (__hip_atomic_load(p.sk_flags + perm + SG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
          __hip_atomic_store(p.sk_flags + perm + SG, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_barrier();
        const __amdgpu_buffer_rsrc_t rs = sk_rsrc(p.sk_slabs + (size_t)(perm + SG) * (BM * BN));
        const int voff = (wave * 32 * 64 + lane_e) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {     // 16 quads (64 VGPRs) per batch
          u32x4 v[16];
#pragma unroll
          for (int qd = 0; qd < 16; ++qd) v[qd] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (i * 16 + qd) * 1024, 0, 16);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
              acc[i][j][4 * rq] += __uint_as_float(v[j * 4 + rq].x); acc[i][j][4 * rq + 1] += __uint_as_float(v[j * 4 + rq].y);
              acc[i][j][4 * rq + 2] += __uint_as_float(v[j * 4 + rq].z); acc[i][j][4 * rq + 3] += __uint_as_float(v[j * 4 + rq].w);
            }
        }
      }
      if constexpr ((DBG & 64) != 0) {
        // slab stamps of tile 1 (uint64[2][16] behind the boundary stamps): 0 = entry, 1 + 2j = slab j converted and transposed,
        // 2 + 2j = slab j's stores issued
        auto st = [&](int idx) {
          __builtin_amdgcn_sched_barrier(0);
          if (stamps && it == 1 && lane == 0)
            (reinterpret_cast<uint64_t*>(p.workspace) + 640 + (wave >> 2) * 16)[idx] = __builtin_amdgcn_s_memtime();
          __builtin_amdgcn_sched_barrier(0);
        };
        reg_epilogue<4, EPI, decltype(st), DEFER>(p, acc, lane_e, w.m0 + grp * 128, w.n0 + wc * 64, w.split, st);
      } else {
        reg_epilogue<4, EPI, NoStamp, DEFER>(p, acc, lane_e, w.m0 + grp * 128, w.n0 + wc * 64, w.split);
      }
    }
    if constexpr ((DBG & 64) != 0) __builtin_amdgcn_sched_barrier(0);
    stamp_tile(it, 2);
  }
}

template <bool A_T, bool B_T, int EPI, int DBG = 0>
void launch_phase_one(const GemmKArgs& a, int split_k, hipStream_t stream) {
  static bool attr_set = false;
  auto kern = &gemm_phase_kernel<A_T, B_T, EPI, DBG>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PCfg::SMEM_BYTES);
    attr_set = true;
  }
  const int64_t items = (int64_t)a.tiles_m * a.tiles_n * split_k;
  const int64_t slots = (int64_t)num_cus();
  const int64_t nwg = a.sk_tiles > 0 ? slots : persistent_grid(items, slots);   // stream-K: exactly one workgroup per CU
  dim3 grid((unsigned)nwg, 1, 1), block(PCfg::NT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, PCfg::SMEM_BYTES, stream, a);
}

}  // namespace dvla_gemm

// dit_team.hip -- the evaluation-time DiT action sampler as ONE persistent kernel on the 32 CUs of one XCD.
//
//   What it replaces.  A closed-loop control step samples the executed action with ten DDIM steps of classifier-free guidance
//   through the DiT head (models/dreamvla_model.py:935-987 -> models/action_model/models.py:162-268): per step 12 blocks of
//   LayerNorm -> qkv -> attention -> proj (+x) -> LayerNorm -> fc1 -> GELU -> fc2 (+x) on 2 x bs x 6 token rows (12 rows at one
//   episode).  Launch by launch that is ~600 few-row GEMM / attention launches per control step, each living for two memory round
//   trips behind a ~5 us launch boundary: 4.7 of the 9.9 ms of a single-episode control step (profiles/r04_rollout_step_summary_*).
//   The work itself is a weight stream: 170 MB of bf16 weights per DiT-B forward and ~0.1 GFLOP.
//
//   Why one XCD.  tests/probes/sync_probe.cpp (profiles/r04_sync_probe.jsonl) measures what a dependent exchange between
//   workgroups costs inside a kernel on this chip: 7.6 us across all 256 CUs with release / acquire fences (buffer_wbl2 /
//   buffer_inv: what a cooperative-groups grid sync does), 4.2 us when the exchanged data moves with device-scope (sc1)
//   accesses and nothing is flushed -- and 1.03 us among the 32 workgroups of ONE XCD, whose exchange stays in that XCD's L2.
//   One XCD streams 1.32 TB/s (170 MB in 134 us); the whole chip 6.5 TB/s (27 us) -- but a DiT forward is a chain of 60
//   dependent exchanges, so the chip-wide variant pays 60 x 4.2 = 250 us of latency per forward to save 107 us of streaming.
//   The team of one XCD is the faster machine for this problem: ~134 us of weight stream per forward with the exchanges
//   (60 x ~1 us) overlapped by prefetching the next phase's weights across each barrier.
//
//   Team.  The grid is one workgroup per CU; consecutive workgroups go to consecutive XCCs (round-robin placement: the probe
//   finds workgroup b on XCC b % 8; inside a process that has launched other kernels the rotation starts elsewhere, but every
//   eighth workgroup still shares an XCC), the 32 workgroups with b % 8 == 0 form the team, the others exit at once.  CORRECTNESS DOES NOT DEPEND ON THE PLACEMENT: every
//   access to data another workgroup wrote is a device-scope (sc1) access, coherent across XCDs as well -- a team that straddles
//   XCDs is only slower.  All spins are bounded; a timeout raises `status` and poisons the output with NaN.
//
//   Phase = one exchange.  Every GEMM of the chain is cut into tiles of 16 output columns; tile i belongs to team member i % 32.
//   A workgroup's 8 waves split K eight ways; a wave's weight fragments for ALL its workgroup's tiles of the phase (at most 32 x
//   16 B per lane) are requested BEFORE the barrier that ends the previous phase is waited on -- the weights do not depend on
//   the activations -- so the stream keeps running through the exchange.  After the barrier the wave reads its K-share of the
//   (at most 32) activation rows, LayerNorm is applied on the fly where the chain has one (row statistics exchanged through
//   LDS), the 16x16x32 MFMAs run, the eight partial tiles meet in LDS, and wave j finishes tile j: bias, GELU, residual,
//   rounding points exactly where the launch-by-launch path has them (csrc/gemm_impl.h epilogue_oct, csrc/gemm_skinny.h).
//   Attention (6 tokens, head_dim 64) is one wave per (sample, head) on the VALU with the rounding points of csrc/attention.hip
//   (integer running maximum in the log2 domain, probabilities rounded to bf16 before P.V).  The final LayerNorm + linear, the
//   guidance + DDIM update (csrc/elementwise.hip ddim_cfg_step_kernel, operation by operation) and the next step's token
//   embedding run on team member 0 between two steps.  One launch = the whole sampler: steps x (1 + 5 x depth) exchanges.
#include "common.h"
#include "../../include/dvla.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int TEAM = 32;        // workgroups of the team = CUs of one XCD
constexpr int NWAVES = 8;
constexpr int LDP = 20;         // row stride (floats) of a partial tile in LDS
constexpr int MAX_L = 8;        // tokens per sample the attention phase holds (2 x action_pred_steps)
constexpr unsigned SPIN_LIMIT = 1u << 22;

// ---- memory operations issued by hand.  The compiler does not see them as memory operations in flight: nothing it emits waits
// for them, so a phase can leave its successor's weight loads outstanding across barriers; every consumer sits behind vm_wait()
// + pin().  (Compiler-tracked loads / atomics in between stay correct: s_waitcnt vmcnt(n) retires in order, uncounted younger
// operations only make it wait longer.)
// (OFF: byte offset in the instruction's immediate field -- one address register pair per row, not one per load)
template <int OFF = 0>
__device__ __forceinline__ u32x4 ldg16(const void* p) {          // weights: read-only for the whole launch, ordinary caching
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF) : "memory");
  return v;
}
template <int OFF = 0>
__device__ __forceinline__ u32x4 ldd16(const void* p) {          // device scope: data another workgroup wrote in this launch
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(v) : "v"(p), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ u32x2 ldg8(const void* p) {
  u32x2 v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ u32x2 ldd8(const void* p) {
  u32x2 v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void std16(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void std8(void* p, u32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void std4(void* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <class T>
__device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }      // uses of v stay behind the preceding vm_wait()
__device__ __forceinline__ void wg_barrier() { __builtin_amdgcn_s_barrier(); }  // no fence: LDS hand-offs add their own lgkmcnt wait
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

// dynamic LDS, in floats: [row statistics 8 x 32 x 2][model output 32 x 16][sampler state 256][flags 16][partial tiles | attention]
// (the phase functions are not inlined: each declares the array itself, so that its accesses are LDS instructions, not flat ones)
constexpr int STAT_OFF = 0, MO_OFF = 512, XS_OFF = 1024, DEAD_OFF = 1280, PART_OFF = 1296;
#define DIT_LDS() extern __shared__ float dit_lds[]

struct Team {
  unsigned* ctr;       // arrivals since the launch
  unsigned* status;    // != 0: a wait timed out
  unsigned epoch;      // exchanges this workgroup has arrived at
  int rank;
  unsigned long long* stamps;   // measurement: 8 wall-clock stamps (100 MHz) per exchange of members 0 and 17, or null
};
// stamp k of the current exchange: 0 weights requested, 1 producers arrived, 2 operands landed, 3 partial tiles in LDS, 4 stored
__device__ __forceinline__ void stamp(const Team& tm, int k) {
  if (tm.stamps && threadIdx.x == 0 && (tm.rank == 0 || tm.rank == 17))
    tm.stamps[((size_t)tm.epoch * 2 + (tm.rank == 17)) * 8 + k] = wall_clock64();
}

// every store of this workgroup to shared buffers has completed (vm_wait + workgroup barrier by the caller).  Called at the very end
// of a phase function -- before it restores its callee-saved registers; the caller counts the exchange (tm.epoch += 1).
__device__ __forceinline__ void team_arrive(const Team& tm) {
  if (threadIdx.x == 0) __hip_atomic_fetch_add(tm.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave 0: until all team members have arrived `epoch` times
__device__ __forceinline__ void team_wait(const Team& tm) {
  DIT_LDS();
  int* dead = reinterpret_cast<int*>(dit_lds + DEAD_OFF);      // this workgroup stopped waiting (after a timeout)
  if (threadIdx.x == 0 && !*dead) {
    const unsigned target = (unsigned)TEAM * tm.epoch;
    unsigned spins = 0;
    while (__hip_atomic_load(tm.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > SPIN_LIMIT) {
        *dead = 1;
        __hip_atomic_store(tm.status, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // (ordered before this member's leave: team_leave)
        break;
      }
    }
  }
}

// The last member to leave finds nobody polling: it zeroes the counters for the next launch (no memset node: see the file header)
// and RETIRES a timeout of this launch -- word 32 (which member 0 has already turned into a NaN output) is moved to the history
// words 33 (launches that timed out so far) and 34 (the last non-zero status) and cleared, so that one timeout on a busy GPU no
// longer poisons every later launch on this workspace (round-4 ADVICE; the host compares word 33 with the count it saw last).
__device__ __forceinline__ void team_leave(unsigned* ctr) {
  if (threadIdx.x == 0) {
    // acq_rel on the leave counter: every member's status store (release, or relaxed + this release) happens-before the LAST leaver's
    // read of word 32, so a timeout raised late by another member is retired by THIS launch and charged to it -- with relaxed
    // accesses it could stay in word 32, make the next launch return NaN and be counted there (round-5 ADVICE).  The status word
    // is taken with an exchange: read and cleared in one step.
    const unsigned left = __hip_atomic_fetch_add(ctr + 16, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (left == TEAM - 1) {
      const unsigned st = __hip_atomic_exchange(ctr + 32, 0u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (st != 0u) {
        __hip_atomic_store(ctr + 34, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(ctr + 33, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ctr + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct Args {
  const dvla_dit_block_weights* blocks;
  const bf16_t *xemb_w, *xemb_b, *final_w, *final_b, *pos, *cond;
  const float* coef;
  const float* noise;
  float* out;
  bf16_t *X, *QKV, *O, *HID;
  unsigned long long* stamps;
  unsigned* ctr;       // [0] team counter, [16] workgroups that have left, [32] status of the running launch, [33] launches that timed
                       // out so far, [34] last non-zero status, [64 + r] XCC id of member r (last launch)
  float cfg, eps;
  int depth, D, H, C, T, bs, steps;
  int inject;          // test hook (dvla_dit_sample_inject_timeouts): member 1 reports a timeout at the start of the launch
};

enum { EPI_BIAS = 0, EPI_BIAS_RES = 1, EPI_BIAS_GELU = 2, EPI_FINAL = 3 };

// a block's eight weight addresses through the SCALAR cache (the table is constant for the launch): a vector load here would be
// waited for with vmcnt(0), i.e. drain whatever weight requests the wave has in flight
__device__ __forceinline__ dvla_dit_block_weights block_weights(const dvla_dit_block_weights* table, int l) {
  typedef const unsigned long long __attribute__((address_space(4))) * const_q;
  const_q q = (const_q)(table + l);
  dvla_dit_block_weights w;
  w.qkv_w = (const void*)q[0]; w.qkv_b = (const void*)q[1]; w.proj_w = (const void*)q[2]; w.proj_b = (const void*)q[3];
  w.fc1_w = (const void*)q[4]; w.fc1_b = (const void*)q[5]; w.fc2_w = (const void*)q[6]; w.fc2_b = (const void*)q[7];
  return w;
}

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void unpack8(u32x4 u, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(u[i] << 16); v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}

// KS consecutive k32-steps of one operand row: 16 B per lane each, 64 B apart, offsets as immediates
template <int KS, bool DEV, int S = 0>
__device__ __forceinline__ void load_steps(u32x4 (&f)[KS], const bf16_t* p) {
  if constexpr (S < KS) {
    f[S] = DEV ? ldd16<64 * S>(p) : ldg16<64 * S>(p);
    load_steps<KS, DEV, S + 1>(f, p);
  }
}

// One GEMM phase:  out[R, N] = epilogue( [LayerNorm](A[R, K]) . W[N, K]^T ),  K = 8 waves x KS x 32.
// KS: k32-steps per wave; MT: tiles per workgroup (>= ceil(N / 16 / TEAM), <= 8); RB: blocks of 16 rows (R <= 16 RB).
// the weight fragments of one phase for one wave (all tiles of its workgroup, its eighth of K) + the bias quad of the tile the
// wave finishes: MT x KS + 1 loads in flight between weights_request() and the phase body
template <int KS, int MT>
struct WSet {
  u32x4 w[MT][KS];
  u32x2 b;
  static constexpr int LOADS = MT * KS + 1;
};
template <int N_>
__device__ __forceinline__ void vm_wait_but() {      // all but the N_ youngest vector-memory operations of this wave have completed
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <int KS, int MT>
__device__ __forceinline__ void weights_request(const Team& tm, WSet<KS, MT>& ws, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias, int N) {
  constexpr int K = NWAVES * KS * 32;
  const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kb = wave * (KS * 32) + 8 * g;
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    int n = (tm.rank + TEAM * j) * 16 + l15;
    n = n < N ? n : N - 1;                                        // (tiles / columns past the end: a valid address, zeroed in the body)
    const bf16_t* wp = W + (int64_t)n * K + kb;
    load_steps<KS, false>(ws.w[j], wp);
  }
  // the epilogue's bias quad of wave j's tile (lane: row lane / 4, columns 4 (lane % 4) .. + 3 of the tile)
  const int en = (tm.rank + TEAM * wave) * 16 + (lane & 3) * 4;
  ws.b = ldg8(bias + ((wave < MT && en + 3 < N) ? en : 0));
}

// The body of a phase, its weights already requested.  `next()` requests the weights of a LATER phase: it runs after this phase's
// stores have been issued, NEXT_LOADS loads per wave, and the stores are then waited for with a counted s_waitcnt that leaves
// those loads in flight (vector memory operations of a wave retire in order).
template <int KS, int MT, int RB, bool LN, int EPI, int NEXT_LOADS, class Next>
__device__ __forceinline__ void gemm_body(const Team& tm, WSet<KS, MT>& ws, const bf16_t* __restrict__ bias, int N, const bf16_t* A, int lda,
                                          bf16_t* out, int ldo, int R, float eps, Next&& next) {
  static_assert(MT <= NWAVES, "wave j finishes tile j");
  DIT_LDS();
  float* part = dit_lds + PART_OFF;
  float* stat = dit_lds + STAT_OFF;
  float* mo = dit_lds + MO_OFF;
  constexpr int K = NWAVES * KS * 32;
  const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntiles = (N + 15) >> 4;
  const int kb = wave * (KS * 32) + 8 * g;
  u32x4 (&wf)[MT][KS] = ws.w;
  u32x2& bq2 = ws.b;
  const int etile = tm.rank + TEAM * wave, erow = lane >> 2, ec = (lane & 3) * 4;
  const bool ewave = wave < MT && etile < ntiles;
  const int en = etile * 16 + ec;
  const bool evec = ewave && en + 3 < N;
  // 2. the producers of A have arrived
  if (wave == 0) team_wait(tm);
  stamp(tm, 1);
  wg_barrier();
  // 3. this wave's K-share of the rows; the residual quad of the epilogue
  u32x4 af[RB][KS];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int row = rb * 16 + l15;
    row = row < R ? row : R - 1;
    const bf16_t* ap = A + (int64_t)row * lda + kb;
    load_steps<KS, true>(af[rb], ap);
  }
  u32x2 rq[RB];
  if (EPI == EPI_BIAS_RES) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      int row = rb * 16 + erow;
      row = row < R ? row : R - 1;
      int n = etile * 16 + ec;
      n = n + 3 < N ? n : 0;
      rq[rb] = ldd8(out + (int64_t)row * ldo + n);               // the residual stream is updated in place: out IS the residual
    }
  }
  vm_wait();
  stamp(tm, 2);
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int s = 0; s < KS; ++s) pin(wf[j][s]);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int s = 0; s < KS; ++s) pin(af[rb][s]);
    if (EPI == EPI_BIAS_RES) pin(rq[rb]);
  }
  pin(bq2);
  float bq[4] = {0.f, 0.f, 0.f, 0.f};
  if (evec) {
    bq[0] = __uint_as_float(bq2[0] << 16); bq[1] = __uint_as_float(bq2[0] & 0xffff0000u);
    bq[2] = __uint_as_float(bq2[1] << 16); bq[3] = __uint_as_float(bq2[1] & 0xffff0000u);
  } else if (ewave) {                      // (the final layer's 7 columns)
#pragma unroll
    for (int e = 0; e < 4; ++e) if (en + e < N) bq[e] = bf2f(bias[en + e]);
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const bool ok = (tm.rank + TEAM * j) * 16 + l15 < N;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) wf[j][s][e] = ok ? wf[j][s][e] : 0u;
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const bool ok = rb * 16 + l15 < R;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) af[rb][s][e] = ok ? af[rb][s][e] : 0u;
  }
  if (LN) {
    // row statistics: this lane holds 8 KS elements of row l15 (+ 16 rb); the four 16-lane groups and the eight waves hold the rest
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float v[8];
        unpack8(af[rb][s], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sm += v[e]; sq = fmaf(v[e], v[e], sq); }
      }
      sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
      sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
      if (g == 0) { stat[(wave * 32 + rb * 16 + l15) * 2] = sm; stat[(wave * 32 + rb * 16 + l15) * 2 + 1] = sq; }
    }
    lds_wait();
    wg_barrier();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float S = 0.f, Q = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) { S += stat[(w * 32 + rb * 16 + l15) * 2]; Q += stat[(w * 32 + rb * 16 + l15) * 2 + 1]; }
      const float inv_k = 1.0f / (float)K;
      const float mean = S * inv_k;
      const float var = fmaxf(Q * inv_k - mean * mean, 0.f);
      const float rstd = rsqrtf(var + eps);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float v[8];
        unpack8(af[rb][s], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[rb][s][i] = pack2bf(v[2 * i], v[2 * i + 1]);
      }
    }
  }
  // 4. products
  f32x4 acc[MT][RB];
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      acc[j][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) acc[j][rb] = mfma16(af[rb][s], wf[j][s], acc[j][rb]);
    }
  // acc[j][rb][r] = C(16 rb + 4 g + r, 16 tile_j + l15), this wave's eighth of K
#pragma unroll
  for (int j = 0; j < MT; ++j)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(((wave * MT + j) * RB + rb) * 16 + 4 * g + r) * LDP + l15] = acc[j][rb][r];
  lds_wait();
  wg_barrier();
  stamp(tm, 3);
  // 5. wave j finishes tile j
  if (ewave) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int row = rb * 16 + erow;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) {                          // fixed order: deterministic
        const float4 q = *reinterpret_cast<const float4*>(part + (((w * MT + wave) * RB + rb) * 16 + erow) * LDP + ec);
        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bq[e];
      if (EPI == EPI_BIAS_GELU) {
        const f32x2 a = act_fwd2(f32x2{v[0], v[1]}, ACT_GELU_TANH), b = act_fwd2(f32x2{v[2], v[3]}, ACT_GELU_TANH);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
      }
      if (EPI == EPI_BIAS_RES) {
        // the branch value is rounded before the residual joins it (the reference materialises it as a bf16 tensor)
        const float r0 = __uint_as_float(rq[rb][0] << 16), r1 = __uint_as_float(rq[rb][0] & 0xffff0000u);
        const float r2 = __uint_as_float(rq[rb][1] << 16), r3 = __uint_as_float(rq[rb][1] & 0xffff0000u);
        v[0] = bf2f(f2bf(v[0])) + r0; v[1] = bf2f(f2bf(v[1])) + r1; v[2] = bf2f(f2bf(v[2])) + r2; v[3] = bf2f(f2bf(v[3])) + r3;
      }
      if (EPI == EPI_FINAL) {
        if (row < R) {
#pragma unroll
          for (int e = 0; e < 4; ++e) mo[row * 16 + ec + e] = bf2f(f2bf(v[e]));      // the model's bf16 output, as floats
        }
      } else if (row < R) {
        const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        std8(out + (int64_t)row * ldo + etile * 16 + ec, o);
      }
    }
  }
  next();
  vm_wait_but<NEXT_LOADS>();
  lds_wait();
  wg_barrier();
  stamp(tm, 4);
  if (EPI != EPI_FINAL) team_arrive(tm);
}

// request + body behind one call: the variants whose register sets do not fit side by side (two row blocks, hidden 1024) keep
// every phase's registers to itself
template <int KS, int MT, int RB, bool LN, int EPI>
__device__ __attribute__((noinline)) void gemm_phase(const Team tm, const bf16_t* __restrict__ W, const bf16_t* __restrict__ bias, int N, const bf16_t* A,
                                                     int lda, bf16_t* out, int ldo, int R, float eps) {
  WSet<KS, MT> ws;
  weights_request<KS, MT>(tm, ws, W, bias, N);
  stamp(tm, 0);
  gemm_body<KS, MT, RB, LN, EPI, 0>(tm, ws, bias, N, A, lda, out, ldo, R, eps, [] {});
}

// attention of one (sample, head) per wave: L <= 8 tokens, head_dim 64; qkv rows (3 D wide: q | k | v), o rows (D wide).
// Scores on two 16x16x32 MFMAs from fragments loaded in operand layout (lane (l15, g): row l15 of q / k, columns 8 g + 32 s .. + 7);
// S arrives as lane (j = l15, group g), register r = query 4 g + r; softmax across the 16 lanes of a group; P.V on the VALU with
// lane = output column: P(i, j) is broadcast from its lane with v_readlane, V(j, lane) comes from L coalesced 2-byte loads.
template <int NEXT_LOADS, class Next>
__device__ __forceinline__ void attention_body(const Team& tm, const bf16_t* QKV, bf16_t* O, int nsamp, int H, int L, int D, Next&& next) {
  const int t = threadIdx.x, lane = t & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if (wave == 0) team_wait(tm);
  stamp(tm, 1);
  wg_barrier();
  const int units = nsamp * H;
  for (int u = tm.rank + TEAM * wave; u < units; u += TEAM * NWAVES) {
    const int smp = u / H, h = u - smp * H;
    const int rc = l15 < L ? l15 : L - 1;
    const bf16_t* qb = QKV + (int64_t)(smp * L + rc) * (3 * D) + h * 64 + 8 * g;
    u32x4 q0 = ldd16<0>(qb), q1 = ldd16<64>(qb), k0 = ldd16<0>(qb + D), k1 = ldd16<64>(qb + D);
    uint32_t vv[MAX_L];
#pragma unroll
    for (int j = 0; j < MAX_L; ++j) {
      const int jc = j < L ? j : L - 1;
      asm volatile("global_load_ushort %0, %1, off sc1" : "=v"(vv[j]) : "v"(QKV + (int64_t)(smp * L + jc) * (3 * D) + 2 * D + h * 64 + lane) : "memory");
    }
    vm_wait();
    pin(q0); pin(q1); pin(k0); pin(k1);
#pragma unroll
    for (int j = 0; j < MAX_L; ++j) pin(vv[j]);
    f32x4 sc = {0.f, 0.f, 0.f, 0.f};
    sc = mfma16(q0, k0, sc);                     // A = q (rows = queries), B = k (columns = keys)
    sc = mfma16(q1, k1, sc);
    // sc[r] = q(4 g + r) . k(l15)
    float pb[4], lsum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool vis = l15 < L && 4 * g + r < L;
      const float s2 = vis ? sc[r] * (0.125f * 1.4426950408889634f) : -INFINITY;      // head_dim 64: scale 1/8, log2 domain
      float mx = s2;
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64)); mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
      const float M = ceilf(mx);                                                  // integer running maximum (csrc/attention.hip)
      const float p = vis ? fast_exp2(s2 - M) : 0.f;
      float l = p;
      l += __shfl_xor(l, 1, 64); l += __shfl_xor(l, 2, 64); l += __shfl_xor(l, 4, 64); l += __shfl_xor(l, 8, 64);
      pb[r] = bf2f(f2bf(p));
      lsum[r] = l;
    }
    float vf[MAX_L];
#pragma unroll
    for (int j = 0; j < MAX_L; ++j) vf[j] = __uint_as_float(vv[j] << 16);
#pragma unroll
    for (int i = 0; i < MAX_L; ++i) {
      if (i < L) {
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < MAX_L; ++j) {
          if (j < L) {
            const float pij = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(pb[i & 3]), j + 16 * (i >> 2)));
            o = fmaf(pij, vf[j], o);
          }
        }
        const float li = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lsum[i & 3]), 16 * (i >> 2)));
        o = o / li;
        const float hi = __shfl_down(o, 1, 64);
        if ((lane & 1) == 0) std4(O + (int64_t)(smp * L + i) * D + h * 64 + lane, pack2bf(o, hi));
      }
    }
  }
  next();
  vm_wait_but<NEXT_LOADS>();
  wg_barrier();
  stamp(tm, 4);
  team_arrive(tm);
}
__device__ __attribute__((noinline)) void attention_phase(const Team tm, const bf16_t* QKV, bf16_t* O, int nsamp, int H, int L, int D) {
  stamp(tm, 0);
  attention_body<0>(tm, QKV, O, nsamp, H, L, D, [] {});
}

// team member 0, between two sampler steps: [final LayerNorm + linear of step j - 1 -> guidance + DDIM update] -> token embedding
// of step j.  xs: the sampler state (bs, T, C) fp32 in LDS; mo: the model output (R, 16) in LDS.
template <int KS, int RB>
__device__ __attribute__((noinline)) void step_boundary(const Team tm, const Args& a, int j) {
  DIT_LDS();
  float* mo = dit_lds + MO_OFF;
  float* xs = dit_lds + XS_OFF;
  const int t = threadIdx.x;
  const int L = 2 * a.T, R = 2 * a.bs * L, per = a.T * a.C, n = a.bs * per;
  if (j > 0) {
    gemm_phase<KS, 1, RB, true, EPI_FINAL>(tm, a.final_w, a.final_b, a.C, a.X, a.D, nullptr, 0, R, a.eps);
    const float ca = a.coef[4 * (j - 1)], cb = a.coef[4 * (j - 1) + 1], sp = a.coef[4 * (j - 1) + 2], sq = a.coef[4 * (j - 1) + 3];
    for (int i = t; i < n; i += 64 * NWAVES) {
      const int s = i / per, r = i - s * per, tok = r / a.C, c = r - tok * a.C;
      const float cond = mo[(s * L + a.T + tok) * 16 + c], unc = mo[((s + a.bs) * L + a.T + tok) * 16 + c];
      // csrc/elementwise.hip ddim_cfg_step_kernel, operation by operation
      const float d = bf2f(f2bf(__fsub_rn(cond, unc)));
      const float sd = bf2f(f2bf(__fmul_rn(a.cfg, d)));
      const float e = bf2f(f2bf(__fadd_rn(unc, sd)));
      const float ax = __fmul_rn(ca, xs[i]);
      const float px = __fsub_rn(ax, __fmul_rn(cb, e));
      const float e2 = __fdiv_rn(__fsub_rn(ax, px), cb);
      const float xn = __fadd_rn(__fmul_rn(px, sp), __fmul_rn(sq, e2));
      xs[i] = xn;
      if (j == a.steps) {
        const unsigned st = __hip_atomic_load(tm.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.out[i] = st ? __uint_as_float(0x7fc00000u) : xn;
      }
    }
    __syncthreads();
  } else {
    if (tm.rank == 0) {
      for (int i = t; i < n; i += 64 * NWAVES) xs[i] = a.noise[i];
    }
    __syncthreads();
  }
  if (j == a.steps) return;
  // tokens of step j: row (s, i): i < T the condition tokens (z_emb + t_emb[j], precomputed) + pos; i >= T x_embedder(x) + pos
  const int octs = a.D >> 3;
  for (int idx = t; idx < R * octs; idx += 64 * NWAVES) {
    const int row = idx / octs, oc = idx - row * octs, s = row / L, i = row - s * L, d0 = oc * 8;
    float v[8], pv[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.pos + (int64_t)i * a.D + d0), pv);
    if (i < a.T) {
      unpack8(*reinterpret_cast<const u32x4*>(a.cond + ((int64_t)(j * 2 * a.bs + s) * a.T + i) * a.D + d0), v);
    } else {
      const float* xr = xs + ((s % a.bs) * a.T + (i - a.T)) * a.C;
      float bv[8];
      unpack8(*reinterpret_cast<const u32x4*>(a.xemb_b + d0), bv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float acc = 0.f;
        for (int k = 0; k < a.C; ++k) acc = fmaf(bf2f(f2bf(xr[k])), bf2f(a.xemb_w[(int64_t)(d0 + e) * a.C + k]), acc);
        v[e] = bf2f(f2bf(acc + bv[e]));
      }
    }
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = pack2bf(v[2 * q] + pv[2 * q], v[2 * q + 1] + pv[2 * q + 1]);
    std16(a.X + (int64_t)row * a.D + d0, o);
  }
  vm_wait();
  wg_barrier();
  team_arrive(tm);
}

// KS = hidden / 256 (3: DiT-B, 4: DiT-L); RB = blocks of 16 token rows
template <int KS, int RB>
__global__ __launch_bounds__(64 * NWAVES) void dit_team_kernel(Args a) {
  DIT_LDS();
  if ((blockIdx.x & 7) != 0) return;                 // the team: workgroups placed on XCC 0
  constexpr int D = KS * 256;
  constexpr int MT_QKV = (3 * D / 16 + TEAM - 1) / TEAM, MT_D = (D / 16 + TEAM - 1) / TEAM, MT_FC1 = (4 * D / 16 + TEAM - 1) / TEAM;
  Team tm;
  tm.ctr = a.ctr; tm.status = a.ctr + 32; tm.epoch = 0; tm.rank = (int)(blockIdx.x >> 3); tm.stamps = a.stamps;
  if (threadIdx.x == 0) {
    *reinterpret_cast<int*>(dit_lds + DEAD_OFF) = 0;
    a.ctr[64 + tm.rank] = xcc_id();                  // diagnostics: where the team runs (all members on one XCC = the fast case)
    if (a.inject && tm.rank == 1) __hip_atomic_store(tm.status, 3u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int L = 2 * a.T, R = 2 * a.bs * L;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  for (int j = 0; j <= a.steps; ++j) {
    // exchange 0 of a step: the step boundary, on member 0.  The other members only pass through it -- but they WAIT for the
    // previous exchange first: one counter serves all exchanges, so nobody may arrive twice before everybody has arrived once.
    if (tm.rank == 0) {
      const Args boundary_args = a;            // (a copy for the call: the kernel's own arguments stay in scalar registers)
      step_boundary<KS, RB>(tm, boundary_args, j);
    } else if (j < a.steps) {
      if (j > 0) {
        if (wave == 0) team_wait(tm);
        wg_barrier();
      }
      team_arrive(tm);
    }
    if (j == a.steps) break;
    tm.epoch += 1;
    for (int l = 0; l < a.depth; ++l) {
      const dvla_dit_block_weights bw = block_weights(a.blocks, l);
      gemm_phase<KS, MT_QKV, RB, true, EPI_BIAS>(tm, (const bf16_t*)bw.qkv_w, (const bf16_t*)bw.qkv_b, 3 * D, a.X, D, a.QKV, 3 * D, R, a.eps);
      tm.epoch += 1;
      attention_phase(tm, a.QKV, a.O, 2 * a.bs, a.H, L, D);
      tm.epoch += 1;
      gemm_phase<KS, MT_D, RB, false, EPI_BIAS_RES>(tm, (const bf16_t*)bw.proj_w, (const bf16_t*)bw.proj_b, D, a.O, D, a.X, D, R, a.eps);
      tm.epoch += 1;
      gemm_phase<KS, MT_FC1, RB, true, EPI_BIAS_GELU>(tm, (const bf16_t*)bw.fc1_w, (const bf16_t*)bw.fc1_b, 4 * D, a.X, D, a.HID, 4 * D, R, a.eps);
      tm.epoch += 1;
      gemm_phase<4 * KS, MT_D, RB, false, EPI_BIAS_RES>(tm, (const bf16_t*)bw.fc2_w, (const bf16_t*)bw.fc2_b, D, a.HID, 4 * D, a.X, D, R, a.eps);
      tm.epoch += 1;
    }
  }
  // The counters are reset BY THE TEAM, from the XCC that counts on them -- not by a memset node in front of the launch: under
  // hipGraph replay a memset's zeros (written by another agent) were not reliably what this XCC's L2 served to the next launch's
  // device-scope atomics (intermittent barriers that did not wait, round 4).  The last member to leave finds nobody polling.
  team_leave(a.ctr);
}

// The same schedule in one function (no call per phase: 2.4 us of register saves, argument traffic and a drained memory queue per
// exchange in the kernel above), with the weight requests AHEAD of the exchange that precedes their use: a phase's end -- after
// its stores have been issued -- requests the weights of the next GEMM phase (qkv's end proj's, the attention's end fc1's: the two
// short phases hide a whole stream; fc1's end fc2's, fc2's end the next block's qkv: hidden behind arrival + exchange + operand
// load).  Requesting fc2's and qkv's a phase earlier as well was built and measured (profiles/r04_dit_team_perf.jsonl): 44
// spilled registers, and SLOWER (33.2 against 30.8 us per block) -- while the XCD's fabric port is saturated by a weight stream,
// every L2 access of the exchange itself (the sc1 stores' acknowledgements, the counter, the operand loads) waits in the same
// queues: stream and exchange do not overlap beyond the first microsecond, they add.  Nothing requested is ever left unconsumed (a load that lands
// after its register was re-used would corrupt it), and no request is in flight across the only real call (step_boundary).
template <int KS, int RB>
__global__ __launch_bounds__(64 * NWAVES) void dit_team_kernel_ahead(Args a) {
  DIT_LDS();
  if ((blockIdx.x & 7) != 0) return;
  constexpr int D = KS * 256;
  constexpr int MT_QKV = (3 * D / 16 + TEAM - 1) / TEAM, MT_D = (D / 16 + TEAM - 1) / TEAM, MT_FC1 = (4 * D / 16 + TEAM - 1) / TEAM;
  typedef WSet<KS, MT_QKV> WQ;
  typedef WSet<KS, MT_D> WP;
  typedef WSet<KS, MT_FC1> W1;
  typedef WSet<4 * KS, MT_D> W2;
  Team tm;
  tm.ctr = a.ctr; tm.status = a.ctr + 32; tm.epoch = 0; tm.rank = (int)(blockIdx.x >> 3); tm.stamps = a.stamps;
  if (threadIdx.x == 0) {
    *reinterpret_cast<int*>(dit_lds + DEAD_OFF) = 0;
    a.ctr[64 + tm.rank] = xcc_id();
    if (a.inject && tm.rank == 1) __hip_atomic_store(tm.status, 3u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int L = 2 * a.T, R = 2 * a.bs * L;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  WQ wq;
  WP wp;
  W1 w1;
  W2 w2;
  for (int j = 0; j <= a.steps; ++j) {
    if (tm.rank == 0) {
      const Args boundary_args = a;            // (a copy for the call: the kernel's own arguments stay in scalar registers)
      step_boundary<KS, RB>(tm, boundary_args, j);
    } else if (j < a.steps) {
      if (j > 0) {
        if (wave == 0) team_wait(tm);
        wg_barrier();
      }
      team_arrive(tm);
    }
    if (j == a.steps) break;
    tm.epoch += 1;
    {
      const dvla_dit_block_weights b0 = block_weights(a.blocks, 0);
      weights_request<KS, MT_QKV>(tm, wq, (const bf16_t*)b0.qkv_w, (const bf16_t*)b0.qkv_b, 3 * D);
    }
    for (int l = 0; l < a.depth; ++l) {
      const dvla_dit_block_weights bw = block_weights(a.blocks, l);
      stamp(tm, 0);
      gemm_body<KS, MT_QKV, RB, true, EPI_BIAS, WP::LOADS>(tm, wq, (const bf16_t*)bw.qkv_b, 3 * D, a.X, D, a.QKV, 3 * D, R, a.eps, [&] {
        weights_request<KS, MT_D>(tm, wp, (const bf16_t*)bw.proj_w, (const bf16_t*)bw.proj_b, D);
      });
      tm.epoch += 1;
      stamp(tm, 0);
      attention_body<W1::LOADS>(tm, a.QKV, a.O, 2 * a.bs, a.H, L, D, [&] {
        weights_request<KS, MT_FC1>(tm, w1, (const bf16_t*)bw.fc1_w, (const bf16_t*)bw.fc1_b, 4 * D);
      });
      tm.epoch += 1;
      stamp(tm, 0);
      const dvla_dit_block_weights bn = block_weights(a.blocks, l + 1 < a.depth ? l + 1 : 0);
      // (the last block requests block 0's qkv weights: what the next step starts with -- and every block ends the same way, so
      // that the qkv registers are written on every path around the loop)
      gemm_body<KS, MT_D, RB, false, EPI_BIAS_RES, 0>(tm, wp, (const bf16_t*)bw.proj_b, D, a.O, D, a.X, D, R, a.eps, [] {});
      tm.epoch += 1;
      stamp(tm, 0);
      gemm_body<KS, MT_FC1, RB, true, EPI_BIAS_GELU, W2::LOADS>(tm, w1, (const bf16_t*)bw.fc1_b, 4 * D, a.X, D, a.HID, 4 * D, R, a.eps, [&] {
        weights_request<4 * KS, MT_D>(tm, w2, (const bf16_t*)bw.fc2_w, (const bf16_t*)bw.fc2_b, D);
      });
      tm.epoch += 1;
      stamp(tm, 0);
      gemm_body<4 * KS, MT_D, RB, false, EPI_BIAS_RES, WQ::LOADS>(tm, w2, (const bf16_t*)bw.fc2_b, D, a.HID, 4 * D, a.X, D, R, a.eps, [&] {
        weights_request<KS, MT_QKV>(tm, wq, (const bf16_t*)bn.qkv_w, (const bf16_t*)bn.qkv_b, 3 * D);
      });
      tm.epoch += 1;
    }
    // nothing stays in flight across the step boundary (member 0 makes a call there): the request above has warmed the L2
    vm_wait();
#pragma unroll
    for (int jj = 0; jj < MT_QKV; ++jj)
#pragma unroll
      for (int ss = 0; ss < KS; ++ss) pin(wq.w[jj][ss]);
    pin(wq.b);
  }
  team_leave(a.ctr);
}

size_t team_smem_bytes(int KS, int RB) {
  const int D = KS * 256, mt_fc1 = (4 * D / 16 + TEAM - 1) / TEAM;
  size_t part = (size_t)NWAVES * mt_fc1 * RB * 16 * LDP;
  const size_t attn = (size_t)NWAVES * (3 * MAX_L * 68 + MAX_L * 8 + MAX_L);
  if (attn > part) part = attn;
  return (PART_OFF + part) * sizeof(float);
}

}  // namespace

extern "C" int64_t dvla_dit_sample_workspace_bytes(int32_t hidden) {
  if (hidden <= 0) return 0;
  return 1024 + (int64_t)32 * (1 + 3 + 1 + 4) * hidden * 2;
}

// measurement hook (tests/gpu_dit_team_perf.py): a device buffer of 2 x 8 x (exchanges + 1) 64-bit words that the next launches
// fill with wall-clock stamps of team members 0 and 17 -- null (the default) switches the stamping off
static unsigned long long* g_dit_stamps = nullptr;
extern "C" void dvla_dit_sample_set_stamps(void* buf) { g_dit_stamps = reinterpret_cast<unsigned long long*>(buf); }

// test hook (tests/rollout_checks.py): the next n launches report a timeout (status 3) -- their outputs are NaN and word 33 of the
// workspace counts them -- so that the callers' recovery path can be exercised without a busy GPU
static int g_dit_inject = 0;
extern "C" void dvla_dit_sample_inject_timeouts(int32_t n) { g_dit_inject = n > 0 ? n : 0; }

extern "C" int dvla_dit_sample(const dvla_dit_sample_params* q, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!q || !q->blocks || !q->xemb_w || !q->xemb_b || !q->final_w || !q->final_b || !q->pos || !q->cond || !q->coef || !q->noise ||
      !q->out || !q->workspace)
    return DVLA_ERR_ARG;
  if (q->depth < 1 || q->steps < 1 || q->bs < 1 || q->tokens < 1 || q->channels < 1 || q->heads < 1) return DVLA_ERR_ARG;
  const int L = 2 * q->tokens, R = 2 * q->bs * L;
  // DiT-B at one episode: the only shape where the team beats the launch-by-launch sampler (profiles/r04_dit_team_perf.jsonl:
  // two row blocks or hidden 1024 do not fit the look-ahead schedule's registers, and the call-per-phase kernel loses there)
  if (q->hidden != 768 || q->heads * 64 != q->hidden || L > MAX_L || R > 16 || q->channels > 16 || q->bs * q->tokens * q->channels > 256)
    return DVLA_ERR_UNSUPPORTED;
  constexpr int KS = 3, RB = 1;
  if (q->workspace_bytes < dvla_dit_sample_workspace_bytes(q->hidden) || (reinterpret_cast<uintptr_t>(q->workspace) & 15))
    return DVLA_ERR_ARG;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return DVLA_ERR_LAUNCH;
  if (cus < 8 * TEAM) return DVLA_ERR_UNSUPPORTED;      // the team is the 32 CUs of one XCD of a whole MI355X
  Args a;
  a.blocks = q->blocks;
  a.xemb_w = (const bf16_t*)q->xemb_w; a.xemb_b = (const bf16_t*)q->xemb_b;
  a.final_w = (const bf16_t*)q->final_w; a.final_b = (const bf16_t*)q->final_b;
  a.pos = (const bf16_t*)q->pos; a.cond = (const bf16_t*)q->cond;
  a.coef = q->coef; a.noise = q->noise; a.out = q->out;
  a.stamps = g_dit_stamps;
  a.inject = 0;
  if (g_dit_inject > 0) { a.inject = 1; --g_dit_inject; }
  char* ws = reinterpret_cast<char*>(q->workspace);
  a.ctr = reinterpret_cast<unsigned*>(ws);
  a.X = reinterpret_cast<bf16_t*>(ws + 1024);
  a.QKV = a.X + (int64_t)32 * q->hidden;
  a.O = a.QKV + (int64_t)32 * 3 * q->hidden;
  a.HID = a.O + (int64_t)32 * q->hidden;
  a.cfg = q->cfg_scale; a.eps = q->ln_eps;
  a.depth = q->depth; a.D = q->hidden; a.H = q->heads; a.C = q->channels; a.T = q->tokens; a.bs = q->bs; a.steps = q->steps;
  const size_t smem = team_smem_bytes(KS, RB);
  // >= 80 KB of LDS per workgroup: at most two workgroups per CU, so that consecutive workgroups spread over the XCCs of an idle chip
  const size_t lds = smem > 81920 ? smem : 81920;
  static int ahead = -1;
  if (ahead < 0) { const char* e = getenv("DVLA_DIT_AHEAD"); ahead = e ? atoi(e) : 1; }      // 0: the call-per-phase kernel (A/B)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)dit_team_kernel_ahead<KS, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)dit_team_kernel<KS, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return DVLA_ERR_LAUNCH;
    attr_set = true;
  }
  const dim3 grid((unsigned)(8 * TEAM)), block(64 * NWAVES);
  if (ahead) hipLaunchKernelGGL((dit_team_kernel_ahead<KS, RB>), grid, block, lds, stream, a);
  else hipLaunchKernelGGL((dit_team_kernel<KS, RB>), grid, block, lds, stream, a);
  return dvla_check_launch();
}

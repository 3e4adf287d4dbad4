// gemm.hip -- bf16 MFMA GEMM with fused epilogue for gfx950 (MI355X).
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T ),   fp32 accumulate on v_mfma_f32_32x32x16_bf16
//
// Tiling: 128x128x32 block tile, 256 threads = 4 waves in a 2(M) x 2(N) grid, each wave a 64x64 patch
// made of 2x2 MFMA 32x32 tiles (64 accumulator VGPRs).  Operands are staged global -> registers -> LDS
// with the next K-tile's global loads issued before the current tile's MFMAs and written to the
// other LDS buffer after them (one barrier per K-tile).
//
// Both operands may have either memory order (the autograd backward GEMMs need every combination):
//   "k-contiguous" : element (r,k) at P[r*ld + k] -> LDS row-major [128][32+8] bf16, fragments by
//                    one ds_read_b128 per lane (rows are 80 B apart: conflict-free for 16 rows).
//   "r-contiguous" : element (r,k) at P[k*ld + r] -> LDS "pair-interleaved" dwords [k/2][128 rows]
//                    (dword = {k even, k odd} of one row): written with two ds_write_b128 per thread
//                    straight from two coalesced 16-B global loads, fragments by four ds_read_b32.
// The MFMA k-slot <-> k mapping is the same for both layouts (slot (g,j) <-> k = 16*ks + 8*g + j).
//
// The MFMA is issued as mfma(a = B-operand fragment (n), b = A-operand fragment (m)) so that a lane
// owns ONE output row m and 4 consecutive n per accumulator quad; the epilogue transposes each 32-row
// half through a wave-private fp32 LDS patch so a lane ends up with 8 consecutive n of one row and
// reads bias / residual / aux and writes C with 16-byte accesses.
#include "common.h"
#include "../../include/dvla.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int RM_STRIDE = BK + 8;           // bf16 elements per row-major LDS row (80 B)
constexpr int OPER_BYTES = BM * RM_STRIDE * 2;  // 10240 B  (pair-interleaved image needs 8192 B)
constexpr int NTHREADS = 256;

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
  int a_vec, b_vec, c_vec, aux_vec;
  int tiles_m, tiles_n;
};

// ---- staging: global -> registers ------------------------------------------------------------------
// k-contiguous operand: thread t loads rows (t>>2) and (t>>2)+64, k-octet (t&3)
// r-contiguous operand: thread t loads k = 2*(t>>4), 2*(t>>4)+1, rows (t&15)*8 .. +7
template <bool TRANS>
__device__ __forceinline__ void stage_load(uint4 (&reg)[2], const bf16_t* __restrict__ P, int64_t ld, int64_t row0,
                                           int64_t rows, int64_t k0, int64_t k_end, bool vec_ok, int t) {
  if (!TRANS) {
    const int r = t >> 2, kc = (t & 3) * 8;
    reg[0] = load8_guard(P, ld, row0 + r, k0 + kc, rows, k_end, vec_ok);
    reg[1] = load8_guard(P, ld, row0 + r + 64, k0 + kc, rows, k_end, vec_ok);
  } else {
    const int kp = t >> 4, r0 = (t & 15) * 8;
    reg[0] = load8_guard(P, ld, k0 + 2 * kp, row0 + r0, k_end, rows, vec_ok);
    reg[1] = load8_guard(P, ld, k0 + 2 * kp + 1, row0 + r0, k_end, rows, vec_ok);
  }
}

// ---- staging: registers -> LDS ---------------------------------------------------------------------
template <bool TRANS>
__device__ __forceinline__ void stage_store(const uint4 (&reg)[2], char* lds, int t) {
  if (!TRANS) {
    const int r = t >> 2, kc = (t & 3) * 8;
    *reinterpret_cast<uint4*>(lds + ((r)*RM_STRIDE + kc) * 2) = reg[0];
    *reinterpret_cast<uint4*>(lds + ((r + 64) * RM_STRIDE + kc) * 2) = reg[1];
  } else {
    const int kp = t >> 4, r0 = (t & 15) * 8;
    const uint32_t a[4] = {reg[0].x, reg[0].y, reg[0].z, reg[0].w};  // k even : rows r0..r0+7 (2 per dword)
    const uint32_t b[4] = {reg[1].x, reg[1].y, reg[1].z, reg[1].w};  // k odd
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = (a[i] & 0xffffu) | (b[i] << 16);             // row r0+2i   : {k even, k odd}
      o[2 * i + 1] = (a[i] >> 16) | (b[i] & 0xffff0000u);     // row r0+2i+1
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds) + kp * BM + r0;
    *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<uint4*>(dst + 4) = make_uint4(o[4], o[5], o[6], o[7]);
  }
}

// ---- LDS -> MFMA fragment: row `row` of the tile, k16-step ks, lane group g ---------------------------
template <bool TRANS>
__device__ __forceinline__ bf16x8 frag_load(const char* lds, int row, int ks, int g) {
  if (!TRANS) {
    return *reinterpret_cast<const bf16x8*>(lds + (row * RM_STRIDE + ks * 16 + g * 8) * 2);
  } else {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(lds) + (ks * 8 + g * 4) * BM + row;
    union { uint32_t w[4]; bf16x8 v; } u;
    u.w[0] = src[0]; u.w[1] = src[BM]; u.w[2] = src[2 * BM]; u.w[3] = src[3 * BM];
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
  if (vec && n + 8 <= N) {
    unpack8f(*reinterpret_cast<const uint4*>(q), f);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (n + e < N) ? bf2f(q[e]) : 0.f;
  }
}
__device__ __forceinline__ void store8_bf16(bf16_t* q, int64_t n, int64_t N, bool vec, const float (&v)[8]) {
  if (vec && n + 8 <= N) {
    *reinterpret_cast<uint4*>(q) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (n + e < N) q[e] = f2bf(v[e]);
  }
}

// eight consecutive outputs of row m: columns n .. n+7 (values arrive in fp32 from the LDS transpose)
__device__ __forceinline__ void epilogue_oct(const GemmKArgs& p, int64_t m, int64_t n, float (&v)[8], int split) {
  if (p.split_k > 1) {  // raw partial sums -> workspace[split][m][n]
    float* w = p.workspace + ((int64_t)split * p.M + m) * p.N + n;
    if (n + 8 <= p.N && (p.N & 3) == 0) {
      *reinterpret_cast<float4*>(w) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(w + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) w[e] = v[e];
    }
    return;
  }
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (n + e < p.N)
        v[e] += p.bias_f32 ? reinterpret_cast<const float*>(p.bias)[n + e]
                           : bf2f(reinterpret_cast<const bf16_t*>(p.bias)[n + e]);
  }
  if (p.preact) {
    store8_bf16(p.preact + m * p.ld_preact + n, n, p.N, p.aux_vec, v);
    // the activation sees the value that was stored (bf16), exactly like act(preact_tensor)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
  }
  if (p.act != ACT_NONE) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_fwd(v[e], p.act);
  }
  if (p.dact_aux) {
    float a[8];
    load8_aux(p.dact_aux + m * p.ld_dact + n, n, p.N, p.aux_vec, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= act_bwd(a[e], p.dact);
  }
  if (p.has_drop) {
    const uint32_t rowkey = drop_rowkey(p.seed_lo, p.seed_hi, (uint32_t)m);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t h = drop_hash_rk(rowkey, (uint32_t)(n + e));
      v[e] = (h >= p.drop_thr) ? v[e] * p.drop_scale : 0.f;
    }
  }
  if (p.residual) {
    float a[8];
    load8_aux(p.residual + (p.res_rows > 0 ? m % p.res_rows : m) * p.ld_res + n, n, p.N, p.aux_vec, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += a[e];
  }
  if (p.c_f32) {
    float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
    if (p.accumulate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) c[e] += v[e];
    } else if (p.c_vec && n + 8 <= p.N) {
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (n + e < p.N) c[e] = v[e];
    }
  } else {
    store8_bf16(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n, n, p.N, p.c_vec, v);
  }
}

template <bool A_T, bool B_T>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmKArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * OPER_BYTES];  // [buf][A|B]
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave & 1, wn = wave >> 1;
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

  f32x16 acc[2][2];  // [n-subtile i][m-subtile j]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (int)((k_end - k_begin + BK - 1) / BK);
  uint4 ra[2], rb[2];
  if (nk > 0) {
    stage_load<A_T>(ra, p.A, p.lda, m0, p.M, k_begin, k_end, p.a_vec, t);
    stage_load<B_T>(rb, p.B, p.ldb, n0, p.N, k_begin, k_end, p.b_vec, t);
    stage_store<A_T>(ra, smem, t);
    stage_store<B_T>(rb, smem + OPER_BYTES, t);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * 2 * OPER_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * 2 * OPER_BYTES;
    const bool more = (kt + 1 < nk);
    if (more) {
      const int64_t k0 = k_begin + (int64_t)(kt + 1) * BK;
      stage_load<A_T>(ra, p.A, p.lda, m0, p.M, k0, k_end, p.a_vec, t);
      stage_load<B_T>(rb, p.B, p.ldb, n0, p.N, k0, k_end, p.b_vec, t);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fa[j] = frag_load<A_T>(cur, wm * 64 + j * 32 + l31, ks, g);
#pragma unroll
      for (int i = 0; i < 2; ++i) fb[i] = frag_load<B_T>(cur + OPER_BYTES, wn * 64 + i * 32 + l31, ks, g);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i], fa[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      stage_store<A_T>(ra, nxt, t);
      stage_store<B_T>(rb, nxt + OPER_BYTES, t);
    }
    __syncthreads();
  }

  // ---- epilogue --------------------------------------------------------------------------------------
  // acc[i][j][r] holds (m = 32j + l31, n = 32i + 8*(r>>2) + 4*g + (r&3)) of the wave's 64x64 patch.  Each
  // 32-row half is transposed through a wave-private fp32 LDS patch [32][64+4] so that a lane then owns 8
  // consecutive n of one row: 16-byte aux loads / C stores, and the (large) epilogue body is emitted once.
  constexpr int PATCH_LD = 68;  // floats per patch row (272 B: 16-B aligned, 4-bank skew per row)
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PATCH_LD);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    __syncthreads();  // operand buffers (j = 0) / previous half (j = 1) no longer read
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<float4*>(patch + l31 * PATCH_LD + 32 * i + 8 * rq + 4 * g) =
            make_float4(acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]);
    __syncthreads();
    for (int it = 0; it < 4; ++it) {
      const int item = it * 64 + lane;
      const int row = item >> 3, cg = item & 7;
      const int64_t m = m0 + wm * 64 + j * 32 + row;
      const int64_t n = n0 + wn * 64 + cg * 8;
      if (m < p.M && n < p.N) {
        const float4 lo = *reinterpret_cast<const float4*>(patch + row * PATCH_LD + cg * 8);
        const float4 hi = *reinterpret_cast<const float4*>(patch + row * PATCH_LD + cg * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        epilogue_oct(p, m, n, v, split);
      }
    }
  }
}

// split-K reduction: C[m,n] (+)= sum_s ws[s][m][n]
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, void* C, int64_t ldc, int c_f32, int accumulate,
                                     int64_t M, int64_t N, int splits) {
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

inline bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

extern "C" int dvla_gemm_bf16(const dvla_gemm_params* q, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!q || !q->A || !q->B || !q->C) return DVLA_ERR_ARG;
  if (q->M < 0 || q->N < 0 || q->K < 0) return DVLA_ERR_ARG;
  if (q->M == 0 || q->N == 0) return DVLA_OK;
  if (q->accumulate && q->c_dtype != DVLA_DT_F32) return DVLA_ERR_ARG;
  if (q->dropout_p < 0.f || q->dropout_p >= 1.f) return DVLA_ERR_ARG;
  const int split_k = q->split_k > 1 ? q->split_k : 1;
  if (split_k > 1) {
    if (!q->workspace || q->bias || q->act || q->preact || q->dact_aux || q->residual || q->dropout_p > 0.f)
      return DVLA_ERR_ARG;
  }
  GemmKArgs a;
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
  a.tiles_m = (int)((q->M + BM - 1) / BM);
  a.tiles_n = (int)((q->N + BN - 1) / BN);

  dim3 grid((unsigned)(a.tiles_m * a.tiles_n), (unsigned)split_k, 1), block(NTHREADS, 1, 1);
  const int combo = (q->a_trans ? 2 : 0) | (q->b_trans ? 1 : 0);
  switch (combo) {
    case 0: hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, stream, a); break;
    case 1: hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, stream, a); break;
  }
  int rc = dvla_check_launch();
  if (rc != DVLA_OK) return rc;
  if (split_k > 1) {
    const int64_t total = q->M * q->N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       a.workspace, q->C, q->ldc, a.c_f32, q->accumulate, q->M, q->N, split_k);
    rc = dvla_check_launch();
  }
  return rc;
}

// attention_small.hip -- softmax attention for SHORT sequences with ANY head_dim <= 128 (forward + backward), gfx950.
//
// The only place the reference produces a head_dim other than 64 is the DiT-S action head
// (/root/reference/models/action_model/action_model.py:12-14: hidden 384 / 4 heads = 96) and its sequences are
// 2 x action_pred_steps = 6 tokens long (models.py:234-251): one wave per (batch, head), lane i owns query i, everything in
// LDS, plain VALU dot products (a 6 x 6 x 96 problem is 3.5 kFLOP -- the MFMA kernels of attention.hip would spend a 32 x 32
// tile on it).  No mask, no dropout (timm Attention in the DiT blocks has neither).  L <= 64, D % 8 == 0, D <= 128.
// Same arithmetic as attention.hip (and as oracle/torch_ref.py::attention_bf16): scores in fp32, integer running maximum in the
// log2 domain, row sum of the unrounded probabilities, P and dS rounded to bf16 where they enter the second product.
#include "common.h"
#include "../../include/dvla.h"

namespace {

constexpr float S_LOG2E = 1.4426950408889634f;
constexpr float S_LN2 = 0.6931471805599453f;
constexpr int SMAX_L = 64, SMAX_D = 128;

struct SmallArgs {
  const bf16_t *q, *k, *v, *o, *dout; bf16_t *out, *dq, *dk, *dv; float* lse;
  int64_t qsb, qst, qsh, ksb, kst, ksh, vsb, vst, vsh, osb, ost, osh, dsb, dst, dsh;
  int64_t dqsb, dqst, dqsh, dksb, dkst, dksh, dvsb, dvst, dvsh;
  int B, H, L, D; float scale;
};

__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

// stage a (L, D) bf16 matrix of one (b, h) into LDS as fp32 [L][D + 1]
__device__ __forceinline__ void stage(float* dst, const bf16_t* src, int64_t st, int L, int D, int lane) {
  for (int idx = lane; idx < L * D; idx += 64) {
    const int r = idx / D, c = idx - r * D;
    dst[r * (D + 1) + c] = bf2f(src[(int64_t)r * st + c]);
  }
}

__global__ __launch_bounds__(64) void attn_small_fwd_kernel(SmallArgs p) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
  const int L = p.L, D = p.D, W = D + 1;
  float *Q = sm, *K = Q + L * W, *V = K + L * W;
  stage(Q, p.q + b * p.qsb + h * p.qsh, p.qst, L, D, lane);
  stage(K, p.k + b * p.ksb + h * p.ksh, p.kst, L, D, lane);
  stage(V, p.v + b * p.vsb + h * p.vsh, p.vst, L, D, lane);
  __syncthreads();
  if (lane >= L) return;
  const float sl = p.scale * S_LOG2E;
  float mx = -INFINITY;
  for (int j = 0; j < L; ++j) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(Q[lane * W + d], K[j * W + d], s);
    mx = fmaxf(mx, s * sl);
  }
  const float M = ceilf(mx);
  float l = 0.f;
  float* PS = V + L * W;            // this lane's row of bf16-rounded probabilities (no per-lane arrays: no scratch)
  for (int j = 0; j < L; ++j) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(Q[lane * W + d], K[j * W + d], s);
    const float pr = exp2f(fmaf(s, sl, -M));
    l += pr;
    PS[lane * (L + 1) + j] = rbf(pr);
  }
  const float inv = 1.0f / l;
  bf16_t* o = p.out + b * p.osb + (int64_t)lane * p.ost + h * p.osh;
  for (int d = 0; d < D; ++d) {
    float acc = 0.f;
    for (int j = 0; j < L; ++j) acc = fmaf(PS[lane * (L + 1) + j], V[j * W + d], acc);
    o[d] = f2bf(acc * inv);
  }
  if (p.lse) p.lse[((int64_t)b * p.H + h) * L + lane] = (M + log2f(l)) * S_LN2;
}

__global__ __launch_bounds__(64) void attn_small_bwd_kernel(SmallArgs p) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
  const int L = p.L, D = p.D, W = D + 1;
  float *Q = sm, *K = Q + L * W, *V = K + L * W, *DO = V + L * W, *PS = DO + L * W, *DS = PS + L * (L + 1);
  stage(Q, p.q + b * p.qsb + h * p.qsh, p.qst, L, D, lane);
  stage(K, p.k + b * p.ksb + h * p.ksh, p.kst, L, D, lane);
  stage(V, p.v + b * p.vsb + h * p.vsh, p.vst, L, D, lane);
  stage(DO, p.dout + b * p.dsb + h * p.dsh, p.dst, L, D, lane);
  __syncthreads();
  const float sl = p.scale * S_LOG2E;
  if (lane < L) {
    const float lse2 = p.lse[((int64_t)b * p.H + h) * L + lane] * S_LOG2E;
    const bf16_t* o = p.o + b * p.osb + (int64_t)lane * p.ost + h * p.osh;
    float delta = 0.f;
    for (int d = 0; d < D; ++d) delta = fmaf(DO[lane * W + d], bf2f(o[d]), delta);
    for (int j = 0; j < L; ++j) {
      float s = 0.f, dp = 0.f;
      for (int d = 0; d < D; ++d) { s = fmaf(Q[lane * W + d], K[j * W + d], s); dp = fmaf(DO[lane * W + d], V[j * W + d], dp); }
      const float pr = exp2f(fmaf(s, sl, -lse2));
      PS[lane * (L + 1) + j] = rbf(pr);
      DS[lane * (L + 1) + j] = rbf(pr * (dp - delta) * p.scale);
    }
    bf16_t* g = p.dq + b * p.dqsb + (int64_t)lane * p.dqst + h * p.dqsh;
    for (int d = 0; d < D; ++d) {
      float acc = 0.f;
      for (int j = 0; j < L; ++j) acc = fmaf(DS[lane * (L + 1) + j], K[j * W + d], acc);
      g[d] = f2bf(acc);
    }
  }
  __syncthreads();
  if (lane < L) {      // lane = key j: dK_j = sum_i dS_ij Q_i, dV_j = sum_i P_ij dO_i
    bf16_t* gk = p.dk + b * p.dksb + (int64_t)lane * p.dkst + h * p.dksh;
    bf16_t* gv = p.dv + b * p.dvsb + (int64_t)lane * p.dvst + h * p.dvsh;
    for (int d = 0; d < D; ++d) {
      float ak = 0.f, av = 0.f;
      for (int i = 0; i < L; ++i) {
        ak = fmaf(DS[i * (L + 1) + lane], Q[i * W + d], ak);
        av = fmaf(PS[i * (L + 1) + lane], DO[i * W + d], av);
      }
      gk[d] = f2bf(ak); gv[d] = f2bf(av);
    }
  }
}

constexpr size_t SMALL_LDS_LIMIT = 160 * 1024;
inline size_t small_fwd_smem(int L, int D) { return ((size_t)3 * L * (D + 1) + (size_t)L * (L + 1)) * 4; }
inline size_t small_bwd_smem(int L, int D) { return ((size_t)4 * L * (D + 1) + (size_t)2 * L * (L + 1)) * 4; }

// hipFuncSetAttribute is per device: remember which devices have it (a process may drive several)
template <class K>
void small_allow_lds(K kernel, uint64_t& done_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (done_mask & (1ull << dev)) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LDS_LIMIT);
  done_mask |= 1ull << dev;
}

int fill(const dvla_attn_params* q, int head_dim, SmallArgs& a) {
  if (!q || !q->q || !q->k || !q->v || !q->o) return DVLA_ERR_ARG;
  if (q->B <= 0 || q->H <= 0 || q->Lq <= 0 || q->Lk != q->Lq || !(q->scale > 0.f)) return DVLA_ERR_ARG;
  if (q->Lq > SMAX_L || head_dim <= 0 || head_dim > SMAX_D || head_dim % 8 != 0) return DVLA_ERR_UNSUPPORTED;
  if (q->tile_map || q->key_index || q->dropout_p > 0.f) return DVLA_ERR_UNSUPPORTED;   // no mask / dropout on this path
  if (q->B > 65535) return DVLA_ERR_UNSUPPORTED;     // gridDim.y
  // the BACKWARD kernel's LDS (the larger of the two) must fit the 160 KiB of a CU: refused here, for forward and backward
  // alike, so that a forward never succeeds whose backward cannot be launched (L = 64, D = 128 needs 165 376 B)
  if (small_bwd_smem(q->Lq, head_dim) > SMALL_LDS_LIMIT) return DVLA_ERR_UNSUPPORTED;
  a.q = (const bf16_t*)q->q; a.k = (const bf16_t*)q->k; a.v = (const bf16_t*)q->v;
  a.o = (const bf16_t*)q->o; a.out = (bf16_t*)q->o;
  a.qsb = q->q_stride_b; a.qst = q->q_stride_t; a.qsh = q->q_stride_h;
  a.ksb = q->k_stride_b; a.kst = q->k_stride_t; a.ksh = q->k_stride_h;
  a.vsb = q->v_stride_b; a.vst = q->v_stride_t; a.vsh = q->v_stride_h;
  a.osb = q->o_stride_b; a.ost = q->o_stride_t; a.osh = q->o_stride_h;
  a.B = q->B; a.H = q->H; a.L = q->Lq; a.D = head_dim; a.scale = q->scale; a.lse = q->lse;
  a.dout = (const bf16_t*)q->dout; a.dsb = q->do_stride_b; a.dst = q->do_stride_t; a.dsh = q->do_stride_h;
  a.dq = (bf16_t*)q->dq; a.dk = (bf16_t*)q->dk; a.dv = (bf16_t*)q->dv;
  a.dqsb = q->dq_stride_b; a.dqst = q->dq_stride_t; a.dqsh = q->dq_stride_h;
  a.dksb = q->dk_stride_b; a.dkst = q->dk_stride_t; a.dksh = q->dk_stride_h;
  a.dvsb = q->dv_stride_b; a.dvst = q->dv_stride_t; a.dvsh = q->dv_stride_h;
  return DVLA_OK;
}

}  // namespace

extern "C" int dvla_attn_small_fwd(const dvla_attn_params* q, int32_t head_dim, void* stream_) {
  SmallArgs a;
  int rc = fill(q, head_dim, a);
  if (rc != DVLA_OK) return rc;
  const size_t smem = small_fwd_smem(a.L, a.D);
  static uint64_t attr_done = 0;
  small_allow_lds(attn_small_fwd_kernel, attr_done);
  hipLaunchKernelGGL(attn_small_fwd_kernel, dim3((unsigned)a.H, (unsigned)a.B), dim3(64), smem, reinterpret_cast<hipStream_t>(stream_), a);
  return dvla_check_launch();
}

extern "C" int dvla_attn_small_bwd(const dvla_attn_params* q, int32_t head_dim, void* stream_) {
  SmallArgs a;
  int rc = fill(q, head_dim, a);
  if (rc != DVLA_OK) return rc;
  if (!q->dout || !q->lse || !q->dq || !q->dk || !q->dv) return DVLA_ERR_ARG;
  const size_t smem = small_bwd_smem(a.L, a.D);
  static uint64_t attr_done = 0;
  small_allow_lds(attn_small_bwd_kernel, attr_done);
  hipLaunchKernelGGL(attn_small_bwd_kernel, dim3((unsigned)a.H, (unsigned)a.B), dim3(64), smem, reinterpret_cast<hipStream_t>(stream_), a);
  return dvla_check_launch();
}
